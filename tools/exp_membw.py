"""Experiment: what plain streaming kernels reach on this GPU (torch fill_ = write only, sum = read only,
copy_ = read + write), to price the record-writing kernels against."""
import time, torch
n = 1 << 30   # 4 GiB of int32
x = torch.empty(n, dtype=torch.int32, device="cuda"); y = torch.empty_like(x)
def t(f, reps=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
b = n * 4
print("fill  (write)      %.2f TB/s" % (b / t(lambda: x.fill_(1)) / 1e12))
print("sum   (read)       %.2f TB/s" % (b / t(lambda: x.sum()) / 1e12))
print("copy  (read+write) %.2f TB/s of traffic" % (2 * b / t(lambda: y.copy_(x)) / 1e12))
