"""Prints the last ~14 kernels of a rocprofv3 --kernel-trace csv with start offsets and durations (us)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
want = int(sys.argv[2]) if len(sys.argv) > 2 else 14
# the last scan kernel that is followed by order kernels
last = max(i for i, nm in enumerate(names) if "order_bucket_kernel<false>" in nm or "order_bucket_kernelILb0" in nm)
lo = max(0, last - 4)
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:lo + want]:
    print("%9.1f us  +%8.1f us  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:90]))
