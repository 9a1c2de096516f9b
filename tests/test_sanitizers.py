"""The host side of the library (acm_host.c + acm_flat.c) under AddressSanitizer and UBSan: GPU
sanitizers are not available on this pool, the host C is where the pointer work lives."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_library_under_asan_and_ubsan(tmp_path):
    csrc = os.path.join(ROOT, "aho-corasick-1975_amd", "csrc")
    exe = str(tmp_path / "asan_driver")
    cmd = ["gcc", "-O1", "-g", "-std=c11", "-Wall", "-Wextra", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-I", os.path.join(ROOT, "include"), "-I", csrc, os.path.join(ROOT, "tests", "helpers", "asan_driver.c"),
           os.path.join(csrc, "acm_host.c"), os.path.join(csrc, "acm_flat.c"), "-o", exe]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks held" in r.stdout
