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


def _build_threads_driver(tmp_path, sanitizer):
    csrc = os.path.join(ROOT, "aho-corasick-1975_amd", "csrc")
    exe = str(tmp_path / ("threads_" + sanitizer))
    cmd = ["gcc", "-O1", "-g", "-std=c11", "-Wall", "-Wextra", "-pthread", "-fsanitize=" + sanitizer, "-fno-sanitize-recover=all",
           "-I", os.path.join(ROOT, "include"), "-I", csrc, os.path.join(ROOT, "tests", "helpers", "threads_driver.c"),
           os.path.join(csrc, "acm_host.c"), os.path.join(csrc, "acm_flat.c"), "-o", exe]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return exe


def test_shared_machine_readers_beside_an_inserter(tmp_path):
    """The reference's threading model (README.md:364): one machine, one cursor per thread,
    insertions in between.  acm_match / acm_get_match take no lock; the host trie publishes with
    release stores and a per-state sequence lock (acm_host.c child_find / child_insert).  Under
    ThreadSanitizer and under AddressSanitizer, with the inserter working beside the readers."""
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1", ASAN_OPTIONS="detect_leaks=1")
    env.pop("LD_PRELOAD", None)
    for sanitizer in ("thread", "address,undefined"):
        exe = _build_threads_driver(tmp_path, sanitizer)
        for mode in ("disjoint", "shared"):
            r = subprocess.run([exe, mode], capture_output=True, text=True, env=env, timeout=600)
            assert r.returncode == 0, (sanitizer, mode, r.stdout + r.stderr)
            assert "threads ok (%s)" % mode in r.stdout
