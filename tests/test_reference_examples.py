"""The reference's own example programs, compiled IN PLACE (never copied) against the drop-in:
`/root/reference/examples/test.c` and `aho_corasick_generic_test.c` include "aho_corasick.h" --
here that is include/aho_corasick.h -- and link libac75_amd.so instead of ../aho_corasick.o + -lmap
(recipe of /root/reference/examples/Makefile:9-25 with those two substitutions).

Build-container test only: /root/reference does not travel to the GPU box, where this module is
skipped.  No GPU is involved (the examples use the per-symbol API)."""
import os
import re
import subprocess

import pytest

from tests.conftest import GOLDEN, ROOT

REF = "/root/reference/examples"
PKG = os.path.join(ROOT, "aho-corasick-1975_amd")

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "test.c")),
                                reason="the reference sources are not on this machine")


def _build(tmp_path, source):
    exe = str(tmp_path / os.path.splitext(source)[0])
    subprocess.run(["gcc", "-O3", "-std=c11", "-I", os.path.join(ROOT, "include"), os.path.join(REF, source), "-o", exe,
                    "-L", PKG, "-lac75_amd", "-Wl,-rpath," + PKG, "-pthread"], check=True)
    return exe


def _run(exe, *args):
    env = dict(os.environ, LC_ALL="C.UTF-8")
    # cwd: the generic test opens "mrs_dalloway.txt" by relative name; tests/golden holds the same text
    p = subprocess.run([exe, *args], cwd=GOLDEN, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    return p.stdout.decode("utf-8")


def test_readme_example_program(tmp_path):
    """examples/test.c: its stdout is given literally in README.md:92-93."""
    out = _run(_build(tmp_path, "test.c")).split("\n")
    assert out[0] == "To ushers: he found his pencil, but she could not find hers."
    assert out[1] == " 6:he 5:she 6:hers 12:he 21:his 38:he 37:she 56:he 56:hers"


@pytest.fixture(scope="module")
def generic(tmp_path_factory):
    return _build(tmp_path_factory.mktemp("generic"), "aho_corasick_generic_test.c")


def test_generic_test_1_asserts_and_matches(generic):
    """generic_test.c:62-164 (wchar_t + case-insensitive alphacmp): every assert passes (exit 0;
    :70 empty machine, :114 first/duplicate registration, :117 value semantics) and the scan
    reports the matches of SURVEY.md Appendix C in order."""
    out = _run(generic, "1")
    assert out.startswith("Incremental string matching (Meyer, 1985) in use.\n")
    assert " [21]\n" in out                                    # 26 inserts -> 21 keywords
    tail = out[out.index("{'He found his pencil"):]
    found = re.findall(r"\{'([^']*)'=\d+\}", tail)
    assert found == ["he", "u", "hi", "his", "pen", "u", "she", "he", "u", "he", "hers", "hi", "u", "she", "he",
                     "ushers", "hers", "abc", "bc", "abcd", "bcd", "cd", "abcde", "bcde", "cde", "bcdef", "cdefg"]
    # acm_print facts of Appendix C: failure links drawn as (v <id>), output counts as [+n]
    assert "[+3]" in out and "[+2]" in out


def test_generic_test_2_incremental_dictionary(generic):
    """generic_test.c:166-239: 6,966 keywords inserted WHILE the novel is scanned (assert :211
    holds: exit 0); counts of Appendix C."""
    out = _run(generic, "2")
    assert "6966 keywords registered." in out
    for kw, n in ((" you ", 116), (" years ", 59), (" yes ", 47)):
        assert "{'%s'=%d}" % (kw, n) in out


def test_generic_test_3_random_keywords(generic):
    """generic_test.c:241-277 (ACM_CMP_DEFAULT over char, glibc's unseeded rand()): new keywords and
    matches per round as recorded in SURVEY.md Appendix C."""
    out = _run(generic, "4")
    new = [int(x.replace(",", "")) for x in re.findall(r"\] ([\d,]+) new keywords added", out)]
    hits = [int(x.replace(",", "")) for x in re.findall(r"\] ([\d,]+) matches found", out)]
    assert new == [25000, 25000, 24999, 24997, 25000, 25000, 25000, 25000, 25000, 24999]
    assert hits == [3, 3, 3, 13, 30, 21, 19, 26, 31, 30]
