"""csrc/libstdcxx_sort.h restates libstdc++'s std::sort step by step (MMS_RANK_TIES_LIBSTDCXX: the order the reference's
unstable sort leaves EQUAL scores in).  It is host-and-device code: here g++ compiles it next to the real std::sort and
20,001 sequences must come out in the same order, payload for payload."""
import os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_restatement_matches_std_sort(tmp_path):
    exe = str(tmp_path / "libstdcxx_sort_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "mms_answer_selection_amd", "csrc"),
                           os.path.join(ROOT, "tests", "libstdcxx_sort_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mismatches 0" in out.stdout, out.stdout
