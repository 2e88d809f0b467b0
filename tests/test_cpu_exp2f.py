"""The device computes std::exp2(float) with a restatement of the libm algorithm
(kmi_extract.hip: exp2f_libm). tests/cpu/exp2f_check.c holds the same restatement in plain C and
compares it with this host's libm exp2f on 2e7 inputs."""
import os
import subprocess


def test_exp2f_restatement_matches_libm(tmp_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "exp2f_check")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-x", "c++", os.path.join(root, "tests", "cpu", "exp2f_check.c"), "-o", exe, "-lm"])
    out = subprocess.check_output([exe], text=True)
    assert "nofma=0 fma=0" in out, out
