"""The spherical tracer's sub-step nodes j / N without a division (prhf_snell.inc substep_node): the reciprocal-and-
correction form equals the IEEE quotient for every pair the kernel uses it for (reference library.py:1656-1657)."""
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_corrected_quotient_is_the_ieee_quotient_up_to_the_kernels_limit(tmp_path):
    src = open(os.path.join(HERE, "..", "pyrayhf_amd", "csrc", "prhf_snell.inc")).read()
    limit = int(re.search(r"constexpr int kQuotientProven = (\d+);", src).group(1))
    exe = str(tmp_path / "substep_quotient")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(HERE, "devtools", "substep_quotient.c"), "-lm"],
                   check=True)
    out = subprocess.run([exe, str(limit)], check=True, capture_output=True, text=True).stdout.split()
    assert int(out[0]) == limit * (limit + 3) // 2 and int(out[1]) == 0
