"""pyrayhf_amd/csrc/prhf_crmath.h compiled for the host: x^3, x^4, sin, cos must be the exactly rounded values.

The header is the arithmetic the reference-order tier of the HIP kernel uses for YT**3, YT**4 (NumPy: pow,
reference library.py:217, :244) and sin / cos (library.py:210-211).  The expected values are computed here
from exact rationals / 80-digit Taylor sums and rounded once, independent of any libm.
"""

import ctypes
import math
import os
import shutil
import subprocess
from decimal import Decimal, getcontext
from fractions import Fraction

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def crlib(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    out = tmp_path_factory.mktemp("crmath") / "libcr.so"
    subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fPIC", "-shared",
                    "-I", os.path.join(REPO, "pyrayhf_amd", "csrc"),
                    os.path.join(REPO, "tests", "devtools", "crmath_host.cpp"), "-o", str(out), "-lm"], check=True)
    lib = ctypes.CDLL(str(out))
    for fn in (lib.cr_sincos, lib.cr_sincos_table, lib.cr_pow34):
        fn.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p]
        fn.restype = None
    return lib


def _exact_sin_cos(x):
    getcontext().prec = 100
    x = Decimal(float(x))
    x2 = x * x
    tiny = abs(x) * Decimal(10) ** -70 if x != 0 else Decimal(0)
    s, t, k = Decimal(0), x, 1
    while abs(t) > tiny:
        s += t
        t = -t * x2 / ((k + 1) * (k + 2))
        k += 2
    c, t, k = Decimal(0), Decimal(1), 0
    while abs(t) > Decimal(10) ** -70:
        c += t
        t = -t * x2 / ((k + 1) * (k + 2))
        k += 2
    return float(Fraction(s)), float(Fraction(c))


def test_sin_cos_are_correctly_rounded(crlib):
    rng = np.random.default_rng(11)
    r = np.concatenate([
        np.deg2rad(rng.uniform(0.0, 90.0, 1200)),                 # field angles of the operator
        rng.uniform(-7.0, 7.0, 400),                              # every quadrant, both signs
        rng.uniform(-25.0, 25.0, 200),                            # n up to 16 (the Taylor reference below loses
                                                                  # e^|x| digits: 100 digits are plenty to 25)
        np.deg2rad(np.array([0.0, 30.0, 45.0, 60.0, 89.999, 90.0, 135.0, 180.0, 270.0, 360.0, 1e-8, 1e-300])),
    ])
    s = np.empty_like(r)
    c = np.empty_like(r)
    crlib.cr_sincos(r.ctypes.data, r.size, s.ctypes.data, c.ctypes.data)
    want = np.array([_exact_sin_cos(v) for v in r])
    # a double-double evaluation misses the exactly rounded value only within ~2^-9 ulp of a rounding
    # boundary; on this sample it never does
    assert np.array_equal(s, want[:, 0])
    assert np.array_equal(c, want[:, 1])


def test_table_sin_cos_are_correctly_rounded_as_often_as_a_good_libm(crlib):
    """sincos_table - what the kernels call - is the exactly rounded value in > 99.7 % of the cases and never more than
    one ulp off; NumPy's own sin / cos (the platform libm) manage ~99.9 % on the same angles."""
    rng = np.random.default_rng(21)
    r = np.concatenate([
        np.deg2rad(rng.uniform(0.0, 90.0, 2500)), np.deg2rad(rng.uniform(0.0, 3.0, 400)),
        np.deg2rad(rng.uniform(87.0, 90.0, 400)), rng.uniform(-7.0, 7.0, 400),
        np.deg2rad(np.array([0.0, 1e-9, 30.0, 45.0, 60.0, 89.999, 90.0, 135.0, 180.0, 270.0, 360.0])),
    ])
    s = np.empty_like(r)
    c = np.empty_like(r)
    crlib.cr_sincos_table(r.ctypes.data, r.size, s.ctypes.data, c.ctypes.data)
    want = np.array([_exact_sin_cos(v) for v in r])
    ws, wc = np.ascontiguousarray(want[:, 0]), np.ascontiguousarray(want[:, 1])
    assert (s == ws).mean() > 0.997 and (c == wc).mean() > 0.997
    assert np.abs(s.view(np.int64) - ws.view(np.int64)).max() <= 1
    assert np.abs(c.view(np.int64) - wc.view(np.int64)).max() <= 1
    # and it agrees with the double-double series wherever that one is exact
    s2 = np.empty_like(r)
    c2 = np.empty_like(r)
    crlib.cr_sincos(r.ctypes.data, r.size, s2.ctypes.data, c2.ctypes.data)
    assert (s == s2).mean() > 0.997 and (c == c2).mean() > 0.997


def test_pow3_pow4_are_correctly_rounded(crlib):
    rng = np.random.default_rng(12)
    x = np.concatenate([rng.uniform(1e-6, 2.0, 3000) * rng.choice([-1.0, 1.0], 3000),
                        10.0 ** rng.uniform(-8, 0, 1000), [0.0, 1.0, -1.0, 0.5]])
    p3 = np.empty_like(x)
    p4 = np.empty_like(x)
    crlib.cr_pow34(x.ctypes.data, x.size, p3.ctypes.data, p4.ctypes.data)
    assert np.array_equal(p3, np.array([float(Fraction(float(v)) ** 3) for v in x]))
    assert np.array_equal(p4, np.array([float(Fraction(float(v)) ** 4) for v in x]))


def test_numpy_agrees_most_of_the_time(crlib):
    """What this buys: NumPy's own sin / x**4 equal the exactly rounded value in > 99 % / ~95 % of the cases, so an
    exactly rounded device implementation reproduces the reference's roundings that often ((x*x)*(x*x): 50 %)."""
    rng = np.random.default_rng(13)
    r = np.deg2rad(rng.uniform(0.0, 90.0, 4000))
    s = np.empty_like(r)
    c = np.empty_like(r)
    crlib.cr_sincos(r.ctypes.data, r.size, s.ctypes.data, c.ctypes.data)
    assert (s == np.sin(r)).mean() > 0.98 and (c == np.cos(r)).mean() > 0.98
    x = rng.uniform(0.01, 1.0, 4000)
    p3 = np.empty_like(x)
    p4 = np.empty_like(x)
    crlib.cr_pow34(x.ctypes.data, x.size, p3.ctypes.data, p4.ctypes.data)
    assert (p4 == x ** 4).mean() > 0.9 and (p3 == x ** 3).mean() > 0.9
    assert math.isclose(float(p4[0]), float(x[0]) ** 4, rel_tol=1e-15)
