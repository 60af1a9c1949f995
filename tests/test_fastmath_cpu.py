"""ulp bounds of the bounded-range elementary functions of lart_amd/csrc/lart_math.hpp (the transport kernel's tan / log / exp /
sincos), host build of the same header (tests/fastmath_shim.cpp, g++ -ffp-contract=off: only the explicit fma()s fuse, as on the
device) against mpmath at 60 digits.  The bounds are the ones DESIGN.md section 3 quotes."""
import ctypes as C
import os
import subprocess
import numpy as np
import pytest

mp = pytest.importorskip("mpmath")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fm(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("fastmath") / "libfastmath.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", os.path.join(ROOT, "tests", "fastmath_shim.cpp"), "-o", so])
    L = C.CDLL(so)
    for f in (L.fm_tan, L.fm_log, L.fm_exp_neg):
        f.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    L.fm_sincos_2pi.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    return L


def call(f, x):
    x = np.ascontiguousarray(x, dtype=np.float64); y = np.zeros_like(x)
    f(len(x), x.ctypes.data, y.ctypes.data); return y


def ulp_err(got, exact):
    """|got - exact| in units of the last place of the correctly rounded result"""
    mp.mp.dps = 60
    out = np.zeros(len(got))
    for i, (g, e) in enumerate(zip(got, exact)):
        r = float(e)
        u = np.spacing(abs(r)) if r != 0.0 else np.spacing(0.0)
        out[i] = float(abs(mp.mpf(float(g)) - e) / mp.mpf(float(u)))
    return out


def test_tan_on_the_open_half_period(fm):
    mp.mp.dps = 60
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-np.pi / 2, np.pi / 2, 6000), np.pi * (rng.uniform(0, 1, 3000) - 0.5) * (1 - 1e-9),      # rand_voigt: tan(pi (u - 1/2))
                        [1e-300, -1e-12, 0.7853981633974483, -0.7853981633974483, 1.5707963267948966 - 1e-9, -1.5707963267948966 + 1e-9]])
    e = ulp_err(call(fm.fm_tan, x), [mp.tan(mp.mpf(float(v))) for v in x])
    assert e.max() <= 2.6, e.max()


def test_log_on_the_unit_interval(fm):
    mp.mp.dps = 60
    rng = np.random.default_rng(2)
    u = np.concatenate([rng.uniform(0, 1, 6000), (np.arange(1, 2001) + 0.5) / 2.0 ** 52, 1.0 - (np.arange(1, 1001) + 0.5) / 2.0 ** 52, [1.0, 0.5, 2.0 ** -1000]])
    e = ulp_err(call(fm.fm_log, u), [mp.log(mp.mpf(float(v))) for v in u])
    assert e.max() <= 1.0, e.max()
    # a few arguments above 1 occur too (1 - u wgt1 rounding, sqrt arguments): the kernel is the plain fdlibm log there
    v = rng.uniform(1.0, 1e6, 2000)
    assert ulp_err(call(fm.fm_log, v), [mp.log(mp.mpf(float(t))) for t in v]).max() <= 1.0


def test_exp_of_non_positive_numbers(fm):
    mp.mp.dps = 60
    rng = np.random.default_rng(3)
    x = -np.concatenate([rng.uniform(0, 30, 5000), rng.uniform(0, 745.2, 3000), 10.0 ** rng.uniform(-12, 0, 1000), [0.0, 745.13, 708.4]])
    got = call(fm.fm_exp_neg, x)
    big = x > -708.0                                        # normal results: <= 1 ulp; below, the result is subnormal (absolute error < 1 subnormal ulp)
    assert ulp_err(got[big], [mp.exp(mp.mpf(float(v))) for v in x[big]]).max() <= 1.0
    sub = ~big
    assert np.all(np.abs(got[sub] - np.exp(x[sub])) <= 2 * np.spacing(0.0) + 4 * np.spacing(np.exp(x[sub])))
    assert call(fm.fm_exp_neg, np.array([-745.2, -800.0, -1e300]))[0] == 0.0 and got[x == 0.0][0] == 1.0
    # the tau >= 745.2 cut of the peel rays relies on this: exp(-745.2) is exactly 0 in f64
    assert np.all(call(fm.fm_exp_neg, np.array([-745.2, -800.0, -1e300])) == 0.0)


def test_sincos_of_two_pi_u(fm):
    mp.mp.dps = 60
    rng = np.random.default_rng(4)
    u = np.concatenate([rng.uniform(0, 1, 8000), (np.arange(1, 1001) + 0.5) / 2.0 ** 52, [0.25, 0.5, 0.75, 0.125, 1.0 - 2.0 ** -53]])
    s = np.zeros_like(u); c = np.zeros_like(u)
    fm.fm_sincos_2pi(len(u), u.ctypes.data, s.ctypes.data, c.ctypes.data)
    twopi = 2 * mp.pi
    es = [mp.sin(twopi * mp.mpf(float(v))) for v in u]; ec = [mp.cos(twopi * mp.mpf(float(v))) for v in u]
    # relative to the larger of |sin|, |cos| ... near a zero of one of them the error is absolute (2 pi u is rounded once): <= 1 ulp of 1
    for got, ex in ((s, es), (c, ec)):
        ex_f = np.array([float(t) for t in ex])
        err = np.array([float(abs(mp.mpf(float(g)) - t)) for g, t in zip(got, ex)])
        assert np.all(err <= np.maximum(np.spacing(np.abs(ex_f)), 2.3e-16)), float((err / np.maximum(np.spacing(np.abs(ex_f)), 2.3e-16)).max())
    assert np.all(np.abs(s * s + c * c - 1.0) < 5e-16)
