"""rm_math*.h against glibc's libm on the host (the library CPython reaches in the reference):
functions marked EXACT in DESIGN.md must agree bit-for-bit on the argument ranges of the path
(SURVEY.md Appendix C)."""
import ctypes

import numpy as np
import pytest

from conftest import build_native

N = 400_000


@pytest.fixture(scope="module")
def m():
    L = ctypes.CDLL(build_native("math_check"))
    return L


def _call2(L, name, x, y):
    dp = ctypes.POINTER(ctypes.c_double)
    out = np.empty_like(x)
    getattr(L, name)(x.ctypes.data_as(dp), y.ctypes.data_as(dp), ctypes.c_size_t(len(x)), out.ctypes.data_as(dp))
    return out


def _call1(L, name, x):
    dp = ctypes.POINTER(ctypes.c_double)
    out = np.empty_like(x)
    getattr(L, name)(x.ctypes.data_as(dp), ctypes.c_size_t(len(x)), out.ctypes.data_as(dp))
    return out


def _bits_equal(a, b):
    return int((a.view(np.uint64) != b.view(np.uint64)).sum())


@pytest.mark.parametrize("y", [0.5, 2.0, 7.0, 8.0])
def test_pow_exact(m, y):
    rng = np.random.default_rng(int(y * 10))
    hi = 4.0 if y >= 7 else 1e6
    sets = [rng.uniform(0, hi, N), np.exp(rng.uniform(-40 if y > 1 else -700, np.log(hi), N)), rng.uniform(0.99, 1.01, N)]
    v = rng.uniform(-4, 4, (N, 3))
    s = (v * v).sum(1)
    sets.append(np.minimum(np.sqrt(s), 4.0) if y >= 7 else s)
    sp = [0.0, 1.0, np.inf, 4.0, np.nan, 1.0000000000000002, 0.9999999999999999, 2.2250738585072014e-308]
    if y == 0.5:
        sp += [5e-324, 1e-310]          # subnormal bases (results stay normal only for y < 1)
    sets.append(np.array(sp))
    for x in sets:
        yy = np.full(len(x), y)
        assert _bits_equal(_call2(m, "rmc_pow", x, yy), _call2(m, "rml_pow", x, yy)) == 0


def test_pow_half_guard_only_passes_roots_that_pow_returns(m):
    """rm_pow_half_guard (the fast form of every `x ** 0.5` of the algebraic scenes, and of team wavefronts): wherever the
    guard accepts the rounded square root, libm's pow(x, 0.5) returns exactly that value; it refuses ~1/32 of spread-out
    arguments, every argument on which pow and sqrt differ (those lie within 0.009 ulp of a rounding midpoint, the guard
    band is 1/64 ulp), roots that are powers of two, and the edges of the exponent range."""
    rng = np.random.default_rng(5)
    dp = ctypes.POINTER(ctypes.c_double)
    m.rmc_pow_half_guard.argtypes = [dp, ctypes.c_size_t, dp, ctypes.POINTER(ctypes.c_ubyte)]
    v = rng.uniform(-4, 4, (4 * N, 3))
    sets = [rng.uniform(0, 40.0, 8 * N), np.exp(rng.uniform(-60, 8, 8 * N)), (v * v).sum(1), rng.uniform(0.999, 1.001, N),
            np.ldexp(rng.uniform(0.5, 1.0, 4 * N), rng.integers(-40, 12, 4 * N))]
    differ = refused = total = 0
    for k, x in enumerate(sets):
        root, safe = np.empty_like(x), np.empty(len(x), np.uint8)
        m.rmc_pow_half_guard(x.ctypes.data_as(dp), len(x), root.ctypes.data_as(dp), safe.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)))
        ref = _call2(m, "rml_pow", x, np.full(len(x), 0.5))
        ne = root.view(np.uint64) != ref.view(np.uint64)
        assert not (ne & (safe != 0)).any()
        differ += int(ne.sum())
        if k in (0, 2):                                    # spread arguments inside the guard's magnitude range
            refused += int((safe == 0).sum()); total += len(x)
    assert differ > 1000                                   # the cases the guard exists for were exercised
    assert 0.025 < refused / total < 0.04
    # how far from a rounding midpoint the arguments on which pow and sqrt differ lie: well inside the 1/32-ulp band
    x = rng.uniform(1.0, 4.0, 16 * N)
    root, safe = np.empty_like(x), np.empty(len(x), np.uint8)
    m.rmc_pow_half_guard(x.ctypes.data_as(dp), len(x), root.ctypes.data_as(dp), safe.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)))
    ne = root.view(np.uint64) != _call2(m, "rml_pow", x, np.full(len(x), 0.5)).view(np.uint64)
    xs, rs = x[ne].astype(np.longdouble), root[ne].astype(np.longdouble)
    dist = 0.5 - np.abs(xs - rs * rs) / (2 * rs * np.longdouble(2.0) ** -52)      # roots in [1, 2): ulp = 2^-52
    assert ne.sum() > 100 and float(dist.max()) < 0.0095          # e_pow.c's own bound: 0.009 + the log term
    # never accepted: zero, subnormal / huge arguments, non-finite, exact powers of four (root = a power of two)
    x = np.array([5e-324, 1e-310, 1e-300, 1e300, np.inf, np.nan, 4.0, 1.0, 0.25, 16.0, 2.0 ** -40, 2.0 ** -61, 2.0 ** 61, -0.0])
    root, safe = np.empty_like(x), np.empty(len(x), np.uint8)
    m.rmc_pow_half_guard(x.ctypes.data_as(dp), len(x), root.ctypes.data_as(dp), safe.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)))
    assert not safe.any()
    # +0 is accepted: pow(+0, 0.5) = +0 (the length of the zero vector inside a box's slabs)
    x = np.array([0.0])
    m.rmc_pow_half_guard(x.ctypes.data_as(dp), 1, root.ctypes.data_as(dp), safe.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)))
    assert safe[0] and root[0] == 0.0 and not np.signbit(root[0])


def _near_midpoint_roots(rng, n):
    """Arguments whose square root lies within ~0.03 ulp of a rounding midpoint: x = RN((s + (0.5 + t) ulp)^2) for random s
    and small t -- the worst case for a guard that must refuse everything pow might round the other way."""
    s = rng.uniform(1.0, 2.0, n).astype(np.longdouble)
    t = rng.uniform(-0.03, 0.03, n).astype(np.longdouble)
    m = s + (np.longdouble(0.5) + t) * np.longdouble(2.0) ** -52
    return (m * m).astype(np.float64) * np.ldexp(1.0, 2 * rng.integers(-12, 12, n))


def test_guards_on_adversarial_arguments(m):
    """The guard on arguments constructed NEAR rounding midpoints (where pow may and does round the other way) and on a
    large spread sample: an accepted value always equals libm's pow, bit for bit."""
    rng = np.random.default_rng(11)
    dp, ub = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_ubyte)
    m.rmc_pow_half_guard.argtypes = [dp, ctypes.c_size_t, dp, ub]
    wrong_h = 0
    for rep in range(6):
        x = np.concatenate([_near_midpoint_roots(rng, 8 * N), rng.uniform(0, 100.0, 4 * N), np.exp(rng.uniform(-30, 30, 4 * N))])
        root, safe = np.empty_like(x), np.empty(len(x), np.uint8)
        m.rmc_pow_half_guard(x.ctypes.data_as(dp), len(x), root.ctypes.data_as(dp), safe.ctypes.data_as(ub))
        ref = _call2(m, "rml_pow", x, np.full(len(x), 0.5))
        ne = root.view(np.uint64) != ref.view(np.uint64)
        assert not (ne & (safe != 0)).any()
        wrong_h += int(ne.sum())
        if rep == 0:
            assert 0.02 < (safe[8 * N:12 * N] == 0).mean() < 0.045
    assert wrong_h > 1000                              # the sample did contain arguments on which pow differs


def test_pow2_shares_the_log_exactly(m):
    rng = np.random.default_rng(8)
    dp = ctypes.POINTER(ctypes.c_double)
    m.rmc_pow2.argtypes = [dp, ctypes.c_size_t, ctypes.c_double, ctypes.c_double, dp, dp]
    for x in [rng.uniform(0, 4.0, N), np.exp(rng.uniform(-30, np.log(4.0), N)), np.array([0.0, 1.0, 4.0, 1e-3])]:
        a, b = np.empty_like(x), np.empty_like(x)
        m.rmc_pow2(x.ctypes.data_as(dp), len(x), 7.0, 8.0, a.ctypes.data_as(dp), b.ctypes.data_as(dp))
        assert _bits_equal(a, _call2(m, "rml_pow", x, np.full(len(x), 7.0))) == 0
        assert _bits_equal(b, _call2(m, "rml_pow", x, np.full(len(x), 8.0))) == 0


def _ranges_sincos(rng):
    return [rng.uniform(-8 * np.pi, 8 * np.pi, N), rng.uniform(-0.2, 0.2, N), rng.uniform(-3, 3, N),
            rng.uniform(-400, 400, N), rng.uniform(-1e8, 1e8, N), np.exp(rng.uniform(-40, 3, N)) * rng.choice([-1, 1], N),
            np.array([0.0, -0.0, 0.126, -0.126, 0.855469, 2.426265, np.pi, -np.pi, np.pi / 2, 1e-9, 105414300.0])]


def test_sin_exact(m):
    for x in _ranges_sincos(np.random.default_rng(3)):
        assert _bits_equal(_call1(m, "rmc_sin", x), _call1(m, "rml_sin", x)) == 0


def test_cos_exact(m):
    for x in _ranges_sincos(np.random.default_rng(4)):
        assert _bits_equal(_call1(m, "rmc_cos", x), _call1(m, "rml_cos", x)) == 0


def test_log_exact(m):
    rng = np.random.default_rng(5)
    for x in [rng.uniform(1e-12, 7e4, N), np.exp(rng.uniform(-740, 700, N)), rng.uniform(0.9, 1.1, N),
              rng.uniform(0.93, 1.07, N), np.array([1.0, 1e-12, 4.0, 5e-324, 1e-310, np.inf, 0.9375, 1.0644])]:
        assert _bits_equal(_call1(m, "rmc_log", x), _call1(m, "rml_log", x)) == 0


def test_acos_exact(m):
    rng = np.random.default_rng(6)
    sets = [rng.uniform(-1, 1, N), rng.uniform(0.96, 1.0, N), -rng.uniform(0.96, 1.0, N), rng.uniform(-0.13, 0.13, N),
            1.0 - np.exp(rng.uniform(-40, -3, N)), -1.0 + np.exp(rng.uniform(-40, -3, N)),
            np.exp(rng.uniform(-60, 0, N)) * rng.choice([-1, 1], N),
            np.array([0.0, -0.0, 1.0, -1.0, 0.125, 0.5, 0.75, 0.921875, 0.953125, 0.96875, -0.125, -0.5, -0.75,
                      -0.921875, -0.953125, -0.96875, 0.25, -0.25, 1e-17, 2.7e-17, 0.9999999999999999])]
    # z.z / r of unit-ish vectors, the actual call-site distribution (catalog.py:277)
    v = rng.normal(size=(N, 3))
    sets.append(np.clip(v[:, 2] / np.sqrt((v * v).sum(1)), -1, 1))
    for x in sets:
        assert _bits_equal(_call1(m, "rmc_acos", x), _call1(m, "rml_acos", x)) == 0


def test_atan2_exact(m):
    rng = np.random.default_rng(7)
    pairs = [(rng.uniform(-4, 4, N), rng.uniform(-4, 4, N)),
             (rng.normal(size=N), rng.normal(size=N)),
             (np.exp(rng.uniform(-30, 30, N)) * rng.choice([-1, 1], N), np.exp(rng.uniform(-30, 30, N)) * rng.choice([-1, 1], N)),
             (np.exp(rng.uniform(-700, 700, N)) * rng.choice([-1, 1], N), np.exp(rng.uniform(-700, 700, N)) * rng.choice([-1, 1], N)),
             (rng.uniform(-1, 1, N) * 1e-3, rng.uniform(-4, 4, N)), (rng.uniform(-4, 4, N), rng.uniform(-1, 1, N) * 1e-3)]
    sp = np.array([0.0, -0.0, 1.0, -1.0, 2.0, -2.0, 1e-300, -1e-300, 1e300, 0.0625, 16.0])
    yy, xx = np.meshgrid(sp, sp)
    pairs.append((yy.ravel().copy(), xx.ravel().copy()))
    for y, x in pairs:
        got, ref = _call2(m, "rmc_atan2", y, x), _call2(m, "rml_atan2", y, x)
        claimed = (np.abs(ref) >= 2.2250738585072014e-308) | (ref == 0.0)      # subnormal quotients are unclaimed
        assert _bits_equal(got[claimed], ref[claimed]) == 0
