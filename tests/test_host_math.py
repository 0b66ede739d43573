"""Arithmetic identities the device code relies on, checked on the CPU against the reference's
double-precision Color semantics (color.h:25-45: truncate toward zero, keep the low byte)."""
import numpy as np
import pytest


def q8(x):
    """Color(char) fed with a double on x86-64: cvttsd2si to 32 bits, low byte."""
    x = np.asarray(x, dtype=np.float64)
    ok = (x > -2147483649.0) & (x < 2147483648.0)
    return np.where(ok, np.trunc(np.where(ok, x, 0.0)).astype(np.int64) & 0xff, 0)


def test_running_mean_over_lights_is_integer_arithmetic():
    """shader.h:93: clr_sum / (k+1) * k + clr / (k+1) per 8-bit channel equals
    ((c_sum * M >> 16) * k + (c * M >> 16)) & 255 with M = 65536 / (k+1) + 1 (0 beyond 256 lights)."""
    c = np.arange(256, dtype=np.int64)
    cs, cc = np.meshgrid(c, c, indexing="ij")
    for k in list(range(0, 300)) + [511, 1000]:
        d = k + 1
        want = (q8(q8(cs.astype(np.float64) / d).astype(np.float64) * k) + q8(cc.astype(np.float64) / d)) & 0xff
        M = (65536 // d + 1) if d <= 256 else 0
        got = (((cs * M) >> 16) * k + ((cc * M) >> 16)) & 0xff
        assert np.array_equal(got, want), k


def _two_prod(a, b):
    """a * b = p + e exactly (Dekker / Veltkamp; stands in for the device's FMA)."""
    p = a * b
    def split(x):
        c = 134217729.0 * x
        hi = c - (c - x)
        return hi, x - hi
    ah, al = split(a); bh, bl = split(b)
    e = ((ah * bh - p) + ah * bl + al * bh) + al * bl
    return p, e


def _dd_mul(a, b):
    p, e = _two_prod(a[0], b[0])
    e = e + a[0] * b[1]  # the device fuses these two updates; the difference is far below the kept precision
    e = e + a[1] * b[0]
    h = p + e
    return h, e - (h - p)


def _pow25(x):
    x1 = (x, np.zeros_like(x))
    x2 = _dd_mul(x1, x1); x4 = _dd_mul(x2, x2); x8 = _dd_mul(x4, x4); x16 = _dd_mul(x8, x8)
    return _dd_mul(_dd_mul(x16, x8), x1)[0]


def test_power_25_in_double_double_is_correctly_rounded():
    """CalcSpecular (shader.h:366) calls pow(s, 25); the device computes it with five squarings and two
    products in double-double arithmetic.  Against exact rational arithmetic the result is the correctly
    rounded power, and it gives the same 8-bit channels as the C library's pow on a dense sample."""
    from fractions import Fraction
    rng = np.random.default_rng(355)
    s = np.concatenate([rng.random(20000), 1.0 - np.logspace(-16, -1, 2000), np.linspace(0.001, 1.0, 2000)])
    got = _pow25(s)
    exact = np.array([float(Fraction(float(v)) ** 25) for v in s])  # float(Fraction) rounds to nearest
    assert np.array_equal(got, exact)
    big = np.concatenate([rng.random(2_000_000), np.linspace(0.0, 1.0, 100001)])
    chain, ref = _pow25(big), np.power(big, 25.0)
    assert np.abs(chain - ref).max() <= np.spacing(ref).max()
    for c in (255.0, 221.0, 200.0, 100.0, 16.0, 1.0):
        assert np.array_equal(q8(c * chain), q8(c * ref))
        assert np.array_equal(q8(c * (1 - chain)), q8(c * (1 - ref)))


def _exact_sincos(x):
    """sin and cos of a double from their Taylor series in rational arithmetic, rounded once."""
    from fractions import Fraction
    X = Fraction(x)
    sv = cv = Fraction(0)
    term, n = Fraction(1), 0
    while n < 30 or abs(term) > Fraction(1, 1 << 260):
        if n % 2 == 0:
            cv += term if (n // 2) % 2 == 0 else -term
        else:
            sv += term if (n // 2) % 2 == 0 else -term
        n += 1
        term = term * X / n
    return float(sv), float(cv)


def test_rotation_sine_and_cosine_are_correctly_rounded(oracle):
    """shading2 / shading3 rotate the direction set with the C library's sin and cos (shader.h:115,119), whose
    last bit differs between libraries; the gather path is defined on the correctly rounded functions.  The
    device evaluates them in double-double arithmetic, and rt355_sincos is that same code compiled for the
    host: against exact rational arithmetic every result is the correctly rounded one, it equals the
    checker's (binary128, rounded once), and the C library of this host differs in the last bit now and
    then."""
    import math
    from raytracer_amd import api
    rng = np.random.default_rng(356)
    xs = list(rng.random(1500) * 2 * 3.1415927)
    xs += [k * math.pi / 2 for k in range(5)] + [math.nextafter(k * math.pi / 2, 9.0) for k in range(5)]
    xs += [math.nextafter(k * math.pi / 2, 0.0) for k in range(1, 5)] + [0.0, 1e-300, 1e-9, 2 * 3.1415927]
    off = 0
    for x in xs:
        got = api.sincos(float(x))
        assert got == _exact_sincos(float(x)), x
        assert got == oracle.sincos(float(x), rounded=True), x
        off += got != oracle.sincos(float(x))
    assert off <= len(xs) // 50
    with pytest.raises(api.Rt355Error):
        api.sincos(8.5)
