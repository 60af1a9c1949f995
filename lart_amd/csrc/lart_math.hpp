// lart_math.hpp -- fp64 elementary functions sized for the transport kernel's arguments.
//
// The ROCm device library's tan / sincos / log carry full-range argument reduction and double-double
// evaluation (211 / 191 / 113 instructions on gfx950); the photon loop only ever calls them on bounded
// arguments: tan on (-pi/2, pi/2), sin/cos on 2*pi*u with u in (0,1), log on (0, 1].  These versions use the
// classical Sun fdlibm kernels (minimax polynomials on [-pi/4, pi/4], log via s = f/(2+f)) with explicit fma,
// so host (tests/test_fastmath_cpu.py, g++) and device produce the same bits.  Measured max error against
// libm: tan <= 2.6 ulp, sin/cos <= 1 ulp (abs 2e-16 near zeros of the other), log <= 1 ulp, exp_neg <= 1 ulp.
//
// LART_HD is empty for the host test build.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#ifndef LART_HD
#define LART_HD __host__ __device__ inline
#endif

namespace lartm {

// K(c): a coefficient held in a scalar register pair.  The compiler otherwise materialises every coefficient with two
// v_mov_b32 in front of a two-address v_fmac_f64 (3 vector instructions per polynomial term, in loops that run a few
// thousand times per packet); pinned to SGPRs the term is s_mov x2 on the scalar unit + one v_fma_f64.
#if defined(__HIPCC__)
__host__ __device__ inline double K(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
#ifdef LART_K_HOISTABLE
    asm("" : "+s"(v));
#else
    asm volatile("" : "+s"(v));
#endif
#endif
    return v;
}
#else
static inline double K(double v) { return v; }
#endif

// sin(x), |x| <= pi/4
LART_HD double ksin(double x) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = x * x, v = z * x;
    double r = fma(z, fma(z, fma(z, fma(z, K(S6), K(S5)), K(S4)), K(S3)), K(S2));
    return fma(v, fma(z, r, K(S1)), x);
}
// cos(x), |x| <= pi/4
LART_HD double kcos(double x) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = x * x, w = z * z;
    double r = fma(w * w, fma(z, fma(z, K(C6), K(C5)), K(C4)), z * fma(z, fma(z, K(C3), K(C2)), K(C1)));
    double hz = 0.5 * z, a = 1.0 - hz;
    return a + (((1.0 - a) - hz) + z * r);
}
// tan(t) for |t| <= pi/2 (the range of pi*(u - 1/2) and of values between two arctangents)
LART_HD double tan_halfpi(double t) {
    const double pio4 = 7.85398163397448278999e-01, pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
    double ax = fabs(t);
    bool big = ax > pio4;
    double r = big ? (pio2_hi - ax) + pio2_lo : ax;       // pio2_hi - ax is exact on [pi/4, pi/2]
    double s = ksin(r), c = kcos(r);
    double num = big ? c : s, den = big ? s : c;           // one division for both halves of the range
    return copysign(num / den, t);
}
// sin and cos of 2*pi*u, u in [0, 1]
LART_HD void sincos_2pi(double u, double *sn, double *cs) {
    const double pi = 3.14159265358979311600e+00;
    double x = 2.0 * u;                                    // [0, 2]
    double qf = rint(2.0 * x);                             // nearest quarter turn 0..4
    double r = fma(qf, -0.5, x);                           // exact, |r| <= 1/4
    double th = pi * r;
    double s = ksin(th), c = kcos(th);
    int q = (int)qf & 3;
    double s1 = (q & 1) ? c : s, c1 = (q & 1) ? s : c;
    *sn = (q & 2) ? -s1 : s1;
    *cs = ((q + 1) & 2) ? -c1 : c1;
}
// natural log of a positive normal number
LART_HD double log_pos(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01, sqrt_half = 7.07106781186547524401e-01;
    int e;
    double m = frexp(x, &e);                               // [0.5, 1)
    bool lo = m < sqrt_half;
    m = lo ? 2.0 * m : m; e = lo ? e - 1 : e;              // [sqrt(1/2), sqrt(2))
    double f = m - 1.0, k = (double)e;
    double s = f / (2.0 + f), z = s * s, w = z * z;
    double t1 = w * fma(w, fma(w, K(Lg6), K(Lg4)), K(Lg2));
    double t2 = z * fma(w, fma(w, fma(w, K(Lg7), K(Lg5)), K(Lg3)), K(Lg1));
    double R = t2 + t1, hfsq = 0.5 * f * f;
    return fma(k, K(ln2_hi), -((hfsq - fma(s, hfsq + R, k * K(ln2_lo))) - f));
}

// exp(x) for x <= 0 (the acceptance test exp(-v^2) and the escape fractions exp(-tau)); 0 below the double range.
// x = n ln2 + r, |r| <= ln2/2, degree-11 near-minimax polynomial (Chebyshev fit, |err| < 3.2e-18), then ldexp.
LART_HD double exp_neg(double x) {
    const double log2e = 1.44269504088896338700e+00, ln2_hi = 6.93147180559945286227e-01, ln2_lo = 2.31904681384629955842e-17;
    x = fmax(x, -1100.0);                                  // keeps n in int range; exp(-1100) underflows to 0 anyway
    double n = rint(x * K(log2e));
    double r = fma(-n, K(ln2_hi), x);
    r = fma(-n, K(ln2_lo), r);
    double p = fma(r, K(2.5110037605963777712e-8), K(2.76326396390410297493e-7));
    p = fma(r, p, K(2.7557240918578969823e-6));
    p = fma(r, p, K(2.4801485482328492419e-5));
    p = fma(r, p, K(1.98412698900471137067e-4));
    p = fma(r, p, K(1.38888889523147746524e-3));
    p = fma(r, p, K(8.33333333331960061091e-3));
    p = fma(r, p, K(4.16666666664880954954e-2));
    p = fma(r, p, K(1.66666666666666808057e-1));
    p = fma(r, p, K(5.00000000000001838553e-1));
    p = fma(r, p, 1.0);
    p = fma(r, p, 1.0);
    return ldexp(p, (int)n);
}

}  // namespace lartm
