// mhd_math.hpp -- small FP64 elementary functions for the secular-function
// kernels.  Written for straight-line VALU code on gfx950: no tables, no
// divisions, no slow paths inside the argument ranges the kernels use.  The
// same source compiles for the host (tests/test_host_logic.py builds tests/host/math_check.cpp
// with g++) so accuracy is checked against long-double libm without a GPU.
//
//   sincos_pio2(x, s, c)  |x| <= 1e5: three-constant Cody-Waite reduction by
//                         pi/2 (exact products through fma) + the classic
//                         degree-13 / degree-14 minimax kernels; 1.5 ulp at most
//                         (measured over |x| in [1e-2, 1e5], tests/host/math_check.cpp).
//   exp_neg(y)            exp(-y) for 0 <= y <= 700: ln2 hi/lo reduction +
//                         degree-13 polynomial + ldexp; < 1 ulp.
//   rsqrt_pos(x)          1/sqrt(x), x > 0 normal: hardware seed (device) or a
//                         single-precision one (host) refined by one step of third
//                         order; < 1 ulp.
//
// Every multiply-add is written out (fma_) and the compiler contracts nothing by itself: the
// routines are inlined into several evaluations of the period equation, which must round alike
// wherever they stand -- a search decides by the sign of a value that can be all rounding.
#pragma once
#include <cmath>
#if defined(__clang__)
#pragma clang fp contract(off)
#endif

#if defined(__HIPCC__)
#define MHD_HD __host__ __device__ __forceinline__
#else
#define MHD_HD inline
#endif

namespace mhd {

MHD_HD double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

// a * b + k for a constant k of a polynomial.  On the device the instruction is written out:
// left to itself the compiler turns every Horner step into a register copy of the constant
// followed by a two-address multiply-add, a third of the instructions of sincos and exp.
// SK: the constant in scalar registers (set up by scalar instructions inside the loop, which
// cost the vector pipe nothing) instead of vector registers; the better choice where vector
// registers are short, as in the evaluation of two abscissae per lane.
template <bool SK>
MHD_HD double fma_k(double a, double b, double k)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MHD_POLY_PLAIN)
    double d;
    if (SK)
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(k));
    else
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(k));
    return d;
#else
    return __builtin_fma(a, b, k);
#endif
}

// The constants of the Horner steps that are written out as instructions (they must sit in
// registers), for a caller that sets them up once before a loop instead of in every pass: with
// K = nullptr the routines use their own literals, and the arithmetic is the same either way.
struct MathK {
    double s1, s2, s3, s4, c1, c2, c3, c4;
    double e3, e4, e5, e6, e7, e8, e9, e10, e11, e12;
    // the other constants that are no inline operands of the instruction set
    double inv_pio2, p1, p2, p3, s5, s6, c5, c6, log2e, ln2_hi, ln2_lo, e13;
};
MHD_HD MathK math_k()
{
    return {-1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04,
            2.75573137070700676789e-06, 4.16666666666666019037e-02, -1.38888888888741095749e-03,
            2.48015872894767294178e-05, -2.75573143513906633035e-07,
            1.0 / 6.0, 1.0 / 24.0, 1.0 / 120.0, 1.0 / 720.0, 1.0 / 5040.0, 1.0 / 40320.0,
            1.0 / 362880.0, 1.0 / 3628800.0, 1.0 / 39916800.0, 1.0 / 479001600.0,
            6.36619772367581382433e-01, 1.57079632679489655800e+00, 6.12323399573676603587e-17,
            -1.49738490485916983294e-33, -2.50507602534068634195e-08, 1.58969099521155010221e-10,
            2.08757232129817482790e-09, -1.13596475577881948265e-11, 1.44269504088896338700e+00,
            6.93147180369123816490e-01, 1.90821492927058770002e-10, 1.0 / 6227020800.0};
}

template <bool SK = false>
MHD_HD void sincos_pio2(double x, double& s, double& c, const MathK* K = nullptr)
{
    // pi/2 = P1 + P2 + P3 (each the nearest double of the running remainder)
    const double INV_PIO2 = 6.36619772367581382433e-01;
    const double P1 = 1.57079632679489655800e+00;
    const double P2 = 6.12323399573676603587e-17;
    const double P3 = -1.49738490485916983294e-33;
    const double kf = __builtin_rint(x * (K ? K->inv_pio2 : INV_PIO2));
    double r = fma_(-kf, K ? K->p1 : P1, x);
    r = fma_(-kf, K ? K->p2 : P2, r);
    r = fma_(-kf, K ? K->p3 : P3, r);
    const int q = (int)kf;
    const double z = r * r;
    // sin kernel on [-pi/4, pi/4]
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double ps = fma_(z, K ? K->s6 : S6, K ? K->s5 : S5);
    ps = fma_k<SK>(z, ps, K ? K->s4 : S4);
    ps = fma_k<SK>(z, ps, K ? K->s3 : S3);
    ps = fma_k<SK>(z, ps, K ? K->s2 : S2);
    ps = fma_k<SK>(z, ps, K ? K->s1 : S1);
    const double sr = fma_(r * z, ps, r);
    // cos kernel
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double pc = fma_(z, K ? K->c6 : C6, K ? K->c5 : C5);
    pc = fma_k<SK>(z, pc, K ? K->c4 : C4);
    pc = fma_k<SK>(z, pc, K ? K->c3 : C3);
    pc = fma_k<SK>(z, pc, K ? K->c2 : C2);
    pc = fma_k<SK>(z, pc, K ? K->c1 : C1);
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    const double cr = w + fma_(z * z, pc, (1.0 - w) - hz);
    // quadrant
    const double ss = (q & 1) ? cr : sr;
    const double cs = (q & 1) ? sr : cr;
    s = (q & 2) ? -ss : ss;
    c = ((q + 1) & 2) ? -cs : cs;
}

template <bool SK = false>
MHD_HD double exp_neg(double y, const MathK* K = nullptr)
{
    // exp(-y), 0 <= y <= 700
    const double LOG2E = 1.44269504088896338700e+00;
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double x = -y;
    const double nf = __builtin_rint(x * (K ? K->log2e : LOG2E));
    double r = fma_(-nf, K ? K->ln2_hi : LN2_HI, x);
    r = fma_(-nf, K ? K->ln2_lo : LN2_LO, r);
    // e^r, |r| <= ln2/2: Taylor to degree 13 (remainder < 5e-18)
    double p = K ? K->e13 : 1.0 / 6227020800.0;
    p = fma_k<SK>(p, r, K ? K->e12 : 1.0 / 479001600.0);
    p = fma_k<SK>(p, r, K ? K->e11 : 1.0 / 39916800.0);
    p = fma_k<SK>(p, r, K ? K->e10 : 1.0 / 3628800.0);
    p = fma_k<SK>(p, r, K ? K->e9 : 1.0 / 362880.0);
    p = fma_k<SK>(p, r, K ? K->e8 : 1.0 / 40320.0);
    p = fma_k<SK>(p, r, K ? K->e7 : 1.0 / 5040.0);
    p = fma_k<SK>(p, r, K ? K->e6 : 1.0 / 720.0);
    p = fma_k<SK>(p, r, K ? K->e5 : 1.0 / 120.0);
    p = fma_k<SK>(p, r, K ? K->e4 : 1.0 / 24.0);
    p = fma_k<SK>(p, r, K ? K->e3 : 1.0 / 6.0);
    p = fma_(p, r, 0.5);
    p = fma_(p, r, 1.0);
    p = fma_(p, r, 1.0);
    return __builtin_ldexp(p, (int)nf);
}

MHD_HD double rsqrt_pos(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const double y = __builtin_amdgcn_rsq(x);
#else
    const double y = (double)(1.0f / __builtin_sqrtf((float)x));  // a seed of similar quality
#endif
    // one step of third order: with e = 1 - x y^2 (|e| < 2^-22 for either seed)
    // 1/sqrt(x) = y (1 - e)^(-1/2) = y (1 + e/2 + 3 e^2/8 + O(e^3))
    const double e = fma_(-(x * y), y, 1.0);
    const double q = e * fma_(e, 0.375, 0.5);
    return fma_(y, q, y);
}

}  // namespace mhd
