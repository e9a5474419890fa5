// mhd_device.hpp -- device-side building blocks for gfx950 (wave64).
//
//   * LayerTable: one chain's layered half-space, staged in LDS
//   * secular_rayleigh / secular_love: the period equation Delta(c)
//   * WaveSearch: the sequential bracketing + interval-halving root search of
//     disba's surf96 (reached from src/inversion/model_params.py:201-206 of the
//     reference), executed by one 64-lane wave: the control flow is the
//     sequential algorithm's, every secular value it consumes has really been
//     evaluated at exactly the floating-point abscissa the sequential algorithm
//     would use, and the 64 lanes evaluate the points it is about to need
//     (the next scan points, or a 6-level tree of interval midpoints) at once.
//
// Floating-point discipline: everything that fixes an abscissa (start value,
// scan steps, midpoints, stopping test) is compiled without fused multiply-add
// so that the abscissae are bit-identical to a plain IEEE evaluation; only the
// inside of the secular function, of which just the sign is used, may contract.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mcmc_amd.h"
#include "mhd_math.hpp"

namespace mhd {

constexpr int MAXL = MHD_MAX_LAYERS + 1;  // rows, last = half-space
constexpr int MAXNP = 2 * MHD_MAX_LAYERS + 1;
constexpr int MAXP = MHD_MAX_PERIODS;
constexpr double TWOPI = 2.0 * 3.14159265358979323846;

struct LayerTable {
    double d[MAXL];      // thickness (km), last row unused
    double a[MAXL];      // vp
    double b[MAXL];      // vs
    double rho[MAXL];    // density
    double inv_a[MAXL];  // 1/vp
    double inv_b[MAXL];  // 1/vs
    double inv_rho[MAXL];
    double twob2[MAXL];  // 2*vs^2
    double cc;           // search floor: 0.9 * half-space Rayleigh velocity of the slowest layer
    double betmx;        // largest vs
    int L;
    // The tail of the last period's halvings that solve_period_wave inferred from an
    // interpolated root instead of evaluating it (see there): abscissae, the signs the
    // inference gave them, the period's angular frequency.  Evaluated with the next period's
    // first round, or by flush_deferred().
    int pend_n;
    int pend_want[8];
    double pend_x[8];
    double pend_omega;
};

// Out-of-line routines share one register budget: VGPRs + AGPRs <= 256, so that two waves fit
// on a SIMD (512 registers per lane) and hide each other's FP64 latency.
#ifndef MHD_WAVES_PER_SIMD
#define MHD_WAVES_PER_SIMD 2
#endif
#define MHD_REGS __attribute__((amdgpu_waves_per_eu(MHD_WAVES_PER_SIMD, MHD_WAVES_PER_SIMD)))
#ifndef MHD_NOINLINE
#define MHD_NOINLINE __attribute__((noinline))
#endif
// the one-abscissa evaluation of the period equation: one out-of-line body
#ifndef MHD_EVAL_FN
#define MHD_EVAL_FN __forceinline__
#endif
// Polynomial constants kept in registers across the layers of an evaluation (see
// rayleigh_compound, secular_rayleigh2); 0: set up again in every layer.
// MHD_PIN_K, one abscissa per lane: 4 all in scalar registers; 1 those of the written-out Horner
// steps in vector registers, 2 the others in vector registers too, 3 the others in scalar
// registers.  (Measured: 3 makes the evaluation 3 % faster than 4 and the chains 1.5 %, but it
// takes 80 vector registers from the callers, which then spill 75 instead of 19 of theirs: 3.3
// instead of 0.2 GB of scratch traffic per 250 steps of 1024 chains.)
// MHD_PIN_K2, two abscissae per lane (all constants in scalar registers there): 1 those of the
// written-out Horner steps, 2 all.
#ifndef MHD_PIN_K
#define MHD_PIN_K 4
#endif
#ifndef MHD_PIN_K2
#define MHD_PIN_K2 2
#endif
#ifndef MHD_PIN_LOVE
#define MHD_PIN_LOVE 1  // the same in the Love-wave equation (with MHD_PIN_K 4)
#endif

// Scratch of one wave in dynamic LDS: the layer table of the model it works on, then the state
// of the predicted search (mhd_search2.hpp).  A kernel that calls the forward model is launched
// with WAVE_LDS_BYTES of dynamic LDS per wave of its workgroups; the out-of-line routines reach
// the instance of the calling wave through g_T / g_PS, with LDS instructions and no pointer
// argument.
#ifndef MHD_WAVE_LDS_BYTES
#define MHD_WAVE_LDS_BYTES (9 * 1024)
#endif
constexpr int WAVE_LDS_BYTES = MHD_WAVE_LDS_BYTES;
constexpr int WAVE_LDS_SEARCH_OFFSET = (sizeof(LayerTable) + 15) / 16 * 16;
extern __shared__ __align__(16) unsigned char mhd_wave_lds[];
#define g_T (*reinterpret_cast<LayerTable*>(mhd_wave_lds + (threadIdx.x >> 6) * WAVE_LDS_BYTES))

// Orders the LDS accesses of ONE wave (its lanes run in lock step, LDS operations of a wave
// complete in order): a compiler fence, no hardware barrier.  Waves of a workgroup that work
// on different candidates must not meet in __syncthreads().
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int sgni(double x) { return (x > 0.0) - (x < 0.0); }

// A condition that is the same in every lane, made visible as such: the branch on it is a
// scalar branch instead of an exec-masked region.
__device__ __forceinline__ bool uni(bool c) { return __ballot(c) != 0ull; }

__device__ __forceinline__ double readlane_f64(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// A double that is the same in every lane, moved to scalar registers.
__device__ __forceinline__ double uniform_value(double x)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// ---------------------------------------------------------------------------
// Secular functions.  Only the sign of the result steers the search.
// ---------------------------------------------------------------------------
// Every multiply-add of the period equations is written out (fma_) and the compiler contracts
// nothing by itself.  The equations are inlined in several places -- one abscissa per lane, two
// abscissae per lane, the ellipticity -- and a search decides by the SIGN of a value that can be
// all rounding next to a root: only if every copy rounds alike do the searches, the candidates
// and the launches of a chain agree on such a sign whichever copy they went through.
#pragma clang fp contract(off)

// One wave type (P or S) in one layer.  Input: k, xk = omega/v, dpth.
// Output: cs (cos or scaled cosh), over_r (sin/r or scaled sinh/r), times_r
// (-r*sin or r*scaled sinh), em = exp(-r*dpth) when evanescent (1 otherwise),
// ex = r*dpth when evanescent (0 otherwise).
__device__ __forceinline__ void osc_or_evan(double k, double xk, double dpth, double& cs,
                                            double& over_r, double& times_r, double& em,
                                            double& ex, const MathK* K = nullptr)
{
    const double s2 = (k + xk) * fabs(k - xk);
    const double ir = rsqrt_pos(fmax(s2, 1.0e-300));
    const double r = s2 * ir;
    const double arg = r * dpth;
    if (k < xk) {
        double s, c;
        if (arg < 1.0e5) sincos_pio2<MHD_PIN_K == 4>(arg, s, c, K); else sincos(arg, &s, &c);
        cs = c;
        over_r = s * ir;
        times_r = -r * s;
        em = 1.0;
        ex = 0.0;
    } else if (k == xk) {
        cs = 1.0;
        over_r = dpth;
        times_r = 0.0;
        em = 1.0;
        ex = 0.0;
    } else {
        em = exp_neg<MHD_PIN_K == 4>(fmin(arg, 700.0), K);
        const double fac = (arg < 16.0) ? em * em : 0.0;
        cs = (1.0 + fac) * 0.5;
        const double sh = (1.0 - fac) * 0.5;
        over_r = sh * ir;
        times_r = r * sh;
        ex = arg;
    }
}

// a * b + c of the period equations' algebra: fused.  (MHD_ALGEBRA_UNFUSED: two roundings, as
// the reference's arithmetic has them.  Over 163 840 random models the same 8 curves differ
// from the CPU restatement's either way -- the disagreements come from lattice points next to
// roots of ill-conditioned equations, not from the fusing -- and the steps are 5 % slower.)
#ifdef MHD_ALGEBRA_UNFUSED
__device__ __forceinline__ double mad_(double a, double b, double c) { return a * b + c; }
#else
__device__ __forceinline__ double mad_(double a, double b, double c) { return fma_(a, b, c); }
#endif

// The compound vector of the half-space (the start of the Dunkin recursion).
__device__ __forceinline__ void compound_halfspace(double k, double k2, double xka, double xkb,
                                                   double gammk, double rho1, double& e0,
                                                   double& e1, double& e2, double& e3, double& e4)
{
    const double ra = sqrt((k + xka) * fabs(k - xka));
    const double rb = sqrt((k + xkb) * fabs(k - xkb));
    const double gam = gammk * k2;
    const double gamm1 = gam - 1.0;
    const double rarb = ra * rb;
    const double gkrr = gammk * rarb;
    e0 = (rho1 * rho1) * mad_(-gam, gkrr, gamm1 * gamm1);
    e1 = -rho1 * ra;
    e2 = rho1 * (gamm1 - gkrr);
    e3 = rho1 * rb;
    e4 = mad_(-ra, rb, k2);
}

// One layer of the Dunkin recursion: e <- e * ca(layer), then rescaled by an exact power of two
// (sign-preserving, no rounding) whose exponent is added to sexp.  (cosp, w, x, ep) and
// (cosq, y, z, eq) are the outputs of osc_or_evan for the P and the S wave of the layer.
__device__ __forceinline__ void compound_layer(double k2, double gammk, double rho1, double irho,
                                               double cosp, double w, double x, double ep,
                                               double cosq, double y, double z, double eq,
                                               double& e0, double& e1, double& e2, double& e3,
                                               double& e4, int& sexp)
{
    // exp(-(pex + sex)); the reference sets it to 0 from an exponent of 60 on, where it is
    // below 1e-26 of the cosine product it is subtracted from: the same number in double
    const double a0 = ep * eq;
    const double gam = gammk * k2;
    const double cpcq = cosp * cosq, cpy = cosp * y, cpz = cosp * z, cqw = cosq * w,
                 cqx = cosq * x, xy = x * y, xz = x * z, wy = w * y, wz = w * z;
    const double gamm1 = gam - 1.0;
    const double twgm1 = gam + gamm1, gmgmk = gam * gammk, gmgm1 = gam * gamm1,
                 gm1sq = gamm1 * gamm1, rho2 = rho1 * rho1, a0pq = a0 - cpcq, t = -2.0 * k2,
                 k2wy = k2 * wy;
    // ca00 = cpcq - 2 gmgm1 a0pq - gmgmk xz - k2 gm1sq wy
    double ca00 = mad_(-2.0 * gmgm1, a0pq, cpcq);
    ca00 = mad_(-gmgmk, xz, ca00);
    ca00 = mad_(-gm1sq, k2wy, ca00);
    const double ca01 = mad_(k2, cpy, -cqx) * irho;
    // ca02 = -(twgm1 a0pq + gammk xz + k2 gamm1 wy) / rho
    double s02 = twgm1 * a0pq;
    s02 = mad_(gammk, xz, s02);
    s02 = mad_(gamm1, k2wy, s02);
    const double ca02 = -s02 * irho;
    const double ca03 = mad_(-k2, cqw, cpz) * irho;
    // ca04 = -(2 k2 a0pq + xz + k2 k2 wy) / rho^2
    double s04 = mad_(2.0 * k2, a0pq, xz);
    s04 = mad_(k2, k2wy, s04);
    const double ca04 = -s04 * (irho * irho);
    const double ca10 = mad_(gmgmk, cpz, -(gm1sq * cqw)) * rho1;
    const double ca11 = cpcq;
    const double ca12 = mad_(gammk, cpz, -(gamm1 * cqw));
    const double ca13 = -wz;
    const double ca30 = mad_(gm1sq, cpy, -(gmgmk * cqx)) * rho1;
    const double ca31 = -xy;
    const double ca32 = mad_(gamm1, cpy, -(gammk * cqx));
    // ca40 = -(2 gmgmk gm1sq a0pq + gmgmk^2 xz + gm1sq^2 wy) rho^2
    double s40 = (2.0 * gmgmk * gm1sq) * a0pq;
    s40 = mad_(gmgmk * gmgmk, xz, s40);
    s40 = mad_(gm1sq * gm1sq, wy, s40);
    const double ca40 = -s40 * rho2;
    // ca42 = -(gammk gamm1 twgm1 a0pq + gam gammk^2 xz + gamm1 gm1sq wy) rho
    double s42 = (gammk * gamm1 * twgm1) * a0pq;
    s42 = mad_(gmgmk * gammk, xz, s42);
    s42 = mad_(gamm1 * gm1sq, wy, s42);
    const double ca42 = -s42 * rho1;
    const double ca20 = t * ca42, ca21 = t * ca32, ca22 = mad_(2.0, cpcq - ca00, a0),
                 ca23 = t * ca12, ca24 = t * ca02;
    // ee_i = sum_j e_j ca_ji with ca14=ca03, ca33=ca11, ca34=ca01, ca41=ca30,
    // ca43=ca10, ca44=ca00
    const double n0 = mad_(e4, ca40, mad_(e3, ca30, mad_(e2, ca20, mad_(e1, ca10, e0 * ca00))));
    const double n1 = mad_(e4, ca30, mad_(e3, ca31, mad_(e2, ca21, mad_(e1, ca11, e0 * ca01))));
    const double n2 = mad_(e4, ca42, mad_(e3, ca32, mad_(e2, ca22, mad_(e1, ca12, e0 * ca02))));
    const double n3 = mad_(e4, ca10, mad_(e3, ca11, mad_(e2, ca23, mad_(e1, ca13, e0 * ca03))));
    const double n4 = mad_(e4, ca00, mad_(e3, ca01, mad_(e2, ca24, mad_(e1, ca03, e0 * ca04))));
    const double t1 = fmax(fmax(fmax(fabs(n0), fabs(n1)), fmax(fabs(n2), fabs(n3))), fabs(n4));
    const int ex = (t1 < 1.0e-40) ? 0 : __builtin_amdgcn_frexp_exp(t1);
    sexp += ex;
    e0 = ldexp(n0, -ex);
    e1 = ldexp(n1, -ex);
    e2 = ldexp(n2, -ex);
    e3 = ldexp(n3, -ex);
    e4 = ldexp(n4, -ex);
}

// Rayleigh period equation (Dunkin compound matrix).  Returns the first
// component of the propagated, rescaled vector; *scale_exp receives the sum of
// the binary exponents taken out, so that value * 2^(*scale_exp) is the
// unscaled (smooth) secular function up to one positive constant.  rest, if given, receives
// the other four components at the surface (on the same scale).
__device__ __forceinline__ double rayleigh_compound(const LayerTable& T, double omega,
                                                    double inv_om2, double c, int* scale_exp,
                                                    double* rest)
{
    const int L = T.L;
    const double k = omega / c;
    const double k2 = k * k;
    double e0, e1, e2, e3, e4;
    {
        const int h = L - 1;
        compound_halfspace(k, k2, omega * T.inv_a[h], omega * T.inv_b[h], T.twob2[h] * inv_om2,
                           T.rho[h], e0, e1, e2, e3, e4);
    }
    int sexp = 0;
    // the polynomial constants in vector registers for all layers (the compiler is told not to
    // hoist out of loops, see mcmc_amd/build.py, and would set them up again for every layer)
#if MHD_PIN_K == 4
    MathK KK = math_k();
    asm volatile("" : "+s"(KK.s1), "+s"(KK.s2), "+s"(KK.s3), "+s"(KK.s4), "+s"(KK.c1), "+s"(KK.c2),
                      "+s"(KK.c3), "+s"(KK.c4));
    asm volatile("" : "+s"(KK.e3), "+s"(KK.e4), "+s"(KK.e5), "+s"(KK.e6), "+s"(KK.e7), "+s"(KK.e8),
                      "+s"(KK.e9), "+s"(KK.e10), "+s"(KK.e11), "+s"(KK.e12));
    asm volatile("" : "+s"(KK.inv_pio2), "+s"(KK.p1), "+s"(KK.p2), "+s"(KK.p3), "+s"(KK.s5), "+s"(KK.s6),
                      "+s"(KK.c5), "+s"(KK.c6), "+s"(KK.log2e), "+s"(KK.ln2_hi), "+s"(KK.ln2_lo),
                      "+s"(KK.e13));
    const MathK* const K = &KK;
#elif MHD_PIN_K > 0
    MathK KK = math_k();
    asm volatile("" : "+v"(KK.s1), "+v"(KK.s2), "+v"(KK.s3), "+v"(KK.s4), "+v"(KK.c1), "+v"(KK.c2),
                      "+v"(KK.c3), "+v"(KK.c4));
    asm volatile("" : "+v"(KK.e3), "+v"(KK.e4), "+v"(KK.e5), "+v"(KK.e6), "+v"(KK.e7), "+v"(KK.e8),
                      "+v"(KK.e9), "+v"(KK.e10), "+v"(KK.e11), "+v"(KK.e12));
#if MHD_PIN_K == 2
    asm volatile("" : "+v"(KK.inv_pio2), "+v"(KK.p1), "+v"(KK.p2), "+v"(KK.p3), "+v"(KK.s5), "+v"(KK.s6),
                      "+v"(KK.c5), "+v"(KK.c6), "+v"(KK.log2e), "+v"(KK.ln2_hi), "+v"(KK.ln2_lo),
                      "+v"(KK.e13));
#elif MHD_PIN_K == 3
    asm volatile("" : "+s"(KK.inv_pio2), "+s"(KK.p1), "+s"(KK.p2), "+s"(KK.p3), "+s"(KK.s5), "+s"(KK.s6),
                      "+s"(KK.c5), "+s"(KK.c6), "+s"(KK.log2e), "+s"(KK.ln2_hi), "+s"(KK.ln2_lo),
                      "+s"(KK.e13));
#endif
    const MathK* const K = &KK;
#else
    const MathK* const K = nullptr;
#endif

#pragma nounroll
    for (int m = L - 2; m >= 0; --m) {
        const double xka = omega * T.inv_a[m];
        const double xkb = omega * T.inv_b[m];
        const double dpth = T.d[m];
        double cosp, w, x, ep, pex, cosq, y, z, eq, sex;
        osc_or_evan(k, xka, dpth, cosp, w, x, ep, pex, K);
        osc_or_evan(k, xkb, dpth, cosq, y, z, eq, sex, K);
        (void)pex;
        (void)sex;
        compound_layer(k2, T.twob2[m] * inv_om2, T.rho[m], T.inv_rho[m], cosp, w, x, ep, cosq, y,
                       z, eq, e0, e1, e2, e3, e4, sexp);
    }
    if (scale_exp) *scale_exp = sexp;
    if (rest) {
        rest[0] = e1;
        rest[1] = e2;
        rest[2] = e3;
        rest[3] = e4;
    }
    return e0;
}

__device__ __forceinline__ double secular_rayleigh(const LayerTable& T, double omega,
                                                   double inv_om2, double c, int* scale_exp)
{
    return rayleigh_compound(T, omega, inv_om2, c, scale_exp, nullptr);
}

__device__ __forceinline__ double secular_love(const LayerTable& T, double omega, double c,
                                               int* scale_exp)
{
    const int L = T.L;
    const double k = omega / c;
    int h = L - 1;
    double xkb = omega * T.inv_b[h];
    double rb = sqrt((k + xkb) * fabs(k - xkb));
    double e1 = T.rho[h] * rb;
    double e2 = T.inv_b[h] * T.inv_b[h];
    int sexp = 0;
    // (the polynomial constants in scalar registers for all layers, see rayleigh_compound)
#if MHD_PIN_K == 4
    MathK KK = math_k();
    asm volatile("" : "+s"(KK.s1), "+s"(KK.s2), "+s"(KK.s3), "+s"(KK.s4), "+s"(KK.c1), "+s"(KK.c2),
                      "+s"(KK.c3), "+s"(KK.c4));
    asm volatile("" : "+s"(KK.e3), "+s"(KK.e4), "+s"(KK.e5), "+s"(KK.e6), "+s"(KK.e7), "+s"(KK.e8),
                      "+s"(KK.e9), "+s"(KK.e10), "+s"(KK.e11), "+s"(KK.e12));
    asm volatile("" : "+s"(KK.inv_pio2), "+s"(KK.p1), "+s"(KK.p2), "+s"(KK.p3), "+s"(KK.s5), "+s"(KK.s6),
                      "+s"(KK.c5), "+s"(KK.c6), "+s"(KK.log2e), "+s"(KK.ln2_hi), "+s"(KK.ln2_lo),
                      "+s"(KK.e13));
    const MathK* const K = MHD_PIN_LOVE ? &KK : nullptr;
#else
    const MathK* const K = nullptr;
#endif
#pragma nounroll
    for (int m = L - 2; m >= 0; --m) {
        const double xmu = T.rho[m] * T.b[m] * T.b[m];
        xkb = omega * T.inv_b[m];
        const double dpth = T.d[m];
        double cosq, y, z, eq, sex;
        osc_or_evan(k, xkb, dpth, cosq, y, z, eq, sex, K);
        const double e10 = mad_(e2 * xmu, z, e1 * cosq);
        const double e20 = mad_(e2, cosq, e1 * y / xmu);
        const double xnor = fmax(fabs(e10), fabs(e20));
        const int ex = (xnor < 1.0e-40) ? 0 : __builtin_amdgcn_frexp_exp(xnor);
        sexp += ex;
        e1 = ldexp(e10, -ex);
        e2 = ldexp(e20, -ex);
    }
    if (scale_exp) *scale_exp = sexp;
    return e1;
}

struct SecVal {
    double m;  // value after rescaling
    int e;     // binary exponent taken out: m * 2^e is smooth in c
};

// The ONE out-of-line body of the period equation (keeps the kernels within the
// instruction cache: every search stage calls this instead of inlining a copy).
__device__ MHD_EVAL_FN SecVal eval_secular(int wave, double omega, double c)
{
    SecVal v;
    omega = fmax(omega, 1.0e-4);
    if (wave == MHD_LOVE)
        v.m = secular_love(g_T, omega, c, &v.e);
    else
        v.m = secular_rayleigh(g_T, omega, 1.0 / (omega * omega), c, &v.e);
    return v;
}

// Rayleigh-wave ellipticity ur/uz at the free surface for phase velocity c.  Numbering the
// motion-stress components 1 = ur, 2 = uz, 3 = normal stress, 4 = shear stress, the compound
// vector holds the minors e0 = Y34, e1 = Y24, e2 = Y14 / k = -Y23 / k, e3 = Y13, e4 = Y12 of the
// two solutions that decay in the half-space.  The combination free of shear stress at the
// surface has ur : uz = Y14 : Y24, the one free of normal stress Y13 : Y23; on a root of the
// period equation (Y34 = 0) they are one solution and the ratios agree.  The better conditioned
// one is taken: Y13 / Y23 next to a pole (uz = 0, |ur/uz| > 1), Y14 / Y24 next to a zero.
// Positive for retrograde motion (a homogeneous Poisson half-space gives +0.6812).  The
// reference only names this datum (README.md:4, the empty EllipticityParams of
// src/inversion/model_params.py:212-213): there are no reference values for it.
__device__ MHD_NOINLINE double eval_ellipticity(double omega, double c)
{
    omega = fmax(omega, 1.0e-4);
    double r[4];
    int se;
    rayleigh_compound(g_T, omega, 1.0 / (omega * omega), c, &se, r);
    const double k = omega / c;
    return (fabs(r[0]) >= fabs(r[2])) ? -k * r[1] / r[0] : r[2] / (k * r[1]);
}

// Two abscissae in one lane: the same operations as secular_rayleigh for each, written side by
// side so that the two dependency chains fill each other's FP64 latency (one wave issues a
// dependent FP64 instruction every ~8 cycles, the pipe takes one every 4).  The branches on the
// kind of wave (oscillating / evanescent) are taken for the pair: both sides alike -- the rule
// when the two abscissae are neighbours of one search -- run interleaved, a mixed pair runs the
// one-abscissa routine twice.
__device__ __forceinline__ void osc_or_evan2(const double k[2], const double xk[2], double dpth,
                                             double cs[2], double over_r[2], double times_r[2],
                                             double em[2], double ex[2], const MathK* K = nullptr)
{
    const bool osc0 = k[0] < xk[0], osc1 = k[1] < xk[1];
    const bool ev0 = k[0] > xk[0], ev1 = k[1] > xk[1];
    if (osc0 && osc1) {
        double s2[2], ir[2], r[2], arg[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            s2[u] = (k[u] + xk[u]) * fabs(k[u] - xk[u]);
            ir[u] = rsqrt_pos(fmax(s2[u], 1.0e-300));
            r[u] = s2[u] * ir[u];
            arg[u] = r[u] * dpth;
        }
        if (arg[0] < 1.0e5 && arg[1] < 1.0e5) {
            double sn[2], cn[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) sincos_pio2<true>(arg[u], sn[u], cn[u], K);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                cs[u] = cn[u];
                over_r[u] = sn[u] * ir[u];
                times_r[u] = -r[u] * sn[u];
                em[u] = 1.0;
                ex[u] = 0.0;
            }
            return;
        }
    } else if (ev0 && ev1) {
        double s2[2], ir[2], r[2], arg[2], e1[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            s2[u] = (k[u] + xk[u]) * fabs(k[u] - xk[u]);
            ir[u] = rsqrt_pos(fmax(s2[u], 1.0e-300));
            r[u] = s2[u] * ir[u];
            arg[u] = r[u] * dpth;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) e1[u] = exp_neg<true>(fmin(arg[u], 700.0), K);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            em[u] = e1[u];
            const double fac = (arg[u] < 16.0) ? e1[u] * e1[u] : 0.0;
            cs[u] = (1.0 + fac) * 0.5;
            const double sh = (1.0 - fac) * 0.5;
            over_r[u] = sh * ir[u];
            times_r[u] = r[u] * sh;
            ex[u] = arg[u];
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
        osc_or_evan(k[u], xk[u], dpth, cs[u], over_r[u], times_r[u], em[u], ex[u], K);
}

__device__ __forceinline__ void secular_rayleigh2(const LayerTable& T, const double omega[2],
                                                  const double c[2], double val[2], int sexp[2])
{
    const int L = T.L;
    double k[2], k2[2], inv_om2[2], e0[2], e1[2], e2[2], e3[2], e4[2];
    {
        const int h = L - 1;
        const double ia = T.inv_a[h], ib = T.inv_b[h], tb2 = T.twob2[h], rho1 = T.rho[h];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            inv_om2[u] = 1.0 / (omega[u] * omega[u]);
            k[u] = omega[u] / c[u];
            k2[u] = k[u] * k[u];
            compound_halfspace(k[u], k2[u], omega[u] * ia, omega[u] * ib, tb2 * inv_om2[u], rho1,
                               e0[u], e1[u], e2[u], e3[u], e4[u]);
            sexp[u] = 0;
        }
    }
    // (the polynomial constants in scalar registers for all layers, see rayleigh_compound)
#if MHD_PIN_K2 > 0
    MathK KK = math_k();
    asm volatile("" : "+s"(KK.s1), "+s"(KK.s2), "+s"(KK.s3), "+s"(KK.s4), "+s"(KK.c1), "+s"(KK.c2),
                      "+s"(KK.c3), "+s"(KK.c4));
    asm volatile("" : "+s"(KK.e3), "+s"(KK.e4), "+s"(KK.e5), "+s"(KK.e6), "+s"(KK.e7), "+s"(KK.e8),
                      "+s"(KK.e9), "+s"(KK.e10), "+s"(KK.e11), "+s"(KK.e12));
#if MHD_PIN_K2 > 1
    asm volatile("" : "+s"(KK.inv_pio2), "+s"(KK.p1), "+s"(KK.p2), "+s"(KK.p3), "+s"(KK.s5), "+s"(KK.s6),
                      "+s"(KK.c5), "+s"(KK.c6), "+s"(KK.log2e), "+s"(KK.ln2_hi), "+s"(KK.ln2_lo),
                      "+s"(KK.e13));
#endif
    const MathK* const K = &KK;
#else
    const MathK* const K = nullptr;
#endif
#pragma nounroll
    for (int m = L - 2; m >= 0; --m) {
        const double ia = T.inv_a[m], ib = T.inv_b[m], tb2 = T.twob2[m];
        const double dpth = T.d[m], rho1 = T.rho[m], irho = T.inv_rho[m];
        double xka[2], xkb[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            xka[u] = omega[u] * ia;
            xkb[u] = omega[u] * ib;
        }
        double cosp[2], w[2], x[2], ep[2], pex[2], cosq[2], y[2], z[2], eq[2], sex[2];
        osc_or_evan2(k, xka, dpth, cosp, w, x, ep, pex, K);
        osc_or_evan2(k, xkb, dpth, cosq, y, z, eq, sex, K);
#pragma unroll
        for (int u = 0; u < 2; ++u)
            compound_layer(k2[u], tb2 * inv_om2[u], rho1, irho, cosp[u], w[u], x[u], ep[u],
                           cosq[u], y[u], z[u], eq[u], e0[u], e1[u], e2[u], e3[u], e4[u],
                           sexp[u]);
    }
    val[0] = e0[0];
    val[1] = e0[1];
}

struct SecVal2 {
    double m0, m1;
    int e0, e1;
};

// A pair of abscissae (Love waves: the one-abscissa routine twice, the Love period equation is a
// tenth of the Rayleigh one).  Inlined into the out-of-line verification pass, its one user in
// the library: as a function of its own it saved and restored 52 registers per call.
__device__ __forceinline__ SecVal2 eval_secular2_inline(int wave, double omega0, double c0,
                                                        double omega1, double c1)
{
    SecVal2 v;
    const double om[2] = {fmax(omega0, 1.0e-4), fmax(omega1, 1.0e-4)};
    const double cc[2] = {c0, c1};
    if (wave == MHD_LOVE) {
        v.m0 = secular_love(g_T, om[0], c0, &v.e0);
        v.m1 = secular_love(g_T, om[1], c1, &v.e1);
    } else {
        double val[2];
        int se[2];
        secular_rayleigh2(g_T, om, cc, val, se);
        v.m0 = val[0];
        v.m1 = val[1];
        v.e0 = se[0];
        v.e1 = se[1];
    }
    return v;
}

// The same out of line (the evaluation microbenchmark calls it).
__device__ MHD_NOINLINE SecVal2 eval_secular2(int wave, double omega0, double c0, double omega1,
                                              double c1)
{
    return eval_secular2_inline(wave, omega0, c0, omega1, c1);
}

__device__ __forceinline__ double secular(const LayerTable&, int wave, double omega, double c)
{
    return eval_secular(wave, omega, c).m;
}

// Value with its binary scale: value * 2^(*scale_exp) is a smooth function of c whose roots
// are the roots of the period equation (used by the root predictor, which interpolates).
__device__ __forceinline__ double secular_tracked(const LayerTable&, int wave, double omega,
                                                  double c, int* scale_exp)
{
    const SecVal v = eval_secular(wave, omega, c);
    *scale_exp = v.e;
    return v.m;
}

// ---------------------------------------------------------------------------
// Everything below fixes abscissae: no contraction.
// ---------------------------------------------------------------------------
#pragma clang fp contract(off)

// Half-space Rayleigh velocity, five Newton steps from 0.95*vs.
__device__ inline double halfspace_rayleigh(double a, double b)
{
    double c = 0.95 * b;
    for (int i = 0; i < 5; ++i) {
        double gamma = b / a;
        double kappa = c / b;
        double k2 = kappa * kappa;
        double gk2 = (gamma * kappa) * (gamma * kappa);
        double fac1 = sqrt(1.0 - gk2);
        double fac2 = sqrt(1.0 - k2);
        double fr = (2.0 - k2) * (2.0 - k2) - 4.0 * fac1 * fac2;
        double frp = -4.0 * (2.0 - k2) * kappa;
        frp += 4.0 * fac2 * gamma * gamma * kappa / fac1;
        frp += 4.0 * fac1 * kappa / fac2;
        frp /= b;
        c -= fr / frp;
    }
    return c;
}

// Build the LDS layer table from a parameter vector
// [n_layers interface depths, n_layers+1 vs]  (model_params.py:187-198).
// Called by every lane of the block; lanes < L fill one row each.
// vs_at: the slot of the first layer's vs (n_layers, or the padded maximum when the batch
// holds models of different layer counts).
__device__ inline void build_layer_table(LayerTable& T, const double* m, int n_layers, double vpvs,
                                         int lane, int vs_at)
{
    const int L = n_layers + 1;
    if (lane < L) {
        double dep = (lane < n_layers) ? m[lane] : 0.0;
        double prev = (lane > 0 && lane < n_layers) ? m[lane - 1] : 0.0;
        double d = (lane < n_layers) ? dep - prev : 0.0;
        double vs = m[vs_at + lane];
        double vp = vs * vpvs;
        double rho = (1741.0 * (double)sgni(vp) * pow(fabs(vp), 0.25)) / 1000.0;
        T.d[lane] = d;
        T.a[lane] = vp;
        T.b[lane] = vs;
        T.rho[lane] = rho;
        T.inv_a[lane] = 1.0 / vp;
        T.inv_b[lane] = 1.0 / vs;
        T.inv_rho[lane] = 1.0 / rho;
        T.twob2[lane] = 2.0 * vs * vs;
    }
    wave_sync();
    // extremal velocities and start value (uniform, every lane computes the same)
    double betmx = -1.0e20, betmn = 1.0e20;
    int jmn = 0, jsol = 1;
    for (int i = 0; i < L; ++i) {
        double bi = T.b[i], ai = T.a[i];
        if (bi > 0.01 && bi < betmn) {
            betmn = bi;
            jmn = i;
            jsol = 1;
        } else if (bi <= 0.01 && ai < betmn) {
            betmn = ai;
            jmn = i;
            jsol = 0;
        }
        if (bi > betmx) betmx = bi;
    }
    double cc = jsol ? halfspace_rayleigh(T.a[jmn], T.b[jmn]) : betmn;
    cc *= 0.9;
    if (lane == 0) {
        T.cc = cc;
        T.betmx = betmx;
        T.L = L;
        T.pend_n = 0;
    }
    wave_sync();
}

struct EvalCount {
    unsigned long long lit;   // evaluations the sequential algorithm performs
    unsigned long long exec;  // evaluations executed by lanes
    unsigned rounds;          // evaluation rounds of the wave (one evaluation per lane each; a
                              // replay pass counts as a round per four periods): what a curve
                              // costs in time, whatever the lanes' occupancy -- a count that
                              // does not depend on the clock (the launch schedule's estimates)
    // one round of `lanes` evaluations (two per lane: two rounds)
    __device__ __forceinline__ void round(unsigned long long lanes, unsigned n = 1)
    {
        exec += lanes;
        rounds += n;
    }
};

__device__ __forceinline__ int mask_bit_sign(unsigned long long neg, unsigned long long pos,
                                             int bit)
{
    return (int)((pos >> bit) & 1) - (int)((neg >> bit) & 1);
}

// One advance of the sequential scan: from (c1, idir) to the next abscissa c2.
__device__ __forceinline__ double scan_next(double& c1, int& idir, double clow, double dc)
{
    double c2;
    for (;;) {
        c2 = (idir > 0) ? c1 + dc : c1 - dc;
        if (c2 <= clow) {
            idir = +1;
            c1 = clow;
            continue;
        }
        break;
    }
    return c2;
}

// Bracket and refine one period (getsol), one wave.  All arguments are
// wave-uniform.  s1st carries sign(del1st) between periods.
// Returns 1 (root found), 0 (the search fails) or, when deferring, -1 (see DEFER).
//
// Round 1 evaluates the first 8 scan abscissae (lanes 0..7) and, for each of the 7
// cells between them, the 7 nodes of a 3-level midpoint tree (lanes 8..56): when the
// sign change lies in one of those cells -- the usual case -- three halvings are
// already done.  Further rounds evaluate 64 scan abscissae, or the 63 nodes of a
// 6-level tree.
//
// max_levels < 100 stops the halving after that many levels (a predictor's use): br[] then
// receives the bracket {c1, c2, sign at c1} and c1io its midpoint.
//
// DEFER: after the tree round that takes the bracket to 9 levels (1e-5 km/s wide, far below
// the spacing of neighbouring modes) the root is taken by linear interpolation of the tracked
// values at the bracket's ends, the remaining halvings (at most 7) follow from it without being
// evaluated, and their abscissae with the signs inferred are left in the layer table's
// pend_* fields: two rounds per period instead of three.  They are really evaluated in lanes
// 57..63 of the first round of the NEXT call (any call evaluates what is pending), or by
// flush_deferred(); a sign that is not the inferred one makes that call return -1 before it does
// anything else (flush_deferred: false), and the caller redoes the period the tail belongs to
// without defer and goes on from there.  Every abscissa of the sequential algorithm is thus
// evaluated before a curve is returned.
template <bool DEFER>
__device__ __forceinline__ int solve_period_impl(int wave, double t1, double& c1io, double clow,
                                                 double dc, double cm, double betmx, bool ifirst,
                                                 int& s1st, EvalCount& cnt, int max_levels,
                                                 double* br)
{
    constexpr bool defer = DEFER;
    LayerTable& T = g_T;  // (the calling wave's table: the deferred tail lives beside it)
    const int lane = threadIdx.x & 63;
    const double omega = TWOPI / t1;
    double c1 = c1io, c2 = c1io;
    int idir = +1, s1 = 0, s2 = 0;
    bool bracketed = false;
    int pre_levels = 0;               // halving levels already evaluated for the bracket
    unsigned long long pneg = 0, ppos = 0;
    int pbase = 0;                    // first lane of the bracket's 3-level tree
    // tracked values (mantissa, binary exponent) at the bracket's ends c1 and c2, and at the end
    // that was replaced last (c0), the third point of the interpolation
    double f1m = 0.0, f2m = 0.0, f0m = 0.0, c0 = 0.0;
    int f1e = 0, f2e = 0, f0e = 0;
    bool have0 = false;
    auto lane_value = [&](const SecVal& v, int l, double& m, int& e) {
        if (!DEFER) return;
        m = readlane_f64(v.m, l);
        e = __builtin_amdgcn_readlane(v.e, l);
    };

    // ---- first pass of the wave: 8 scan abscissae + 7 small trees (upward scan assumed) ----
    {
        // scan abscissa j (0..7) of the upward sequence
        double q1 = c1, pt = c1;
        int qd = +1;
        const int jmine = (lane < 8) ? lane : (lane - 8) / 7 + 1;  // right end of my cell
        double left = c1;
        for (int i = 0; i < jmine; ++i) {
            left = pt;
            pt = scan_next(q1, qd, clow, dc);
            // scan_next may move the left end to clow (reflection): the sequential code
            // brackets between its current c1 and the new abscissa
            left = q1;
            q1 = pt;
        }
        double x = pt;
        if (lane >= 8 && lane < 57) {
            const int node = (lane - 8) % 7 + 1;  // 1..7, heap order
            double a = left, b = pt;
            const int depth = 31 - __clz(node);
            for (int lvl = depth - 1; lvl >= 0; --lvl) {
                const double mid = 0.5 * (a + b);
                if ((node >> lvl) & 1) a = mid; else b = mid;
            }
            x = 0.5 * (a + b);
        }
        // the deferred tail of the period before, in the lanes the trees leave free
        const int np = DEFER ? __builtin_amdgcn_readfirstlane(T.pend_n) : 0;
        const bool pend = lane >= 57 && lane - 57 < np;
        double om = omega;
        int want = 0;
        if (pend) {
            x = T.pend_x[lane - 57];
            want = T.pend_want[lane - 57];
            om = T.pend_omega;
        }
        const bool act = lane < 57 || pend;
        SecVal v = {0.0, 0};
        if (act) v = eval_secular(wave, om, x);
        const unsigned long long neg = __ballot(act && v.m < 0.0), pos = __ballot(act && v.m > 0.0);
        cnt.round(57 + np);
        if (np) {
            const bool bad = pend && sgni(v.m) != want;
            wave_sync();
            if (lane == 0) T.pend_n = 0;
            wave_sync();
            if (__ballot(bad) != 0ull) return -1;
        }
        s1 = mask_bit_sign(neg, pos, 0);
        lane_value(v, 0, f1m, f1e);
        cnt.lit += 1;
        if (ifirst) s1st = s1;
        idir = (ifirst || s1st * s1 >= 0) ? +1 : -1;
        if (idir > 0) {
            for (int j = 1; j < 8; ++j) {
                c2 = scan_next(c1, idir, clow, dc);
                s2 = mask_bit_sign(neg, pos, j);
                lane_value(v, j, f2m, f2e);
                cnt.lit += 1;
                if (s1 != s2) {
                    bracketed = true;
                    pre_levels = 3;
                    pneg = neg;
                    ppos = pos;
                    pbase = 8 + (j - 1) * 7;
                    break;
                }
                c1 = c2;
                s1 = s2;
                f1m = f2m;
                f1e = f2e;
                if (uni(c1 < cm || c1 >= betmx + dc)) {
                    if (br) br[2] = (double)s1;
                    return 0;
                }
            }
        }
        if (bracketed) {
            // the three levels of the bracket's own tree, with this round's values still in
            // the lanes: c1 / c2 / s1 and the values at the ends
            double c3;
            int nctrl0 = 2, levels0 = 0;
            int node = 1;
            bool done0 = false;
            while (node < 8) {
                c3 = 0.5 * (c1 + c2);
                const int s3 = mask_bit_sign(pneg, ppos, pbase + node - 1);
                double m3;
                int e3;
                lane_value(v, pbase + node - 1, m3, e3);
                cnt.lit += 1;
                have0 = true;
                if (s3 * s1 < 0) {
                    c0 = c2;
                    f0m = f2m;
                    f0e = f2e;
                    c2 = c3;
                    f2m = m3;
                    f2e = e3;
                    node = 2 * node;
                } else {
                    c0 = c1;
                    f0m = f1m;
                    f0e = f1e;
                    c1 = c3;
                    s1 = s3;
                    f1m = m3;
                    f1e = e3;
                    node = 2 * node + 1;
                }
                if (uni(fabs(c1 - c2) <= 1.0e-6 * c1) || ++levels0 >= max_levels) {
                    done0 = true;
                    break;
                }
                ++nctrl0;
            }
            pre_levels = levels0;
            if (done0) {
                if (br) {
                    br[0] = c1;
                    br[1] = c2;
                    br[2] = (double)s1;
                    c3 = 0.5 * (c1 + c2);
                }
                c1io = c3;
                return uni(c3 <= betmx) ? 1 : 0;
            }
        }
    }

    // ---- scan on, 64 abscissae per round ----
    while (!bracketed) {
        double pc1 = c1, pt = c1;
        int pdir = idir;
        for (int i = 0; i <= lane; ++i) {
            pt = scan_next(pc1, pdir, clow, dc);
            pc1 = pt;
        }
        const SecVal v = eval_secular(wave, omega, pt);
        const unsigned long long neg = __ballot(v.m < 0.0), pos = __ballot(v.m > 0.0);
        cnt.round(64);
        for (int j = 0; j < 64; ++j) {
            c2 = scan_next(c1, idir, clow, dc);
            s2 = mask_bit_sign(neg, pos, j);
            lane_value(v, j, f2m, f2e);
            cnt.lit += 1;
            if (s1 != s2) {
                bracketed = true;
                break;
            }
            c1 = c2;
            s1 = s2;
            f1m = f2m;
            f1e = f2e;
            if (uni(c1 < cm || c1 >= betmx + dc)) {
                if (br) br[2] = (double)s1;
                return 0;
            }
        }
    }

    // ---- refine: pure interval halving ----
    double c3 = 0.0;
    int levels = pre_levels, nctrl = 2 + pre_levels;
    bool done = false;
    int tree_rounds = 0;
    while (!done) {
        double a = c1, b = c2;
        const int t = lane ? lane : 1;
        const int depth = 31 - __clz(t);
        for (int lvl = depth - 1; lvl >= 0; --lvl) {
            const double mid = 0.5 * (a + b);
            if ((t >> lvl) & 1) a = mid; else b = mid;
        }
        const double pt = 0.5 * (a + b);
        const SecVal v = eval_secular(wave, omega, pt);
        const unsigned long long neg = __ballot(v.m < 0.0), pos = __ballot(v.m > 0.0);
        cnt.round(64);
        ++tree_rounds;
        int node = 1;
        while (node < 64) {
            c3 = 0.5 * (c1 + c2);
            const int s3 = mask_bit_sign(neg, pos, node);
            double m3;
            int e3;
            lane_value(v, node, m3, e3);
            cnt.lit += 1;
            // the sequential loop applies at most 98 updates; the 99th midpoint is
            // evaluated and returned without being applied
            if (nctrl >= 100) {
                done = true;
                break;
            }
            have0 = true;
            if (s3 * s1 < 0) {
                c0 = c2;
                f0m = f2m;
                f0e = f2e;
                c2 = c3;
                f2m = m3;
                f2e = e3;
                node = 2 * node;
            } else {
                c0 = c1;
                f0m = f1m;
                f0e = f1e;
                c1 = c3;
                s1 = s3;
                f1m = m3;
                f1e = e3;
                node = 2 * node + 1;
            }
            if (uni(fabs(c1 - c2) <= 1.0e-6 * c1) || ++levels >= max_levels) {
                done = true;
                break;
            }
            ++nctrl;
        }
        if (done || !defer || max_levels < 1000 || tree_rounds > 1) continue;
        // ---- the rest of the halvings from the interpolated root, to be verified later ----
        if (!(f1m != 0.0 && f2m != 0.0 && (f1m < 0.0) != (f2m < 0.0))) continue;
        int de = f2e - f1e;
        de = de > 1000 ? 1000 : (de < -1000 ? -1000 : de);
        const double rho = ldexp(f2m / f1m, de);  // negative
        double root = c1 + (c2 - c1) / (1.0 - rho);
        if (have0) {
            // inverse quadratic interpolation through the three points (values on f1's scale);
            // taken if it lands inside the bracket
            int d0 = f0e - f1e;
            d0 = d0 > 1000 ? 1000 : (d0 < -1000 ? -1000 : d0);
            const double g1 = f1m, g2 = ldexp(f2m, de), g0 = ldexp(f0m, d0);
            const double q = c0 * (g1 * g2) / ((g0 - g1) * (g0 - g2)) +
                             c1 * (g0 * g2) / ((g1 - g0) * (g1 - g2)) +
                             c2 * (g0 * g1) / ((g2 - g0) * (g2 - g1));
            if (uni(q > fmin(c1, c2) && q < fmax(c1, c2))) root = q;
        }
#ifdef MHD_DEFER_SABOTAGE
        // test build: every third inference starts from a wrong root, so that the verification of
        // the tails and the searches it sends back are exercised (the results must not change)
        if ((cnt.lit % 3ull) == 0ull) root = c1 + 0.37 * (c2 - c1);
#endif
        const bool below1 = uni(c1 <= root);
        double d1 = c1, d2 = c2, d3 = c3, px = 0.0;
        int ds1 = s1, n = 0, dctrl = nctrl, pw = 0;
        bool settled = false;
        while (n < 7) {
            d3 = 0.5 * (d1 + d2);
            const bool same = uni((d3 <= root) == below1);
            const int s3 = same ? ds1 : -ds1;
            if (lane == 57 + n) {
                px = d3;
                pw = s3;
            }
            ++n;
            if (dctrl >= 100) {
                settled = true;
                break;
            }
            if (same) d1 = d3; else d2 = d3;
            if (uni(fabs(d1 - d2) <= 1.0e-6 * d1)) {
                settled = true;
                break;
            }
            ++dctrl;
        }
        if (!settled) continue;  // more than 7 halvings left: by another tree round
        // An answer above the ceiling ends the curve ("no root") at once, before anything could
        // verify the tail it was inferred from: that verdict is only given on evaluated halvings.
        // (An inferred answer below the ceiling whose true one lies above is caught like any
        // wrong tail: the period is searched again, and fails then.)
        if (!uni(d3 <= betmx)) continue;
        wave_sync();
        if (lane >= 57 && lane - 57 < n) {
            T.pend_x[lane - 57] = px;
            T.pend_want[lane - 57] = pw;
        }
        if (lane == 0) {
            T.pend_n = n;
            T.pend_omega = omega;
        }
        wave_sync();
        // (counted with the period they belong to; a caller that has to search the period again
        // takes the period's whole count back)
        cnt.lit += n;
        c3 = d3;
        done = true;
    }
    if (br) {
        br[0] = c1;
        br[1] = c2;
        br[2] = (double)s1;
        c3 = 0.5 * (c1 + c2);
    }
    c1io = c3;
    return uni(c3 <= betmx) ? 1 : 0;
}

__device__ MHD_NOINLINE bool solve_period_wave(const LayerTable&, int wave, double t1, double& c1io,
                                         double clow, double dc, double cm, double betmx,
                                         bool ifirst, int& s1st, EvalCount& cnt,
                                         int max_levels = 1000, double* br = nullptr)
{
    return solve_period_impl<false>(wave, t1, c1io, clow, dc, cm, betmx, ifirst, s1st, cnt,
                                    max_levels, br) > 0;
}

// The same search leaving a deferred tail (and verifying the one it finds): 1, 0 or -1 as above.
__device__ MHD_NOINLINE int solve_period_deferring(int wave, double t1, double& c1io, double clow,
                                             double dc, double cm, double betmx, bool ifirst,
                                             int& s1st, EvalCount& cnt)
{
    return solve_period_impl<true>(wave, t1, c1io, clow, dc, cm, betmx, ifirst, s1st, cnt, 1000,
                                   nullptr);
}

// Really evaluates the tail a deferring solve_period_wave left (if any).  false: a sign is not
// the inferred one, the period it belongs to has to be searched again without defer.
__device__ MHD_NOINLINE bool flush_deferred(int wave, EvalCount& cnt)
{
    LayerTable& T = g_T;
    const int lane = threadIdx.x & 63;
    const int np = __builtin_amdgcn_readfirstlane(T.pend_n);
    if (np == 0) return true;
    const bool pend = lane < np;
    SecVal v = {0.0, 0};
    int want = 0;
    if (pend) {
        want = T.pend_want[lane];
        v = eval_secular(wave, T.pend_omega, T.pend_x[lane]);
    }
    const bool bad = pend && sgni(v.m) != want;
    cnt.round(np);
    wave_sync();
    if (lane == 0) T.pend_n = 0;
    wave_sync();
    return __ballot(bad) == 0ull;
}

// Periods k0..k1-1 of the fundamental mode, one after the other, by the deferring search (two
// rounds per period); c[k0-1] (if k0 > 0) and s1st are exact inputs.  stop(k, ck), asked after
// every period, ends the run early.  Returns the index after the last period done (all of them
// verified: a tail that was inferred wrongly makes its period be searched again, every halving
// evaluated), or -1 when a period has no root.
template <class Stop>
__device__ __forceinline__ int sequential_run(int wave, const double* t, double tscale, int k0,
                                              int k1, double dc, double* c, int& s1st,
                                              EvalCount& cnt, Stop stop)
{
    const LayerTable& T = g_T;
    const double cc = T.cc, betmx = T.betmx, cm = T.cc;
    const double onea = 1.5;
    // the sequential algorithm's evaluation count before the search of the current period and of
    // the one before (a period that is searched again must not be counted twice)
    unsigned long long lit_cur = cnt.lit, lit_prev = cnt.lit;
    // period k again, exactly
    auto exact = [&](int k) -> bool {
        double cp = (k == 0) ? cc : c[k - 1] - onea * dc;
        if (!solve_period_wave(T, wave, t[k] / tscale, cp, (k == 0) ? cc : cm, dc, cm, betmx,
                               k == 0, s1st, cnt))
            return false;
        c[k] = cp;
        wave_sync();
        return true;
    };
    int k = k0;
    while (k < k1) {
        const bool ifirst = (k == 0);
        double c1 = ifirst ? cc : c[k - 1] - onea * dc;
        lit_prev = lit_cur;
        lit_cur = cnt.lit;
        const int ok = solve_period_deferring(wave, t[k] / tscale, c1, ifirst ? cc : cm, dc, cm,
                                              betmx, ifirst, s1st, cnt);
        if (ok < 0) {
            // the tail inferred for period k-1 was wrong: that period again, then this one from
            // its true start (unless the run ends with the true answer)
            cnt.lit = lit_prev;
            lit_cur = lit_prev;
            if (!exact(k - 1)) return -1;
            if (stop(k - 1, c[k - 1])) return k;
            continue;
        }
        if (!ok) return -1;
        c[k] = c1;
        wave_sync();
        ++k;
        if (k == k1 || stop(k - 1, c1)) {
            if (!flush_deferred(wave, cnt)) {
                cnt.lit = lit_cur;
                if (!exact(k - 1)) return -1;
                if (k < k1 && !stop(k - 1, c[k - 1])) {
                    lit_cur = cnt.lit;
                    continue;
                }
            }
            return k;
        }
    }
    return k;
}

// One phase-velocity curve (getc) for the table in T; t[] ascending periods,
// c[] in LDS.  Wave-uniform.  Returns MHD_ST_OK or MHD_ST_NOROOT.
// kstart > 0 resumes the fundamental mode at period kstart: c[0..kstart-1] and
// the sign of the first secular value of period 0 (s1st) are then inputs.
__device__ MHD_NOINLINE int phase_curve_wave(const LayerTable& T, int wave, const double* t, int P,
                                       double tscale, int mode, double dc, double* c,
                                       EvalCount& cnt, int kstart = 0, int s1st = 0)
{
    if (T.b[0] <= 0.0) return MHD_ST_NOROOT;  // water layer: not part of this workload
    const double betmx = T.betmx, cm = T.cc;
    dc = fabs(dc);
    const double one = 1.0e-2;
    for (int k = kstart; k < P; ++k) c[k] = 0.0;
    wave_sync();
    // the fundamental mode: every period starts 1.5 dc below the answer of the one before
    if (sequential_run(wave, t, tscale, kstart, P, dc, c, s1st, cnt,
                       [](int, double) { return false; }) < 0)
        return MHD_ST_NOROOT;
    // higher modes: above the mode below, from the period before
    for (int iq = 1; iq <= mode; ++iq) {
        for (int k = 0; k < P; ++k) {
            const bool ifirst = (k == 0);
            double c1, clow;
            if (ifirst) {
                if (uni(c[0] == 0.0)) break;  // the mode below has no root at all: neither has this
                c1 = c[0] + one * dc;
                clow = c1;
            } else {
                // c[k] is still the mode below: past its cut-off this mode has none either
                // (cut-off periods shrink with the mode number), and a search from c = 0 upward
                // would run through wavenumbers where the secular function is rounding noise
                if (uni(c[k] == 0.0)) {
                    for (int i = k; i < P; ++i) c[i] = 0.0;
                    break;
                }
                clow = c[k] + one * dc;
                c1 = c[k - 1];
                if (c1 < clow) c1 = clow;
            }
            if (!solve_period_wave(T, wave, t[k] / tscale, c1, clow, dc, cm, betmx, ifirst, s1st,
                                   cnt)) {
                for (int i = k; i < P; ++i) c[i] = 0.0;
                break;
            }
            c[k] = c1;
        }
    }
    return MHD_ST_OK;
}

// The ellipticity at each period's phase velocity c[k] (0 where the mode has no root), one
// period per lane; c and ell in LDS.
__device__ inline void ellipticity_curve(const double* t, int P, const double* c, double* ell)
{
    const int lane = threadIdx.x & 63;
    wave_sync();
    for (int k0 = 0; k0 < P; k0 += 64) {
        const int k = k0 + lane;
        double v = 0.0;
        if (k < P && c[k] > 0.0) v = eval_ellipticity(TWOPI / t[k], c[k]);
        if (k < P) ell[k] = v;
    }
    wave_sync();
}

// forward_model for the table in T: phase or group velocity into vel[] (LDS), or both
// (MHD_PHASE_AND_GROUP: phase in vel[0..P), group in vel[P..2P)), or phase velocity and
// ellipticity (MHD_PHASE_AND_ELL: phase in vel[0..P), ellipticity in vel[P..2P)), or the
// ellipticity alone (MHD_ELL).
// scratch must hold 2*P doubles (LDS).  Wave-uniform.
__device__ inline int forward_wave(const LayerTable& T, int wave, int mode, int itype,
                                   const double* t, int P, double dc, double dt, double* vel,
                                   double* scratch, EvalCount& cnt)
{
    if (itype == MHD_PHASE) return phase_curve_wave(T, wave, t, P, 1.0, mode, dc, vel, cnt);
    if (itype == MHD_PHASE_AND_ELL || itype == MHD_ELL) {
        double* c = (itype == MHD_ELL) ? scratch : vel;
        const int st0 = phase_curve_wave(T, wave, t, P, 1.0, mode, dc, c, cnt);
        if (st0 != MHD_ST_OK) return st0;
        ellipticity_curve(t, P, c, (itype == MHD_ELL) ? vel : vel + P);
        return MHD_ST_OK;
    }
    if (itype == MHD_PHASE_AND_GROUP) {
        const int st0 = phase_curve_wave(T, wave, t, P, 1.0, mode, dc, vel, cnt);
        if (st0 != MHD_ST_OK) return st0;
        vel += P;
    }
    double* c1 = scratch;
    double* c2 = scratch + P;
    int st = phase_curve_wave(T, wave, t, P, 1.0 + dt, mode, dc, c1, cnt);
    if (st != MHD_ST_OK) return st;
    st = phase_curve_wave(T, wave, t, P, 1.0 - dt, mode, dc, c2, cnt);
    if (st != MHD_ST_OK) return st;
    for (int k = 0; k < P; ++k) {
        if (c1[k] > 0.0 && c2[k] > 0.0) {
            const double t1 = t[k] / (1.0 + dt), t2 = t[k] / (1.0 - dt);
            const double f1 = 1.0 / t1, f2 = 1.0 / t2;
            vel[k] = (f1 - f2) / (f1 / c1[k] - f2 / c2[k]);
        } else {
            vel[k] = 0.0;
        }
    }
    return MHD_ST_OK;
}

// numpy float64 add.reduce order for a contiguous vector of n <= 128 entries.
__device__ inline double npsum_small(const double* a, int n)
{
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += a[i];
        return res;
    }
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        r0 += a[i + 0];
        r1 += a[i + 1];
        r2 += a[i + 2];
        r3 += a[i + 3];
        r4 += a[i + 4];
        r5 += a[i + 5];
        r6 += a[i + 6];
        r7 += a[i + 7];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res += a[i];
    return res;
}

// Entries of the data vectors (obs, pred, noise) of a problem.
__device__ __host__ inline int data_length(const mhd_problem_t& p)
{
    return (p.itype == MHD_PHASE_AND_GROUP || p.itype == MHD_PHASE_AND_ELL) ? 2 * p.n_periods
                                                                             : p.n_periods;
}

// Model.get_likelihood for one predicted curve in LDS; term[] is LDS scratch [ND].
__device__ inline double loglik_wave(const mhd_problem_t& p, const double* pred, double* term,
                                     int lane)
{
    const int P = data_length(p);
    for (int k = lane; k < P; k += 64) {
        const double r = p.obs[k] - pred[k];
        if (p.noise_dist == MHD_NORMAL) {
            const double s = p.noise_a[p.n_noise_a == 1 ? 0 : k];
            term[k] = (r * r) / (2.0 * (s * s));
        } else {
            const double lam = p.noise_a[p.n_noise_a == 1 ? 0 : k];
            const double kap = p.noise_b[p.n_noise_b == 1 ? 0 : k];
            const int si = sgni(r);
            const double s = (double)si;
            const double kps = (si > 0) ? kap : ((si < 0) ? 1.0 / kap : 1.0);
            term[k] = log(lam / (kap + (1.0 / kap))) - (((r * lam) * s) * kps);
        }
    }
    wave_sync();
    const double sum = npsum_small(term, P);
    wave_sync();
    return (p.noise_dist == MHD_NORMAL) ? -sum : sum;
}

}  // namespace mhd
