// mhd_search2.hpp -- period-parallel dispersion curve (fundamental mode, phase).
//
// The sequential search couples the periods only through the abscissa lattice:
// period k starts at c[k-1] - 1.5*dc.  Which lattice points it visits depends on
// c[k-1] exactly, but the SIGN it sees at any point follows from where the root
// of period k lies.  So the curve is computed in four stages, one wave per chain:
//
//   1. predict_coarse: walk the periods in order, several at a time; the 64
//      lanes evaluate windows of dc-spaced abscissae for the next 1..8 periods
//      and bracket each period's root to one dc cell (continuation in period,
//      like the sequential search, but on approximate grids and without halving).
//   2. predict_refine: one lane per period finds its root to ~1e-12 relative with
//      secant steps on the exponent-tracked (smooth) secular value.
//      (predict_hint does 1 and 2 in one when the curve of a similar model is
//      known and the new curve is close to it: secant steps from that curve, all
//      periods at once.)
//   3. replay: the sequential algorithm is executed with signs inferred from the
//      roots; no evaluations.  What it did for each period is kept as a summary
//      from which every abscissa it consumed can be regenerated.
//   4. verify: the lanes really evaluate every abscissa the replay consumed.  If
//      all signs agree, the replay WAS the sequential algorithm, bit for bit.  At
//      the first disagreement the periods before it are exact; that period is
//      redone by the wave-sequential search of mhd_device.hpp and the rest is
//      replayed (or predicted) again.
//
// Stages 1-2 are only a predictor: any failure there costs time, never accuracy.
#pragma once
#include "mhd_device.hpp"

namespace mhd {

#ifndef MHD_MAX_REPAIRS
#define MHD_MAX_REPAIRS 8
#endif
constexpr int MAX_REPAIRS = MHD_MAX_REPAIRS;  // then the wave-sequential search finishes the curve
// Periods searched sequentially after a repair while the answers differ from the predictions
// (measured: 4 -> 10 gave 2 % on ordinary chains and 9 % where one chain's curve jumps; 16 and
// more lose again, the sequential search being the slower one per period).
#ifndef RESYNC_PERIODS
#define RESYNC_PERIODS 10
#endif
constexpr int MAX_SCAN = 62;    // scan abscissae of one period the replay follows; longer scans are mostly
                                // downward ones that end at the floor, which the wave-sequential search handles

struct PredictScratch {
    double lo[MAXP], hi[MAXP];    // root bracket per period (stage 1: one dc cell)
    float fam[MAXP], fbm[MAXP];   // tracked values at the bracket ends: mantissa ...
    short fae[MAXP], fbe[MAXP];   // ... and binary exponent (start values of stage 2 only)
    signed char slo[MAXP], shi[MAXP];
    double fm[64];  // lane values of the current stage-1 round
    int fe[64];
    double wbase[8];  // windows of the current stage-1 round: first abscissa ...
    double wstep[8];  // ... spacing ...
    int woff[9];      // ... and first lane
    // what the replay did for each period (see replay_period)
    double rstart[MAXP];
    int rmeta[MAXP];
    unsigned short roff[MAXP + 1];  // number of abscissae of the pass before the period's
    int trouble;  // repairs the last predicted curve needed (9: the predictor gave up)
    int hinted;   // predict_hint for the last curve: 0 not tried, 1 settled, 2 gave up
    int prof_slot;  // where a profiling build books this wave's cycles (the chain's index)
    int abort;      // set by another wave of the workgroup: the curve this wave works on is no
                    // longer needed (see abort_requested)
};

static_assert(WAVE_LDS_SEARCH_OFFSET + sizeof(PredictScratch) <= WAVE_LDS_BYTES,
              "WAVE_LDS_BYTES is too small");
#define g_PS                                                                              \
    (*reinterpret_cast<PredictScratch*>(mhd_wave_lds + (threadIdx.x >> 6) * WAVE_LDS_BYTES + \
                                        WAVE_LDS_SEARCH_OFFSET))
// the same block of wave w of this workgroup
#define g_PS_OF(w)                                                                         \
    (*reinterpret_cast<PredictScratch*>(mhd_wave_lds + (w) * WAVE_LDS_BYTES + WAVE_LDS_SEARCH_OFFSET))

// A predicted curve can be given up between two rounds of evaluations: the Metropolis-Hastings
// kernels run several candidates of a chain at once, and once one of them is accepted the later
// ones started from a state the chain never has.  The wave that finds its candidate acceptable
// sets `abort` of the waves after it; they look at it once per round and return
// MHD_ST_ABORTED.  A request only ever saves time: a curve that is returned is complete.
constexpr int MHD_ST_ABORTED = -1;  // internal: never leaves a kernel
__device__ __forceinline__ bool abort_requested()
{
    return __builtin_amdgcn_readfirstlane(*(volatile int*)&g_PS.abort) != 0;
}

// A secant iterate x(n+1) is within about C |x(n+1)-x(n)| |x(n+1)-x(n-1)| of the root; with this
// bound on the product it is taken as the root without being evaluated (C, the curvature over
// twice the slope, is some tens per km/s).  The replay infers a wrong sign only for an abscissa
// between the root and its estimate; the halving abscissae come no closer to a root than some
// 1e-7 of it as a rule, so an estimate good to a few 1e-11 fails about one curve in a thousand,
// which the verification repairs.
#ifndef ROOT_ERR_PRODUCT
#define ROOT_ERR_PRODUCT 1.0e-12
#endif

// half-width, in abscissae, of the first window of a round that the hint places
#ifndef SHAPED_BASE_M
#define SHAPED_BASE_M 1
#endif

// hint steps (in dc) from which predict_coarse spaces the abscissae of a window 2 and 4 dc apart
#ifndef WIDE_STEP_2
#define WIDE_STEP_2 3.0
#endif
#ifndef WIDE_STEP_4
#define WIDE_STEP_4 8.0
#endif

// predict_coarse widens the windows it places by the hint: by HINT_STRETCH of the way along the
// hint's curve since the last known root (the new model stretches or shrinks its steps), and by
// HINT_TREND_SLACK of the trend carried on (a shift that changes fast is less certain)
#ifndef HINT_STRETCH
#define HINT_STRETCH 0.0
#endif
#ifndef HINT_TREND_SLACK
#define HINT_TREND_SLACK 1.5
#endif

// share of the last period-to-period change of (root - hint) that predict_coarse carries on
#ifndef HINT_TREND
#define HINT_TREND 0.8
#endif

#pragma clang fp contract(off)

// The verification with two abscissae per lane (eval_secular2) or one.  Two per lane halve the
// rounds but not the time: with two waves on a SIMD the FP64 pipe is saturated either way (an
// isolated evaluation kernel gains 10 %, scratch/evalbench); measured on the benchmark: +3 %
// on long calls, -2 % on 20-step calls.
#ifndef VERIFY_PAIRS
#define VERIFY_PAIRS 1
#endif

// -DMHD_PROFILE: per-stage cycle counts and event counts (they slow the kernels by a quarter:
// every wave adds to the same few counters); with -DMHD_PROFILE_LIGHT only where and when each
// chain's launches begin and end (g_times, g_blk[..][5]), which costs nothing measurable
#if defined(MHD_PROFILE) && !defined(MHD_PROFILE_LIGHT)
#define MHD_PROFILE_STAGES 1
#endif
#ifdef MHD_PROFILE
__device__ unsigned long long g_prof[64];
__device__ unsigned long long g_times[4096];
__device__ unsigned long long g_blk[1024 * 8];
__device__ unsigned int g_chain[1024 * 32];  // event counts per chain, see CHAIN_EVENT
#endif
#ifdef MHD_PROFILE_STAGES
#define CHAIN_EVENT(i) { const int _pc = g_PS.prof_slot; if ((threadIdx.x & 63) == 0 && _pc >= 0 && _pc < 1024) atomicAdd(&g_chain[_pc * 32 + (i)], 1u); }
#define PROF_T0 long long _t0 = clock64();
#define PROF_ADD(i) { long long _t1 = clock64(); if ((threadIdx.x & 63) == 0) { atomicAdd(&g_prof[i], (unsigned long long)(_t1 - _t0)); const int _ps = g_PS.prof_slot; if (_ps >= 0 && _ps < 1024) atomicAdd(&g_blk[_ps * 8 + i], (unsigned long long)(_t1 - _t0)); } _t0 = _t1; }
#define HINT_WHY(i) if ((threadIdx.x & 63) == 0) atomicAdd(&g_prof[i], 1ull);
#define REPLAY_T0 long long _rt0 = clock64();
#define REPLAY_ADD(i) { const long long _rt1 = clock64(); if ((threadIdx.x & 63) == 0) atomicAdd(&g_prof[i], (unsigned long long)(_rt1 - _rt0)); _rt0 = _rt1; }
#else
#define PROF_T0
#define PROF_ADD(i)
#define CHAIN_EVENT(i) {}
#define REPLAY_T0
#define REPLAY_ADD(i)
#define HINT_WHY(i)
#endif

__device__ __forceinline__ int mask_sign(unsigned long long neg, unsigned long long pos, int bit)
{
    return (int)((pos >> bit) & 1) - (int)((neg >> bit) & 1);
}

// ---- stage 1 ----------------------------------------------------------------
// Brackets the roots of periods kbeg..n-1 to one dc cell each and returns n (= P normally);
// kbeg > 0 continues from the root rprev_in of period kbeg-1 (sigma0_in = sign of the secular
// function of period 0 at the floor).
// n < P: period n-1 has no sign change below the ceiling (its bracket is "constant sign"),
// the sequential search fails there.  -1: give up (the caller uses the sequential search).
// hint (if use_hint): a curve of a similar model (the chain's current one).  Its shape -- the
// differences between its periods -- places small windows around the expected roots, eight
// periods per round; where that fails, and without a hint, the windows start at the expected
// start of the sequential scan and are 8, 16, 32 or 64 abscissae wide.
__device__ __forceinline__ int predict_coarse(int wave, const double* t, int kbeg, int P,
                                           double tscale, double dc, double rprev_in,
                                           int sigma0_in, const double* hint, bool use_hint,
                                           EvalCount& cnt)
{
    PredictScratch& S = g_PS;
    const int lane = threadIdx.x & 63;
    const double cc = g_T.cc, top = g_T.betmx + dc;
    int k0 = kbeg, sigma0 = sigma0_in, carry_sign = 0, rounds = 0;
    double rprev = rprev_in, inc = 1.0, carry_base = 0.0;
    bool have_inc = false, carry = false;
    bool hint_ok = true;  // the next round may use the hint's shape
    double trend = 0.0;   // change of (root - hint) from one period to the next, lately
    double shift_prev = 0.0;
    int shift_at = -2;    // the period shift_prev belongs to
    int hint_misses = 0;  // rounds in which it bracketed nothing
#ifdef MHD_PROFILE_STAGES
    int prof_first = 0, prof_shaped = 0, prof_shaped_res = 0, prof_lat = 0, prof_lat_res = 0, prof_carry = 0, prof_down = 0;
#endif

    while (k0 < P) {
        if (++rounds > 96 || abort_requested()) return -1;
        // ---- the windows of this round: period k0+q gets the lanes woff[q]..woff[q+1]-1 and
        // the abscissae wbase[q] + j*dc
        int nq = 0, g = 1;
        const bool shaped = use_hint && hint_ok && k0 > 0 && !carry;
#ifdef MHD_PROFILE_STAGES
        prof_first += (k0 == 0 && !carry);
        prof_shaped += shaped;
        prof_carry += carry;
        prof_lat += (!shaped && k0 > 0 && !carry);
#endif
        if (shaped) {
            // around the expected root of each period, wider with the distance from the last
            // known root -- in periods (5, 5, 7, 7, ... abscissae) and along the hint's curve,
            // whose steps the new model stretches or shrinks; as many periods as 64 lanes hold
            // (the shift against the hint changes smoothly with the period: its last change is
            // carried on, damped)
            // Where the hint's curve is steep (long periods) the new model's steps differ from
            // the hint's by several dc: there the abscissae of a window are 2 or 4 dc apart
            // (the refinement starts from a bracket of that width just as well).
            int off = 0;
            double path = 0.0;
            for (int q = 0; q < 8 && k0 + q < P; ++q) {
                const double rq = rprev + (hint[k0 + q] - hint[k0 - 1]) + (q + 1) * trend;
                const double hstep = fabs(hint[k0 + q] - hint[k0 + q - 1]);
                path += hstep;
                const double gs = (hstep < WIDE_STEP_2 * dc ? 1.0 : (hstep < WIDE_STEP_4 * dc ? 2.0 : 4.0)) * dc;
                int m = SHAPED_BASE_M + q / 2 + (int)((HINT_STRETCH * path + HINT_TREND_SLACK * (q + 1) * fabs(trend)) / gs);
                m = m > 15 ? 15 : m;
                if (off + 2 * m + 1 > 64) {
                    if (q > 0) break;
                    m = 31;
                }
                if (lane == 0) {
                    S.woff[q] = off;
                    S.wbase[q] = rq - (m + 0.5) * gs;
                    S.wstep[q] = gs;
                }
                off += 2 * m + 1;
                nq = q + 1;
            }
            if (lane == 0) S.woff[nq] = off;
        } else {
            int lg = 0;  // log2 of the number of periods in this round
            if (!(k0 == 0 || carry)) {
                const double ai = fabs(inc);
                lg = !have_inc ? 2 : (ai <= 2.0 ? 3 : (ai <= 5.0 ? 2 : (ai <= 12.0 ? 1 : 0)));
            }
            g = 1 << lg;
            const int W = 64 >> lg;
            nq = (P - k0 < g) ? P - k0 : g;
            const double pinc = have_inc ? inc : 1.0;
            if (lane == 0) {
                for (int q = 0; q < nq; ++q) {
                    double b;
                    if (carry && q == 0) {
                        b = carry_base;
                    } else if (k0 + q == 0) {
                        b = cc;
                    } else {
                        int mlo = (q == 0) ? 0 : (1 + (q + 1) / 2);
                        if (W >= 16) mlo *= 2;
                        b = rprev + q * pinc * dc - 1.5 * dc - mlo * dc;
                        if (b < cc - 2.0 * dc) b = cc - 2.0 * dc;
                    }
                    S.woff[q] = q * W;
                    S.wbase[q] = b;
                    S.wstep[q] = dc;
                }
                S.woff[nq] = nq * W;
            }
        }
        wave_sync();
        int jj = 0;
        for (int q = 1; q < nq; ++q) jj += (lane >= S.woff[q]);
        const int nlanes = S.woff[nq];
        const bool active = lane < nlanes;
        SecVal v = {0.0, 0};
        if (active) {
            const double omega = TWOPI / (t[k0 + jj] / tscale);
            v = eval_secular(wave, omega, S.wbase[jj] + (lane - S.woff[jj]) * S.wstep[jj]);
        }
        const unsigned long long neg = __ballot(active && v.m < 0.0);
        const unsigned long long pos = __ballot(active && v.m > 0.0);
        wave_sync();
        S.fm[lane] = v.m;
        S.fe[lane] = v.e;
        wave_sync();
        cnt.round((unsigned long long)nlanes);
        if (k0 == 0 && !carry) sigma0 = mask_sign(neg, pos, 0);

        int resolved = 0;
        bool carried_now = false;
        if (shaped) {
            // a period is bracketed when its window holds a sign change (whether the sequential
            // scan meets another one before it is for the verification to find out)
            for (int q = 0; q < nq; ++q) {
                const int k = k0 + q, bit0 = S.woff[q], W = S.woff[q + 1] - bit0;
                const unsigned long long wm = (1ull << W) - 1ull;
                const unsigned long long ng = (neg >> bit0) & wm, ps = (pos >> bit0) & wm;
                const unsigned long long chg = ((ng << 1) & ps) | ((ps << 1) & ng);
#ifdef MHD_PROFILE_STAGES
                if ((__popcll(chg) != 1 || __popcll(ng | ps) != W) && lane == 0) {
                    atomicAdd(&g_prof[40 + (__popcll(chg) == 0 ? 0 : 1)], 1ull);
                    atomicAdd(&g_prof[42], (unsigned long long)q);
                    if (q < 2) atomicAdd(&g_prof[44 + (k < 10 ? 0 : (k < 20 ? 1 : (k < 30 ? 2 : 3)))], 1ull);
                }
#endif
                if (chg == 0ull || __popcll(ng | ps) != W) break;
                // (of several sign changes the lowest: the scan comes from below)
                const int j = (int)__ffsll((long long)chg) - 1;  // sign change between j-1 and j
                // The sequential scan goes upward only if the sign at its start is the one below
                // the first root of the first period; a root with the other sign below it is one
                // the scan will not arrive at from below (it turns downward at its start): the
                // next round looks at this period from its scan start.
                if (sigma0 != 0 && (int)((ps >> (j - 1)) & 1ull) - (int)((ng >> (j - 1)) & 1ull) != sigma0)
                    break;
                const double sb = S.wbase[q], gs = S.wstep[q];
                S.lo[k] = sb + (j - 1) * gs;
                S.fam[k] = (float)S.fm[bit0 + j - 1];
                S.fae[k] = (short)S.fe[bit0 + j - 1];
                S.hi[k] = sb + j * gs;
                S.fbm[k] = (float)S.fm[bit0 + j];
                S.fbe[k] = (short)S.fe[bit0 + j];
                // the root inside the bracket, by the secant of the tracked values at its ends
                int de = S.fe[bit0 + j] - S.fe[bit0 + j - 1];
                de = de > 60 ? 60 : (de < -60 ? -60 : de);
                const double rho = ldexp(S.fm[bit0 + j] / S.fm[bit0 + j - 1], de);  // negative
                double frac = 1.0 / (1.0 - rho);
                frac = frac > 0.05 ? (frac < 0.95 ? frac : 0.95) : 0.05;
                const double nr = sb + (j - 1 + frac) * gs;
                inc = (nr - rprev) / dc;
                have_inc = true;
                if (use_hint) {
                    const double sh = nr - hint[k];
                    trend = (shift_at == k - 1) ? HINT_TREND * (sh - shift_prev) : 0.0;
                    shift_prev = sh;
                    shift_at = k;
                }
                rprev = nr;
                ++resolved;
            }
            nq = 0;  // skips the lattice pass below
        }
        for (int q = 0; q < nq; ++q) {
            const int k = k0 + q, bit0 = S.woff[q], W = S.woff[q + 1] - bit0;
            const double s = S.wbase[q];
            int start, s1;
            if (carry && q == 0) {
                start = -1;
                s1 = carry_sign;
            } else if (k == 0) {
                start = 0;
                s1 = mask_sign(neg, pos, bit0);
            } else {
                const double st = rprev - 1.5 * dc;
                const double off = (st - s) / dc;
                if (off < -1.0e-9) break;
                start = (int)floor(off + 1.0e-9);
                if (start >= W - 1) break;
                s1 = mask_sign(neg, pos, bit0 + start);
                if (s1 * sigma0 < 0) {
                    // reversed dispersion: the root lies below the start
                    int j2 = start - 1;
                    for (; j2 >= 0; --j2)
                        if (mask_sign(neg, pos, bit0 + j2) != s1) break;
                    if (j2 >= 0) {
                        S.lo[k] = s + j2 * dc;
                        S.fam[k] = (float)S.fm[bit0 + j2];
                        S.fae[k] = (short)S.fe[bit0 + j2];
                        S.hi[k] = s + (j2 + 1) * dc;
                        S.fbm[k] = (float)S.fm[bit0 + j2 + 1];
                        S.fbe[k] = (short)S.fe[bit0 + j2 + 1];
                        const double nr = s + (j2 + 0.5) * dc;
                        inc = (nr - rprev) / dc;
                        have_inc = true;
                        if (use_hint) {
                            const double sh = nr - hint[k];
                            trend = (shift_at == k - 1) ? HINT_TREND * (sh - shift_prev) : 0.0;
                            shift_prev = sh;
                            shift_at = k;
                        }
                        rprev = nr;
                        ++resolved;
                        continue;
                    }
                    if (resolved > 0) break;
                    // its own round: 64 abscissae downwards from the start
                    if (++rounds > 96 || abort_requested()) return -1;
                    const double xd = st - lane * dc;
                    const bool act = xd > cc;
                    SecVal vd = {0.0, 0};
                    if (act) vd = eval_secular(wave, TWOPI / (t[k] / tscale), xd);
                    const unsigned long long ng = __ballot(act && vd.m < 0.0);
                    const unsigned long long ps = __ballot(act && vd.m > 0.0);
                    const unsigned long long am = __ballot(act);
                    wave_sync();
                    S.fm[lane] = vd.m;
                    S.fe[lane] = vd.e;
                    wave_sync();
                    cnt.round((unsigned long long)__popcll(am));
                    const int sd = mask_sign(ng, ps, 0);
                    int j3 = 1;
                    for (; j3 < 64; ++j3) {
                        if (!((am >> j3) & 1)) return -1;  // reached the floor
                        if (mask_sign(ng, ps, j3) != sd) break;
                    }
                    if (j3 >= 64) return -1;
                    S.lo[k] = st - j3 * dc;
                    S.fam[k] = (float)S.fm[j3];
                    S.fae[k] = (short)S.fe[j3];
                    S.hi[k] = st - (j3 - 1) * dc;
                    S.fbm[k] = (float)S.fm[j3 - 1];
                    S.fbe[k] = (short)S.fe[j3 - 1];
                    const double nr = st - (j3 - 0.5) * dc;
                    inc = (nr - rprev) / dc;
                    have_inc = true;
                    if (use_hint) {
                        const double sh = nr - hint[k];
                        trend = (shift_at == k - 1) ? HINT_TREND * (sh - shift_prev) : 0.0;
                        shift_prev = sh;
                        shift_at = k;
                    }
                    rprev = nr;
                    ++resolved;
                    break;
                }
            }
            int jn = start + 1;
            bool got = false;
            for (; jn < W; ++jn) {
                if (mask_sign(neg, pos, bit0 + jn) != s1) {
                    got = true;
                    break;
                }
                if (s + jn * dc >= top) {
                    // no sign change below the ceiling: the sequential search fails here
                    S.lo[k] = 1.0e300;
                    S.hi[k] = 1.0e301;
                    S.fam[k] = (float)s1;
                    S.fbm[k] = (float)s1;
                    S.fae[k] = 0;
                    S.fbe[k] = 0;
                    wave_sync();
                    return k + 1;
                }
            }
            if (!got) {
                if (q > 0) break;  // next round gives this period its own windows
                carry = true;
                carried_now = true;
                carry_base = s + W * dc;
                carry_sign = s1;
                S.lo[k] = s + (W - 1) * dc;
                S.fam[k] = (float)S.fm[bit0 + W - 1];
                S.fae[k] = (short)S.fe[bit0 + W - 1];
                break;
            }
            if (jn - 1 >= 0) {
                S.lo[k] = s + (jn - 1) * dc;
                S.fam[k] = (float)S.fm[bit0 + jn - 1];
                S.fae[k] = (short)S.fe[bit0 + jn - 1];
            }
            S.hi[k] = s + jn * dc;
            S.fbm[k] = (float)S.fm[bit0 + jn];
            S.fbe[k] = (short)S.fe[bit0 + jn];
            const double nr = s + (jn - 0.5) * dc;
            if (k > 0) {
                inc = (nr - rprev) / dc;
                have_inc = true;
            }
            if (use_hint) {
                const double sh = nr - hint[k];
                trend = (shift_at == k - 1) ? HINT_TREND * (sh - shift_prev) : 0.0;
                shift_prev = sh;
                shift_at = k;
            }
            rprev = nr;
            ++resolved;
            carry = false;
        }
#ifdef MHD_PROFILE_STAGES
        if (shaped) prof_shaped_res += resolved; else if (k0 > 0) prof_lat_res += resolved;
#endif
        k0 += resolved;
        if (resolved > 0 && !shaped && hint_misses < 4) hint_ok = true;
        if (resolved == 0 && !carried_now) {
            if (shaped) {
                hint_ok = false;  // this period from its scan start
                ++hint_misses;
            } else {
                if (g == 1) return -1;
                inc = (fabs(inc) + 1.0) * 4.0;  // widen the windows
            }
        }
    }
    wave_sync();
#ifdef MHD_PROFILE_STAGES
    if (lane == 0) {
        atomicAdd(&g_prof[14], (unsigned long long)rounds);
        atomicAdd(&g_prof[15], 1ull);
        atomicAdd(&g_prof[32], (unsigned long long)prof_first);
        atomicAdd(&g_prof[33], (unsigned long long)prof_shaped);
        atomicAdd(&g_prof[34], (unsigned long long)prof_shaped_res);
        atomicAdd(&g_prof[35], (unsigned long long)prof_lat);
        atomicAdd(&g_prof[36], (unsigned long long)prof_lat_res);
        atomicAdd(&g_prof[37], (unsigned long long)prof_carry);
        atomicAdd(&g_prof[38], (unsigned long long)(use_hint ? 1 : 0));
        atomicAdd(&g_prof[39], (unsigned long long)hint_misses);
    }
#endif
    return P;
}

// ---- stage 1, period by period ------------------------------------------------
// The same brackets from the sequential search itself, stopped after `levels` halvings of each
// period (3: one round of the wave, 9: two) and continued from the bracket's midpoint.  Its
// abscissae are within dc / 2^(levels+1) of the lattice the sequential algorithm will use, so
// it follows that algorithm through densely spaced modes, where the windows of predict_coarse
// lose it; it costs two to four times as many rounds.  The bracket ends carry signs only:
// predict_refine evaluates them first.
// kbeg > 0 continues from cprev = c[kbeg-1].  Returns like predict_coarse.
__device__ MHD_NOINLINE int predict_stepwise(int wave, const double* t, int kbeg, int P,
                                             double tscale, double dc, double cprev, int s1st_in,
                                             int levels, EvalCount& cnt)
{
    PredictScratch& S = g_PS;
    const LayerTable& T = g_T;
    const int lane = threadIdx.x & 63;
    const double cc = T.cc, betmx = T.betmx, cm = T.cc;
    int s1st = s1st_in;
    double cp = cprev;
    EvalCount mine = {0, 0, 0};
    for (int k = kbeg; k < P; ++k) {
        const bool ifirst = (k == 0);
        double c1 = ifirst ? cc : cp - 1.5 * dc;
        double br[3] = {0.0, 0.0, 0.0};
        const bool ok = solve_period_wave(T, wave, t[k] / tscale, c1, ifirst ? cc : cm, dc, cm,
                                          betmx, ifirst, s1st, mine, levels, br);
        if (!ok) {
            // the sequential search will fail here as well (unless the lattice decides otherwise)
            if (lane == 0) {
                S.lo[k] = 1.0e300;
                S.hi[k] = 1.0e301;
                S.slo[k] = (signed char)(int)br[2];
                S.shi[k] = (signed char)(int)br[2];
            }
            cnt.exec += mine.exec;
            cnt.rounds += mine.rounds;
            wave_sync();
            return k + 1;
        }
        const bool up = br[0] < br[1];
        const int sa = (int)br[2];
        if (lane == 0) {
            S.lo[k] = up ? br[0] : br[1];
            S.hi[k] = up ? br[1] : br[0];
            S.slo[k] = (signed char)(up ? sa : -sa);
            S.shi[k] = (signed char)(up ? -sa : sa);
        }
        cp = c1;
    }
    cnt.exec += mine.exec;
    cnt.rounds += mine.rounds;
    wave_sync();
    return P;
}

// ---- stage 2 ----------------------------------------------------------------
// eval_ends: the brackets come with signs only (predict_stepwise); the tracked values at their
// ends are evaluated first.
__device__ __forceinline__ void predict_refine(int wave, const double* t, int kbeg, int P,
                                            double tscale, EvalCount& cnt, bool eval_ends = false)
{
    PredictScratch& S = g_PS;
    const int lane = threadIdx.x & 63;
    for (int kb = kbeg; kb < P; kb += 64) {
        const int k = kb + lane;
        const bool active = k < P;
        double a = 1.0, b = 1.0, fa = 1.0, fb = -1.0, omega = 1.0;
        int ea = 0, eb = 0;
        if (active) {
            a = S.lo[k];
            b = S.hi[k];
            fa = S.fam[k];
            fb = S.fbm[k];
            ea = S.fae[k];
            eb = S.fbe[k];
            omega = TWOPI / (t[k] / tscale);
        }
        if (eval_ends) {
            const bool real = active && a < 1.0e299;
            if (real) {
                const SecVal va = eval_secular(wave, omega, a);
                const SecVal vb = eval_secular(wave, omega, b);
                fa = va.m;
                ea = va.e;
                fb = vb.m;
                eb = vb.e;
            } else if (active) {
                fa = (double)S.slo[k];
                fb = (double)S.shi[k];
                ea = eb = 0;
            }
            cnt.round(2ull * (unsigned long long)__popcll(__ballot(real)), 2);
        }
        const int sa = sgni(fa);
        int sb = sgni(fb);
        bool done = !active || sa == sb || sa == 0 || sb == 0;
        // Secant steps on the two latest iterates (x0, x1), kept inside the bracket [a, b] of
        // evaluated points (a step that leaves it becomes an Illinois step on the bracket).
        // The errors fall like e(n+1) ~ C e(n) e(n-1), so when a step is shorter than 1e-9 a the
        // NEXT iterate is far closer to the root than the 1e-12 that the replay needs, and is
        // taken without being evaluated: the replay infers signs from it alone and the
        // verification pass checks every one of them anyway.
        double x0 = a, f0 = fa, x1 = b, f1 = fb, root = 0.5 * (a + b);
        int e0 = ea, e1 = eb, side = 0;
        for (int it = 0; it < 48; ++it) {
            double x = root;
            if (!done) {
                auto secant = [](double xa, double fxa, int exa, double xb, double fxb, int exb) {
                    int de = exb - exa;
                    de = de > 1000 ? 1000 : (de < -1000 ? -1000 : de);
                    const double rho = ldexp(fxb / fxa, de);
                    return xa + (xb - xa) / (1.0 - rho);
                };
                x = secant(x0, f0, e0, x1, f1, e1);
                const double w = b - a;
                const double mgn = 1.0e-3 * w;
                bool plain_step = true;
                if (!(x > a + mgn && x < b - mgn)) {
                    x = secant(a, fa, ea, b, fb, eb);  // Illinois-weighted ends
                    if (!(x > a + mgn)) x = a + mgn;
                    if (!(x < b - mgn)) x = b - mgn;
                    plain_step = false;
                }
                if (it > 0 && (fabs(x - x1) <= 1.0e-9 * a ||
                               (plain_step && fabs(x - x1) * fabs(x - x0) <= ROOT_ERR_PRODUCT))) {
                    root = x;
                    done = true;
                }
                if (w <= 1.0e-12 * a) {
                    root = 0.5 * (a + b);
                    done = true;
                }
            }
            const unsigned long long todo = __ballot(!done);
            if (todo == 0ull || abort_requested()) break;
            cnt.round((unsigned long long)__popcll(todo));
            if (!done) {
                const SecVal v = eval_secular(wave, omega, x);
                x0 = x1;
                f0 = f1;
                e0 = e1;
                x1 = x;
                f1 = v.m;
                e1 = v.e;
                root = x;
                if (sgni(v.m) == sa) {
                    a = x;
                    fa = v.m;
                    ea = v.e;
                    if (side == -1) fb *= 0.5;
                    side = -1;
                } else {
                    b = x;
                    fb = v.m;
                    eb = v.e;
                    sb = sgni(v.m);
                    if (side == +1) fa *= 0.5;
                    side = +1;
                }
            }
        }
        if (active) {
            const bool bracketed = sa != sb && sa != 0 && sb != 0;
            S.lo[k] = bracketed ? root : a;
            S.hi[k] = bracketed ? root : b;
            S.slo[k] = (signed char)sa;
            S.shi[k] = (signed char)sb;
        }
    }
    wave_sync();
}

// second abscissa of predict_hint: this many dc above the hint
#ifndef HINT_SECOND_POINT
#define HINT_SECOND_POINT 0.1
#endif

// longest extrapolation step of predict_hint, in dc
#ifndef HINT_STEP_MAX
#define HINT_STEP_MAX 12.0
#endif

// ---- stages 1 and 2 in one, from a hint ---------------------------------------------
// All periods at once, one lane each: secant iterations started at the hint's value for the
// period (the curve of a similar model) find the nearby root to ~1e-12 relative in five to
// seven rounds, without the dc lattice.  Whether that root is the one the sequential search
// arrives at is for the verification to find out, as with predict_coarse.  Returns P, or -1 when
// a period does not settle (the caller then uses predict_coarse): an iterate leaves
// (floor, ceiling) or strays more than 16 dc from the hint, or the iteration is not done after
// ten rounds.
__device__ __forceinline__ int predict_hint(int wave, const double* t, int P, double tscale,
                                         double dc, const double* hint, EvalCount& cnt)
{
    PredictScratch& S = g_PS;
    const int lane = threadIdx.x & 63;
    const double cc = g_T.cc, top = g_T.betmx + dc;
    int first_slo = 0;
    for (int kb = 0; kb < P; kb += 64) {
        const int k = kb + lane;
        const bool active = k < P;
        const double h = hint[active ? k : 0];
        const double omega = TWOPI / (t[active ? k : 0] / tscale);
        bool done = !active, failed = false;
        if (active && !(h > cc + dc && h < top - dc)) failed = true;
        if (__ballot(failed)) return -1;
        // iterates (x0, x1), the latest in x1; [a, b] brackets the root once a sign change
        // has been seen (sa = sign at a, below the root)
        double x0 = h, f0 = 0.0, x1 = h, f1 = 0.0, a = 0.0, b = 0.0, fa = 0.0, fb = 0.0;
        double root = h, x = h;
        int e0 = 0, e1 = 0, ea = 0, eb = 0, sa = 0, side = 0;
        bool bracketed = false;
        auto secant = [](double xa, double fxa, int exa, double xb, double fxb, int exb) {
            int de = exb - exa;
            de = de > 1000 ? 1000 : (de < -1000 ? -1000 : de);
            const double rho = ldexp(fxb / fxa, de);
            return xa + (xb - xa) / (1.0 - rho);
        };
        for (int it = 0; it < 11; ++it) {
            // the next abscissa, computed by every lane (no divergent flow before the votes)
            if (it == 0) {
                x = h;
            } else if (it == 1) {
                x = h + HINT_SECOND_POINT * dc;
            } else {
                double xs = secant(x0, f0, e0, x1, f1, e1);
                // bracketed: inside the bracket, else an Illinois step on its ends
                const double mgn = 1.0e-3 * (b - a);
                const bool inside = xs > a + mgn && xs < b - mgn;
                double xi = secant(a, fa, ea, b, fb, eb);
                xi = (xi > a + mgn) ? xi : a + mgn;
                xi = (xi < b - mgn) ? xi : b - mgn;
                // not bracketed: an extrapolation; a long one means a large shift (or a poor
                // linear model), where the windows of predict_coarse are the better tool
                const double xn = bracketed ? (inside ? xs : xi) : xs;
                const bool bad = !(xn > cc && xn < top) ||
                                 (!bracketed && !(fabs(xn - x1) <= HINT_STEP_MAX * dc)) ||
                                 !(fabs(xn - h) <= 16.0 * dc);
                const bool conv = fabs(xn - x1) <= 1.0e-9 * x1 ||
                                  (it > 2 && (inside || !bracketed) &&
                                   fabs(xn - x1) * fabs(xn - x0) <= ROOT_ERR_PRODUCT);
                failed = failed || (!done && (bad || it >= 10));
                if (!done) {
                    x = xn;
                    root = conv ? xn : root;
                }
                done = done || conv;
            }
            const unsigned long long nfail = __ballot(failed);
            const unsigned long long todo = __ballot(!done);
            if (it >= 2 && abort_requested()) return -1;
            if (nfail != 0ull) {
                HINT_WHY(16)
#ifdef MHD_PROFILE_STAGES
                if (lane == 0) atomicAdd(&g_prof[19], (unsigned long long)it);
#endif
                return -1;
            }
            if (todo == 0ull) {
                HINT_WHY(18)
#ifdef MHD_PROFILE_STAGES
                if (lane == 0) atomicAdd(&g_prof[17], (unsigned long long)it);
#endif
                break;
            }
            cnt.round((unsigned long long)__popcll(todo));
            SecVal v = {0.0, 0};
            if (!done) v = eval_secular(wave, omega, x);
            const int sv = sgni(v.m);
            if (!done) {
                if (sv == 0) failed = true;
                if (bracketed) {
                    if (sv == sa) {
                        a = x;
                        fa = v.m;
                        ea = v.e;
                        if (side == -1) fb *= 0.5;
                        side = -1;
                    } else {
                        b = x;
                        fb = v.m;
                        eb = v.e;
                        if (side == +1) fa *= 0.5;
                        side = +1;
                    }
                } else if (it > 0 && sv != sgni(f1)) {
                    // the first sign change: between the latest iterate and this one
                    bracketed = true;
                    const bool up = x > x1;
                    a = up ? x1 : x;
                    fa = up ? f1 : v.m;
                    ea = up ? e1 : v.e;
                    b = up ? x : x1;
                    fb = up ? v.m : f1;
                    eb = up ? v.e : e1;
                    sa = sgni(fa);
                }
                x0 = x1;
                f0 = f1;
                e0 = e1;
                x1 = x;
                f1 = v.m;
                e1 = v.e;
            }
        }
        // never bracketed: every iterate lies on one side of the root
        const int s1 = sgni(f1);
        const int slo = bracketed ? sa : (root > x1 ? s1 : -s1);
        // every root must have the sign below it that the first period's has: the sequential scan
        // does not arrive from below at a root with the other one (see predict_coarse)
        if (kb == 0) first_slo = __builtin_amdgcn_readfirstlane(slo);
        if (__ballot(active && slo != first_slo) != 0ull) {
            HINT_WHY(16)
            return -1;
        }
        if (active) {
            S.lo[k] = root;
            S.hi[k] = root;
            S.slo[k] = (signed char)slo;
            S.shi[k] = (signed char)-slo;
        }
    }
    wave_sync();
    return P;
}

// ---- stage 3 ----------------------------------------------------------------
// getsol of periods kstart..nP-1, one after the other, with signs inferred from the roots (same
// control flow and abscissa arithmetic as solve_period_wave; every value is the same in all
// lanes).  Only the plain case: a scan that meets neither the floor nor the ceiling, then
// halvings, no abscissa inside the root's bracket.  What it did for a period is kept as a
// summary -- start, direction, number of scan and halving abscissae -- from which
// replay_abscissae() regenerates any abscissa it consumed.
//
// This is one chain of dependent operations through all periods (a period starts 1.5 dc below
// the answer of the one before), so it is written for latency: the brackets are loaded once,
// lane j holding period kb + j, and reach the chain through v_readlane; the summaries are
// gathered in the lanes' registers and stored after the chain; the chain itself touches no LDS.
// A halving is an add, a multiply, a compare and two selects.  The stopping test
// |c1 - c2| <= 1e-6 c1 cannot succeed while the ends are more than 2^-19 of the larger one
// apart (2^-19 > 1e-6): for that many halvings, counted from the binary exponents, it is left
// out.  The verification regenerates every abscissa with the plain arithmetic of
// replay_abscissae and checks the answer of every period against it.
struct ReplayPass {
    int kend;    // periods kstart..kend-1 were replayed (summaries and roff[] are valid)
    int kfail;   // >= 0: the search of this period (= kend-1) ends above the ceiling
    int kplain;  // >= 0: this period (= kend) is not a plain case; nothing recorded for it
    int npts;    // abscissae consumed by the periods replayed
};

// Scan and halvings of one plain period.  BELOW: the start lies on the root's lower side (the scan
// goes upward); otherwise on its upper side (it goes downward).  c1 is the end of the bracket on
// the start's side throughout.  Written once per side so that a halving is an add, a multiply,
// a compare and two selects, with no flag to combine.
template <bool BELOW>
__device__ __forceinline__ bool replay_search(double start, double lo, double hi, double dc,
                                              double cc, double top, int& ns_out, int& nh_out,
                                              double& c3_out)
{
    const double step = BELOW ? dc : -dc;
    double c1 = start, c2 = start + step;
    int ns = 2;
    // the abscissae rise (fall) monotonically: the floor and the ceiling need to be looked at
    // for the first and the last one only
    if (BELOW ? uni(!(c2 > cc) || !(lo < top)) : uni(!(start < top))) return false;
    while (uni(BELOW ? (c2 <= lo) : !(c2 <= lo))) {
        if (ns >= MAX_SCAN) return false;
        c1 = c2;
        c2 = c2 + step;
        ++ns;
    }
    if (!BELOW && !uni(c2 > cc)) return false;
    const int e_apart = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_frexp_exp(fabs(c1 - c2)));
    const int e_ends = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_frexp_exp(fmax(c1, c2)));
    // |c1 - c2| >= 2^(e_apart - 1), the ends stay below 2^e_ends: after n halvings the ends are
    // at least 2^(e_apart - 1 - n - e_ends) of the larger one apart
    int sure = e_apart - 1 - e_ends + 19;
    sure = sure < 0 ? 0 : (sure > 38 ? 38 : sure);
    double c3 = 0.0;
    int nh = 0;
#pragma unroll 4
    for (; nh < sure; ++nh) {
        c3 = 0.5 * (c1 + c2);
        const bool low = c3 <= lo;
        // BELOW: a midpoint on the lower side replaces c1; otherwise it replaces c2
        c1 = (low == BELOW) ? c3 : c1;
        c2 = (low == BELOW) ? c2 : c3;
    }
    for (;;) {
        if (nh >= 40) return false;
        c3 = 0.5 * (c1 + c2);
        const bool low = c3 <= lo;
        ++nh;
        c1 = (low == BELOW) ? c3 : c1;
        c2 = (low == BELOW) ? c2 : c3;
        if (uni(fabs(c1 - c2) <= 1.0e-6 * c1)) break;
    }
    // an abscissa inside the root's bracket [lo, hi] would have become an end of [c1, c2]
    // (c1 is on lo's side by construction)
    if (!uni(BELOW ? c2 >= hi : c1 >= hi)) return false;
    ns_out = ns;
    nh_out = nh;
    c3_out = c3;
    return true;
}

__device__ __forceinline__ ReplayPass replay_pass(int kstart, int nP, double* c, double cc,
                                                  double dc, double top, double betmx, int& s1st)
{
    PredictScratch& S = g_PS;
    const int lane = threadIdx.x & 63;
    kstart = __builtin_amdgcn_readfirstlane(kstart);
    nP = __builtin_amdgcn_readfirstlane(nP);
    ReplayPass o = {kstart, -1, -1, 0};
    double cprev = kstart > 0 ? uniform_value(c[kstart - 1]) : 0.0;
    int s1 = s1st;
    for (int kb = kstart; kb < nP; kb += 64) {
        const int kmine = kb + lane;
        const bool mine = kmine < nP;
        const double lo_v = mine ? S.lo[kmine] : 0.0, hi_v = mine ? S.hi[kmine] : 0.0;
        // bit 0: the bracket's ends have opposite, non-zero signs; bit 1: the lower one is positive
        int sg_v = 0;
        if (mine) {
            const int slo = S.slo[kmine], shi = S.shi[kmine];
            sg_v = (int)(slo != 0 && shi != 0 && slo != shi) | ((int)(slo > 0) << 1);
        }
        double r_start = 0.0, r_c = 0.0;
        int r_meta = 0, r_off = 0;
        const int kchunk = (nP - kb < 64) ? nP - kb : 64;
        bool stop = false;
        for (int kk = 0; kk < kchunk; ++kk) {
            const int k = kb + kk;
            const double lo = readlane_f64(lo_v, kk), hi = readlane_f64(hi_v, kk);
            const int sg = __builtin_amdgcn_readlane(sg_v, kk);
            const int slo = (sg & 2) ? 1 : -1;
            const bool ifirst = (k == 0);
            const double start = ifirst ? cc : cprev - 1.5 * dc;
            const bool below1 = uni(start <= lo);
            const int sfirst = below1 ? slo : -slo;
            const bool upward = ifirst || s1 * sfirst >= 0;
            int ns = 0, nh = 0;
            double c3 = 0.0;
            // the scan arrives at the root only when it moves towards it
            bool plain = (sg & 1) && (upward == below1);
            if (plain)
                plain = below1 ? replay_search<true>(start, lo, hi, dc, cc, top, ns, nh, c3)
                               : replay_search<false>(start, lo, hi, dc, cc, top, ns, nh, c3);
            if (!plain) {
                o.kplain = k;
                stop = true;
                break;
            }
            if (lane == kk) {
                r_start = start;
                r_meta = ns | (nh << 8) | ((int)upward << 16);
                r_off = o.npts;
                r_c = c3;
            }
            o.npts += ns + nh;
            o.kend = k + 1;
            if (ifirst) s1 = sfirst;
            cprev = c3;
            if (!uni(c3 <= betmx)) {
                o.kfail = k;
                stop = true;
                break;
            }
        }
        if (mine && kmine < o.kend) {
            S.rstart[kmine] = r_start;
            S.rmeta[kmine] = r_meta;
            S.roff[kmine] = (unsigned short)r_off;
            if (kmine != o.kfail) c[kmine] = r_c;
        }
        if (stop) break;
    }
    if (lane == 0) S.roff[o.kend] = (unsigned short)o.npts;
    s1st = s1;
    wave_sync();
    return o;
}

// Abscissae q and q + 1 that replay_pass consumed for period k (any lane, any k and q), by the
// arithmetic of the sequential search: the scan adds the step, a halving takes the midpoint and
// moves the end that lies on the midpoint's side of the root.  xb is meaningless when q is the
// period's last abscissa.
__device__ __forceinline__ void replay_abscissae(int k, int q, double dc, double& xa, double& xb)
{
    const PredictScratch& S = g_PS;
    const int meta = S.rmeta[k];
    const int ns = meta & 255;
    const double step = (meta >> 16) ? dc : -dc;
    const double lo = S.lo[k];
    const double start = S.rstart[k];
    const bool below1 = start <= lo;
    double c1 = start, c2 = start, c3 = start, x = start;
    xa = start;
    for (int j = 1; j <= q + 1; ++j) {
        if (j < ns) {
            c1 = c2;
            c2 = c2 + step;
            x = c2;
        } else {
            if (j > ns) {
                const bool left = (c3 <= lo) != below1;
                c2 = left ? c3 : c2;
                c1 = left ? c1 : c3;
            }
            c3 = 0.5 * (c1 + c2);
            x = c3;
        }
        if (j == q) xa = x;
    }
    xb = x;
}

// The period that abscissa i of a replay pass over periods kstart..kend-1 belongs to.
__device__ __forceinline__ int period_of_abscissa(int i, int kstart, int kend)
{
    const PredictScratch& S = g_PS;
    int a = kstart, b = kend - 1;
    while (a < b) {
        const int mid = (a + b + 1) >> 1;
        if ((int)S.roff[mid] <= i) a = mid; else b = mid - 1;
    }
    return a;
}

#ifndef MHD_CURVE_FN
#define MHD_CURVE_FN MHD_NOINLINE
#endif
#ifndef VERIFY_OUT_OF_LINE
#define VERIFY_OUT_OF_LINE 1
#endif
#if VERIFY_OUT_OF_LINE
#define VERIFY_INLINE MHD_NOINLINE
#else
#define VERIFY_INLINE __forceinline__
#endif

// ---- stage 4 ----------------------------------------------------------------
// Really evaluates abscissae vbeg..npts-1 of the replay pass over periods kstart..kend-1 and
// returns the first whose sign is not the inferred one (or whose value is not the answer the
// replay gave for its period), npts if there is none, -1 if the curve was given up (see
// abort_requested).  Out of line, with the pair evaluation
// inlined: its registers are saved once per pass, not once per round.
__device__ VERIFY_INLINE int verify_pass(int wave, const double* t, double tscale, double dc,
                                        const double* c, int kstart, int kend, int kfail,
                                        int vbeg, int npts, EvalCount& cnt)
{
    const PredictScratch& S = g_PS;
    const int lane = threadIdx.x & 63;
    auto period_of = [&](int i) -> int { return period_of_abscissa(i, kstart, kend); };
    unsigned long long exec = 0;
    unsigned rounds = 0;
    int first_bad = npts;
#if !VERIFY_PAIRS
            for (int base = vbeg; base < npts; base += 64) {
                if (abort_requested()) {
                    first_bad = -1;
                    break;
                }
                const int i = base + lane;
                const bool act = i < npts;
                bool bad = false;
                if (act) {
                    const int k = period_of(i);
                    const int q = i - (int)S.roff[k];
                    double x, xnext;
                    replay_abscissae(k, q, dc, x, xnext);
                    const int want = (x <= S.lo[k]) ? (int)S.slo[k] : (int)S.shi[k];
                    const double omega = TWOPI / (t[k] / tscale);
                    bad = sgni(eval_secular(wave, omega, x).m) != want;
                    // the answer the replay gave for the period: its last abscissa
                    if (i == (int)S.roff[k + 1] - 1 && k != kfail) bad = bad || x != c[k];
                }
                const unsigned long long mask = __ballot(bad);
                exec += (unsigned long long)__popcll(__ballot(act));
                rounds += 1;
                if (mask) {
                    first_bad = base + (int)__ffsll((long long)mask) - 1;
                    break;
                }
            }
#else
            // two abscissae per lane, neighbours of one period's search as a rule
            for (int base = vbeg; base < npts; base += 128) {
                if (abort_requested()) {
                    first_bad = -1;
                    break;
                }
                const int i0 = base + 2 * lane, i1 = i0 + 1;
                const bool act0 = i0 < npts, act1 = i1 < npts;
                bool bad0 = false, bad1 = false;
                if (act0) {
                    const int k0 = period_of(i0);
                    const int q0 = i0 - (int)S.roff[k0];
                    const bool same = !act1 || i1 < (int)S.roff[k0 + 1];
                    const int k1 = same ? k0 : k0 + 1;
                    double x0, x1;
                    replay_abscissae(k0, q0, dc, x0, x1);
                    if (!same) x1 = S.rstart[k1];
                    if (!act1) x1 = x0;
                    const double omega0 = TWOPI / (t[k0] / tscale);
                    const double omega1 = same ? omega0 : TWOPI / (t[k1] / tscale);
                    const SecVal2 v = eval_secular2_inline(wave, omega0, x0, omega1, x1);
                    const int want0 = (x0 <= S.lo[k0]) ? (int)S.slo[k0] : (int)S.shi[k0];
                    const int want1 = (x1 <= S.lo[k1]) ? (int)S.slo[k1] : (int)S.shi[k1];
                    bad0 = sgni(v.m0) != want0;
                    bad1 = act1 && sgni(v.m1) != want1;
                    // the answer the replay gave for the period: its last abscissa
                    const int last0 = (int)S.roff[k0 + 1] - 1, last1 = (int)S.roff[k1 + 1] - 1;
                    if (i0 == last0 && k0 != kfail) bad0 = bad0 || x0 != c[k0];
                    if (act1 && i1 == last1 && k1 != kfail) bad1 = bad1 || x1 != c[k1];
                }
                const unsigned long long m0 = __ballot(bad0), m1 = __ballot(bad1);
                exec += (unsigned long long)(__popcll(__ballot(act0)) + __popcll(__ballot(act1)));
                rounds += 2;
                if (m0 | m1) {
                    const int f0 = m0 ? base + 2 * ((int)__ffsll((long long)m0) - 1) : npts;
                    const int f1 = m1 ? base + 2 * ((int)__ffsll((long long)m1) - 1) + 1 : npts;
                    first_bad = f0 < f1 ? f0 : f1;
                    break;
                }
            }
#endif
    cnt.exec += exec;
    cnt.rounds += rounds;
    return first_bad;
}

// ---- the curve ---------------------------------------------------------------
// Fundamental-mode phase velocities c[0..P-1] (LDS) of the table in g_T, bit-identical to
// phase_curve_wave(..., mode = 0, ...).  Returns MHD_ST_OK or MHD_ST_NOROOT.
// stepwise > 0: predict with predict_stepwise(levels = stepwise) instead of predict_coarse.
// stepwise < 0: the first -stepwise periods by the deferring sequential search (two rounds per
// period, nothing to repair), the others predicted from where it ends -- for models whose modes
// lie closer together than the search's step at short periods, where the sequential search hops
// between branches and every prediction of it costs a repair.
// hint: see predict_coarse; use_hint 0: none, 1: for predict_coarse, 2: predict_hint first.
__device__ MHD_CURVE_FN int phase_curve_predicted(int wave, const double* t, int P, double tscale,
                                                  double dc, double* c, EvalCount& cnt,
                                                  int stepwise, const double* hint,
                                                  int use_hint)
{
    const LayerTable& T = g_T;
    PredictScratch& S = g_PS;
    if (T.b[0] <= 0.0) return MHD_ST_NOROOT;
    const int lane = threadIdx.x & 63;
    dc = fabs(dc);
    PROF_T0
    int nP = -1;
    S.hinted = 0;
    CHAIN_EVENT(0)
    if (stepwise) CHAIN_EVENT(9)
    const double cc = T.cc, betmx = T.betmx, cm = T.cc;
    int s1st = 0, kstart = 0, repairs = 0;
    if (stepwise < 0) {
        const int kpre = -stepwise < P ? -stepwise : P;
        auto given_up = [&](int, double) -> bool { return abort_requested(); };
        const int kdone = sequential_run(wave, t, tscale, 0, kpre, dc, c, s1st, cnt, given_up);
        PROF_ADD(4)
        if (kdone < 0) return MHD_ST_NOROOT;
        if (abort_requested()) return MHD_ST_ABORTED;
        S.trouble = 0;
        if (kdone >= P) return MHD_ST_OK;
        kstart = kdone;
        nP = predict_coarse(wave, t, kstart, P, tscale, dc, uniform_value(c[kstart - 1]), s1st,
                            hint, use_hint != 0, cnt);
        PROF_ADD(0)
        if (abort_requested()) return MHD_ST_ABORTED;
        if (nP < 0) {
            const int r = phase_curve_wave(T, wave, t, P, tscale, 0, dc, c, cnt, kstart, s1st);
            PROF_ADD(5)
            return r;
        }
        predict_refine(wave, t, kstart, nP, tscale, cnt, false);
        PROF_ADD(1)
        stepwise = 0;
    } else
    if (!stepwise && use_hint == 2) {
        nP = predict_hint(wave, t, P, tscale, dc, hint, cnt);
        S.hinted = nP < 0 ? 2 : 1;
        CHAIN_EVENT(1)
        if (nP < 0) CHAIN_EVENT(2)
        PROF_ADD(0)
#ifdef MHD_PROFILE_STAGES
        if (lane == 0) { atomicAdd(&g_prof[11], 1ull); atomicAdd(&g_prof[12], (unsigned long long)(nP < 0)); }
#endif
    }
    if (abort_requested()) return MHD_ST_ABORTED;
    if (nP < 0 && kstart == 0) {
        nP = stepwise ? predict_stepwise(wave, t, 0, P, tscale, dc, 0.0, 0, stepwise, cnt)
                      : predict_coarse(wave, t, 0, P, tscale, dc, 0.0, 0, hint, use_hint != 0, cnt);
        PROF_ADD(0)
        if (abort_requested()) return MHD_ST_ABORTED;
        if (nP < 0) {
            S.trouble = 9;
            const int r = phase_curve_wave(T, wave, t, P, tscale, 0, dc, c, cnt);
            PROF_ADD(5)
            return r;
        }
        predict_refine(wave, t, 0, nP, tscale, cnt, stepwise > 0);
        PROF_ADD(1)
    }
    S.trouble = 0;

    for (;;) {
        if (abort_requested()) return MHD_ST_ABORTED;
        // replay periods kstart..kend-1 (c[kstart-1] and s1st are exact); period kplain, if
        // any, is not a plain case and ends the pass
        const ReplayPass rp = replay_pass(kstart, nP, c, cc, dc, betmx + dc, betmx, s1st);
        const int npts = rp.npts, kfail = rp.kfail, kplain = rp.kplain, kend = rp.kend;
        cnt.rounds += (unsigned)(kend - kstart + 3) >> 2;
#ifdef MHD_PROFILE_STAGES
        if (kplain >= 0) CHAIN_EVENT(7)
#endif
        PROF_ADD(2)
        auto period_of = [&](int i) -> int { return period_of_abscissa(i, kstart, kend); };
        // verify: really evaluate every abscissa the replay consumed
        int vbeg = 0, klit = kstart;  // first unverified abscissa; first period not yet counted
        bool again = false;           // replay again from kstart
        for (;;) {
            const int first_bad = verify_pass(wave, t, tscale, dc, c, kstart, kend, kfail, vbeg,
                                              npts, cnt);
            PROF_ADD(3)
            if (first_bad < 0) return MHD_ST_ABORTED;
            if (first_bad == npts) break;
            // periods before the first wrong sign are exact
            const int kbad = period_of(first_bad);
            for (int k = klit; k < kbad; ++k) cnt.lit += (int)S.roff[k + 1] - (int)S.roff[k];
            klit = kbad + 1;
#ifdef MHD_PROFILE_STAGES
            if (lane == 0) { atomicAdd(&g_prof[6], 1ull); const int _ps = g_PS.prof_slot; if (_ps >= 0 && _ps < 1024) atomicAdd(&g_blk[_ps * 8 + 6], 1ull); }
#endif
            S.trouble = repairs + 1;
            CHAIN_EVENT(4)
            if (S.hinted == 1) CHAIN_EVENT(27)
            if (++repairs > MAX_REPAIRS) {
                CHAIN_EVENT(10)
                const int r = phase_curve_wave(T, wave, t, P, tscale, 0, dc, c, cnt, kbad, s1st);
                PROF_ADD(4)
#ifdef MHD_PROFILE_STAGES
                if (lane == 0) atomicAdd(&g_prof[7], (unsigned long long)(P - kbad));
#endif
                return r;
            }
            // redo that one period with the wave-sequential search
            const bool ifirst = (kbad == 0);
            const double cold = (kfail == kbad) ? -1.0 : uniform_value(c[kbad]);
            double cr = ifirst ? cc : uniform_value(c[kbad - 1]) - 1.5 * dc;
            const bool ok = solve_period_wave(T, wave, t[kbad] / tscale, cr, ifirst ? cc : cm, dc,
                                              cm, betmx, ifirst, s1st, cnt);
            PROF_ADD(4)
            if (!ok) return MHD_ST_NOROOT;
            if (abort_requested()) return MHD_ST_ABORTED;
#ifdef MHD_PROFILE_STAGES
            {
                // where the sequential search ended against the predicted root, and where in the
                // period the first wrong sign was
                const double pr = uniform_value(S.lo[kbad]);
                if (uni(cr > pr + 0.5 * dc)) CHAIN_EVENT(16)
                else if (uni(cr < pr - 0.5 * dc)) CHAIN_EVENT(17)
                else CHAIN_EVENT(18)
                const int qbad = first_bad - (int)S.roff[kbad];
                if (qbad < (S.rmeta[kbad] & 255)) CHAIN_EVENT(19)
                else CHAIN_EVENT(20)
                if (uni(cr > pr + 8.0 * dc)) CHAIN_EVENT(21)
                if (kbad >= 30) CHAIN_EVENT(22)
                else if (kbad >= 20) CHAIN_EVENT(23)
                else if (kbad >= 10) CHAIN_EVENT(24)
                else CHAIN_EVENT(25)
            }
#endif
            c[kbad] = cr;
            wave_sync();
            if (kbad + 1 >= P) return MHD_ST_OK;
            if (uni(cr == cold)) {
                CHAIN_EVENT(8)
                // same answer: the replay of the later periods stands; verify on from the
                // first abscissa of the next period
                vbeg = (int)S.roff[kbad + 1];
                continue;
            }
            kstart = kbad + 1;
            again = true;
            const double plo = uniform_value(S.lo[kbad]), phi = uniform_value(S.hi[kbad]);
            if (kbad < nP - 1 && uni(cr >= plo - 2.0e-6 * cr && cr <= phi + 2.0e-6 * cr)) {
                CHAIN_EVENT(6)
                // same root on another lattice: the predictions for the later periods stand
                break;
            }
            // The sequential search is on another branch than the predictor's here.  In the models
            // where that happens (a slow branch that ends, the next one above it) the two mostly
            // differ in the period at which the search changes over, by one or two periods: the
            // next periods are searched sequentially while that lasts, and once an answer is the
            // predicted root again the predictions for the later periods stand.
            {
                // (two rounds per period: the deferring search, every tail verified on return)
                const int klast = kstart + RESYNC_PERIODS < nP ? kstart + RESYNC_PERIODS : nP;
                auto back = [&](int k, double ck) -> bool {
                    return uni(fabs(ck - uniform_value(S.lo[k])) <= 2.0e-6 * ck);
                };
                // (a request to give the curve up ends the run like an answer that is back)
                auto back_or_given_up = [&](int k, double ck) -> bool {
                    return back(k, ck) || abort_requested();
                };
                const int kdone = sequential_run(wave, t, tscale, kstart, klast, dc, c, s1st, cnt,
                                                 back_or_given_up);
                PROF_ADD(4)
                if (kdone < 0) return MHD_ST_NOROOT;
                if (abort_requested()) return MHD_ST_ABORTED;
                bool resynced = false;
                if (kdone > kstart) {
                    cr = uniform_value(c[kdone - 1]);
                    resynced = back(kdone - 1, cr);
                }
                kstart = kdone;
                if (kstart >= P) return MHD_ST_OK;
                if (resynced && kstart < nP) {
                    CHAIN_EVENT(26)
                    break;
                }
            }
            CHAIN_EVENT(5)
            // the sequential search moved to another branch than the predictor's: predict the
            // remaining periods again, continuing from the root it found
            nP = stepwise ? predict_stepwise(wave, t, kstart, P, tscale, dc, cr, s1st, stepwise, cnt)
                          : predict_coarse(wave, t, kstart, P, tscale, dc, cr, s1st, hint,
                                           use_hint != 0, cnt);
            PROF_ADD(0)
            if (abort_requested()) return MHD_ST_ABORTED;
            if (nP < 0) {
                const int r = phase_curve_wave(T, wave, t, P, tscale, 0, dc, c, cnt, kstart, s1st);
                PROF_ADD(5)
                return r;
            }
            predict_refine(wave, t, kstart, nP, tscale, cnt, stepwise > 0);
            PROF_ADD(1)
            break;
        }
        if (again) continue;
        for (int k = klit; k < kend; ++k) cnt.lit += (int)S.roff[k + 1] - (int)S.roff[k];
#ifdef MHD_PROFILE_STAGES
        if (lane == 0 && (kend >= nP || kfail >= 0)) atomicAdd(&g_prof[8], 1ull);
#endif
        if (kfail >= 0) return MHD_ST_NOROOT;
        if (kplain >= 0) {
            // not a plain case (floor, ceiling, an abscissa inside the bracket): this period by
            // the wave-sequential search, then on with the predictions for the later ones
            const bool ifirst = (kplain == 0);
            double cr = ifirst ? cc : uniform_value(c[kplain - 1]) - 1.5 * dc;
            const bool ok = solve_period_wave(T, wave, t[kplain] / tscale, cr, ifirst ? cc : cm,
                                              dc, cm, betmx, ifirst, s1st, cnt);
            PROF_ADD(4)
            if (!ok) return MHD_ST_NOROOT;
            c[kplain] = cr;
            wave_sync();
            kstart = kplain + 1;
            if (kstart >= P) return MHD_ST_OK;
            if (kstart < nP) continue;
            // fall through: the predictions end here
            nP = kstart;
        }
        if (nP == P) return MHD_ST_OK;
        // the predictor saw no root for period nP-1 but the search found one: go on
        const int r = phase_curve_wave(T, wave, t, P, tscale, 0, dc, c, cnt, nP, s1st);
        PROF_ADD(4)
        return r;
    }
}

// forward_model with the period-parallel curve where it applies (fundamental mode).
// hint: the phase curve of a similar model (MHD_PHASE, MHD_PHASE_AND_GROUP, MHD_PHASE_AND_ELL,
// MHD_ELL); hint2: its two
// phase curves at t/(1+dt) and t/(1-dt), P values each (MHD_GROUP, MHD_PHASE_AND_GROUP).
__device__ inline int forward_predicted(const LayerTable& T, int wave, int mode, int itype,
                                        const double* t, int P, double dc, double dt,
                                        double* vel, double* scratch, EvalCount& cnt,
                                        int stepwise, const double* hint, const double* hint2,
                                        int use_hint)
{
    // (the hints are never tested against null: a null test of a pointer into LDS trips the
    // compiler; without a hint the callers pass any valid pointer and use_hint = 0)
    if (mode != 0) {
        // the callers read these after every forward: higher modes use no predictor
        if ((threadIdx.x & 63) == 0) {
            g_PS.trouble = 0;
            g_PS.hinted = 0;
        }
        wave_sync();
        return forward_wave(T, wave, mode, itype, t, P, dc, dt, vel, scratch, cnt);
    }
    if (itype == MHD_PHASE_AND_ELL || itype == MHD_ELL) {
        double* c = (itype == MHD_ELL) ? scratch : vel;
        const int st = phase_curve_predicted(wave, t, P, 1.0, dc, c, cnt, stepwise, hint, use_hint);
        if (st != MHD_ST_OK) return st;
        ellipticity_curve(t, P, c, (itype == MHD_ELL) ? vel : vel + P);
        return MHD_ST_OK;
    }
    int trouble = 0;
    if (itype != MHD_GROUP) {
        const int st = phase_curve_predicted(wave, t, P, 1.0, dc, vel, cnt, stepwise, hint, use_hint);
#ifdef MHD_PROFILE_STAGES
        if (use_hint && st == MHD_ST_OK) {
            // largest shift of a root against the hint, in dc: buckets <1/4, <1/2, <1, <2, ... 
            double mx = 0.0;
            for (int k = 0; k < P; ++k) mx = fmax(mx, fabs(vel[k] - hint[k]) / fabs(dc));
            int bkt = 0;
            for (double lim = 0.25; mx >= lim && bkt < 10; lim *= 2.0) ++bkt;
            if ((threadIdx.x & 63) == 0) atomicAdd(&g_prof[21 + bkt], 1ull);
        }
#endif
        if (itype == MHD_PHASE || st != MHD_ST_OK) return st;
        trouble = __builtin_amdgcn_readfirstlane(g_PS.trouble);
        vel += P;
    }
    double* c1 = scratch;
    double* c2 = scratch + P;
    int st = phase_curve_predicted(wave, t, P, 1.0 + dt, dc, c1, cnt, stepwise,
                                   use_hint ? hint2 : t, use_hint);
    if (st != MHD_ST_OK) return st;
    trouble += __builtin_amdgcn_readfirstlane(g_PS.trouble);
    st = phase_curve_predicted(wave, t, P, 1.0 - dt, dc, c2, cnt, stepwise,
                               use_hint ? hint2 + P : t, use_hint);
    if (st != MHD_ST_OK) return st;
    // (the repairs of all the curves of this forward)
    if ((threadIdx.x & 63) == 0) g_PS.trouble += trouble;
    wave_sync();
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

}  // namespace mhd
