// mhd_api.hip -- kernels and the extern "C" entry points declared in
// include/mcmc_amd.h.  gfx950 only; one 64-lane wave per chain.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <mutex>

#include "mhd_device.hpp"
#include "mhd_search2.hpp"

using namespace mhd;

namespace {

thread_local char g_err[256] = "";

int fail(int code, const char* what)
{
    snprintf(g_err, sizeof(g_err), "%s", what);
    return code;
}

int hip_fail(hipError_t e, const char* where)
{
    snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
    return MHD_EHIP;
}

int check_problem(const mhd_problem_t* p, bool need_data)
{
    if (!p) return fail(MHD_EINVAL, "problem is NULL");
    if (p->n_layers < 1 || p->n_layers > MHD_MAX_LAYERS)
        return fail(MHD_EINVAL, "n_layers outside 1..MHD_MAX_LAYERS");
    if (p->n_periods < 1 || p->n_periods > MHD_MAX_PERIODS)
        return fail(MHD_EINVAL, "n_periods outside 1..MHD_MAX_PERIODS");
    if (p->wave != MHD_RAYLEIGH && p->wave != MHD_LOVE) return fail(MHD_EINVAL, "bad wave");
    if (p->itype < MHD_PHASE || p->itype > MHD_ELL) return fail(MHD_EINVAL, "bad itype");
    if (data_length(*p) > MHD_MAX_PERIODS)
        return fail(MHD_EINVAL, "a joint datum: 2 * n_periods beyond MHD_MAX_PERIODS");
    if ((p->itype == MHD_PHASE_AND_ELL || p->itype == MHD_ELL) && p->wave != MHD_RAYLEIGH)
        return fail(MHD_EINVAL, "ellipticity is a property of Rayleigh waves");
    if (p->mode < 0 || p->mode > 64) return fail(MHD_EINVAL, "bad mode");
    if (!(p->dc > 0.0) && !(p->dc < 0.0)) return fail(MHD_EINVAL, "dc must be non-zero");
    if (!p->periods) return fail(MHD_EINVAL, "periods is NULL");
    if (p->search != 0 && p->search != 1 && p->search != 2)
        return fail(MHD_EINVAL, "search must be 0 (default), 1 or 2");
    if (p->candidates != 0 && p->candidates != 1 && p->candidates != 2 && p->candidates != 4)
        return fail(MHD_EINVAL, "candidates must be 0 (default), 1, 2 or 4");
    if (need_data) {
        if (!p->obs || !p->noise_a || !p->bounds)
            return fail(MHD_EINVAL, "obs / noise_a / bounds is NULL");
        if (p->noise_dist != MHD_NORMAL && p->noise_dist != MHD_ASYM_LAPLACE)
            return fail(MHD_EINVAL, "bad noise_dist");
        if (p->noise_dist == MHD_ASYM_LAPLACE && !p->noise_b)
            return fail(MHD_EINVAL, "asym-laplace needs noise_b (kappa)");
        const int nd = data_length(*p);
        if ((p->n_noise_a != 1 && p->n_noise_a != nd) ||
            (p->noise_dist == MHD_ASYM_LAPLACE && p->n_noise_b != 1 && p->n_noise_b != nd))
            return fail(MHD_EINVAL, "noise arrays must have length 1 or that of the data");
    }
    return MHD_OK;
}

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------

// Finite layers of model ib of the batch (see mhd_problem_t.n_layers_b).
__device__ __forceinline__ int layers_of(const mhd_problem_t& p, int ib)
{
    if (!p.n_layers_b) return p.n_layers;
    const int n = p.n_layers_b[ib];
    return n < 1 ? 1 : (n > p.n_layers ? p.n_layers : n);
}

__global__ __launch_bounds__(64) MHD_REGS void secular_kernel(mhd_problem_t p, const double* params, int B,
                                                     const int32_t* chain, const int32_t* period,
                                                     const double* c, int N, double* delta)
{
    // one wave per query keeps the table code identical to the other kernels
    LayerTable& T = g_T;
    if ((threadIdx.x & 63) == 0) {
        g_PS.prof_slot = (int)blockIdx.x;
        g_PS.abort = 0;
    }
    __shared__ double m[MAXNP];
    const int q = blockIdx.x;
    if (q >= N) return;
    const int lane = threadIdx.x;
    const int NP = 2 * p.n_layers + 1;
    const int ib = chain[q], ip = period[q];
    // (the indices come from device memory the host has not seen: a query that points outside
    // the batch or the periods answers NaN)
    if (ib < 0 || ib >= B || ip < 0 || ip >= p.n_periods) {
        if (lane == 0) delta[q] = __builtin_nan("");
        return;
    }
    if (lane < NP) m[lane] = params[(size_t)ib * NP + lane];
    __syncthreads();
    build_layer_table(T, m, layers_of(p, ib), p.vpvs, lane, p.n_layers);
    const double omega = TWOPI / p.periods[ip];
    const double d = secular(T, p.wave, omega, c[q]);
    if (lane == 0) delta[q] = d;
}

// The same query through the evaluation of two abscissae per lane (the verification pass's):
// c and delta hold two values per query.
__global__ __launch_bounds__(64) MHD_REGS void secular_pair_kernel(
    mhd_problem_t p, const double* params, int B, const int32_t* chain, const int32_t* period,
    const double* c, int N, double* delta)
{
    LayerTable& T = g_T;
    if ((threadIdx.x & 63) == 0) {
        g_PS.prof_slot = (int)blockIdx.x;
        g_PS.abort = 0;
    }
    __shared__ double m[MAXNP];
    const int q = blockIdx.x;
    if (q >= N) return;
    const int lane = threadIdx.x;
    const int NP = 2 * p.n_layers + 1;
    const int ib = chain[q], ip = period[q];
    if (ib < 0 || ib >= B || ip < 0 || ip >= p.n_periods) {
        if (lane == 0) delta[2 * q] = delta[2 * q + 1] = __builtin_nan("");
        return;
    }
    if (lane < NP) m[lane] = params[(size_t)ib * NP + lane];
    __syncthreads();
    build_layer_table(T, m, layers_of(p, ib), p.vpvs, lane, p.n_layers);
    const double omega = TWOPI / p.periods[ip];
    const SecVal2 v = eval_secular2(p.wave, omega, c[2 * q], omega, c[2 * q + 1]);
    if (lane == 0) {
        delta[2 * q] = v.m0;
        delta[2 * q + 1] = v.m1;
    }
}

__global__ __launch_bounds__(64) MHD_REGS void forward_kernel(mhd_problem_t p, const double* params, int B,
                                                     double* vel, int32_t* status,
                                                     unsigned long long* n_evals, int search)
{
    LayerTable& T = g_T;
    if ((threadIdx.x & 63) == 0) {
        g_PS.prof_slot = (int)blockIdx.x;
        g_PS.abort = 0;
    }
    __shared__ double m[MAXNP];
    __shared__ double t[MAXP];
    __shared__ double out[MAXP];
    __shared__ double scratch[2 * MAXP];
    const int ib = blockIdx.x;
    const int lane = threadIdx.x;
    const int NP = 2 * p.n_layers + 1, P = p.n_periods, ND = data_length(p);
    if (lane < NP) m[lane] = params[(size_t)ib * NP + lane];
    for (int k = lane; k < P; k += 64) t[k] = p.periods[k];
    __syncthreads();
    build_layer_table(T, m, layers_of(p, ib), p.vpvs, lane, p.n_layers);
    EvalCount cnt = {0, 0, 0};
    const int st = (search == 2)
                       ? forward_predicted(T, p.wave, p.mode, p.itype, t, P, p.dc, p.dt, out,
                                           scratch, cnt, 0, t, t, 0)
                       : forward_wave(T, p.wave, p.mode, p.itype, t, P, p.dc, p.dt, out, scratch,
                                      cnt);
    __syncthreads();
    for (int k = lane; k < ND; k += 64) vel[(size_t)ib * ND + k] = (st == MHD_ST_OK) ? out[k] : 0.0;
    if (lane == 0) {
        status[ib] = st;
        if (n_evals) {
            atomicAdd(&n_evals[0], cnt.lit);
            atomicAdd(&n_evals[1], cnt.exec);
        }
    }
}

__global__ __launch_bounds__(64) void loglik_kernel(mhd_problem_t p, const double* pred, int B,
                                                    double* logL)
{
    __shared__ double pr[MAXP];
    __shared__ double term[MAXP];
    const int ib = blockIdx.x;
    const int lane = threadIdx.x;
    const int P = data_length(p);
    for (int k = lane; k < P; k += 64) pr[k] = pred[(size_t)ib * P + k];
    __syncthreads();
    const double l = loglik_wave(p, pr, term, lane);
    if (lane == 0) logL[ib] = l;
}

#pragma clang fp contract(off)

// update_covariance_matrix for the chain of this block; mn[] normalised params in LDS.
__device__ inline void cov_accumulate_wave(const mhd_chains_t& ch, int ib, int NP,
                                           const double* mn, double* diff, int lane)
{
    const double* pp = ch.params_prev + (size_t)ib * NP;
    bool nz = false;
    if (lane < NP) {
        diff[lane] = mn[lane] - pp[lane];
        nz = diff[lane] != 0.0;
    }
    wave_sync();
    if (__ballot(nz) == 0ull) return;
    double* cs = ch.cov_sum + (size_t)ib * NP * NP;
    for (int e = lane; e < NP * NP; e += 64) {
        const int i = e / NP, j = e - i * NP;
        cs[e] += diff[i] * diff[j];
    }
    if (lane == 0) ch.n_cov[ib] += 1;
    wave_sync();
}

// Optional per-step records beside hist_m / hist_logL / hist_pred (struct mhd_diag): the five
// counters of the chain, its per-parameter counters and its covariance matrix cov_sum / n_cov as
// they stand after step j, the h-th recorded one.  base[] holds the chain's counters at the
// start of the launch, d_* what the launch has added so far.  Called by the chain's first wave.
__device__ inline void record_diag(const mhd_diag_t& dg, const mhd_chains_t& ch, int B, int ib,
                                   int NP, size_t h, const long long* base, int d_acc, int d_rej,
                                   int d_bnd, int d_fm, int lane, long long d_lit = 0)
{
    if (dg.hist_counters && lane < 5) {
        // (slot 3, the reference's n_physics_err, stays 0: no proposal of this parameterisation
        // fails that check.  MHD_LIT_DEBUG, a debugging build: the sequential algorithm's
        // evaluations of the chain so far in this launch instead)
        long long d = lane == 0 ? d_acc : (lane == 1 ? d_rej : (lane == 2 ? d_bnd : (lane == 4 ? d_fm : 0)));
#ifdef MHD_LIT_DEBUG
        if (lane == 3) d = d_lit;
#else
        (void)d_lit;
#endif
        dg.hist_counters[(h * B + ib) * 5 + lane] = base[lane] + (long long)d;
    }
    if (dg.hist_counters_ind && ch.counters_ind) {
        wave_sync();  // (the per-parameter counters are updated in place by lane 0)
        const int64_t* ci = ch.counters_ind + (size_t)ib * NP * 5;
        int64_t* out = dg.hist_counters_ind + (h * B + ib) * (size_t)NP * 5;
        for (int e = lane; e < NP * 5; e += 64) out[e] = ci[e];
    }
    if (dg.hist_cov && ch.cov_sum && ch.n_cov) {
        wave_sync();
        const double* cs = ch.cov_sum + (size_t)ib * NP * NP;
        const long long n = ch.n_cov[ib];
        double* out = dg.hist_cov + (h * B + ib) * (size_t)NP * NP;
        for (int e = lane; e < NP * NP; e += 64) out[e] = n > 0 ? cs[e] / (double)n : 0.0;
    }
}

// How a chain's curves are searched, adapted to how the predictors fare on its models:
// level 0 = windowed predictor (predict_coarse), 1 and 2 = period-by-period predictor
// (predict_stepwise) with 3 and 9 halvings, 3 = the wave-sequential search without prediction.
// A curve that needed two or more repairs moves the chain one level up for the next `left`
// forwards; a level that runs out falls back one level.  Every level returns the same curve.
// Proposals that move the curve little start their forward model with predict_hint: those
// whose Cauchy draw is at most hint_limit[perturbed parameter] in magnitude.  The limit adapts,
// per chain and parameter: it grows by half when predict_hint settles and drops to 0.7 of the
// draw that made it give up (giving up costs a third of what settling saves, so the limit may
// be bold).  (A tuning state only: every path returns the same curve.)
#ifndef HINT_SECANT_CAUCHY
#define HINT_SECANT_CAUCHY 1.0
#endif
#ifndef HLIM_GROW
#define HLIM_GROW 1.5f
#endif
#ifndef HLIM_SHRINK
#define HLIM_SHRINK 0.7f
#endif
__device__ __forceinline__ void hint_limit_update(float* limit, int kk, double cz, int hinted)
{
    if (hinted == 1) limit[kk] = fminf(limit[kk] * HLIM_GROW, 64.0f);
    if (hinted == 2) limit[kk] = fmaxf(HLIM_SHRINK * (float)fabs(cz), 0.01f);
}

// Chains whose curves the predictor keeps getting wrong although it has the chain's current
// curve to go by (densely spaced modes: the sequential search hops between branches): a curve
// that needed this many repairs moves the chain to the period-by-period predictor, up to this level
#ifndef HINTED_MAX_LEVEL
#define HINTED_MAX_LEVEL 0
#endif
#ifndef HINTED_TROUBLE_MIN
#define HINTED_TROUBLE_MIN 3
#endif
// Chains in models whose modes lie closer together than the search's step at short periods: the
// sequential search hops between branches there and every prediction of it costs a repair.  When
// the curves of a chain's first candidate have needed more than DENSE_REPAIRS_UP repairs each on
// average lately (one or two chains in a thousand do; the 99th percentile is 0.6), the chain's
// curves start with DENSE_PREFIX(P) periods by the deferring sequential search and are predicted
// from there on, for DENSE_HOLD forwards; then the plain predictor is tried again.  (The one
// such chain among the benchmark's 1024 needs 2.3 repairs per curve, most of them in the first
// half of the periods, and was the last to finish every call by a fifth of the call's length;
// with this it finishes among the others.  Measured: 1000-step runs +13 %, 800-step +6 %,
// eight runs of 1500 ... 3000 steps +1 %, calls of 20 steps unchanged; a threshold of 1.5 with
// an average over 16 curves lets ordinary chains in now and then and gains nothing.)
#ifndef DENSE_REPAIRS_UP
#define DENSE_REPAIRS_UP 1.8f
#endif
#ifndef DENSE_HOLD
#define DENSE_HOLD 64
#endif
#ifndef DENSE_PREFIX
#define DENSE_PREFIX(P) ((P) / 2)
#endif
#ifndef DENSE_WINDOW
#define DENSE_WINDOW 32.0f
#endif
// A chain takes the free waves of its workgroup on when it has at least this many steps left
// (the test build: always)
#ifndef ADOPT_MIN_STEPS
#ifdef MHD_ABORT_SABOTAGE
#define ADOPT_MIN_STEPS 1
#else
#define ADOPT_MIN_STEPS 2
#endif
#endif
// Accept tests u <= exp(dl) that were settled with u within CLOSE_ULPS units in the last place of
// exp(dl) (mhd_close_decisions): two correctly working exponentials may differ by a unit or two
// there, so only such a decision could come out differently with the reference's numpy.
__device__ unsigned long long g_close_decisions;
constexpr double CLOSE_ULPS = 16.0;
__device__ __forceinline__ bool close_decision(double u, double e)
{
    // (u = 0 passes whatever the exponential gives, and an infinite one passes every u)
    return u > 0.0 && e < 1.0e308 && fabs(u - e) <= CLOSE_ULPS * 2.220446049250313e-16 * e;
}
struct SearchPolicy {
    int level, left;
    int max_level;  // 3, or 0 where the predictor has the chain's current curve to go by: there
                    // repairing its curves costs less than the stepwise predictions do
    __device__ bool predicted() const { return level < 3; }
    __device__ int stepwise() const { return level == 0 ? 0 : (level == 1 ? 3 : 9); }
    __device__ void update(int forwards, bool trouble)
    {
        if (level < max_level && trouble) {
            ++level;
            left = 8;
        } else if (level > 0) {
            left -= forwards;
            if (left <= 0) {
                --level;
                left = level > 0 ? 8 : 0;
            }
        }
    }
};

// One proposal (model.py:196-222), sorted by layer depth (model_params.py:157-176) and checked
// against the bounds (model.py:107-144), by one wave: test[] = the candidate parameters made from
// the normalised current parameters mn[].  Returns MHD_ST_OK or MHD_ST_BOUNDS.
struct ProposalFrame {
    const double *lo, *hi, *rg, *wid;  // LDS copies of the bounds and proposal widths
    const double* R;                   // rotation matrix of the chain, or nullptr
    int nl, NP;                        // finite layers of THIS chain; slots of a row
    int nlmax;                         // slot of the first layer's vs = (NP - 1) / 2
    bool uniform;                      // MHD_PROPOSE_UNIFORM: the draw replaces the parameter
    // the slots this chain's model uses (all of them unless the batch mixes layer counts)
    __device__ bool used(int slot) const
    {
        return slot < nl || (slot >= nlmax && slot <= nlmax + nl);
    }
};

__device__ inline int propose_wave(const ProposalFrame& f, const double* mn, int kk, double cz,
                                   double* test, double* tmp, int lane)
{
    const int nl = f.nl, NP = f.NP, vs_at = f.nlmax;
    if (f.R) {
        const double* R = f.R;
        if (lane < NP) {
            double acc = 0.0;
            for (int j = 0; j < NP; ++j) acc += R[j * NP + lane] * mn[j];
            // model.py:207 sets the (normalised, rotated) parameter to the physical lower bound
            // plus the draw; model.py:210 adds the scaled Cauchy deviate
            if (lane == kk) acc = f.uniform ? f.lo[kk] + cz : acc + f.wid[kk] * cz;
            tmp[lane] = acc;
        }
        wave_sync();
        if (lane < NP) {
            double acc = 0.0;
            for (int j = 0; j < NP; ++j) acc += R[lane * NP + j] * tmp[j];
            test[lane] = (acc * f.rg[lane]) + f.lo[lane];
        }
    } else if (lane < NP) {
        double v = mn[lane];
        if (lane == kk) v = f.uniform ? f.lo[kk] + cz : v + f.wid[kk] * cz;
        test[lane] = (v * f.rg[lane]) + f.lo[lane];
    }
    wave_sync();

    // stable rank sort of the layers by depth
    double dep = 0.0, vsl = 0.0;
    int rank = 0;
    if (lane < nl) {
        dep = test[lane];
        vsl = test[vs_at + lane];
        for (int j = 0; j < nl; ++j) {
            const double dj = test[j];
            rank += (dj < dep) || (dj == dep && j < lane);
        }
    }
    wave_sync();
    if (lane < nl) {
        test[rank] = dep;
        test[vs_at + rank] = vsl;
    }
    wave_sync();

    bool inb = true;
    if (lane < NP && f.used(lane)) inb = (test[lane] >= f.lo[lane]) && (test[lane] <= f.hi[lane]);
    return (__ballot(!inb) != 0ull) ? MHD_ST_BOUNDS : MHD_ST_OK;
}

__global__ __launch_bounds__(64) void cov_kernel(mhd_problem_t p, mhd_chains_t ch, int B)
{
    __shared__ double mn[MAXNP];
    __shared__ double diff[MAXNP];
    const int ib = blockIdx.x;
    const int lane = threadIdx.x;
    const int NP = 2 * p.n_layers + 1;
    if (lane < NP) {
        const double lo = p.bounds[3 * lane], rg = p.bounds[3 * lane + 2];
        mn[lane] = (ch.m[(size_t)ib * NP + lane] - lo) / rg;
    }
    __syncthreads();
    cov_accumulate_wave(ch, ib, NP, mn, diff, lane);
}

__global__ __launch_bounds__(64) MHD_REGS void mh_steps_kernel(
    mhd_problem_t p, mhd_chains_t ch, int B, int S, const int32_t* idx, const double* cauchy,
    const double* u_acc, double Temp, int rotate, int sample_prior, uint8_t* accepted,
    int32_t* status, int hist_every, double* hist_m, double* hist_logL, double* hist_pred,
    mhd_diag_t dg, unsigned long long* n_evals, int search)
{
    LayerTable& T = g_T;
    if ((threadIdx.x & 63) == 0) {
        g_PS.prof_slot = (int)blockIdx.x;
        g_PS.abort = 0;
    }
    __shared__ long long cbase[5];  // the chain's counters when the launch starts
    __shared__ double m[MAXNP];     // current parameters
    __shared__ double mn[MAXNP];    // normalised current parameters
    __shared__ double test[MAXNP];  // proposal
    __shared__ double tmp[MAXNP];
    __shared__ double lo[MAXNP], hi[MAXNP], rg[MAXNP], wid[MAXNP];
    __shared__ double t[MAXP];
    __shared__ double pred[MAXP];
    __shared__ double newpred[MAXP];
    __shared__ double scratch[2 * MAXP];
    __shared__ float hlim[MAXNP];  // see hint_limit_update

    const int ib = blockIdx.x;
    const int lane = threadIdx.x;
    const int nlmax = p.n_layers, NP = 2 * nlmax + 1, P = p.n_periods, ND = data_length(p);
    const int nl = layers_of(p, ib);

    if (lane < NP) {
        m[lane] = ch.m[(size_t)ib * NP + lane];
        lo[lane] = p.bounds[3 * lane];
        hi[lane] = p.bounds[3 * lane + 1];
        rg[lane] = p.bounds[3 * lane + 2];
        wid[lane] = ch.width[(size_t)ib * NP + lane];
        hlim[lane] = (float)HINT_SECANT_CAUCHY;
    }
    if (lane < 5) cbase[lane] = ch.counters[(size_t)ib * 5 + lane];
    for (int k = lane; k < P; k += 64) t[k] = p.periods[k];
    for (int k = lane; k < ND; k += 64) pred[k] = ch.pred[(size_t)ib * ND + k];
    double logL = ch.logL[ib];
    const double beta = ch.beta ? ch.beta[ib] : 1.0;
    const double* R = (rotate & MHD_ROT_FRAME) ? ch.rot + (size_t)ib * NP * NP : nullptr;
    const ProposalFrame frame = {lo, hi, rg, wid, R, nl, NP, nlmax,
                                 (rotate & MHD_PROPOSE_UNIFORM) != 0};
    long long c_acc = 0, c_rej = 0, c_bnd = 0, c_fm = 0;
#ifdef MHD_PROFILE
    const long long k_t0 = clock64(), k_w0 = wall_clock64();
    if (lane == 0 && ib < 2048) g_times[2 * ib] = (unsigned long long)k_w0;
#endif
    EvalCount cnt = {0, 0, 0};
    // a chain whose curves keep defeating the predictor (densely spaced modes) uses the
    // wave-sequential search for a while, then tries the predictor again
    // (the chain's current phase curve guides the predictor when the chain keeps one)
    const bool phase_hint = p.itype == MHD_PHASE || p.itype == MHD_PHASE_AND_ELL;
    SearchPolicy policy = {search == 2 ? 0 : 3, 0, phase_hint ? 0 : 3};
    __syncthreads();

    for (int s = 0; s < S; ++s) {
        const int kk = idx[(size_t)s * B + ib];
        const double cz = cauchy[(size_t)s * B + ib];
        const double ua = u_acc[(size_t)s * B + ib];

        if (lane < NP) mn[lane] = (m[lane] - lo[lane]) / rg[lane];
        __syncthreads();
        if (rotate & MHD_ROT_ACCUMULATE) cov_accumulate_wave(ch, ib, NP, mn, tmp, lane);

        int st = propose_wave(frame, mn, kk, cz, test, tmp, lane);
        if (st == MHD_ST_OK && !sample_prior) {
            build_layer_table(T, test, nl, p.vpvs, lane, nlmax);
            if (policy.predicted()) {
                st = forward_predicted(T, p.wave, p.mode, p.itype, t, P, p.dc, p.dt, newpred,
                                       scratch, cnt, policy.stepwise(), pred, pred,
                                       !phase_hint ? 0 : (!frame.uniform && fabs(cz) <= hlim[kk] ? 2 : 1));
                if (lane == 0)
                    hint_limit_update(hlim, kk, cz, __builtin_amdgcn_readfirstlane(g_PS.hinted));
                if (search == 2)
                    policy.update(1, __builtin_amdgcn_readfirstlane(g_PS.trouble) >= 2);
            } else {
                st = forward_wave(T, p.wave, p.mode, p.itype, t, P, p.dc, p.dt, newpred, scratch,
                                  cnt);
                if (search == 2) policy.update(1, false);
            }
            __syncthreads();
        }
        int acc = 0;
        if (st == MHD_ST_OK) {
            const double lnew = sample_prior ? 1.0 : loglik_wave(p, newpred, scratch, lane);
            double dl = lnew - logL;
            dl = dl / Temp;
            if (ch.beta) dl = beta * dl;
            const double edl = exp(dl);
            acc = ua <= edl;
            if (lane == 0 && close_decision(ua, edl)) atomicAdd(&g_close_decisions, 1ull);
            if (acc) {
                if (lane < NP) {
                    ch.params_prev[(size_t)ib * NP + lane] = mn[lane];
                    if (frame.used(lane)) m[lane] = test[lane];
                }
                logL = lnew;
                if (!sample_prior)
                    for (int k = lane; k < ND; k += 64) pred[k] = newpred[k];
            }
        }
        __syncthreads();
        c_acc += acc;
        c_rej += !acc;
        c_bnd += (st == MHD_ST_BOUNDS);
        c_fm += (st >= MHD_ST_NOROOT);
        if (lane == 0) {
            if (ch.counters_ind) {
                int64_t* ci = ch.counters_ind + ((size_t)ib * NP + kk) * 5;
                ci[0] += acc;
                ci[1] += !acc;
                ci[2] += (st == MHD_ST_BOUNDS);
                ci[4] += (st >= MHD_ST_NOROOT);
            }
            if (accepted) accepted[(size_t)s * B + ib] = (uint8_t)acc;
            if (status) status[(size_t)s * B + ib] = st;
        }
        if (hist_every > 0 && (s + 1) % hist_every == 0) {
            const size_t h = (size_t)((s + 1) / hist_every - 1);
            if (hist_m && lane < NP) hist_m[(h * B + ib) * NP + lane] = m[lane];
            if (hist_logL && lane == 0) hist_logL[h * B + ib] = logL;
            if (hist_pred)
                for (int k = lane; k < ND; k += 64) hist_pred[(h * B + ib) * ND + k] = pred[k];
            record_diag(dg, ch, B, ib, NP, h, cbase, (int)c_acc, (int)c_rej, (int)c_bnd, (int)c_fm,
                        lane);
        }
    }

#ifdef MHD_PROFILE
    if (lane == 0) {
#ifdef MHD_PROFILE_STAGES
        atomicAdd(&g_prof[9], (unsigned long long)(clock64() - k_t0));
        atomicAdd(&g_prof[10], (unsigned long long)(wall_clock64() - k_w0));
#else
        (void)k_t0;
#endif
        if (ib < 2048) g_times[2 * ib + 1] = (unsigned long long)wall_clock64();
    }
#endif
    if (lane < NP) ch.m[(size_t)ib * NP + lane] = m[lane];
    for (int k = lane; k < ND; k += 64) ch.pred[(size_t)ib * ND + k] = pred[k];
    if (lane == 0) {
        ch.logL[ib] = logL;
        int64_t* c5 = ch.counters + (size_t)ib * 5;
        c5[0] += c_acc;
        c5[1] += c_rej;
        c5[2] += c_bnd;
        c5[4] += c_fm;
        if (n_evals) {
            atomicAdd(&n_evals[0], cnt.lit);
            atomicAdd(&n_evals[1], cnt.exec);
        }
    }
}

// ---- parallel tempering -------------------------------------------------------
// One exchange sweep (src/inversion/inversion.py:560-596) for every temperature ladder: one lane
// per ladder walks its adjacent pairs in order, rung t with rung t+1, with the betas as the
// earlier pairs of the same sweep left them.  A pair with different betas is exchanged when
// u <= exp((beta2 - beta1) * (logL1 - logL2)); the betas move, the states stay.  Replica (l, t)
// sits at index l * ladder_stride + t * rung_stride of logL and beta.
__global__ __launch_bounds__(64) void tempering_sweep_kernel(const double* logL, double* beta,
                                                             const double* u, int n_temps,
                                                             int n_ladders, long long rung_stride,
                                                             long long ladder_stride,
                                                             long long* counts)
{
    const int l = blockIdx.x * 64 + threadIdx.x;
    if (l >= n_ladders) return;
    long long acc = 0, prop = 0;
    const long long base = (long long)l * ladder_stride;
    double b1 = beta[base];
    for (int t = 0; t + 1 < n_temps; ++t) {
        const long long i1 = base + t * rung_stride, i2 = i1 + rung_stride;
        const double b2 = beta[i2];
        double keep = b2;  // what rung t+1 holds after this pair
        if (b1 != b2) {
            const double logratio = (b2 - b1) * (logL[i1] - logL[i2]);
            const bool one = (b1 == 1.0) || (b2 == 1.0);
            if (u[(long long)l * (n_temps - 1) + t] <= exp(logratio)) {
                beta[i1] = b2;
                beta[i2] = b1;
                keep = b1;
                acc += one;
            }
            prop += one;
        }
        b1 = keep;
    }
    if (counts) {
        if (acc) atomicAdd((unsigned long long*)&counts[0], (unsigned long long)acc);
        if (prop) atomicAdd((unsigned long long*)&counts[1], (unsigned long long)prop);
    }
}

// Model.stepsize_tuning (src/inversion/model.py:538-556) for every chain: one thread per
// (chain, parameter).  The acceptance rate is n_acc / (n_acc + n_rej) once a proposal has been
// rejected and 0 before (the reference refreshes its rates only while n_rej > 0, :483, :520);
// above 0.4 the proposal width grows by half, in (0, 0.2) it is halved.
__global__ __launch_bounds__(256) void stepsize_tuning_kernel(mhd_chains_t ch, int B, int NP)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= B * NP) return;
    const int ib = e / NP;
    const int64_t* c = ch.counters_ind ? ch.counters_ind + (size_t)e * 5 : ch.counters + (size_t)ib * 5;
    const long long n_acc = c[0], n_rej = c[1];
    const double rate = n_rej > 0 ? (double)n_acc / (double)(n_acc + n_rej) : 0.0;
    double w = ch.width[e];
    if (rate > 0.4) w = w * 1.5;
    if (rate > 0.0 && rate < 0.2) w = w * 0.5;
    ch.width[e] = w;
}

// ---- cost-aware launch schedule ---------------------------------------------
// mhd_chains_t.sched, kept between calls: what a step of each chain costs and what the next
// launch is made from.  The cost of a chain is counted, not timed: the evaluation rounds its
// forward models take (EvalCount::rounds of the candidate that is always needed, averaged) times
// the share of its steps that have a forward model at all (a proposal outside the bounds costs
// nothing) -- "evaluation rounds per step with one candidate in flight"; 0 = not known yet.  It
// depends on the chain's path only, so the same run is scheduled the same way every time.  With
// nw candidates in flight a chain settles consumed(nw, p) forwards per round, p being the share
// of its forwards it accepted lately.  Only the deadline of a launch needs the clock: the ticks
// of the 100 MHz counter that one evaluation round took on average in the launch before (one
// number for all chains, summed in the header).
// The launch is made from the chains in order of decreasing expected time: the n4 chains at the
// head of that order run with four candidates in flight, the 4 * n1 = 2 * n4 chains at its tail
// with one, all others with two, so a launch keeps two waves per chain on average, and the
// expensive chains finish closer to the others.  Every workgroup has four waves, whatever it
// holds, so the launch fills the wave slots exactly whichever way the chains are split.
constexpr int SCHED_MAX_CHAINS = 4096;  // one workgroup sorts them in LDS
constexpr int SCHED_MIN_CHAINS = 64;    // fewer chains leave the GPU empty however they are ordered
// in a call's first launch a chain gets four candidates when with two it is expected to take
// longer than this multiple of the median chain's time; at most SCHED_FOUR_MAX(B) chains do
// (measured with 1024 chains: thresholds 1.15 ... 1.3 and 2 ... 8 chains are within 1 % of each
// other, 16 and more cost what they gain)
#ifndef SCHED_FOUR_ABOVE
#define SCHED_FOUR_ABOVE 1.2f
#endif
#ifndef SCHED_FOUR_MAX
#define SCHED_FOUR_MAX(B) ((B) / 256 > 2 ? (B) / 256 : 2)
#endif
#ifndef SCHED_GAIN
#define SCHED_GAIN 0.5f
#endif
#ifndef SCHED_LATER_FOUR
#define SCHED_LATER_FOUR 0
#endif
// share of a chain's steps that have a forward model, before the chain has shown its own
#ifndef SCHED_PVALID_PRIOR
#define SCHED_PVALID_PRIOR 0.85f
#endif
#ifndef SCHED_PVALID_WEIGHT
#define SCHED_PVALID_WEIGHT 16.0f
#endif
#ifndef SCHED_COST_PRIOR
#define SCHED_COST_PRIOR 20.0f  // evaluation rounds per step of a chain that has not run a forward yet
#endif
#ifndef SPEC_FIRST_A
#define SPEC_FIRST_A 0.42f
#endif
// Forwards a chain settles per round with nw candidates in flight when it accepts the share p of
// its forwards: the root of the candidates' tree always, every other node with the probability
// of the path to it (see mh_steps_multi_kernel for the trees).
__host__ __device__ inline float sched_consumed(int nw, float p)
{
    const float q = 1.0f - p;
    if (nw <= 1) return 1.0f;
    if (nw == 2) return 1.0f + (p > 0.5f ? p : q);
    if (p < SPEC_FIRST_A) return 1.0f + q + q * q + q * q * q;  // R RR RRR
    if (p < 0.5f) return 1.0f + q + q * q + p;                  // R RR A
    if (p < 0.68f) return 1.0f + p + q + p * p;                 // A R AA
    return 1.0f + p + p * p + p * p * p;                        // A AA AAA
}

#ifdef MHD_ABORT_SABOTAGE
constexpr bool ABORT_SABOTAGE_BUILD = true;
#else
constexpr bool ABORT_SABOTAGE_BUILD = false;
#endif
struct SchedView {
    float* cost;  // [B]
    int* n4;      // [0] n4, [1] n1: workgroups of four one-candidate chains, [2..3] the deadline of the launch,
                  // [4] n4 of the call's first launch, [5] chains the last launch found unfinished
                  // (diagnostics), [6..7] as floats: ticks and evaluation rounds of the launches
                  // since the last phase 0 (their ratio prices a round for the deadlines)
    int* order;   // [B]
    int* progress;         // [B] steps of the current call each chain has done
    unsigned char* level;  // [B] bits 0-1: SearchPolicy level each chain ended its last launch
                           // with; bits 2-7: the share of its forwards it accepted lately, in
                           // 63rds (0: not known yet), see mh_steps_multi_kernel
};

#ifndef SCHED_FREE4
#define SCHED_FREE4 1
#endif
// Chains that can have four candidates without another chain paying for it: the wave slots
// the GPU has beyond two per chain, two more per such chain.
__host__ __device__ inline int sched_free4(int B, int slots)
{
    const int f = SCHED_FREE4 ? (slots - 2 * B) / 2 : 0;
    return f < 0 ? 0 : (f > B ? B : f);
}

__host__ __device__ inline int sched_cap4(int B, int phase, int slots)
{
    const int c = (phase == 0 ? B / 8 : B / 2) + sched_free4(B, slots);
    return c > B ? B : (c > 0 ? c : 1);
}

__host__ __device__ inline SchedView sched_view(void* buf, int B)
{
    SchedView v;
    v.cost = (float*)buf;
    v.n4 = (int*)(v.cost + B);
    v.order = v.n4 + 8;
    v.progress = v.order + B;
    v.level = (unsigned char*)(v.progress + B);
    return v;
}

// One call of mhd_mh_steps is made of SCHED_PHASES launches.  All but the last end at a deadline
// (every chain stops after the round it is in and leaves the number of steps it has done in
// `progress`); the next launch gives the chains their candidates anew, by the work each has
// left: as the cheap chains finish, the places they leave go to the expensive ones, which then
// run with four candidates instead of two.  phase 0 also resets `progress`.
#ifndef SCHED_PHASES_N
#define SCHED_PHASES_N 2
#endif
constexpr int SCHED_PHASES = SCHED_PHASES_N;
// Calls of fewer steps are one launch: chains that share a workgroup hand their waves on when
// they are done (mh_steps_multi_kernel), which evens a short call out better than a deadline and
// a second launch do.  (Measured from prior draws, one launch against two: calls of 24 steps
// 3.32 against 2.89 M steps/s, 32 steps 3.37 against 3.04 M, 40 steps 3.51 against 3.66 M, 60
// steps 3.53 against 3.70 M.)
#ifndef SCHED_MIN_STEPS
#define SCHED_MIN_STEPS 48
#endif
constexpr int SCHED_PHASES_MIN_STEPS = SCHED_MIN_STEPS;
#ifndef PHASE_DEADLINE_LATER
#define PHASE_DEADLINE_LATER 1.0f  // later deadlines, in units of the median unfinished chain's
#endif
#ifndef PHASE_DEADLINE_FRACTION
#define PHASE_DEADLINE_FRACTION 0.7f  // first deadline, in units of the median chain's expected time
#endif

// the share of its forwards chain i accepted lately (SPEC_PACC_START while not known)
__device__ __forceinline__ float sched_pacc(const SchedView& v, int i)
{
    const int q = (int)v.level[i] >> 2;
    return q > 0 ? (float)q * (1.0f / 63.0f) : 0.3f;
}

__global__ __launch_bounds__(1024) void schedule_kernel(void* buf, int B, int S, int phase,
                                                        int phases, int slots)
{
    __shared__ float key[SCHED_MAX_CHAINS];
    __shared__ int val[SCHED_MAX_CHAINS];
    __shared__ int unknown;
    const SchedView v = sched_view(buf, B);
    int n = 1;
    while (n < B) n <<= 1;
    if (threadIdx.x == 0) unknown = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        float c = i < B ? v.cost[i] : -1.0f;
        if (i < B && !(c > 0.0f)) unknown = 1;
        if (i < B && phase == 0) v.progress[i] = 0;
        // evaluation rounds per step with two candidates in flight ...
        if (i < B) c /= sched_consumed(2, sched_pacc(v, i));
        // ... later phases: of the work that is left
        if (i < B && phase > 0) c *= (float)(S - v.progress[i]);
        key[i] = c;
        val[i] = i < B ? i : -1;
    }
    __syncthreads();
    // bitonic sort, descending by that (ties: by chain index, so that the order is definite)
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                const int l = i ^ j;
                if (l > i) {
                    const bool desc = (i & k) == 0;
                    const float a = key[i], b = key[l];
                    const int ia = val[i], ibv = val[l];
                    const bool a_first = a > b || (a == b && ia < ibv);
                    if (desc != a_first) {
                        key[i] = b;
                        key[l] = a;
                        val[i] = ibv;
                        val[l] = ia;
                    }
                }
            }
            __syncthreads();
        }
    }
    // the candidates of the launch
    if (threadIdx.x == 0) {
        int n4 = 0, n1 = 0;
        unsigned long long deadline = 0;
        // what an evaluation round took lately, in ticks (0: not known, no deadline)
        float* calib = (float*)(v.n4 + 6);
        const float tick = calib[1] > 0.0f ? calib[0] / calib[1] : 0.0f;
        // the same chain's expected rounds per step with four candidates
        auto with_four = [&](int j) {
            const float p = sched_pacc(v, val[j]);
            return key[j] * sched_consumed(2, p) / sched_consumed(4, p);
        };
        // (wave slots beyond two per chain go to the chains at the head of the order, two more
        // each, whatever is known about them)
        // (the test build gives only half of them away, so that its launches mix every kind of
        // workgroup)
        const int free4 = sched_free4(B, slots) / (ABORT_SABOTAGE_BUILD ? 2 : 1);
        if (unknown) n4 = free4;
        if (!unknown && phase == 0) {
            // four candidates for the chains that stand out from the others; each takes two
            // wave slots more, which two of the cheapest chains give up
            const int cap = sched_cap4(B, 0, slots);
            n4 = free4;
#ifdef MHD_ABORT_SABOTAGE
            const bool whatever_they_cost = true;  // the test build runs every kind of workgroup
#else
            const bool whatever_they_cost = false;
#endif
            // (with more chains than the GPU holds at once the launch is a queue of
            // two-candidate workgroups in this order: whoever finishes makes room)
            while (2 * B <= slots && n4 < cap && n4 - free4 < SCHED_FOUR_MAX(B) &&
                   4 * (n4 - free4) + 8 < B - free4 &&
                   (whatever_they_cost || key[n4] > SCHED_FOUR_ABOVE * key[(B + free4) / 2]))
                ++n4;
            n1 = (n4 - free4 + 1) / 2;
            if (PHASE_DEADLINE_FRACTION > 0.0f && phases > 1 && tick > 0.0f)
                deadline = (unsigned long long)wall_clock64() +
                           (unsigned long long)(PHASE_DEADLINE_FRACTION * key[B / 2] * (float)S *
                                                tick);
            calib[0] = 0.0f;
            calib[1] = 0.0f;
        } else if (!unknown) {
            // the chains with work left (they come first in the order): four candidates for as
            // many of them as the wave slots allow beside two for the others
            int left = 0;
            while (left < B && key[left] > 0.0f) ++left;
            v.n4[5] = left;
            // (SCHED_LATER_FOUR 0: none -- the chains with most work left share their workgroups
            // with the finished ones and take their waves on at once, see mh_steps_multi_kernel)
            const int room = 2 * B > slots ? 0 : (SCHED_LATER_FOUR ? (slots - 2 * left) / 2 : free4);
            n4 = room < 0 ? 0 : (room < left ? room : left);
            const int cap = sched_cap4(B, phase, slots);
            if (n4 > cap) n4 = cap;
            if (phase < phases - 1 && left > 0 && tick > 0.0f)
                deadline = (unsigned long long)wall_clock64() +
                           (unsigned long long)(PHASE_DEADLINE_LATER * tick *
                                                (left / 2 < n4 ? with_four(left / 2)
                                                               : key[left / 2]));
        }
        v.n4[0] = n4;
        v.n4[1] = n1;
        if (phase == 0) v.n4[4] = n4;
        v.n4[2] = (int)(deadline & 0xFFFFFFFFull);
        v.n4[3] = (int)(deadline >> 32);
    }
    for (int i = threadIdx.x; i < B; i += blockDim.x) v.order[i] = val[i];
}

// The same chain of Metropolis-Hastings steps with several candidates in flight: NW waves per
// chain.  From the current state, wave w takes the (w+1)-th proposal that passes the bounds
// (steps without one are rejections that cost nothing); all run the forward model at the same
// time.  The steps are then settled strictly in order: a wave's result is used only if the
// candidates before it were all rejected, because only then did its proposal start from the
// state the sequential chain has at that step.  States, likelihoods, counters and the draw each
// step consumes are those of the one-candidate kernel.
//
// scheduled (NW = 4): the workgroups take their chains from the launch schedule (see SchedView)
// and hold one chain with four candidates, two with two each or four with one each.  Of two
// chains that share a workgroup the one that is done first (or has nothing to do in this launch)
// hands its two waves to the other, which runs four candidates from its next round on: the
// workgroups pair the k-th chain of the order with the k-th from its end, so the chains expected
// to take longest get that help soonest.
#ifndef ABORT_MOOT
#define ABORT_MOOT 1
#endif
#ifndef SPEC_TREE
#define SPEC_TREE 1
#endif
// (a candidate that follows an acceptance has a poorer guide for its curve -- the current
// curve is two moves away -- and takes longer: with four candidates the first of them joins
// the rejection chain later than its probability alone says, 0.32)
#ifndef SPEC_FIRST_A
#define SPEC_FIRST_A 0.42f
#endif
#ifndef SPEC_HINT_SMALL
#define SPEC_HINT_SMALL 1
#endif
#ifndef SPEC_GROUP_PENALTY
#define SPEC_GROUP_PENALTY 0.1f
#endif
#ifndef SPEC_BIAS
#define SPEC_BIAS 0.0f
#endif
#ifndef SPEC_PACC_START
#define SPEC_PACC_START 0.3f
#endif
#ifndef SPEC_PACC_WINDOW
#define SPEC_PACC_WINDOW 32
#endif
#ifndef SPEC_PACC_KEPT
#define SPEC_PACC_KEPT 8
#endif
// Test build: every shape of the candidates' tree in turn, whatever the chain accepts; and
// requests to give a curve up that are NOT justified -- every wave that finishes a
// curve sends one to the waves after it, whatever its accept test says, and in every third round
// the later waves find one before they start.  Steps whose curve is missing when the settlement
// reaches them are run again, so nothing but the time taken may change
// (tests/test_gpu_deferred_tails.py).  g_abort_stats: curves given up, and how many of them the
// settlement reached, steps settled by a candidate that follows an acceptance.
// Waits between the waves of a workgroup that ran into their limit (mhd_wait_timeouts): never, in
// a working library; a call in which one did has wrong results.
__device__ unsigned long long g_wait_timeouts;
#ifdef MHD_ABORT_SABOTAGE
constexpr bool ABORT_SABOTAGE = true;
__device__ unsigned long long g_abort_stats[3];
#else
constexpr bool ABORT_SABOTAGE = false;
#endif
template <int NW>
__global__ __launch_bounds__(64 * NW) MHD_REGS void mh_steps_multi_kernel(
    mhd_problem_t p, mhd_chains_t ch, int B, int S, const int32_t* idx, const double* cauchy,
    const double* u_acc, double Temp, int rotate, uint8_t* accepted, int32_t* status,
    int hist_every, double* hist_m, double* hist_logL, double* hist_pred, mhd_diag_t dg,
    unsigned long long* n_evals, int search, int scheduled)
{
    constexpr int NG = (NW == 4) ? 4 : 1;  // chains a workgroup can hold (a scheduled launch
                                           // packs one, two or four into a workgroup of four waves)
    __shared__ long long cbase_[NG][5];    // the chain's counters when the launch starts
    __shared__ unsigned pair_arrivals[2];  // see pair_sync
    // Two chains that share a workgroup hand their waves on: the waves of the one that is done
    // first (or that has nothing to do in this launch) run two more candidates of the other.
    // help_wait_[g]: the waves of slot g are free; help_state_[g]: 0 the chain of slot g runs
    // with its own two waves, 1 it has taken the other two on, 2 it is done.
    struct ChainShare {
        int ib, s, round_no, pacc_n, level, left, max_level, hinted;
        float pacc, dense_avg;
        double logL;
    };
    __shared__ ChainShare share_[2];
    __shared__ int help_wait_[2], help_state_[2];
    __shared__ int r_adopt[NW];
    __shared__ double lo[MAXNP], hi[MAXNP], rg[MAXNP];
    __shared__ double t[MAXP];
    __shared__ double m_[NG][MAXNP];    // current parameters
    __shared__ double mn_[NG][MAXNP];   // normalised current parameters
    __shared__ double wid_[NG][MAXNP];
    __shared__ double pred_[NG][MAXP];
    __shared__ float hlim_[NG][MAXNP];  // see hint_limit_update
    __shared__ double test[NW][MAXNP];  // candidate of each wave
    __shared__ double tmp[NW][MAXNP];
    __shared__ double newpred[NW][MAXP];
    __shared__ double scratch[NW][3 * MAXP];  // two phase curves (group velocity), likelihood terms
    __shared__ double ghint_[NG][2 * MAXP];   // group velocity: the two phase curves of the current state
    __shared__ int ghint_ok_[NG];
    __shared__ double mw_[NW][MAXNP];   // the state a wave's candidate starts from when that is
    __shared__ double mnw_[NW][MAXNP];  // not the chain's current one: parameters, normalised
    __shared__ int r_step[NW], r_st[NW], r_trouble[NW], r_stop[NW];
    __shared__ double r_logL[NW];
    __shared__ unsigned long long r_lit[NW];

    // (wave-uniform values are moved to scalar registers explicitly: what the compiler cannot
    // prove uniform it keeps per lane, and those registers are spilled around every forward)
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    // the chain of this wave: the workgroup's own index, or its place in the launch schedule.
    // A scheduled launch (NW = 4 only) is made of workgroups of one size: the n4 most expensive
    // chains have one to themselves (four candidates), the 4 * n1 cheapest ones sit four to a
    // workgroup (one wave, one candidate each), all others two to a workgroup (two candidates
    // each); a workgroup's chains meet in no barrier but the one below.
    int ib = blockIdx.x;
    int g = 0, w0 = 0, nw = NW;  // chain slot in the workgroup, the chain's first wave, its waves
    bool idle = false;           // nothing for this wave in this launch
    bool pairs_wg = false;       // the workgroup holds two chains with two candidates each
    if (NW == 4 && scheduled) {
        const SchedView sv = sched_view(ch.sched, B);
        const int n4 = sv.n4[0], n1 = sv.n4[1];
        const int pairs = B - n4 - 4 * n1;  // chains that run with two candidates
        const int np = (pairs + 1) >> 1;    // their workgroups
        int slot = ib;                      // the chain's place in the launch order
        if (ib < n4) {
        } else if (ib < n4 + np) {
            // (the k-th of them with the k-th from the end: whoever is done first hands its
            // waves to the other, and the chains that take longest get the most of that)
            g = w >> 1;
            w0 = 2 * g;
            nw = 2;
            pairs_wg = true;
            const int k = ib - n4;
            slot = g == 0 ? n4 + k : n4 + pairs - 1 - k;
            idle = g == 1 && pairs - 1 - k <= k;
        } else if (ib < n4 + np + n1) {
            g = w;
            w0 = w;
            nw = 1;
            slot = n4 + pairs + 4 * (ib - n4 - np) + w;
        } else {
            idle = true;
        }
        if (!idle) {
            ib = __builtin_amdgcn_readfirstlane(sv.order[slot]);
            // nothing left of this call for the chain (a later phase)
            idle = sv.progress[ib] >= S;
        }
        if (idle) ib = 0;
    }
    if (NW == 2 && scheduled) {
        // more chains than the GPU holds at once: a workgroup each, the expensive ones first
        const SchedView sv = sched_view(ch.sched, B);
        ib = __builtin_amdgcn_readfirstlane(sv.order[ib]);
        idle = sv.progress[ib] >= S;
    }
    bool writer = (w == w0);
    const auto wave_of = [&](int q) { return (w0 + q) & (NW - 1); };  // the chain's q-th wave
    // The waves of a chain meet after every round: all waves of the workgroup in its barrier,
    // two of its four by counting arrivals in LDS (pair_sync), one alone not at all.
    unsigned pair_target = 0;
    auto pair_sync = [&]() {
        pair_target += 2;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0)
            __hip_atomic_fetch_add(&pair_arrivals[g], 1u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
        // (both waves are resident, so the other one arrives; the limit -- some ten seconds --
        // only keeps a defect from hanging the GPU: results are then wrong, not late)
        bool met = false;
        for (unsigned spins = 0; spins < (1u << 27); ++spins) {
            const unsigned seen = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(
                &pair_arrivals[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            met = (int)(seen - pair_target) >= 0;
            if (met) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if (!met && lane == 0) atomicAdd(&g_wait_timeouts, 1ull);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    auto group_sync = [&]() {
        if (nw == 1) wave_sync();
        else if (nw == NW) __syncthreads();
        else pair_sync();
    };
    double *m, *mn, *wid, *pred, *ghint;
    float* hlim;
    const auto bind_slot = [&]() {
        m = m_[g];
        mn = mn_[g];
        wid = wid_[g];
        pred = pred_[g];
        hlim = hlim_[g];
        ghint = ghint_[g];
    };
    bind_slot();

    const long long t_begin = wall_clock64();
    const int nlmax = p.n_layers, NP = 2 * nlmax + 1, P = p.n_periods, ND = data_length(p);
    int nl = layers_of(p, ib);
    LayerTable& T = g_T;
    if (lane == 0) {
        g_PS.prof_slot = ib;
        g_PS.abort = 0;
    }

    if (threadIdx.x < NP) {
        lo[threadIdx.x] = p.bounds[3 * threadIdx.x];
        hi[threadIdx.x] = p.bounds[3 * threadIdx.x + 1];
        rg[threadIdx.x] = p.bounds[3 * threadIdx.x + 2];
    }
    for (int k = threadIdx.x; k < P; k += 64 * NW) t[k] = p.periods[k];
    if (threadIdx.x < 2) {
        pair_arrivals[threadIdx.x] = 0u;
        help_wait_[threadIdx.x] = 0;
        help_state_[threadIdx.x] = 0;
    }
    __syncthreads();  // the only barrier that chains sharing a workgroup share from the start
    if (idle && !pairs_wg) return;
    int pass = 0;      // 1: this wave runs candidates of the workgroup's other chain
    bool run = !idle;  // (a pair of waves without a chain of its own goes straight to helping)
again:
    if (run) {
    const ChainShare sh = share_[g];  // (read by the waves that were taken on only)
    if (pass == 0 && writer) {
        if (lane < NP) {
            m[lane] = ch.m[(size_t)ib * NP + lane];
            wid[lane] = ch.width[(size_t)ib * NP + lane];
            mn[lane] = (m[lane] - lo[lane]) / rg[lane];
            hlim[lane] = (float)HINT_SECANT_CAUCHY;
        }
        for (int k = lane; k < ND; k += 64) pred[k] = ch.pred[(size_t)ib * ND + k];
        if (lane < 5) cbase_[g][lane] = ch.counters[(size_t)ib * 5 + lane];
    }
    double logL = pass ? sh.logL : uniform_value(ch.logL[ib]);
    const double beta = uniform_value(ch.beta ? ch.beta[ib] : 1.0);
    const double* R = (rotate & MHD_ROT_FRAME) ? ch.rot + (size_t)ib * NP * NP : nullptr;
    const ProposalFrame frame = {lo, hi, rg, wid, R, nl, NP, nlmax,
                                 (rotate & MHD_PROPOSE_UNIFORM) != 0};
    int c_acc = 0, c_rej = 0, c_bnd = 0, c_fm = 0;
    unsigned long long lit = 0, exec = 0;
    // evaluation rounds of the forwards of the chain's first candidate (the one that is always
    // needed and never given up), and how many there were: the chain's cost for the schedule
    unsigned root_rounds = 0, root_forwards = 0;
    // how the curves of this chain are searched (see SearchPolicy); a chain goes on at the
    // level its last launch ended with
    const bool keeps = ch.sched && B <= SCHED_MAX_CHAINS;
    // the chain's current phase curve guides the predictor when the chain keeps one; for group
    // velocity the two phase curves behind it do (ghint)
    const bool phase_hint = p.itype == MHD_PHASE || p.itype == MHD_PHASE_AND_ELL;
    const bool group_kind = p.itype == MHD_GROUP || p.itype == MHD_PHASE_AND_GROUP;
    SearchPolicy policy = {search == 2 ? 0 : 3, 0, phase_hint ? 0 : 3};
    // share of the chain's forwards that were accepted lately (which candidates a round runs
    // depends on it, see below); kept between launches beside the search level
    // (a running mean that starts from SPEC_PACC_START, or from what the last launch left,
    // and becomes an exponential average over the last SPEC_PACC_WINDOW forwards: quick to
    // learn in a chain's first steps, steady in a long run)
    float pacc = SPEC_PACC_START;
    int pacc_n = 2;
    if (keeps) {
        const int q = (int)sched_view(ch.sched, B).level[ib] >> 2;
        if (q > 0) {
            pacc = (float)q * (1.0f / 63.0f);
            pacc_n = SPEC_PACC_KEPT;
        }
    }
    if (search == 2 && keeps) {
        int lv = (int)sched_view(ch.sched, B).level[ib] & 3;
        // (chains with a guide keep level 1 only: the sequential start of their curves)
        const int lv_max = phase_hint ? 1 : policy.max_level;
        if (lv > lv_max) lv = lv_max;
        policy.level = lv;
        policy.left = lv > 0 ? 8 : 0;
    }
#ifdef MHD_PROFILE
    const long long k_w0 = wall_clock64();
    if (writer && lane == 0 && ib < 2048) g_times[2 * ib] = (unsigned long long)k_w0;
    if (writer && lane == 0 && ib < 1024) {
        // where this workgroup runs: HW_ID (register 4) and XCC_ID (register 20)
        const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
        const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));
        (void)hw; (void)xcc;
    }
#endif
    // Group velocity: the two phase curves of the current state guide the predictor as the
    // current curve does for phase velocity, but they are not part of a chain's state: the
    // chain's first wave computes them once per launch (and on every acceptance they are the
    // candidate's).
    if (pass == 0 && writer && lane == 0) ghint_ok_[g] = 0;
    if (pass == 0 && group_kind && p.mode == 0 && search == 2 && writer) {
        wave_sync();
        build_layer_table(T, m, nl, p.vpvs, lane, nlmax);
        EvalCount extra = {0, 0, 0};
        const int st0 = forward_predicted(T, p.wave, p.mode, MHD_GROUP, t, P, p.dc, p.dt, newpred[w],
                                          scratch[w], extra, 0, t, t, 0);
        wave_sync();
        if (st0 == MHD_ST_OK)
            for (int k = lane; k < 2 * P; k += 64) ghint[k] = scratch[w][k];
        if (lane == 0) ghint_ok_[g] = (st0 == MHD_ST_OK);
        exec += extra.exec;
        wave_sync();
    }
    // (waves that were taken on meet the chain's own here, see the end of the round loop)
    if (pass == 0) group_sync(); else __syncthreads();
    const bool hinted = pass ? sh.hinted != 0 : p.mode == 0 && (phase_hint || ghint_ok_[g] != 0);
    // (hinted chains: level 1 stands for the sequential start of the curves, see DENSE_REPAIRS_UP)
    float dense_avg = 0.0f;  // repairs per curve of the first candidate, lately
    if (search == 2 && hinted) {
        policy.max_level = HINTED_MAX_LEVEL;
        if (policy.level > 1) policy.level = 1;
        policy.left = policy.level > 0 ? DENSE_HOLD : 0;
        dense_avg = policy.level > 0 ? 2.0f * DENSE_REPAIRS_UP : 0.0f;
    }
    if (pass) {
        pacc = sh.pacc;
        pacc_n = sh.pacc_n;
        policy.level = sh.level;
        policy.left = sh.left;
        policy.max_level = sh.max_level;
        dense_avg = sh.dense_avg;
    }

    // everything one step leaves behind besides the state itself (the chain's first wave writes)
    auto book = [&](int j, int st, int acc) {
        c_acc += acc;
        c_rej += !acc;
        c_bnd += (st == MHD_ST_BOUNDS);
        c_fm += (st >= MHD_ST_NOROOT);
        if (!writer) return;
        if (lane == 0) {
            if (ch.counters_ind) {
                const int kk = idx[(size_t)j * B + ib];
                int64_t* ci = ch.counters_ind + ((size_t)ib * NP + kk) * 5;
                ci[0] += acc;
                ci[1] += !acc;
                ci[2] += (st == MHD_ST_BOUNDS);
                ci[4] += (st >= MHD_ST_NOROOT);
            }
            if (accepted) accepted[(size_t)j * B + ib] = (uint8_t)acc;
            if (status) status[(size_t)j * B + ib] = st;
        }
        if (hist_every > 0 && (j + 1) % hist_every == 0) {
            const size_t h = (size_t)((j + 1) / hist_every - 1);
            if (hist_m && lane < NP) hist_m[(h * B + ib) * NP + lane] = m[lane];
            if (hist_logL && lane == 0) hist_logL[h * B + ib] = logL;
            if (hist_pred)
                for (int k = lane; k < ND; k += 64) hist_pred[(h * B + ib) * ND + k] = pred[k];
            record_diag(dg, ch, B, ib, NP, h, cbase_[g], c_acc, c_rej, c_bnd, c_fm, lane,
                        (long long)lit);
        }
    };
    // the start of step j: update_covariance_matrix sees the state the step starts from
    auto begin_step = [&]() {
        if ((rotate & MHD_ROT_ACCUMULATE) && writer)
            cov_accumulate_wave(ch, ib, NP, mn, tmp[w0], lane);
    };
    // accept test (model.py:449-467) of a candidate for step j with likelihood lnew
    // (settling: the test decides a step of the chain, not only which curves are given up)
    auto accept_test = [&](double lnew, int j, bool settling) -> int {
        double dl = lnew - logL;
        dl = dl / Temp;
        if (ch.beta) dl = beta * dl;
        const double u = u_acc[(size_t)j * B + ib], edl = exp(dl);
        if (settling && writer && lane == 0 && close_decision(u, edl))
            atomicAdd(&g_close_decisions, 1ull);
        return __builtin_amdgcn_readfirstlane((int)(u <= edl));
    };
    // the candidate of wave c settles step j: the accept test and, if it passes, the move
    auto settle = [&](int c, int j) -> int {
        const int st = r_st[c];
        int acc = 0;
        if (st == MHD_ST_OK) {
            const double lnew = r_logL[c];
            acc = accept_test(lnew, j, true);
            if (acc) {
                if (writer) {
                    if (lane < NP) {
                        ch.params_prev[(size_t)ib * NP + lane] = mn[lane];
                        if (frame.used(lane)) m[lane] = test[c][lane];
                        mn[lane] = (m[lane] - lo[lane]) / rg[lane];
                    }
                    for (int k = lane; k < ND; k += 64) pred[k] = newpred[c][k];
                    if (group_kind)
                        for (int k = lane; k < 2 * P; k += 64) ghint[k] = scratch[c][k];
                    wave_sync();
                }
                logL = uniform_value(lnew);
            }
        }
        book(j, st, acc);
        return acc;
    };

    // a scheduled launch goes on where the call's earlier launches left the chain and ends at
    // the deadline, if there is one
    int s = 0;
    unsigned long long deadline = 0;
    if (scheduled) {
        const SchedView sv = sched_view(ch.sched, B);
        s = sv.progress[ib];
        deadline = (unsigned long long)(unsigned)sv.n4[2] | ((unsigned long long)(unsigned)sv.n4[3] << 32);
    }
    if (pass) s = sh.s;
    const int s_begin = s;
#ifdef MHD_PROFILE
    // where the chain's last launch of the call started: step, candidates
    if (writer && lane == 0 && ib < 1024 && s < S && s > 0)
        g_blk[ib * 8 + 5] = ((unsigned long long)s << 32) | (unsigned long long)nw;
    if (writer && lane == 0 && ib < 1024 && s == 0) g_times[2 * (1024 + ib)] = (unsigned long long)wall_clock64();
#endif
    int round_no = pass ? sh.round_no : 0;
    while (s < S) {
        // ---- this wave's candidate: the (w-w0+1)-th proposal from step s on that passes the
        // bounds
#ifdef MHD_PROFILE_STAGES
        const long long _r0 = clock64();
#endif
        // Which proposals the waves of the chain work on.  A candidate is a node of the tree of
        // what the chain may do next: its path says, for each candidate before it on the way
        // from the current state, whether that one is taken to be accepted (bit i of the path:
        // the i-th) -- its proposal then starts from that candidate's parameters -- or rejected.
        // The first wave takes the root (the next proposal from the current state); the others
        // take the likeliest nodes given the share of forwards the chain accepted lately:
        // after rejections only (R, RR, RRR) for a chain that rejects most, after acceptances
        // (A, AA, AAA) for one that accepts most -- two or three accepted steps then settle in
        // one round -- and mixtures in between.  Every node's draws are those of its steps, so
        // whichever path the settlement follows, the chain is the sequential one.
        int nlen[NW], nbits[NW];
        nlen[0] = 0;
        nbits[0] = 0;
        // (the test build runs every shape of the tree in turn, whatever the chain accepts)
        // (group velocity is three curves per forward, and the guide of a candidate after an
        // acceptance is poorer for each: such candidates have to be likelier to be worth it)
        const float psel = ABORT_SABOTAGE ? 0.2f + 0.25f * (float)(round_no & 3)
                                          : pacc - (group_kind ? SPEC_GROUP_PENALTY : 0.0f);
        if (NW == 2 || nw <= 2) {
            nlen[1] = 1;
            nbits[1] = (SPEC_TREE && psel > 0.5f + SPEC_BIAS) ? 1 : 0;
#pragma unroll
            for (int q = 2; q < NW; ++q) {
                nlen[q] = 0;
                nbits[q] = 0;
            }
        } else {
            const int kind = !SPEC_TREE ? 0 : (psel < SPEC_FIRST_A ? 0 : (psel < 0.5f + SPEC_BIAS ? 1 : (psel < 0.68f + SPEC_BIAS ? 2 : 3)));
            // kind 0: R RR RRR   1: R RR A   2: A R AA   3: A AA AAA
            nlen[1] = 1;
            nbits[1] = kind >= 2 ? 1 : 0;
            nlen[2] = kind == 2 ? 1 : 2;
            nbits[2] = kind == 3 ? 3 : 0;
            nlen[3] = kind == 1 ? 1 : (kind == 2 ? 2 : 3);
            nbits[3] = kind == 0 ? 0 : (kind == 1 ? 1 : (kind == 2 ? 3 : 7));
        }
        const int wi = (w - w0) & (NW - 1);
        int mylen = 0, mybits = 0;
#pragma unroll
        for (int q = 0; q < NW; ++q)
            if (q == wi) {
                mylen = nlen[q];
                mybits = nbits[q];
            }
        // walk the path: the proposals of the candidates before this one, from the states the
        // path gives them, then this wave's own
        int jc = S, kkc = 0;
        double czc = 0.0;
        bool small_moves = true;  // every move on the way is one that predict_hint copes with
        {
            const double* cur_m = m;
            const double* cur_mn = mn;
            int pos = s;
            for (int i = 0; i <= mylen; ++i) {
                int ji = S;
                for (int j = pos; j < S; ++j) {
                    const int kk = idx[(size_t)j * B + ib];
                    const double cz = cauchy[(size_t)j * B + ib];
                    if (propose_wave(frame, cur_mn, kk, cz, test[w], tmp[w], lane) == MHD_ST_OK) {
                        ji = j;
                        czc = cz;
                        kkc = kk;
                        break;
                    }
                }
                if (ji >= S) break;  // the steps run out on the way: nothing for this wave
                if (i == mylen) {
                    jc = ji;
                    break;
                }
                if ((mybits >> i) & 1) {
                    small_moves = small_moves && !frame.uniform && fabs(czc) <= hlim[kkc];
                    // taken to be accepted: the state moves as the settlement would move it
                    if (lane < NP) {
                        const double mv = frame.used(lane) ? test[w][lane] : cur_m[lane];
                        mw_[w][lane] = mv;
                        mnw_[w][lane] = (mv - lo[lane]) / rg[lane];
                    }
                    wave_sync();
                    cur_m = mw_[w];
                    cur_mn = mnw_[w];
                }
                pos = ji + 1;
            }
        }
        int st = MHD_ST_BOUNDS, trouble = 0;
        double lnew = 0.0;
        EvalCount cnt = {0, 0, 0};
        if (lane == 0) g_PS.abort = (ABORT_SABOTAGE && wi > 0 && round_no % 3 == 1) ? 1 : 0;
        ++round_no;
        if (jc < S) {
            build_layer_table(T, test[w], nl, p.vpvs, lane, nlmax);
            if (policy.predicted()) {
                st = forward_predicted(T, p.wave, p.mode, p.itype, t, P, p.dc, p.dt, newpred[w],
                                       scratch[w], cnt,
                                       hinted ? (policy.level > 0 ? -DENSE_PREFIX(P) : 0)
                                              : policy.stepwise(),
                                       pred, ghint,
                                       // (after an acceptance on the path the current curve is
                                       // two moves away: a guide for the windows only)
                                       !hinted ? 0 : ((mybits == 0 || SPEC_HINT_SMALL) && small_moves && fabs(czc) <= hlim[kkc] ? 2 : 1));
                trouble = st == MHD_ST_ABORTED ? 0 : __builtin_amdgcn_readfirstlane(g_PS.trouble);
                // (the waves of a chain update the limits unordered; they are heuristics)
                if (lane == 0 && st != MHD_ST_ABORTED)
                    hint_limit_update(hlim, kkc, czc, __builtin_amdgcn_readfirstlane(g_PS.hinted));
            } else {
                st = forward_wave(T, p.wave, p.mode, p.itype, t, P, p.dc, p.dt, newpred[w],
                                  scratch[w], cnt);
            }
            wave_sync();
            if (st == MHD_ST_OK) lnew = loglik_wave(p, newpred[w], scratch[w] + 2 * MAXP, lane);
            exec += cnt.exec;
            if (mylen == 0) {
                root_rounds += cnt.rounds;
                root_forwards += 1;
            }
            // A candidate that passes its accept test ends the round's steps wherever it is
            // reached, and if it is not reached an earlier one did: either way the candidates
            // after it are not needed, and their waves are told to give their curves up.
            // (For a candidate that follows rejections only the accept test is known here, the
            // chain's likelihood being the current one.  Passed: whatever takes it to be
            // rejected is moot -- if it is reached it is accepted, if not, an earlier one was.
            // Failed: whatever takes it to be accepted is moot, by the same argument.)
            if (ABORT_MOOT && nw > 1 && mybits == 0 && st != MHD_ST_ABORTED && !ABORT_SABOTAGE) {
                const int out = (st == MHD_ST_OK) ? accept_test(lnew, jc, false) : 0;
#pragma unroll
                for (int q = 0; q < NW; ++q)
                    if (q != wi && q < nw && nlen[q] > mylen &&
                        (nbits[q] & ((1 << mylen) - 1)) == 0 && ((nbits[q] >> mylen) & 1) != out &&
                        lane == 0)
                        g_PS_OF(wave_of(q)).abort = 1;
            }
            if (ABORT_SABOTAGE && lane == 0)
                for (int q = wi + 1; q < nw; ++q) g_PS_OF(wave_of(q)).abort = 1;
#ifdef MHD_ABORT_SABOTAGE
            if (st == MHD_ST_ABORTED && lane == 0) atomicAdd(&g_abort_stats[0], 1ull);
#endif
        }
        if (lane == 0) {
            r_step[w] = jc;
            r_st[w] = st;
            r_trouble[w] = trouble;
            r_logL[w] = lnew;
            r_lit[w] = cnt.lit;
        }
#ifdef MHD_PROFILE_STAGES
        const long long _b0 = clock64();
#endif
        group_sync();
#ifdef MHD_PROFILE_STAGES
        if (lane == 0) {
            atomicAdd(&g_prof[9], (unsigned long long)(clock64() - _b0));   // wait for the partner
            atomicAdd(&g_prof[10], (unsigned long long)(_b0 - _r0));        // scan + forward
        }
#endif

        // ---- settle the steps in order (all waves of the chain follow; the first one writes)
        int ran = 0;
        bool trouble2 = false;
        for (int q = 0; q < nw; ++q) {
            const int c = wave_of(q);
            ran += (r_step[c] < S);
            trouble2 |= r_trouble[c] >= (hinted ? HINTED_TROUBLE_MIN : 2);
        }
        // follow the tree from the root by what really happens
        int cl = 0, cb = 0;
        for (int n = 0; n < nw; ++n) {
            int c = -1;
#pragma unroll
            for (int q = 0; q < NW; ++q)
                if (q < nw && nlen[q] == cl && nbits[q] == cb) c = wave_of(q);
            if (c < 0) break;         // nobody ran this node
            const int j = r_step[c];  // <= S
            for (; s < j; ++s) {
                begin_step();
                book(s, MHD_ST_BOUNDS, 0);
            }
            if (j >= S) break;
            // (a curve that was given up is never reached: its node is off the path the chain
            // takes.  Should it be, the step is simply run again in the next round.)
            if (r_st[c] == MHD_ST_ABORTED) {
#ifdef MHD_ABORT_SABOTAGE
                if (writer && lane == 0) atomicAdd(&g_abort_stats[1], 1ull);
#endif
                break;
            }
            begin_step();
            lit += r_lit[c];
            s = j + 1;
            const int acc = settle(c, j);
#ifdef MHD_ABORT_SABOTAGE
            if (writer && lane == 0 && cb != 0) atomicAdd(&g_abort_stats[2], 1ull);
#endif
            pacc_n += pacc_n < SPEC_PACC_WINDOW;
            pacc += ((float)acc - pacc) / (float)pacc_n;
            cb |= acc << cl;
            ++cl;
        }
#ifdef MHD_PROFILE
        if (writer && lane == 0 && ib < 1024) {
            g_blk[ib * 8 + 7] += 1ull << 40;                       // rounds
            g_blk[ib * 8 + 7] += (unsigned long long)ran << 20;    // forwards run
            g_blk[ib * 8 + 7] += (policy.level == 3) ? ran : 0;    // of which wave-sequential
        }
#endif
        if (search == 2 && hinted) {
            // (the first candidate's curve is always needed and never given up)
            if (r_step[w0] < S && r_st[w0] != MHD_ST_ABORTED)
                dense_avg += ((float)r_trouble[w0] - dense_avg) * (1.0f / DENSE_WINDOW);
            if (policy.level == 0) {
                // (the test build sends every chain there now and then, whatever its curves cost)
                if (dense_avg > DENSE_REPAIRS_UP || (ABORT_SABOTAGE && (round_no & 15) == 15)) {
                    policy.level = 1;
                    policy.left = ABORT_SABOTAGE ? 12 : DENSE_HOLD;
                }
            } else {
                policy.left -= ran;
                if (policy.left <= 0) {
                    policy.level = 0;
                    dense_avg = 0.9f * DENSE_REPAIRS_UP;
                }
            }
        } else if (search == 2) {
            policy.update(ran, trouble2);
        }
        if (writer && lane == 0) {
            const int stop = deadline != 0 && (unsigned long long)wall_clock64() >= deadline;
            r_stop[w0] = stop;
            // the other two waves of the workgroup are free: from the next round on they run
            // two more candidates of this chain
            int adopt = 0;
            if (NW == 4 && pairs_wg && nw == 2 && !stop && S - s >= ADOPT_MIN_STEPS &&
                __hip_atomic_load(&help_wait_[1 - g], __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_WORKGROUP) != 0) {
                adopt = 1;
                ChainShare& out = share_[g];
                out.ib = ib;
                out.s = s;
                out.round_no = round_no;
                out.pacc_n = pacc_n;
                out.level = policy.level;
                out.left = policy.left;
                out.max_level = policy.max_level;
                out.hinted = hinted;
                out.pacc = pacc;
                out.dense_avg = dense_avg;
                out.logL = logL;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __hip_atomic_store(&help_state_[g], 1, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            r_adopt[w0] = adopt;
        }
        group_sync();
        if (r_stop[w0]) break;
        if (NW == 4 && r_adopt[w0]) {
            // (set for a chain with two waves only; the waves taken on wait in this barrier,
            // which is the workgroup's from here on)
            nw = NW;
            __syncthreads();
        }
    }

#ifdef MHD_PROFILE
    if (writer && lane == 0 && ib < 2048) g_times[2 * ib + 1] = (unsigned long long)wall_clock64();
#endif
#ifdef MHD_PROFILE
    // the call's first launch: steps done, ticks taken, waves per workgroup, split, candidates
    if (writer && lane == 0 && ib < 1024 && s_begin == 0)
        g_times[2 * (1024 + ib) + 1] = ((unsigned long long)s << 48) |
                                       (((unsigned long long)(wall_clock64() - t_begin) & 0xFFFFFFFFull) << 12) |
                                       ((unsigned long long)NW << 4) | ((unsigned long long)(nw == 1) << 3) | (unsigned long long)nw;
#endif
    if (writer && lane == 0 && keeps) {
        {
            int q = (int)(pacc * 63.0f + 0.5f);
            q = q < 1 ? 1 : (q > 63 ? 63 : q);
            const int lv = search == 2 ? policy.level : ((int)sched_view(ch.sched, B).level[ib] & 3);
            sched_view(ch.sched, B).level[ib] = (unsigned char)(lv | (q << 2));
        }
        if (scheduled) sched_view(ch.sched, B).progress[ib] = s;
        if (s > s_begin && root_forwards > 0) {
            // evaluation rounds per forward times the share of the steps that had one (with the
            // prior's weight while the chain has shown few steps); a launch that did only part
            // of the call's steps counts by that part
            float* cost = sched_view(ch.sched, B).cost;
            const float done = (float)(s - s_begin);
            const float pvalid = (done - (float)c_bnd + SCHED_PVALID_PRIOR * SCHED_PVALID_WEIGHT) /
                                 (done + SCHED_PVALID_WEIGHT);
            const float now = (float)root_rounds / (float)root_forwards * pvalid;
            const float old = cost[ib];
            cost[ib] = old > 0.0f ? old + SCHED_GAIN * (done / (float)S) * (now - old) : now;
            // what a round took, for the deadlines of the next call (all chains together)
            float* calib = (float*)(sched_view(ch.sched, B).n4 + 6);
            atomicAdd(&calib[0], (float)(wall_clock64() - t_begin));
            atomicAdd(&calib[1], (float)root_rounds);
        } else if (s > s_begin && !(sched_view(ch.sched, B).cost[ib] > 0.0f)) {
            // no forward model yet (every proposal so far left the bounds): an ordinary chain
            sched_view(ch.sched, B).cost[ib] = SCHED_COST_PRIOR;
        }
    }
    if (writer) {
        if (lane < NP) ch.m[(size_t)ib * NP + lane] = m[lane];
        for (int k = lane; k < ND; k += 64) ch.pred[(size_t)ib * ND + k] = pred[k];
    }
    if (lane == 0) {
        if (writer) {
            ch.logL[ib] = logL;
            int64_t* c5 = ch.counters + (size_t)ib * 5;
            c5[0] += c_acc;
            c5[1] += c_rej;
            c5[2] += c_bnd;
            c5[4] += c_fm;
#ifdef MHD_LIT_DEBUG
            c5[3] += (int64_t)lit;
#endif
            if (n_evals) atomicAdd(&n_evals[0], lit);
        }
        if (n_evals) atomicAdd(&n_evals[1], exec);
    }
    }  // if (run)
    if (NW == 4 && pairs_wg && pass == 0 && nw == 2) {
        // This pair of waves is done with its own chain (or had none): it says so, and waits
        // for the workgroup's other chain to take it on or to end.  (That chain ends, or takes
        // the waves on, after the round it is in; the limit only keeps a defect from hanging
        // the GPU.)
        if (w == w0 && lane == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __hip_atomic_store(&help_state_[g], 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&help_wait_[g], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        int other = 0;
        for (unsigned spins = 0; spins < (1u << 25); ++spins) {
            other = __builtin_amdgcn_readfirstlane(__hip_atomic_load(
                &help_state_[1 - g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            if (other != 0) break;
            __builtin_amdgcn_s_sleep(8);
        }
        if (other == 0 && lane == 0) atomicAdd(&g_wait_timeouts, 1ull);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (other == 1) {
            g = 1 - g;
            w0 = 2 * g;
            nw = NW;
            writer = false;
            ib = __builtin_amdgcn_readfirstlane(share_[g].ib);
            nl = layers_of(p, ib);
            bind_slot();
            pass = 1;
            run = true;
            goto again;
        }
    }
}

}  // namespace

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
namespace {

constexpr size_t sched_bytes_of(size_t B)
{
    return sizeof(float) * B + sizeof(int) * (8 + 2 * B) + B + 64;
}

}  // namespace

// Launch-schedule scratch of mhd_mh_steps_static: what mhd_chains_t.sched is for mhd_mh_steps,
// for up to STATIC_MAX_CHAINS chains, one copy per GPU the library runs on.  In host code the
// name only stands for the symbol: the address on a device comes from hipGetSymbolAddress
// (DeviceState.scratch), and only that pointer is handed to kernels and memsets.
constexpr int STATIC_MAX_CHAINS = SCHED_MAX_CHAINS;
constexpr size_t STATIC_SCRATCH_BYTES = (sched_bytes_of(STATIC_MAX_CHAINS) + 255) / 256 * 256;
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];

namespace {

// What the library keeps per device, made by mhd_init() and kept until the process ends: the
// number of compute units and the address of the static scratch.  `mu` serialises the set-up
// and, for mhd_mh_steps_static, the decision whose chains the scratch belongs to with the
// launches that follow it.
struct DeviceState {
    std::mutex mu;
    bool ready = false;
    int cus = 0;
    void* scratch = nullptr;              // this device's copy of `scratch` (mhd_mh_steps_static)
    int scratch_B = -1;                   // what the scratch was last laid out for: chains ...
                                          // (SCRATCH_FRESH: cleared by mhd_static_reset)
    const void* scratch_owner = nullptr;  // ... and whose (mhd_chains_t.m)
};
constexpr int SCRATCH_FRESH = -2;
constexpr int MAX_DEVICES = 64;
DeviceState g_dev[MAX_DEVICES];

// The state of `device` (< 0: the current one), set up if it is not yet and `create` allows it.
DeviceState* device_state(int device, bool create)
{
    hipError_t e = hipSuccess;
    int cur = 0;
    e = hipGetDevice(&cur);
    if (e != hipSuccess) {
        hip_fail(e, "hipGetDevice");
        return nullptr;
    }
    if (device < 0) device = cur;
    if (device >= MAX_DEVICES) {
        fail(MHD_EINVAL, "device index beyond the library's table");
        return nullptr;
    }
    DeviceState& sd = g_dev[device];
    std::lock_guard<std::mutex> lock(sd.mu);
    if (sd.ready) return &sd;
    if (!create) {
        fail(MHD_EINVAL, "mhd_init() has not been called for this device");
        return nullptr;
    }
    if (device != cur) e = hipSetDevice(device);
    if (e == hipSuccess)
        e = hipDeviceGetAttribute(&sd.cus, hipDeviceAttributeMultiprocessorCount, device);
    if (e == hipSuccess) e = hipGetSymbolAddress(&sd.scratch, HIP_SYMBOL(scratch));
    if (device != cur) {
        const hipError_t e2 = hipSetDevice(cur);
        if (e == hipSuccess) e = e2;
    }
    if (e != hipSuccess) {
        hip_fail(e, "mhd_init");
        return nullptr;
    }
    sd.ready = true;
    return &sd;
}

}  // namespace

extern "C" {

size_t mhd_sched_bytes(int32_t B)
{
    if (B < 0) B = 0;
    return sched_bytes_of((size_t)B);
}

int mhd_abi_version(void) { return MHD_ABI_VERSION; }

// Resident workgroups per CU of the two main kernels (diagnostic).
int mhd_wait_timeouts(uint64_t* count)
{
    if (!count) return fail(MHD_EINVAL, "count is NULL");
    unsigned long long n = 0;
    const hipError_t e = hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_wait_timeouts), sizeof(n));
    if (e != hipSuccess) return hip_fail(e, "mhd_wait_timeouts");
    *count = (uint64_t)n;
    return MHD_OK;
}

int mhd_close_decisions(uint64_t* count)
{
    if (!count) return fail(MHD_EINVAL, "count is NULL");
    unsigned long long n = 0;
    const hipError_t e = hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_close_decisions), sizeof(n));
    if (e != hipSuccess) return hip_fail(e, "mhd_close_decisions");
    *count = (uint64_t)n;
    return MHD_OK;
}

int mhd_debug_occupancy(int* mh_blocks, int* fwd_blocks)
{
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(mh_blocks, mh_steps_kernel, 64, 0);
    if (e != hipSuccess) return hip_fail(e, "occupancy");
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(fwd_blocks, forward_kernel, 64, 0);
    return e == hipSuccess ? MHD_OK : hip_fail(e, "occupancy");
}

#ifdef MHD_PROFILE
int mhd_debug_blk(unsigned long long* out8192, int reset)
{
    hipDeviceSynchronize();
    if (out8192) hipMemcpyFromSymbol(out8192, HIP_SYMBOL(g_blk), 8192 * sizeof(unsigned long long));
    if (reset) { static unsigned long long z[8192]; hipMemcpyToSymbol(HIP_SYMBOL(g_blk), z, sizeof(z)); }
    return 0;
}
int mhd_debug_times(unsigned long long* out4096)
{
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out4096, HIP_SYMBOL(g_times), 4096 * sizeof(unsigned long long));
    return 0;
}
int mhd_debug_profile(unsigned long long* out16, int reset)
{
    hipDeviceSynchronize();
    if (out16) hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_prof), 16 * sizeof(unsigned long long));
    if (reset) { unsigned long long z[64] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)); }
    return 0;
}
int mhd_debug_chain(unsigned int* out32768)
{
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out32768, HIP_SYMBOL(g_chain), 32768 * sizeof(unsigned int));
    return 0;
}
int mhd_debug_profile_hi3(unsigned long long* out16)
{
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_prof), 16 * sizeof(unsigned long long), 48 * sizeof(unsigned long long));
    return 0;
}
int mhd_debug_profile_hi2(unsigned long long* out16)
{
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_prof), 16 * sizeof(unsigned long long), 32 * sizeof(unsigned long long));
    return 0;
}
int mhd_debug_profile_hi(unsigned long long* out16)
{
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_prof), 16 * sizeof(unsigned long long), 16 * sizeof(unsigned long long));
    return 0;
}
#endif

#ifdef MHD_ABORT_SABOTAGE
int mhd_debug_aborts(unsigned long long* out2)  // out2[3]
{
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess)
        e = hipMemcpyFromSymbol(out2, HIP_SYMBOL(g_abort_stats), 3 * sizeof(unsigned long long));
    return e == hipSuccess ? MHD_OK : hip_fail(e, "mhd_debug_aborts");
}
#endif

int mhd_init(int32_t device) { return device_state(device, true) ? MHD_OK : MHD_EHIP; }

const char* mhd_last_error(void) { return g_err; }

int mhd_secular_batch(const mhd_problem_t* prob, const double* params, int32_t B,
                      const int32_t* chain, const int32_t* period, const double* c, int32_t N,
                      double* delta, void* stream)
{
    if (int rc = check_problem(prob, false)) return rc;
    if (B < 0 || N < 0) return fail(MHD_EINVAL, "negative size");
    if (N == 0) return MHD_OK;  // (an empty batch may come with NULL arrays)
    if (!params || !chain || !period || !c || !delta) return fail(MHD_EINVAL, "NULL argument");
    hipLaunchKernelGGL(secular_kernel, dim3(N), dim3(64), WAVE_LDS_BYTES, (hipStream_t)stream, *prob, params, B,
                       chain, period, c, N, delta);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MHD_OK : hip_fail(e, "mhd_secular_batch launch");
}

int mhd_debug_secular_pairs(const mhd_problem_t* prob, const double* params, int32_t B,
                            const int32_t* chain, const int32_t* period, const double* c,
                            int32_t N, double* delta, void* stream)
{
    if (int rc = check_problem(prob, false)) return rc;
    if (B < 0 || N < 0) return fail(MHD_EINVAL, "negative size");
    if (N == 0) return MHD_OK;
    if (!params || !chain || !period || !c || !delta) return fail(MHD_EINVAL, "NULL argument");
    hipLaunchKernelGGL(secular_pair_kernel, dim3(N), dim3(64), WAVE_LDS_BYTES, (hipStream_t)stream,
                       *prob, params, B, chain, period, c, N, delta);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MHD_OK : hip_fail(e, "mhd_debug_secular_pairs launch");
}

int mhd_forward_batch(const mhd_problem_t* prob, const double* params, int32_t B, double* vel,
                      int32_t* status, uint64_t* n_evals, void* stream)
{
    if (int rc = check_problem(prob, false)) return rc;
    if (B < 0) return fail(MHD_EINVAL, "negative batch");
    if (B == 0) return MHD_OK;  // (an empty batch may come with NULL arrays)
    if (!params || !vel || !status) return fail(MHD_EINVAL, "NULL argument");
    hipLaunchKernelGGL(forward_kernel, dim3(B), dim3(64), WAVE_LDS_BYTES, (hipStream_t)stream, *prob, params, B,
                       vel, status, (unsigned long long*)n_evals, prob->search ? prob->search : 2);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MHD_OK : hip_fail(e, "mhd_forward_batch launch");
}

int mhd_loglik_batch(const mhd_problem_t* prob, const double* pred, int32_t B, double* logL,
                     void* stream)
{
    if (int rc = check_problem(prob, true)) return rc;
    if (B < 0) return fail(MHD_EINVAL, "negative batch");
    if (B == 0) return MHD_OK;  // (an empty batch may come with NULL arrays)
    if (!pred || !logL) return fail(MHD_EINVAL, "NULL argument");
    hipLaunchKernelGGL(loglik_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, *prob, pred, B,
                       logL);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MHD_OK : hip_fail(e, "mhd_loglik_batch launch");
}

static int check_chains(const mhd_chains_t* ch, int rotate, const mhd_problem_t* prob = nullptr)
{
    if (prob && prob->n_layers_b && (rotate & (MHD_ROT_ACCUMULATE | MHD_ROT_FRAME)))
        return fail(MHD_EINVAL, "rotated proposals are not available with n_layers_b");
    if (!ch) return fail(MHD_EINVAL, "chains is NULL");
    if (!ch->m || !ch->logL || !ch->pred || !ch->params_prev || !ch->width || !ch->counters)
        return fail(MHD_EINVAL, "chain state pointer is NULL");
    if ((rotate & MHD_ROT_FRAME) && !ch->rot) return fail(MHD_EINVAL, "rotate needs rot");
    if ((rotate & MHD_ROT_ACCUMULATE) && (!ch->cov_sum || !ch->n_cov))
        return fail(MHD_EINVAL, "covariance accumulation needs cov_sum and n_cov");
    return MHD_OK;
}

// The launches of one mhd_mh_steps call, for checked arguments, all on the caller's stream.
// sd: the per-device state when the call uses the launch schedule, else nullptr.
static int mh_steps_enqueue(const mhd_problem_t* prob, const mhd_chains_t* chains, int32_t B,
                            int32_t n_steps, const int32_t* idx, const double* cauchy,
                            const double* u_acc, double T, int32_t rotate, int32_t sample_prior,
                            uint8_t* accepted, int32_t* status, int32_t hist_every, double* hist_m,
                            double* hist_logL, double* hist_pred, const mhd_diag_t& dg,
                            uint64_t* n_evals, hipStream_t s, DeviceState* sd)
{
    unsigned long long* ne = (unsigned long long*)n_evals;
    const int search = prob->search ? prob->search : 2;
    const int candidates = prob->candidates ? prob->candidates : 2;
    if (sd) {
        // one launch per phase, of workgroups of four waves that hold one, two or four chains
        // (schedule_kernel says which); the grid covers the most workgroups a phase can need,
        // those beyond what it does need return at once
        const int slots = sd->cus * 4 * MHD_WAVES_PER_SIMD;  // waves of these kernels the GPU holds
        // (a short call is one launch, see SCHED_MIN_STEPS)
        const int phases = n_steps >= SCHED_PHASES_MIN_STEPS ? SCHED_PHASES : 1;
        for (int phase = 0; phase < phases; ++phase) {
            hipLaunchKernelGGL(schedule_kernel, dim3(1), dim3(1024), 0, s, chains->sched, B,
                               n_steps, phase, phases, slots);
            if (2 * B > slots) {
                // more chains than fit at once: a queue of two-candidate workgroups, one chain
                // each (a workgroup of two chains would keep its place until both are done)
                hipLaunchKernelGGL(mh_steps_multi_kernel<2>, dim3(B), dim3(128),
                                   2 * WAVE_LDS_BYTES, s, *prob, *chains, B, n_steps, idx, cauchy,
                                   u_acc, T, rotate, accepted, status, hist_every, hist_m,
                                   hist_logL, hist_pred, dg, ne, search, 1);
                continue;
            }
            const int cap = sched_cap4(B, phase, slots);
            const int grid = cap + (B - cap + 1) / 2 + 2;
            hipLaunchKernelGGL(mh_steps_multi_kernel<4>, dim3(grid), dim3(256),
                               4 * WAVE_LDS_BYTES, s, *prob, *chains, B, n_steps, idx, cauchy,
                               u_acc, T, rotate, accepted, status, hist_every, hist_m, hist_logL,
                               hist_pred, dg, ne, search, 1);
        }
    } else if (candidates == 2 && !sample_prior) {
        hipLaunchKernelGGL(mh_steps_multi_kernel<2>, dim3(B), dim3(128), 2 * WAVE_LDS_BYTES, s,
                           *prob, *chains, B, n_steps, idx, cauchy, u_acc, T, rotate, accepted,
                           status, hist_every, hist_m, hist_logL, hist_pred, dg, ne, search, 0);
    } else if (candidates == 4 && !sample_prior) {
        hipLaunchKernelGGL(mh_steps_multi_kernel<4>, dim3(B), dim3(256), 4 * WAVE_LDS_BYTES, s,
                           *prob, *chains, B, n_steps, idx, cauchy, u_acc, T, rotate, accepted,
                           status, hist_every, hist_m, hist_logL, hist_pred, dg, ne, search, 0);
    } else {
        hipLaunchKernelGGL(mh_steps_kernel, dim3(B), dim3(64), WAVE_LDS_BYTES, s, *prob, *chains,
                           B, n_steps, idx, cauchy, u_acc, T, rotate, sample_prior, accepted,
                           status, hist_every, hist_m, hist_logL, hist_pred, dg, ne, search);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MHD_OK : hip_fail(e, "mhd_mh_steps launch");
}

// Whether a call with these arguments uses the launch schedule (mhd_chains_t.sched).
static bool uses_schedule(const mhd_problem_t* prob, const mhd_chains_t* chains, int32_t B,
                          int32_t sample_prior)
{
    const int candidates = prob->candidates ? prob->candidates : 2;
    return candidates == 2 && !sample_prior && chains->sched && B >= SCHED_MIN_CHAINS &&
           B <= SCHED_MAX_CHAINS;
}

// 0: go on; 1: nothing to do; < 0: refused
static int mh_steps_check(const mhd_problem_t* prob, const mhd_chains_t* chains, int32_t B,
                          int32_t n_steps, const int32_t* idx, const double* cauchy,
                          const double* u_acc, double T, int32_t rotate, int32_t hist_every,
                          const mhd_diag_t* diag)
{
    if (int rc = check_problem(prob, true)) return rc;
    if (int rc = check_chains(chains, rotate, prob)) return rc;
    if (B < 0 || n_steps < 0 || hist_every < 0) return fail(MHD_EINVAL, "negative size");
    if (!(T > 0.0)) return fail(MHD_EINVAL, "T must be positive");
    if (rotate & ~(MHD_ROT_ACCUMULATE | MHD_ROT_FRAME | MHD_PROPOSE_UNIFORM))
        return fail(MHD_EINVAL, "unknown bit in the rotate mask");
    if (diag && diag->hist_counters_ind && !chains->counters_ind)
        return fail(MHD_EINVAL, "hist_counters_ind needs counters_ind");
    if (diag && diag->hist_cov && (!chains->cov_sum || !chains->n_cov))
        return fail(MHD_EINVAL, "hist_cov needs cov_sum and n_cov");
    if (B == 0 || n_steps == 0) return 1;  // (no steps may come with NULL draw arrays)
    if (!idx || !cauchy || !u_acc) return fail(MHD_EINVAL, "draw array is NULL");
    return 0;
}

int mhd_sched_min_steps(void) { return SCHED_PHASES_MIN_STEPS; }

int mhd_mh_steps(const mhd_problem_t* prob, const mhd_chains_t* chains, int32_t B,
                 int32_t n_steps, const int32_t* idx, const double* cauchy, const double* u_acc,
                 double T, int32_t rotate, int32_t sample_prior, uint8_t* accepted,
                 int32_t* status, int32_t hist_every, double* hist_m, double* hist_logL,
                 double* hist_pred, const mhd_diag_t* diag, uint64_t* n_evals, void* stream)
{
    const int rc = mh_steps_check(prob, chains, B, n_steps, idx, cauchy, u_acc, T, rotate,
                                  hist_every, diag);
    if (rc) return rc < 0 ? rc : MHD_OK;
    const mhd_diag_t dg = diag ? *diag : mhd_diag_t{nullptr, nullptr, nullptr};
    if (!uses_schedule(prob, chains, B, sample_prior))
        return mh_steps_enqueue(prob, chains, B, n_steps, idx, cauchy, u_acc, T, rotate,
                                sample_prior, accepted, status, hist_every, hist_m, hist_logL,
                                hist_pred, dg, n_evals, (hipStream_t)stream, nullptr);
    DeviceState* sd = device_state(-1, true);
    if (!sd) return MHD_EHIP;
    return mh_steps_enqueue(prob, chains, B, n_steps, idx, cauchy, u_acc, T, rotate, sample_prior,
                            accepted, status, hist_every, hist_m, hist_logL, hist_pred, dg,
                            n_evals, (hipStream_t)stream, sd);
}

int mhd_static_reset(void* stream)
{
    DeviceState* sd = device_state(-1, false);
    if (!sd) return MHD_EINVAL;
    std::lock_guard<std::mutex> lock(sd->mu);
    const hipError_t e = hipMemsetAsync(sd->scratch, 0, STATIC_SCRATCH_BYTES, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "mhd_static_reset memset");
    sd->scratch_B = SCRATCH_FRESH;
    sd->scratch_owner = nullptr;
    return MHD_OK;
}

int mhd_mh_steps_static(const mhd_problem_t* prob, const mhd_chains_t* chains, int32_t B,
                        int32_t n_steps, const int32_t* idx, const double* cauchy,
                        const double* u_acc, double T, int32_t rotate, int32_t sample_prior,
                        uint8_t* accepted, int32_t* status, int32_t hist_every, double* hist_m,
                        double* hist_logL, double* hist_pred, const mhd_diag_t* diag,
                        uint64_t* n_evals, void* stream)
{
    const int rc = mh_steps_check(prob, chains, B, n_steps, idx, cauchy, u_acc, T, rotate,
                                  hist_every, diag);
    if (rc) return rc < 0 ? rc : MHD_OK;
    if (B > STATIC_MAX_CHAINS)
        return fail(MHD_EINVAL, "mhd_mh_steps_static: more than 4096 chains, the static scratch "
                                "holds the launch schedule of 4096");
    DeviceState* sd = device_state(-1, false);  // made by mhd_init(); nothing is created here
    if (!sd) return MHD_EINVAL;
    const mhd_diag_t dg = diag ? *diag : mhd_diag_t{nullptr, nullptr, nullptr};
    mhd_chains_t ch = *chains;
    ch.sched = sd->scratch;
    // one call at a time per device from here to the last launch: the owner of the scratch, its
    // clearing and the launches that use it belong together
    std::lock_guard<std::mutex> lock(sd->mu);
    if (sd->scratch_B == SCRATCH_FRESH) {
        // cleared by mhd_static_reset(): whoever comes first owns it, nothing to enqueue
        sd->scratch_B = B;
        sd->scratch_owner = (const void*)chains->m;
    } else if (sd->scratch_B != B || sd->scratch_owner != (const void*)chains->m) {
        // other chains than the last call's (by the address of their parameters): their costs
        // and search levels must not be taken for these, and the layout depends on B
        const hipError_t e =
            hipMemsetAsync(sd->scratch, 0, sched_bytes_of((size_t)B), (hipStream_t)stream);
        if (e != hipSuccess) return hip_fail(e, "mhd_mh_steps_static memset");
        sd->scratch_B = B;
        sd->scratch_owner = (const void*)chains->m;
    }
    return mh_steps_enqueue(prob, &ch, B, n_steps, idx, cauchy, u_acc, T, rotate, sample_prior,
                            accepted, status, hist_every, hist_m, hist_logL, hist_pred, dg,
                            n_evals, (hipStream_t)stream,
                            uses_schedule(prob, &ch, B, sample_prior) ? sd : nullptr);
}

int mhd_stepsize_tuning(const mhd_problem_t* prob, const mhd_chains_t* chains, int32_t B,
                        void* stream)
{
    if (int rc = check_problem(prob, false)) return rc;
    if (!chains || !chains->width || !chains->counters)
        return fail(MHD_EINVAL, "chains / width / counters is NULL");
    if (B < 0) return fail(MHD_EINVAL, "negative batch");
    if (B == 0) return MHD_OK;
    const int NP = 2 * prob->n_layers + 1;
    hipLaunchKernelGGL(stepsize_tuning_kernel, dim3((B * NP + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, *chains, B, NP);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MHD_OK : hip_fail(e, "mhd_stepsize_tuning launch");
}

int mhd_tempering_sweep(const double* logL, double* beta, const double* u, int32_t n_temps,
                        int32_t n_ladders, int64_t rung_stride, int64_t ladder_stride,
                        int64_t* counts, void* stream)
{
    if (n_temps < 1 || n_ladders < 0) return fail(MHD_EINVAL, "bad ladder shape");
    if (rung_stride < 1 || ladder_stride < 1) return fail(MHD_EINVAL, "strides must be positive");
    if (n_ladders == 0 || n_temps == 1) return MHD_OK;
    if (!logL || !beta || !u) return fail(MHD_EINVAL, "NULL argument");
    hipLaunchKernelGGL(tempering_sweep_kernel, dim3((n_ladders + 63) / 64), dim3(64), 0,
                       (hipStream_t)stream, logL, beta, u, n_temps, n_ladders,
                       (long long)rung_stride, (long long)ladder_stride, (long long*)counts);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MHD_OK : hip_fail(e, "mhd_tempering_sweep launch");
}

int mhd_cov_accumulate(const mhd_problem_t* prob, const mhd_chains_t* chains, int32_t B,
                       void* stream)
{
    if (int rc = check_problem(prob, true)) return rc;
    if (int rc = check_chains(chains, MHD_ROT_ACCUMULATE, prob)) return rc;
    if (B < 0) return fail(MHD_EINVAL, "negative batch");
    if (B == 0) return MHD_OK;
    hipLaunchKernelGGL(cov_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, *prob, *chains, B);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MHD_OK : hip_fail(e, "mhd_cov_accumulate launch");
}

}  // extern "C"
