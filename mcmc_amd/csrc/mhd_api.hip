// mhd_api.hip -- kernels and the extern "C" entry points declared in
// include/mcmc_amd.h.  gfx950 only; one 64-lane wave per chain.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "mhd_device.hpp"
#include "mhd_search2.hpp"

using namespace mhd;

namespace {

thread_local char g_err[256] = "";

// 2 = period-parallel predicted search (default), 1 = wave-sequential search
int g_search = 2;

// candidates evaluated at the same time per chain: 2 (default) or 1
int g_candidates = 2;

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
    if (p->itype != MHD_PHASE && p->itype != MHD_GROUP) return fail(MHD_EINVAL, "bad itype");
    if (p->mode < 0 || p->mode > 64) return fail(MHD_EINVAL, "bad mode");
    if (!(p->dc > 0.0) && !(p->dc < 0.0)) return fail(MHD_EINVAL, "dc must be non-zero");
    if (!p->periods) return fail(MHD_EINVAL, "periods is NULL");
    if (need_data) {
        if (!p->obs || !p->noise_a || !p->bounds)
            return fail(MHD_EINVAL, "obs / noise_a / bounds is NULL");
        if (p->noise_dist != MHD_NORMAL && p->noise_dist != MHD_ASYM_LAPLACE)
            return fail(MHD_EINVAL, "bad noise_dist");
        if (p->noise_dist == MHD_ASYM_LAPLACE && !p->noise_b)
            return fail(MHD_EINVAL, "asym-laplace needs noise_b (kappa)");
        if ((p->n_noise_a != 1 && p->n_noise_a != p->n_periods) ||
            (p->noise_dist == MHD_ASYM_LAPLACE && p->n_noise_b != 1 &&
             p->n_noise_b != p->n_periods))
            return fail(MHD_EINVAL, "noise arrays must have length 1 or n_periods");
    }
    return MHD_OK;
}

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(64) MHD_REGS void secular_kernel(mhd_problem_t p, const double* params, int B,
                                                     const int32_t* chain, const int32_t* period,
                                                     const double* c, int N, double* delta)
{
    // one wave per query keeps the table code identical to the other kernels
    LayerTable& T = g_Tw[threadIdx.x >> 6];
    __shared__ double m[MAXNP];
    const int q = blockIdx.x;
    if (q >= N) return;
    const int lane = threadIdx.x;
    const int NP = 2 * p.n_layers + 1;
    const int ib = chain[q];
    if (lane < NP) m[lane] = params[(size_t)ib * NP + lane];
    __syncthreads();
    build_layer_table(T, m, p.n_layers, p.vpvs, lane);
    const double omega = TWOPI / p.periods[period[q]];
    const double d = secular(T, p.wave, omega, c[q]);
    if (lane == 0) delta[q] = d;
}

__global__ __launch_bounds__(64) MHD_REGS void forward_kernel(mhd_problem_t p, const double* params, int B,
                                                     double* vel, int32_t* status,
                                                     unsigned long long* n_evals, int search)
{
    LayerTable& T = g_Tw[threadIdx.x >> 6];
    __shared__ double m[MAXNP];
    __shared__ double t[MAXP];
    __shared__ double out[MAXP];
    __shared__ double scratch[2 * MAXP];
    const int ib = blockIdx.x;
    const int lane = threadIdx.x;
    const int NP = 2 * p.n_layers + 1, P = p.n_periods;
    if (lane < NP) m[lane] = params[(size_t)ib * NP + lane];
    for (int k = lane; k < P; k += 64) t[k] = p.periods[k];
    __syncthreads();
    build_layer_table(T, m, p.n_layers, p.vpvs, lane);
    EvalCount cnt = {0, 0};
    const int st = (search == 2)
                       ? forward_predicted(T, p.wave, p.mode, p.itype, t, P, p.dc, p.dt, out,
                                           scratch, cnt)
                       : forward_wave(T, p.wave, p.mode, p.itype, t, P, p.dc, p.dt, out, scratch,
                                      cnt);
    __syncthreads();
    for (int k = lane; k < P; k += 64) vel[(size_t)ib * P + k] = (st == MHD_ST_OK) ? out[k] : 0.0;
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
    const int P = p.n_periods;
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

// One proposal (model.py:196-222), sorted by layer depth (model_params.py:157-176) and checked
// against the bounds (model.py:107-144), by one wave: test[] = the candidate parameters made from
// the normalised current parameters mn[].  Returns MHD_ST_OK or MHD_ST_BOUNDS.
struct ProposalFrame {
    const double *lo, *hi, *rg, *wid;  // LDS copies of the bounds and proposal widths
    const double* R;                   // rotation matrix of the chain, or nullptr
    int nl, NP;
};

__device__ inline int propose_wave(const ProposalFrame& f, const double* mn, int kk, double cz,
                                   double* test, double* tmp, int lane)
{
    const int nl = f.nl, NP = f.NP;
    if (f.R) {
        const double* R = f.R;
        if (lane < NP) {
            double acc = 0.0;
            for (int j = 0; j < NP; ++j) acc += R[j * NP + lane] * mn[j];
            if (lane == kk) acc += f.wid[kk] * cz;
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
        if (lane == kk) v += f.wid[kk] * cz;
        test[lane] = (v * f.rg[lane]) + f.lo[lane];
    }
    wave_sync();

    // stable rank sort of the layers by depth
    double dep = 0.0, vsl = 0.0;
    int rank = 0;
    if (lane < nl) {
        dep = test[lane];
        vsl = test[nl + lane];
        for (int j = 0; j < nl; ++j) {
            const double dj = test[j];
            rank += (dj < dep) || (dj == dep && j < lane);
        }
    }
    wave_sync();
    if (lane < nl) {
        test[rank] = dep;
        test[nl + rank] = vsl;
    }
    wave_sync();

    bool inb = true;
    if (lane < NP) inb = (test[lane] >= f.lo[lane]) && (test[lane] <= f.hi[lane]);
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
    unsigned long long* n_evals, int search)
{
    LayerTable& T = g_Tw[threadIdx.x >> 6];
    __shared__ double m[MAXNP];     // current parameters
    __shared__ double mn[MAXNP];    // normalised current parameters
    __shared__ double test[MAXNP];  // proposal
    __shared__ double tmp[MAXNP];
    __shared__ double lo[MAXNP], hi[MAXNP], rg[MAXNP], wid[MAXNP];
    __shared__ double t[MAXP];
    __shared__ double pred[MAXP];
    __shared__ double newpred[MAXP];
    __shared__ double scratch[2 * MAXP];

    const int ib = blockIdx.x;
    const int lane = threadIdx.x;
    const int nl = p.n_layers, NP = 2 * nl + 1, P = p.n_periods;

    if (lane < NP) {
        m[lane] = ch.m[(size_t)ib * NP + lane];
        lo[lane] = p.bounds[3 * lane];
        hi[lane] = p.bounds[3 * lane + 1];
        rg[lane] = p.bounds[3 * lane + 2];
        wid[lane] = ch.width[(size_t)ib * NP + lane];
    }
    for (int k = lane; k < P; k += 64) {
        t[k] = p.periods[k];
        pred[k] = ch.pred[(size_t)ib * P + k];
    }
    double logL = ch.logL[ib];
    const double beta = ch.beta ? ch.beta[ib] : 1.0;
    const double* R = (rotate & MHD_ROT_FRAME) ? ch.rot + (size_t)ib * NP * NP : nullptr;
    const ProposalFrame frame = {lo, hi, rg, wid, R, nl, NP};
    long long c_acc = 0, c_rej = 0, c_bnd = 0, c_fm = 0;
#ifdef MHD_PROFILE
    const long long k_t0 = clock64(), k_w0 = wall_clock64();
    if (lane == 0 && ib < 2048) g_times[2 * ib] = (unsigned long long)k_w0;
#endif
    EvalCount cnt = {0, 0};
    // a chain whose curves keep defeating the predictor (densely spaced modes) uses the
    // wave-sequential search for a while, then tries the predictor again
    int penalty = 0;
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
            build_layer_table(T, test, nl, p.vpvs, lane);
            if (search == 2 && penalty == 0) {
                st = forward_predicted(T, p.wave, p.mode, p.itype, t, P, p.dc, p.dt, newpred,
                                       scratch, cnt);
                if (__builtin_amdgcn_readfirstlane(g_PSw[0].trouble) >= 2) penalty = 8;
            } else {
                st = forward_wave(T, p.wave, p.mode, p.itype, t, P, p.dc, p.dt, newpred, scratch,
                                  cnt);
                if (penalty > 0) --penalty;
#ifdef MHD_PROFILE
                if (lane == 0 && ib < 1024) g_blk[ib * 8 + 7] += 1;
#endif
            }
            __syncthreads();
        }
        int acc = 0;
        if (st == MHD_ST_OK) {
            const double lnew = sample_prior ? 1.0 : loglik_wave(p, newpred, scratch, lane);
            double dl = lnew - logL;
            dl = dl / Temp;
            if (ch.beta) dl = beta * dl;
            acc = ua <= exp(dl);
            if (acc) {
                if (lane < NP) {
                    ch.params_prev[(size_t)ib * NP + lane] = mn[lane];
                    m[lane] = test[lane];
                }
                logL = lnew;
                if (!sample_prior)
                    for (int k = lane; k < P; k += 64) pred[k] = newpred[k];
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
                for (int k = lane; k < P; k += 64) hist_pred[(h * B + ib) * P + k] = pred[k];
        }
    }

#ifdef MHD_PROFILE
    if (lane == 0) {
        atomicAdd(&g_prof[9], (unsigned long long)(clock64() - k_t0));
        atomicAdd(&g_prof[10], (unsigned long long)(wall_clock64() - k_w0));
        if (ib < 2048) g_times[2 * ib + 1] = (unsigned long long)wall_clock64();
    }
#endif
    if (lane < NP) ch.m[(size_t)ib * NP + lane] = m[lane];
    for (int k = lane; k < P; k += 64) ch.pred[(size_t)ib * P + k] = pred[k];
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

// The same chain of Metropolis-Hastings steps with two candidates in flight: a workgroup of two
// waves per chain.  From the current state, wave 0 takes the first proposal that passes the
// bounds (steps before it are rejections that cost nothing) and wave 1 the next one after it;
// both run the forward model at the same time.  The steps are then settled strictly in order:
// wave 1's result is used only if wave 0's candidate was rejected, because only then did its
// proposal start from the state the sequential chain has at that step.  States, likelihoods,
// counters and the draw each step consumes are those of the one-candidate kernel.
__global__ __launch_bounds__(128) MHD_REGS void mh_steps_pair_kernel(
    mhd_problem_t p, mhd_chains_t ch, int B, int S, const int32_t* idx, const double* cauchy,
    const double* u_acc, double Temp, int rotate, uint8_t* accepted, int32_t* status,
    int hist_every, double* hist_m, double* hist_logL, double* hist_pred,
    unsigned long long* n_evals, int search)
{
    __shared__ double m[MAXNP];   // current parameters
    __shared__ double mn[MAXNP];  // normalised current parameters
    __shared__ double lo[MAXNP], hi[MAXNP], rg[MAXNP], wid[MAXNP];
    __shared__ double t[MAXP];
    __shared__ double pred[MAXP];
    __shared__ double test[NWAVE][MAXNP];  // candidate of each wave
    __shared__ double tmp[NWAVE][MAXNP];
    __shared__ double newpred[NWAVE][MAXP];
    __shared__ double scratch[NWAVE][2 * MAXP];
    __shared__ int r_step[NWAVE], r_st[NWAVE], r_trouble[NWAVE];
    __shared__ double r_logL[NWAVE];
    __shared__ unsigned long long r_lit[NWAVE];

    const int ib = blockIdx.x;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nl = p.n_layers, NP = 2 * nl + 1, P = p.n_periods;
    LayerTable& T = g_Tw[w];

    if (w == 0) {
        if (lane < NP) {
            m[lane] = ch.m[(size_t)ib * NP + lane];
            lo[lane] = p.bounds[3 * lane];
            hi[lane] = p.bounds[3 * lane + 1];
            rg[lane] = p.bounds[3 * lane + 2];
            wid[lane] = ch.width[(size_t)ib * NP + lane];
            mn[lane] = (m[lane] - lo[lane]) / rg[lane];
        }
        for (int k = lane; k < P; k += 64) {
            t[k] = p.periods[k];
            pred[k] = ch.pred[(size_t)ib * P + k];
        }
    }
    double logL = ch.logL[ib];
    const double beta = ch.beta ? ch.beta[ib] : 1.0;
    const double* R = (rotate & MHD_ROT_FRAME) ? ch.rot + (size_t)ib * NP * NP : nullptr;
    const ProposalFrame frame = {lo, hi, rg, wid, R, nl, NP};
    long long c_acc = 0, c_rej = 0, c_bnd = 0, c_fm = 0;
    unsigned long long lit = 0, exec = 0;
    int penalty = 0;  // forwards left with the wave-sequential search (see mh_steps_kernel)
    __syncthreads();

    // everything one step leaves behind besides the state itself (wave 0 writes)
    auto book = [&](int j, int st, int acc) {
        c_acc += acc;
        c_rej += !acc;
        c_bnd += (st == MHD_ST_BOUNDS);
        c_fm += (st >= MHD_ST_NOROOT);
        if (w != 0) return;
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
                for (int k = lane; k < P; k += 64) hist_pred[(h * B + ib) * P + k] = pred[k];
        }
    };
    // the start of step j: update_covariance_matrix sees the state the step starts from
    auto begin_step = [&]() {
        if ((rotate & MHD_ROT_ACCUMULATE) && w == 0) cov_accumulate_wave(ch, ib, NP, mn, tmp[0], lane);
    };
    // accept test of a candidate (model.py:449-467) and, if it passes, the move
    auto settle = [&](int c, int j) -> int {
        const int st = r_st[c];
        int acc = 0;
        if (st == MHD_ST_OK) {
            const double lnew = r_logL[c];
            double dl = lnew - logL;
            dl = dl / Temp;
            if (ch.beta) dl = beta * dl;
            acc = u_acc[(size_t)j * B + ib] <= exp(dl);
            if (acc) {
                if (w == 0) {
                    if (lane < NP) {
                        ch.params_prev[(size_t)ib * NP + lane] = mn[lane];
                        m[lane] = test[c][lane];
                        mn[lane] = (m[lane] - lo[lane]) / rg[lane];
                    }
                    for (int k = lane; k < P; k += 64) pred[k] = newpred[c][k];
                    wave_sync();
                }
                logL = lnew;
            }
        }
        book(j, st, acc);
        return acc;
    };

    int s = 0;
    while (s < S) {
        // ---- this wave's candidate: the (w+1)-th proposal from step s on that passes the bounds
        int jc = S, found = 0;
        for (int j = s; j < S; ++j) {
            const int kk = idx[(size_t)j * B + ib];
            const double cz = cauchy[(size_t)j * B + ib];
            if (propose_wave(frame, mn, kk, cz, test[w], tmp[w], lane) == MHD_ST_OK &&
                ++found == w + 1) {
                jc = j;
                break;
            }
        }
        int st = MHD_ST_BOUNDS, trouble = 0;
        double lnew = 0.0;
        EvalCount cnt = {0, 0};
        if (jc < S) {
            build_layer_table(T, test[w], nl, p.vpvs, lane);
            if (search == 2 && penalty == 0) {
                st = forward_predicted(T, p.wave, p.mode, p.itype, t, P, p.dc, p.dt, newpred[w],
                                       scratch[w], cnt);
                trouble = __builtin_amdgcn_readfirstlane(g_PSw[w].trouble);
            } else {
                st = forward_wave(T, p.wave, p.mode, p.itype, t, P, p.dc, p.dt, newpred[w],
                                  scratch[w], cnt);
            }
            wave_sync();
            if (st == MHD_ST_OK) lnew = loglik_wave(p, newpred[w], scratch[w], lane);
            exec += cnt.exec;
        }
        if (lane == 0) {
            r_step[w] = jc;
            r_st[w] = st;
            r_trouble[w] = trouble;
            r_logL[w] = lnew;
            r_lit[w] = cnt.lit;
        }
        __syncthreads();

        // ---- settle the steps in order (both waves follow; wave 0 writes)
        const int jA = r_step[0], jB = r_step[1];
        for (int j = s; j < jA; ++j) {  // jA <= S
            begin_step();
            book(j, MHD_ST_BOUNDS, 0);
        }
        s = jA;
        int ran = 0;
        if (jA < S) {
            begin_step();
            lit += r_lit[0];
            ran = 1 + (jB < S);
            s = jA + 1;
            if (!settle(0, jA)) {
                // wave 1's candidate started from the state that still holds
                for (int j = jA + 1; j < jB; ++j) {  // jB <= S
                    begin_step();
                    book(j, MHD_ST_BOUNDS, 0);
                }
                s = jB;
                if (jB < S) {
                    begin_step();
                    lit += r_lit[1];
                    settle(1, jB);
                    s = jB + 1;
                }
            }
        }
        if (penalty > 0) {
            penalty = (penalty > ran) ? penalty - ran : 0;
        } else if (r_trouble[0] >= 2 || r_trouble[1] >= 2) {
            penalty = 8;
        }
        __syncthreads();
    }

    if (w == 0) {
        if (lane < NP) ch.m[(size_t)ib * NP + lane] = m[lane];
        for (int k = lane; k < P; k += 64) ch.pred[(size_t)ib * P + k] = pred[k];
    }
    if (lane == 0) {
        if (w == 0) {
            ch.logL[ib] = logL;
            int64_t* c5 = ch.counters + (size_t)ib * 5;
            c5[0] += c_acc;
            c5[1] += c_rej;
            c5[2] += c_bnd;
            c5[4] += c_fm;
            if (n_evals) atomicAdd(&n_evals[0], lit);
        }
        if (n_evals) atomicAdd(&n_evals[1], exec);
    }
}

}  // namespace

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" {

int mhd_abi_version(void) { return MHD_ABI_VERSION; }

// Resident workgroups per CU of the two main kernels (diagnostic).
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
    if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)); }
    return 0;
}
#endif

int mhd_set_search(int32_t algorithm)
{
    if (algorithm != 1 && algorithm != 2) return fail(MHD_EINVAL, "algorithm must be 1 or 2");
    g_search = algorithm;
    return MHD_OK;
}

int mhd_set_candidates(int32_t n)
{
    if (n != 1 && n != 2) return fail(MHD_EINVAL, "candidates must be 1 or 2");
    g_candidates = n;
    return MHD_OK;
}

const char* mhd_last_error(void) { return g_err; }

int mhd_secular_batch(const mhd_problem_t* prob, const double* params, int32_t B,
                      const int32_t* chain, const int32_t* period, const double* c, int32_t N,
                      double* delta, void* stream)
{
    if (int rc = check_problem(prob, false)) return rc;
    if (!params || !chain || !period || !c || !delta) return fail(MHD_EINVAL, "NULL argument");
    if (B < 0 || N < 0) return fail(MHD_EINVAL, "negative size");
    if (N == 0) return MHD_OK;
    hipLaunchKernelGGL(secular_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, *prob, params, B,
                       chain, period, c, N, delta);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MHD_OK : hip_fail(e, "mhd_secular_batch launch");
}

int mhd_forward_batch(const mhd_problem_t* prob, const double* params, int32_t B, double* vel,
                      int32_t* status, uint64_t* n_evals, void* stream)
{
    if (int rc = check_problem(prob, false)) return rc;
    if (!params || !vel || !status) return fail(MHD_EINVAL, "NULL argument");
    if (B < 0) return fail(MHD_EINVAL, "negative batch");
    if (B == 0) return MHD_OK;
    hipLaunchKernelGGL(forward_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, *prob, params, B,
                       vel, status, (unsigned long long*)n_evals, g_search);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MHD_OK : hip_fail(e, "mhd_forward_batch launch");
}

int mhd_loglik_batch(const mhd_problem_t* prob, const double* pred, int32_t B, double* logL,
                     void* stream)
{
    if (int rc = check_problem(prob, true)) return rc;
    if (!pred || !logL) return fail(MHD_EINVAL, "NULL argument");
    if (B < 0) return fail(MHD_EINVAL, "negative batch");
    if (B == 0) return MHD_OK;
    hipLaunchKernelGGL(loglik_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, *prob, pred, B,
                       logL);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MHD_OK : hip_fail(e, "mhd_loglik_batch launch");
}

static int check_chains(const mhd_chains_t* ch, int rotate)
{
    if (!ch) return fail(MHD_EINVAL, "chains is NULL");
    if (!ch->m || !ch->logL || !ch->pred || !ch->params_prev || !ch->width || !ch->counters)
        return fail(MHD_EINVAL, "chain state pointer is NULL");
    if ((rotate & MHD_ROT_FRAME) && !ch->rot) return fail(MHD_EINVAL, "rotate needs rot");
    if ((rotate & MHD_ROT_ACCUMULATE) && (!ch->cov_sum || !ch->n_cov))
        return fail(MHD_EINVAL, "covariance accumulation needs cov_sum and n_cov");
    return MHD_OK;
}

int mhd_mh_steps(const mhd_problem_t* prob, const mhd_chains_t* chains, int32_t B,
                 int32_t n_steps, const int32_t* idx, const double* cauchy, const double* u_acc,
                 double T, int32_t rotate, int32_t sample_prior, uint8_t* accepted,
                 int32_t* status, int32_t hist_every, double* hist_m, double* hist_logL,
                 double* hist_pred, uint64_t* n_evals, void* stream)
{
    if (int rc = check_problem(prob, true)) return rc;
    if (int rc = check_chains(chains, rotate)) return rc;
    if (!idx || !cauchy || !u_acc) return fail(MHD_EINVAL, "draw array is NULL");
    if (B < 0 || n_steps < 0 || hist_every < 0) return fail(MHD_EINVAL, "negative size");
    if (!(T > 0.0)) return fail(MHD_EINVAL, "T must be positive");
    if (B == 0 || n_steps == 0) return MHD_OK;
    if (g_candidates == 2 && !sample_prior) {
        hipLaunchKernelGGL(mh_steps_pair_kernel, dim3(B), dim3(128), 0, (hipStream_t)stream, *prob,
                           *chains, B, n_steps, idx, cauchy, u_acc, T, rotate, accepted, status,
                           hist_every, hist_m, hist_logL, hist_pred, (unsigned long long*)n_evals,
                           g_search);
    } else {
        hipLaunchKernelGGL(mh_steps_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, *prob,
                           *chains, B, n_steps, idx, cauchy, u_acc, T, rotate, sample_prior,
                           accepted, status, hist_every, hist_m, hist_logL, hist_pred,
                           (unsigned long long*)n_evals, g_search);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MHD_OK : hip_fail(e, "mhd_mh_steps launch");
}

int mhd_cov_accumulate(const mhd_problem_t* prob, const mhd_chains_t* chains, int32_t B,
                       void* stream)
{
    if (int rc = check_problem(prob, true)) return rc;
    if (int rc = check_chains(chains, MHD_ROT_ACCUMULATE)) return rc;
    if (B < 0) return fail(MHD_EINVAL, "negative batch");
    if (B == 0) return MHD_OK;
    hipLaunchKernelGGL(cov_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, *prob, *chains, B);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MHD_OK : hip_fail(e, "mhd_cov_accumulate launch");
}

}  // extern "C"
