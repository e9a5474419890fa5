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

__global__ __launch_bounds__(64) void secular_kernel(mhd_problem_t p, const double* params, int B,
                                                     const int32_t* chain, const int32_t* period,
                                                     const double* c, int N, double* delta)
{
    // one wave per query keeps the table code identical to the other kernels
    LayerTable& T = g_T;
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

__global__ __launch_bounds__(64) void forward_kernel(mhd_problem_t p, const double* params, int B,
                                                     double* vel, int32_t* status,
                                                     unsigned long long* n_evals, int search)
{
    LayerTable& T = g_T;
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
    __syncthreads();
    if (__ballot(nz) == 0ull) return;
    double* cs = ch.cov_sum + (size_t)ib * NP * NP;
    for (int e = lane; e < NP * NP; e += 64) {
        const int i = e / NP, j = e - i * NP;
        cs[e] += diff[i] * diff[j];
    }
    if (lane == 0) ch.n_cov[ib] += 1;
    __syncthreads();
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

__global__ __launch_bounds__(64) void mh_steps_kernel(
    mhd_problem_t p, mhd_chains_t ch, int B, int S, const int32_t* idx, const double* cauchy,
    const double* u_acc, double Temp, int rotate, int sample_prior, uint8_t* accepted,
    int32_t* status, int hist_every, double* hist_m, double* hist_logL, double* hist_pred,
    unsigned long long* n_evals, int search)
{
    LayerTable& T = g_T;
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

        // ---- proposal (model.py:196-222) ----
        if (R) {
            if (lane < NP) {
                double acc = 0.0;
                for (int j = 0; j < NP; ++j) acc += R[j * NP + lane] * mn[j];
                if (lane == kk) acc += wid[kk] * cz;
                tmp[lane] = acc;
            }
            __syncthreads();
            if (lane < NP) {
                double acc = 0.0;
                for (int j = 0; j < NP; ++j) acc += R[lane * NP + j] * tmp[j];
                test[lane] = (acc * rg[lane]) + lo[lane];
            }
        } else if (lane < NP) {
            double v = mn[lane];
            if (lane == kk) v += wid[kk] * cz;
            test[lane] = (v * rg[lane]) + lo[lane];
        }
        __syncthreads();

        // ---- sort_layers (model_params.py:157-176): stable rank sort ----
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
        __syncthreads();
        if (lane < nl) {
            test[rank] = dep;
            test[nl + rank] = vsl;
        }
        __syncthreads();

        // ---- validate (model.py:107-144) ----
        bool inb = true;
        if (lane < NP) inb = (test[lane] >= lo[lane]) && (test[lane] <= hi[lane]);
        int st = (__ballot(!inb) != 0ull) ? MHD_ST_BOUNDS : MHD_ST_OK;
        if (st == MHD_ST_OK && !sample_prior) {
            build_layer_table(T, test, nl, p.vpvs, lane);
            if (search == 2 && penalty == 0) {
                st = forward_predicted(T, p.wave, p.mode, p.itype, t, P, p.dc, p.dt, newpred,
                                       scratch, cnt);
                if (__builtin_amdgcn_readfirstlane(g_PS.trouble) >= 2) penalty = 8;
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
    hipLaunchKernelGGL(mh_steps_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, *prob, *chains,
                       B, n_steps, idx, cauchy, u_acc, T, rotate, sample_prior, accepted, status,
                       hist_every, hist_m, hist_logL, hist_pred, (unsigned long long*)n_evals,
                       g_search);
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
