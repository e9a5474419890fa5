/*
 * mcmc_amd.h -- C ABI of the MI355X (gfx950) dispersion-inversion hot path.
 *
 * The reference (jamiebyer/mcmc) is pure Python and defines no FFI; its seam
 * for this path is a pair of Python methods.  Each entry point below names the
 * reference code it replaces (paths relative to the reference repository).
 *
 * Conventions
 *   - every pointer marked "device" is caller-owned HBM (for example
 *     torch.Tensor.data_ptr()); the library allocates no device memory and never
 *     frees, and creates no stream or event: every launch of a call goes to the
 *     caller's stream;
 *   - all arrays are dense row-major, float64 unless stated;
 *   - calls are asynchronous on `stream` (a hipStream_t passed as void*; NULL
 *     is the default stream) and return 0 or a negative MHD_E* code;
 *     mhd_last_error() gives the text of the last failure on this thread;
 *   - threads: any number of host threads may call into the library.  Two calls
 *     that work on the same chains, or on the same `sched`, must be ordered by
 *     the caller (same stream, or events);
 *   - per-chain outcome codes (`status`) map onto the reference's counters:
 *       MHD_ST_OK       proposal evaluated
 *       MHD_ST_BOUNDS   n_bounds_err   (src/inversion/model.py:110-117)
 *       MHD_ST_PHYSICS  n_physics_err  (src/inversion/model.py:120-127)
 *       MHD_ST_NOROOT   n_fm_err, disba DispersionError (src/inversion/model.py:134-140)
 *       MHD_ST_ZERODIV  n_fm_err, ZeroDivisionError      (src/inversion/model.py:134-140)
 *     MHD_ST_ZERODIV is reserved: numba raises ZeroDivisionError where disba
 *     divides by a float zero, which needs a layer velocity or density of
 *     exactly zero; the parameter bounds exclude that, no kernel of this library
 *     returns the code, and no vector of the reference pins it.
 */
#ifndef MCMC_AMD_H
#define MCMC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MHD_ABI_VERSION 8

#define MHD_OK 0
#define MHD_EINVAL (-22) /* bad argument (size beyond a compiled limit, NULL pointer) */
#define MHD_EHIP (-5)    /* a HIP runtime call failed */

#define MHD_ST_OK 0
#define MHD_ST_BOUNDS 1
#define MHD_ST_PHYSICS 2
#define MHD_ST_NOROOT 3
#define MHD_ST_ZERODIV 4

#define MHD_RAYLEIGH 0
#define MHD_LOVE 1
#define MHD_PHASE 0
#define MHD_GROUP 1
#define MHD_PHASE_AND_GROUP 2 /* both curves as one datum of length 2P: [phase, group] */
/* Rayleigh-wave ellipticity ur/uz at the free surface, evaluated at each period's phase velocity
 * (positive = retrograde motion; 0 where the mode has no root at that period), as the second half
 * of a joint datum of length 2P [phase, ellipticity], or alone.  The reference names the datum
 * (README.md:4) and holds an empty EllipticityParams class (src/inversion/model_params.py:212-213)
 * but no call site and no values: PARITY UNPINNED; the values are checked against this
 * repository's oracle, and the oracle against an independent propagator-matrix solution.  Not
 * defined for Love waves (MHD_EINVAL). */
#define MHD_PHASE_AND_ELL 3
#define MHD_ELL 4
#define MHD_NORMAL 0
#define MHD_ASYM_LAPLACE 1

/* compiled limits */
#define MHD_MAX_LAYERS 15   /* finite layers; rows = n_layers + 1 <= 16 */
#ifndef MHD_MAX_PERIODS
#define MHD_MAX_PERIODS 128
#endif

/* rotate flags of mhd_mh_steps */
#define MHD_ROT_ACCUMULATE 1 /* update_covariance_matrix before proposing  */
#define MHD_ROT_FRAME 2      /* propose in the rotated (principal-component) frame */
/* in the same mask: the reference's "uniform" proposal (src/inversion/model.py:205-207)
   instead of the Cauchy one.  cauchy[s,b] then holds the uniform draw u itself and the
   perturbed parameter -- normalised by its bounds, in the rotated frame with MHD_ROT_FRAME --
   is SET to bounds_lo[idx] + u: the reference's arithmetic, which adds the physical lower
   bound to a normalised value.  proposal_width is not used. */
#define MHD_PROPOSE_UNIFORM 4

/*
 * Static description of one inversion problem: the parameterisation of
 * DispersionCurveParams (src/inversion/model_params.py:25-55) and the data /
 * noise model that Model.get_likelihood reads (src/inversion/model.py:420-447).
 */
typedef struct mhd_problem {
    int32_t n_layers;   /* finite layers; NP = 2*n_layers+1 parameters =
                           [n_layers interface depths, n_layers+1 vs]   */
    int32_t n_periods;  /* P; the data vectors (obs, pred, noise) have ND = P entries, or
                           2P for MHD_PHASE_AND_GROUP and MHD_PHASE_AND_ELL (then
                           P <= MHD_MAX_PERIODS / 2)                                  */
    int32_t wave;       /* MHD_RAYLEIGH | MHD_LOVE  (disba wave=)        */
    int32_t mode;       /* 0 = fundamental          (disba mode=)        */
    int32_t itype;      /* MHD_PHASE | MHD_GROUP | MHD_PHASE_AND_GROUP |
                           MHD_PHASE_AND_ELL | MHD_ELL                           */
    int32_t noise_dist; /* MHD_NORMAL | MHD_ASYM_LAPLACE                 */
    int32_t n_noise_a;  /* 1 or ND: length of noise_a                    */
    int32_t n_noise_b;  /* 1 or ND: length of noise_b                    */
    double vpvs;        /* vp = vs * vpvs   (model_params.py:95-97)      */
    double dc;          /* bracketing step, disba default 0.005 km/s     */
    double dt;          /* group-velocity period offset, disba 0.025     */
    const double *periods; /* device [P], ascending (src/main.py:29)     */
    const double *obs;     /* device [ND] data_obs; may be NULL for forward-only calls */
    const double *noise_a; /* device: std (normal) | lambd_scale*lambd clipped at
                              lambd_max (asym-laplace, model.py:438-440)  */
    const double *noise_b; /* device: kappa (asym-laplace) or NULL        */
    const double *bounds;  /* device [NP,3] = lo, hi, hi-lo
                              (model_params.py:104-126); NULL for forward-only */
    /* Models of different layer counts in one batch (the reference names a
       trans-dimensional parameterisation, README.md:16, without code for it; the
       layout is this library's): device int32 [B] or NULL.  Model b has
       n_layers_b[b] <= n_layers finite layers; its row of NP = 2*n_layers+1
       parameters keeps the padded positions: interface depths in slots
       0..n_layers_b[b]-1, vs of the layers in slots n_layers..n_layers+n_layers_b[b]-1
       and of the half-space in slot n_layers+n_layers_b[b]; the other slots are
       ignored and never change.  idx[s,b] of mhd_mh_steps must name a used slot;
       bounds, width and the per-parameter counters are indexed by slot.  Not
       available together with the rotate flags (MHD_EINVAL).  The array belongs
       to the batch of the call it is passed with (B models / B chains). */
    const int32_t *n_layers_b;
    /* Tuning and testing switches without a counterpart in the reference; 0 picks
       the default.  Every setting returns the same curves and the same chains. */
    int32_t search;     /* root search of the forward model: 2 (default) = period-
                           parallel (predict, replay, verify), 1 = wave-sequential */
    int32_t candidates; /* proposals of one chain whose forward models mhd_mh_steps
                           runs at the same time: 2 (default; with `sched` the
                           library also picks 4 or 1 per chain), 4, or 1.  The steps
                           are settled in order.  The candidates are the likeliest
                           next steps of the chain: proposals from the current
                           state (counting if the ones before them are rejected)
                           and, for chains that accept most proposals, proposals
                           from an earlier candidate's state (counting if that one
                           is accepted); a candidate that can no longer count
                           gives its forward model up.  sample_prior uses 1. */
} mhd_problem_t;

/*
 * Mutable state of B chains: the attributes Model.perturb_params mutates
 * (src/inversion/model.py:244-256, :263-282) for every chain at once.
 */
typedef struct mhd_chains {
    double *m;           /* device [B,NP] model_params.model_params           */
    double *logL;        /* device [B]                                        */
    double *pred;        /* device [B,ND] data_pred (ND: see n_periods)        */
    double *params_prev; /* device [B,NP] normalised previous accepted params */
    double *width;       /* device [B,NP] proposal_width                      */
    double *rot;         /* device [B,NP,NP] rotation_matrix, or NULL         */
    double *cov_sum;     /* device [B,NP,NP] cov_mat_sum, or NULL             */
    int64_t *n_cov;      /* device [B], or NULL                               */
    int64_t *counters;   /* device [B,5]: n_acc, n_rej, n_bounds_err,
                            n_physics_err, n_fm_err                           */
    int64_t *counters_ind; /* device [B,NP,5] per-parameter counters
                            (individual_acceptance, model.py:46-58), or NULL  */
    const double *beta;  /* device [B] tempering exponent, or NULL for 1      */
    void *sched;         /* device, mhd_sched_bytes(B) bytes, or NULL: scratch the
                            library keeps between calls of mhd_mh_steps (what each
                            chain's steps cost last time, what share of its
                            proposals it accepted) and within one (how far
                            each chain has got); zero it once, before the first
                            call.  It is used by calls with 64 <= B <= 4096 chains,
                            candidates = 0 or 2 and sample_prior = 0: the few chains
                            that take longest get four candidates from the start
                            (as many of the cheapest ones one instead of two; with
                            fewer chains than half the GPU's wave slots the slots
                            left over go to four candidates for the chains at the
                            head of the order), and
                            a call of at least mhd_sched_min_steps() steps is two
                            launches, the first ending at a deadline, the second
                            giving the chains that are behind more candidates in
                            flight (a shorter call is one launch).  Results do not
                            depend on it; no counterpart in the reference */
} mhd_chains_t;

/* Size of mhd_chains_t.sched for B chains. */
size_t mhd_sched_bytes(int32_t B);

/* The number of steps from which a call of mhd_mh_steps that uses mhd_chains_t.sched is two
 * launches instead of one (a diagnostic for callers that describe their launches; no
 * counterpart in the reference). */
int mhd_sched_min_steps(void);

/*
 * Optional per-step records of mhd_mh_steps beside hist_m / hist_logL / hist_pred, written at
 * the same cadence (after every hist_every-th step), each NULL to skip: what
 * Inversion.store_samples keeps per step besides the state
 * (src/inversion/inversion.py:163-231, :448-487).  The reference stores rates; these are the
 * counts they are made of (rate = count / (n_acc + n_rej), refreshed only while n_rej > 0,
 * src/inversion/model.py:483, :520).
 */
typedef struct mhd_diag {
    int64_t *hist_counters;     /* device [S/hist_every,B,5]: n_acc, n_rej, n_bounds_err,
                                   n_physics_err, n_fm_err of the chain after the step   */
    int64_t *hist_counters_ind; /* device [S/hist_every,B,NP,5]: the per-parameter
                                   counters (needs mhd_chains_t.counters_ind)           */
    double *hist_cov;           /* device [S/hist_every,B,NP,NP]: cov_mat = cov_mat_sum /
                                   n_cov (zeros while n_cov = 0; model.py:278-281);
                                   needs cov_sum and n_cov                              */
} mhd_diag_t;

int mhd_abi_version(void);
const char *mhd_last_error(void);

/*
 * Per-device set-up: the number of compute units (scheduled calls of
 * mhd_mh_steps size their launches by it) and the address of the library's
 * static device scratch (mhd_mh_steps_static).  device < 0 means the current
 * device.  Thread-safe; a second call for the same device does nothing.
 * mhd_mh_steps calls it at its first scheduled call on a device if the caller
 * has not; mhd_mh_steps_static requires it to have been called.  It creates no
 * stream, event or memory.  No counterpart in the reference.
 */
int mhd_init(int32_t device);

/*
 * How often a wait between the waves of a workgroup has run into its limit (some
 * ten seconds) on the current device since the library was loaded: the waves of a
 * chain's candidates meet after every round by counting arrivals in local memory,
 * and a limit keeps a defect from hanging the GPU.  It is 0 in a working library;
 * if it is not, the results of the calls since it was last 0 are not valid.
 * Synchronous (it copies one number from the device): callers ask where they wait
 * for the device anyway.  No counterpart in the reference.
 */
int mhd_wait_timeouts(uint64_t *count);

/*
 * How many accept tests u <= exp(dl) (src/inversion/model.py:449-467) have been
 * settled on the current device, since the library was loaded, with u within 16
 * units in the last place of exp(dl).  The device's exponential and numpy's may
 * differ by a unit or two in the last place, so only such a decision could come
 * out differently in the reference; the library counts them instead of leaving
 * the question open (one in some 1e14 steps is expected).  Synchronous, like
 * mhd_wait_timeouts.  No counterpart in the reference.
 */
int mhd_close_decisions(uint64_t *count);

/*
 * Resident workgroups per compute unit of the one-candidate MH kernel and of the
 * forward kernel (a diagnostic of the register budget; no counterpart in the
 * reference).
 */
int mhd_debug_occupancy(int *mh_blocks, int *fwd_blocks);

/*
 * Secular (period-equation) function values, for diagnostics and parity gates:
 * delta[i] = Delta(c[i]) of chain chain[i] at period index period[i]
 * (NaN for a query whose chain is outside [0, B) or whose period is outside
 * [0, n_periods): the indices live on the device, the host cannot check them).
 * The value is the one every search of the library sees at that abscissa, bit
 * for bit.  Replaces nothing the reference calls directly; it is disba's dltar4 / dltar1
 * reached through src/inversion/model_params.py:201-206.
 */
int mhd_secular_batch(const mhd_problem_t *prob, const double *params /* device [B,NP] */,
                      int32_t B, const int32_t *chain /* device [N] */,
                      const int32_t *period /* device [N] */, const double *c /* device [N] */,
                      int32_t N, double *delta /* device [N] */, void *stream);

/*
 * The same queries, two abscissae each, through the library's other inlined copy
 * of the period equation (two abscissae per lane, as the verification pass of
 * the period-parallel search evaluates them): delta[2i], delta[2i+1] =
 * Delta(c[2i]), Delta(c[2i+1]) of chain chain[i] at period period[i].  A test
 * hook: both copies must return the same bits as mhd_secular_batch, or searches
 * that reach an abscissa through different copies could disagree on a sign next
 * to a root.  No counterpart in the reference.
 */
int mhd_debug_secular_pairs(const mhd_problem_t *prob, const double *params /* device [B,NP] */,
                            int32_t B, const int32_t *chain /* device [N] */,
                            const int32_t *period /* device [N] */,
                            const double *c /* device [N,2] */, int32_t N,
                            double *delta /* device [N,2] */, void *stream);

/*
 * DispersionCurveParams.forward_model for B parameter vectors
 * (src/inversion/model_params.py:178-209; the disba call at :201-206).
 * vel[b,:] is the datum prob->itype names (a curve of P values, or two in a row), status[b] is MHD_ST_OK or
 * the code of the exception the reference would raise.  n_evals, if not NULL,
 * is device uint64[2] and is INCREMENTED by {secular evaluations the
 * sequential disba algorithm performs, evaluations actually executed}.
 */
int mhd_forward_batch(const mhd_problem_t *prob, const double *params /* device [B,NP] */,
                      int32_t B, double *vel /* device [B,ND] */,
                      int32_t *status /* device [B] */, uint64_t *n_evals, void *stream);

/*
 * Model.get_likelihood for B predicted curves (src/inversion/model.py:420-447);
 * sums in numpy's float64 add.reduce order.
 */
int mhd_loglik_batch(const mhd_problem_t *prob, const double *pred /* device [B,ND] */,
                     int32_t B, double *logL /* device [B] */, void *stream);

/*
 * n_steps consecutive Model.perturb_params calls for each of B chains
 * (src/inversion/model.py:146-257 with validate_params :107-144,
 * acceptance_criteria :449-467 and the counters of :469-532), the chain x thin
 * nest of Inversion.random_walk (src/inversion/inversion.py:399-412).
 *
 * Draws are host-supplied, one triple per (step, chain):
 *   idx[s,b]    the perturbed parameter      (np.random.randint, model.py:196)
 *   cauchy[s,b] tan(pi*(u-1/2))              (model.py:210-212; with
 *               MHD_PROPOSE_UNIFORM the draw u of model.py:207)
 *   u_acc[s,b]  the accept uniform           (model.py:466)
 * accept iff u_acc <= exp(beta*(logL_new-logL)/T).
 * rotate is a mask of MHD_ROT_* and MHD_PROPOSE_UNIFORM.  sample_prior != 0 skips the forward model
 * and uses logL_new = 1 (model.py:234-236).
 * Optional outputs (NULL to skip): accepted[S,B] (uint8), status[S,B],
 * hist_m[S/hist_every,B,NP], hist_logL[S/hist_every,B], hist_pred[S/hist_every,B,ND]
 * written after every hist_every-th step (the per-step records of
 * Inversion.store_samples, src/inversion/inversion.py:431-457); diag (NULL to skip) names
 * further records at the same cadence, see mhd_diag_t.  n_evals as in mhd_forward_batch.
 */
int mhd_mh_steps(const mhd_problem_t *prob, const mhd_chains_t *chains, int32_t B,
                 int32_t n_steps, const int32_t *idx /* device [S,B] */,
                 const double *cauchy /* device [S,B] */, const double *u_acc /* device [S,B] */,
                 double T, int32_t rotate, int32_t sample_prior,
                 uint8_t *accepted, int32_t *status, int32_t hist_every, double *hist_m,
                 double *hist_logL, double *hist_pred, const mhd_diag_t *diag,
                 uint64_t *n_evals, void *stream);

/*
 * mhd_mh_steps for callers that cannot allocate device memory (code under HIP
 * graph capture, serving loops, harnesses that forbid hipMalloc): the same
 * steps with the same results, but the launch schedule that mhd_chains_t.sched
 * holds for mhd_mh_steps lives in a static device array of the library (53 KiB
 * per GPU, held until the process ends); chains->sched is ignored.  It has no
 * counterpart in the reference.  Limits and rules:
 *   - B <= 4096 chains (MHD_EINVAL beyond);
 *   - mhd_init() must have been called for the current device (MHD_EINVAL
 *     otherwise): this entry point creates no stream, event or memory;
 *   - every call on a GPU uses that GPU's one array.  Whose chains its contents
 *     belong to is told by B and the ADDRESS chains->m alone: a call with another
 *     B or another chains->m than the call before first clears the array (a
 *     memset on `stream`, enqueued under the same per-device lock as the launches);
 *     a caller whose allocator hands a new set of chains the address of an old
 *     one calls mhd_static_reset() in between (what the array holds are tuning
 *     values only: results never depend on them).  Calls that overlap in time on
 *     one GPU race on the array and must be ordered by the caller;
 *   - under HIP graph capture the decision to clear is taken once, at capture
 *     time: call mhd_static_reset() before the capture, so that the captured
 *     calls find the array cleared and adopt their chains without a memset in
 *     the graph (a captured memset would wipe the costs at every replay).
 */
#define MHD_STATIC_MAX_CHAINS 4096
int mhd_mh_steps_static(const mhd_problem_t *prob, const mhd_chains_t *chains, int32_t B,
                        int32_t n_steps, const int32_t *idx /* device [S,B] */,
                        const double *cauchy /* device [S,B] */,
                        const double *u_acc /* device [S,B] */, double T, int32_t rotate,
                        int32_t sample_prior, uint8_t *accepted, int32_t *status,
                        int32_t hist_every, double *hist_m, double *hist_logL, double *hist_pred,
                        const mhd_diag_t *diag, uint64_t *n_evals, void *stream);

/*
 * Clears the static scratch of the current device on `stream` (no allocation; mhd_init() must
 * have been called): the next call of mhd_mh_steps_static adopts its chains without clearing
 * again.  No counterpart in the reference.
 */
int mhd_static_reset(void *stream);

/*
 * One parallel-tempering exchange sweep, Inversion.perform_tempering_swap
 * (src/inversion/inversion.py:560-596), for n_ladders independent temperature
 * ladders of n_temps rungs each.  logL and beta hold ALL replicas of the run
 * (on several GPUs: logL after the all-gather of the ranks' blocks; every rank
 * runs the same sweep on the same numbers and so keeps an identical copy of
 * beta); replica (ladder l, rung t) is element l*ladder_stride + t*rung_stride.
 * Within a ladder the adjacent pairs (t, t+1) are visited in order with the
 * betas as the earlier pairs of this sweep left them; a pair with different
 * betas is exchanged when
 *     u[l, t] <= exp((beta2 - beta1) * (logL1 - logL2))
 * (the reference's rule, :578-586).  An exchange swaps the two betas; chain
 * states never move.  u is device [n_ladders, n_temps-1], host-supplied
 * uniforms (the reference draws np.random.rand(1) per pair, :584).  counts, if
 * not NULL, is device int64[2], incremented by {accepted, proposed} exchanges
 * that involve a beta == 1 replica (swap_acc, swap_prop, :588-596).
 * Deviations from the reference, both required for a working sampler: the
 * reference decides an exchange and then changes nothing, and its flat loop
 * knows one ladder only (here: n_ladders = 1, rung_stride = 1).
 */
int mhd_tempering_sweep(const double *logL /* device [R] */, double *beta /* device [R] */,
                        const double *u /* device [n_ladders, n_temps-1] */, int32_t n_temps,
                        int32_t n_ladders, int64_t rung_stride, int64_t ladder_stride,
                        int64_t *counts, void *stream);

/*
 * Model.stepsize_tuning (src/inversion/model.py:538-556) for B chains: a proposal width grows
 * by half where the acceptance rate is above 0.4 and is halved where it lies in (0, 0.2).  The
 * rate is n_acc / (n_acc + n_rej) of chains->counters -- one rate per chain, every width of the
 * chain scaled alike -- or, when chains->counters_ind is given (individual_acceptance), of each
 * parameter's own counters; it counts as 0 until a proposal has been rejected (the reference
 * refreshes its rates only while n_rej > 0, model.py:483, :520).  The reference defines the
 * method and leaves its call (after every step, model.py:257) commented out.
 */
int mhd_stepsize_tuning(const mhd_problem_t *prob, const mhd_chains_t *chains, int32_t B,
                        void *stream);

/*
 * Model.update_covariance_matrix alone (src/inversion/model.py:263-282), used on
 * the steps where the reference refreshes the rotation matrix between the
 * covariance update and the proposal (model.py:183-193).
 */
int mhd_cov_accumulate(const mhd_problem_t *prob, const mhd_chains_t *chains, int32_t B,
                       void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MCMC_AMD_H */
