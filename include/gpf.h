/*
 * gpf.h -- C ABI of libgpf_hip.so: the MI355X (gfx950) particle-filter hot path behind the
 * pf_initialize / pf_update! / pf_resample! / pf_rejuvenate! API of GenParticleFilters.jl.
 *
 * The reference has no FFI: its boundary is Julia multiple dispatch on
 * ParticleFilterView (reference src/view.jl:32-33).  A drop-in therefore is a new state type
 * whose methods ccall the entry points below (julia/GenParticleFiltersAMD.jl).
 * Every entry point names the reference method it replaces (paths relative to the reference
 * repository root, v0.2.3).
 *
 * Conventions
 *   - plain C types only; no C++/torch types cross this boundary.
 *   - every function returns a gpf_status; nothing throws across the ABI.  The Julia glue maps
 *     a non-zero status to error(gpf_last_error(h)) (ErrorException, like the reference's error()).
 *   - the handle owns all device buffers; host arrays passed in or out belong to the caller.
 *   - one handle = one host thread at a time.  Work is enqueued on one HIP stream; calls that
 *     return a scalar or fill a host array synchronise that stream, all others are asynchronous.
 *   - particle state is Float64; ancestor indices are reported 1-based Int64 into the
 *     PRE-resample particle array, exactly like state.parents (reference test/resample.jl:11).
 *   - there is no CPU fallback: gpf_create fails if no gfx950 device is usable.
 */
#ifndef GPF_H
#define GPF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#define GPF_ABI_VERSION 1

typedef struct gpf_filter* gpf_handle;

/* the library is built with -fvisibility=hidden: the functions this header declares are what it exports */
#pragma GCC visibility push(default)

typedef enum {
    GPF_OK = 0,
    GPF_ERR_INVALID_ARGUMENT = 1,
    GPF_ERR_INVALID_WEIGHTS = 2,   /* error("Invalid weights.")            src/resample.jl:55,92,151 */
    GPF_ERR_UNKNOWN_METHOD = 3,    /* error("... method ... not recognized") src/resample.jl:28, src/rejuvenate.jl:25 */
    GPF_ERR_HIP = 4,               /* a HIP runtime call failed; see gpf_last_error */
    GPF_ERR_NO_DEVICE = 5,
    GPF_ERR_STATE = 6              /* operation not valid in the current state (e.g. update before initialize) */
} gpf_status;

/* native models (the reference runs arbitrary Gen programs; the hot path runs these compiled ones) */
typedef enum {
    GPF_MODEL_LGSSM2 = 1,          /* 2-D linear-Gaussian SSM              (BASELINE configs 2,3) */
    GPF_MODEL_BEARINGS4 = 2,       /* bearings-only tracking, 4-D          (BASELINE config 4)   */
    GPF_MODEL_SV1 = 3,             /* stochastic volatility, 1-D           (BASELINE config 5)   */
    GPF_MODEL_OBJECT_MOTION = 4,   /* reference README.md:43-55            (BASELINE config 1)   */
    GPF_MODEL_LINE = 5             /* line_model, the fixture of the reference's tests (test/runtests.jl:3-16) */
} gpf_model;

/* method::Symbol of pf_resample! (src/resample.jl:19-30) */
typedef enum { GPF_RESAMPLE_MULTINOMIAL = 0, GPF_RESAMPLE_RESIDUAL = 1, GPF_RESAMPLE_STRATIFIED = 2,
               GPF_RESAMPLE_OPTIMAL = 3 /* gpf_resize only: pf_optimal_resize!, src/resize.jl:149-219 */,
               /* OPT-IN extension, not a reference method: pf_multinomial_resample! (src/resample.jl:48-65) with the N uniforms drawn
                * ALREADY SORTED (uniform spacings in exact integers).  Offspring counts ~ Multinomial(N, w) as for :multinomial;
                * state.parents comes out non-decreasing instead of as an i.i.d. sequence (src/resample.jl:59), so slot k of the new
                * population is NOT exchangeable with slot k' any more -- harmless for a filter that treats its particles as a set, visible
                * to code that cuts the population into views by position.  The ancestor search becomes a streaming merge and the row
                * gather reads ascending rows.  gpf_resample (and views) only. */
               GPF_RESAMPLE_MULTINOMIAL_SORTED = 4 } gpf_resample_method;
/* method::Symbol of pf_rejuvenate! (src/rejuvenate.jl:18-27) */
typedef enum { GPF_REJUVENATE_MOVE = 0, GPF_REJUVENATE_REWEIGHT = 1 } gpf_rejuvenate_method;
/* check keyword of the resamplers: true | :warn | false (src/resample.jl:43-46) */
typedef enum { GPF_CHECK_FALSE = 0, GPF_CHECK_WARN = 1, GPF_CHECK_TRUE = 2 } gpf_check;

typedef struct {
    int32_t  abi_version;    /* GPF_ABI_VERSION */
    int32_t  model;          /* gpf_model */
    int32_t  n_params;       /* <= 24 */
    int32_t  keep_prev;      /* 1: rows also carry x_{t-1} (needed by gpf_rejuvenate) */
    const double* params;    /* model parameters, layout in csrc/gpf_models.hpp */
    int64_t  n_particles;    /* particles held by this handle (this shard) */
    int64_t  n_global;       /* particles of the whole filter; == n_particles when unsharded */
    int64_t  gid0;           /* global index of local particle 0 (RNG counters use global ids) */
    uint64_t seed;           /* Philox key */
    int32_t  device;         /* HIP device ordinal */
    int32_t  reserved;
    void*    stream;         /* hipStream_t to enqueue on, or NULL for a library-owned stream */
} gpf_config;

/* ---- lifetime ------------------------------------------------------------------------------ */
int  gpf_abi_version(void);
gpf_status gpf_create(const gpf_config* cfg, gpf_handle* out);
gpf_status gpf_destroy(gpf_handle h);
const char* gpf_last_error(gpf_handle h);         /* valid until the next call on h; h may be NULL */
gpf_status gpf_synchronize(gpf_handle h);

/* ---- the four operations ------------------------------------------------------------------- */
/* pf_initialize(model, model_args, observations, n_particles)      src/initialize.jl:31-44
 * x_1 ~ prior, log_weights[i] = log p(obs | x_1), log_ml_est = 0, parents = 1:N. */
gpf_status gpf_initialize(gpf_handle h, const double* obs, int32_t n_obs);

/* pf_update!(state, new_args, argdiffs, observations)              src/update.jl:12-25
 * x_t ~ p(. | x_{t-1}), log_weights[i] += log p(obs | x_t); buffers swap (update_refs!, src/utils.jl:10-15). */
gpf_status gpf_update(gpf_handle h, const double* obs, int32_t n_obs);

/* One iteration of the reference's README loop (README.md:66-77) in one call:
 *     if effective_sample_size(state) < ess_frac * N   (src/utils.jl:163-164)
 *         pf_resample!(state, resample_method; check, sort_particles)          (src/resample.jl:19-30)
 *         pf_rejuvenate!(state, kern, (), n_iters; method = rejuvenate_method)  (src/rejuvenate.jl:18-27; rejuvenate_method < 0: none)
 *     end
 *     pf_update!(state, new_args, argdiffs, observations)                       (src/update.jl:12-25)
 * Same results as the four calls in that order, bit for bit.  The ESS verdict is also formed on the device and the propagate is
 * enqueued speculatively behind it, so the steps that do not resample never wait for the host's decision.
 * resampled / invalid / ess_out may be NULL. */
gpf_status gpf_step_ess(gpf_handle h, const double* obs, int32_t n_obs, double ess_frac, int32_t resample_method, int32_t sort_particles,
                        int32_t check, int32_t rejuvenate_method, int32_t n_iters, int32_t* resampled, int32_t* invalid, double* ess_out);

/* pf_initialize(model, args, obs, proposal, proposal_args, n)        src/initialize.jl:46-62
 * pf_update!(state, new_args, argdiffs, obs, proposal, proposal_args) src/update.jl:79-96 (+ src/translate.jl:86-105)
 * with a NATIVE proposal: new latents x ~ q(. | x_{t-1}, y_t); log_weights[i] += [log p(x | x_{t-1}) + log p(y | x)] - log q(x).
 * GPF_PROPOSAL_LOCALLY_OPTIMAL: the exact conditional of the linear-Gaussian model (GPF_MODEL_LGSSM2 only).
 * GPF_PROPOSAL_LINE_FIXED: the proposals of the reference's own tests for GPF_MODEL_LINE, slope ~ uniform_discrete(0, 0) at
 * the first step and outlier ~ bernoulli(0.0) at every step (test/initialize.jl:16-19, test/update.jl:42-43). */
typedef enum { GPF_PROPOSAL_LOCALLY_OPTIMAL = 1, GPF_PROPOSAL_LINE_FIXED = 2 } gpf_proposal;
gpf_status gpf_initialize_proposal(gpf_handle h, const double* obs, int32_t n_obs, int32_t proposal);
gpf_status gpf_update_proposal(gpf_handle h, const double* obs, int32_t n_obs, int32_t proposal);

/* Stratified initialisation / update over a discrete latent          src/initialize.jl:92-109, src/update.jl:193-210,
 * stratified_map! src/utils.jl:29-55.  values[k] = the value the model's discrete latent is constrained to in stratum k
 * (GPF_MODEL_OBJECT_MOTION: `moving`, 0 or 1); K = n_strata <= 8, block size B = n div K; particle i < K B belongs to
 * stratum i div B (interleaved = 0, :contiguous) or i mod K (interleaved = 1); the remaining particles draw their stratum
 * uniformly.  log_weights[i] (+)= log p(latent = value | parents) + log p(obs | x) + log K. */
gpf_status gpf_initialize_strata(gpf_handle h, const double* obs, int32_t n_obs, const double* values, int32_t n_strata, int32_t interleaved);
gpf_status gpf_update_strata(gpf_handle h, const double* obs, int32_t n_obs, const double* values, int32_t n_strata, int32_t interleaved);
/* pf_initialize(model, args, obs, strata, proposal, proposal_args, n; layout)   src/initialize.jl:111-129: the stratified latent is
 * constrained per stratum, the model's other choice comes from a native proposal, log_weights[i] = model_weight - prop_weight +
 * log(n_strata).  GPF_MODEL_LINE with GPF_PROPOSAL_LINE_FIXED: strata over `slope`, outlier ~ bernoulli(0.0) -- the form the reference
 * tests (test/initialize.jl:66-90: expected_w = logpdf(bernoulli, false, 0.1) + logpdf(normal, 0, slope, 1)). */
gpf_status gpf_initialize_strata_proposal(gpf_handle h, const double* obs, int32_t n_obs, const double* values, int32_t n_strata, int32_t interleaved,
                                          int32_t proposal);

/* pf_resample!(state, method; priority_fn, check[, sort_particles])  src/resample.jl:19-175
 *   priority_alpha: NaN -> priority_fn = nothing; otherwise priority_fn = w -> priority_alpha * w
 *                   (the tempering family the reference's tests use, test/resample.jl:15).
 *   sort_particles: only read by GPF_RESAMPLE_STRATIFIED (src/resample.jl:143-145, default true).
 *   check:          GPF_CHECK_TRUE returns GPF_ERR_INVALID_WEIGHTS on invalid weights (src/resample.jl:55).
 *   invalid:        if non-NULL receives safe_softmax's `invalid` flag (src/utils.jl:117-140); this
 *                   synchronises the stream.  Pass NULL with check != TRUE for a fully asynchronous call.
 * Weights containing NaN/+Inf always return GPF_ERR_INVALID_WEIGHTS when checked (the reference's
 * Categorical constructor rejects NaN probabilities). */
gpf_status gpf_resample(gpf_handle h, int32_t method, double priority_alpha, int32_t sort_particles,
                        int32_t check, int32_t* invalid);

/* pf_resample!(state[1:n], method; check, sort_particles) on the WHOLE filter (or shard): the sub-state semantics of
 * src/resample.jl:185-187,205-218 -- normalised over its own n particles, log_ml_est untouched, every particle keeps
 * the log-weight logsumexp(log_weights) - log n -- without the copies a view handle makes; the gather stays deferred
 * like gpf_resample's.  On a shard of a sharded filter this is the communication-free "island" resample. */
gpf_status gpf_resample_local(gpf_handle h, int32_t method, int32_t sort_particles, int32_t check, int32_t* invalid);

/* Many small filters in one state -- the batched form of
 *     for b in blocks; if effective_sample_size(state[b]) < ess_frac * length(b); pf_resample!(state[b], method; sort_particles, check); end; end
 * (sub-states: src/view.jl:16-48, src/resample.jl:185-187,205-218; the README loop README.md:60-79 per block; the reference's own
 * tests run N = 100).  The particles are cut into consecutive blocks of block_size (the last block may be shorter).  Up to 2048 particles per
 * block ONE launch
 * resamples every block out of LDS with the sub-state semantics: normalised over the block, log_ml_est untouched, every particle of
 * a resampled block carries logsumexp(block weights) - log(block size), parents local to the block.  Block b's result is
 * bit-identical to the same call on a view of the block (gpf_view_create + gpf_resample), all blocks under the call's one epoch.
 *   priority_alpha: NaN = priority_fn nothing; else priority_fn = w -> priority_alpha * w per block (src/resample.jl:51-52): ancestors from
 *                the priorities, new weights log_ws + (logsumexp(block weights) - logsumexp(log_ws)), log_ws = lw[a] - lp[a] (:213-216).
 *   ess_frac:    >= 0: a block resamples only if its effective sample size is below ess_frac x its size (decided on the device;
 *                an invalid block -- ESS NaN -- does not);  < 0 or NaN: every block resamples.
 *   check / invalid: as gpf_resample, over all blocks; blocks with NaN / +Inf weights (and, with GPF_CHECK_TRUE, all -Inf blocks)
 *                are left as they stand, the other blocks resample, and the call returns GPF_ERR_INVALID_WEIGHTS.  (With ess_frac >= 0 an
 *                invalid block never reaches the resampler -- its ESS is NaN -- and nothing is reported, as in the loop.)
 *   n_resampled: if non-NULL receives the number of blocks that resampled (synchronises).
 * Blocks of more than 2048 particles (no size limit in the reference's loop, test/resample.jl:130-162) are resampled one after the other with the
 * full-size kernels through view handles the filter keeps -- the same results, the same single epoch, one set of launches per block.
 * Not on sharded filters, views or filters with a trajectory store (GPF_ERR_STATE). */
gpf_status gpf_resample_blocks(gpf_handle h, int32_t method, int64_t block_size, double priority_alpha, int32_t sort_particles,
                               double ess_frac, int32_t check, int32_t* invalid, int64_t* n_resampled);
/* which blocks the last gpf_resample_blocks resampled: out[ceil(n / block_size)] (host), 1 / 0 */
gpf_status gpf_block_resampled(gpf_handle h, int32_t* out);
/* effective_sample_size(state[b]) and log_ml_estimate(state[b]) = log_ml_est + logsumexp(block weights) - log(block size) of every block
 * (src/utils.jl:163-178); host arrays of ceil(n / block_size) doubles, either may be NULL */
gpf_status gpf_block_stats(gpf_handle h, int64_t block_size, double* ess_out, double* lml_out);

/* The other steps of the loop over sub-states, block by block in one launch each (blocks as in gpf_resample_blocks):
 *   gpf_initialize_blocks / gpf_update_blocks: block b is initialised / extended with ITS OWN observation vector
 *     (for b in blocks; pf_update!(state[b], new_args, argdiffs, observations[b]); end -- per-view updates, test/update.jl:179-189);
 *     obs = [n_blocks][n_obs] doubles (host), row b for block b.  Default proposal only.
 *   gpf_rejuvenate_blocks: pf_rejuvenate!(state[b], ...) with the block's latest observation; only_resampled != 0: only in the blocks
 *     the last gpf_resample_blocks resampled (the README loop rejuvenates inside its `if`), the others keep particles and weights.
 *     n_accepted as in gpf_rejuvenate (over the blocks that took part).
 * After gpf_update_blocks the whole-filter gpf_rejuvenate also uses the per-block observations; gpf_update / gpf_initialize go back
 * to one observation for all particles.  Each call advances the epoch once; block b's result is bit-identical to the same call on a
 * view of the block.  Not on sharded filters, views or filters with a trajectory store. */
gpf_status gpf_initialize_blocks(gpf_handle h, const double* obs, int32_t n_obs, int64_t block_size);
gpf_status gpf_update_blocks(gpf_handle h, const double* obs, int32_t n_obs, int64_t block_size);
/* the same with a proposal PER BLOCK -- "Update with different proposals per view", test/update.jl:179-189, in one launch:
 * use_proposal[b] != 0 extends block b with the model's native proposal (src/update.jl:79-96, gpf_update_proposal's), 0 with the default
 * one (src/update.jl:12-25).  use_proposal: HOST int32[n_blocks]. */
gpf_status gpf_update_blocks_proposal(gpf_handle h, const double* obs, int32_t n_obs, int64_t block_size, const int32_t* use_proposal, int32_t proposal);
/* stratified initialisation / update of every block by itself (src/initialize.jl:92-109, src/update.jl:193-210 on each sub-state, one launch):
 * a particle's stratum follows from its index INSIDE its block and the block's own particle count (stratified_map!, src/utils.jl:29-55);
 * values / n_strata / interleaved as in gpf_initialize_strata, the same strata for all blocks */
gpf_status gpf_initialize_blocks_strata(gpf_handle h, const double* obs, int32_t n_obs, int64_t block_size, const double* values, int32_t n_strata, int32_t interleaved);
gpf_status gpf_update_blocks_strata(gpf_handle h, const double* obs, int32_t n_obs, int64_t block_size, const double* values, int32_t n_strata, int32_t interleaved);
gpf_status gpf_rejuvenate_blocks(gpf_handle h, int32_t method, int32_t n_iters, int32_t only_resampled, uint64_t* n_accepted);

/* same, with log_priorities = priority_fn.(log_weights) evaluated by the caller (any closure):
 * log_priorities is a HOST array of n_particles doubles. */
gpf_status gpf_resample_with_priorities(gpf_handle h, int32_t method, const double* log_priorities,
                                        int32_t sort_particles, int32_t check, int32_t* invalid);

/* pf_rejuvenate!(state, kern, kern_args, n_iters; method)          src/rejuvenate.jl:18-90
 * native kernels: GPF_REJUVENATE_MOVE     = pf_move_accept!  with kern = Gen.mh(trace, select(current step))
 *                 GPF_REJUVENATE_REWEIGHT = pf_move_reweight! with kern = move_reweight(trace, selection)
 *                                           (src/rejuvenate.jl:125-132)
 * n_accepted (may be NULL; non-NULL synchronises) replaces the per-particle @debug log
 * (src/rejuvenate.jl:47). Requires keep_prev = 1. */
gpf_status gpf_rejuvenate(gpf_handle h, int32_t method, int32_t n_iters, uint64_t* n_accepted);

/* pf_move_reweight!(state, move_reweight, (proposal, proposal_args), n_iters)   src/rejuvenate.jl:74-90 with the PROPOSAL variant
 * of the kernel, move_reweight(trace, proposal, proposal_args) (src/rejuvenate.jl:134-148): the current step's latent is proposed by a
 * NATIVE proposal q, the trace is updated with it, log_weights[i] += weight - fwd_score + bwd_score.
 *   GPF_MOVE_PROPOSAL_LOCALLY_OPTIMAL  (GPF_MODEL_LGSSM2, no parameters): q = p(x_t | x_{t-1}, y_t) -- a Gibbs move, relative weight 0 up to rounding
 *   GPF_MOVE_PROPOSAL_LINE_OUTLIER     (GPF_MODEL_LINE, params = {q, log q, log(1 - q)}): {:line => t => :outlier} ~ bernoulli(q), the
 *                                      outlier_propose of the reference's test (test/rejuvenate.jl:19-27, q = 0.9)
 * Requires keep_prev = 1 like gpf_rejuvenate. */
typedef enum { GPF_MOVE_PROPOSAL_LOCALLY_OPTIMAL = 1, GPF_MOVE_PROPOSAL_LINE_OUTLIER = 2 } gpf_move_proposal;
gpf_status gpf_rejuvenate_proposal(gpf_handle h, int32_t proposal, const double* params, int32_t n_params, int32_t n_iters);
/* the same native proposals under either rejuvenation method (method::Symbol of pf_rejuvenate!, src/rejuvenate.jl:18-27):
 * GPF_REJUVENATE_REWEIGHT = gpf_rejuvenate_proposal above; GPF_REJUVENATE_MOVE = pf_move_accept!(state, mh, (proposal, proposal_args), n_iters)
 * (src/rejuvenate.jl:40-53 with Gen.mh(trace, proposal, proposal_args)): propose, update, assess the reverse move, accept iff
 * log(rand()) < weight - fwd_score + bwd_score; the weights stay.  n_accepted (may be NULL): accepted proposals over all particles and sweeps. */
gpf_status gpf_rejuvenate_with_proposal(gpf_handle h, int32_t method, int32_t proposal, const double* params, int32_t n_params, int32_t n_iters,
                                        uint64_t* n_accepted);

/* "Lazy search" (csrc/gpf_k_fused.hpp): with enable != 0 a plain pf_resample!(state, :multinomial) enqueues the weight scan
 * only and the pf_update! that follows finds the ancestors, gathers, propagates and writes state.parents in ONE kernel; any other consumer
 * runs the stand-alone search first.  Same results bit for bit.  OFF by default (GPF_LAZY_SEARCH=1 in the environment turns it on for new
 * handles): measured on MI355X it is no faster than the two kernels (profiles/r04_lazy_search.txt). */
gpf_status gpf_set_lazy_search(gpf_handle h, int32_t enable);

/* ---- weight summaries ---------------------------------------------------------------------- */
/* effective_sample_size(state) / get_ess                            src/utils.jl:163-164,171 */
gpf_status gpf_effective_sample_size(gpf_handle h, double* out);
/* Gen.log_ml_estimate(state) / get_lml_est                          src/utils.jl:186 */
gpf_status gpf_log_ml_estimate(gpf_handle h, double* out);
/* Gen.get_log_weights(state) */
gpf_status gpf_get_log_weights(gpf_handle h, double* out, int64_t n);
/* get_log_norm_weights(state)                                       src/utils.jl:148 */
gpf_status gpf_get_log_norm_weights(gpf_handle h, double* out, int64_t n);
/* get_norm_weights(state)                                           src/utils.jl:156 */
gpf_status gpf_get_norm_weights(gpf_handle h, double* out, int64_t n);
/* state.parents (1-based, global)                                   test/resample.jl:11 */
gpf_status gpf_get_parents(gpf_handle h, int64_t* out, int64_t n);

/* ---- state access (Gen.get_traces: one latent address = one column) ------------------------- */
gpf_status gpf_state_dim(gpf_handle h, int32_t* dim, int32_t* row_width);
gpf_status gpf_get_column(gpf_handle h, int32_t column, double* out, int64_t n);
gpf_status gpf_get_rows(gpf_handle h, double* out, int64_t n_doubles);          /* n_particles * row_width */
gpf_status gpf_set_rows(gpf_handle h, const double* rows, int64_t n_doubles);   /* ParticleFilterState(trs, ws), src/initialize.jl:8-10 */
gpf_status gpf_set_log_weights(gpf_handle h, const double* lw, int64_t n);

/* Gen.sample_unweighted_traces(state, n_samples)                    src/utils.jl:7,189-194
 * n_samples i.i.d. draws from the normalised weights; the filter is not modified.  rows_out: HOST [n_samples][row_width];
 * idx_out (may be NULL): HOST 1-based indices of the drawn particles. */
gpf_status gpf_sample_unweighted(gpf_handle h, int64_t n_samples, double* rows_out, int64_t* idx_out);

/* ---- statistics ---------------------------------------------------------------------------- */
/* mean(state, addr)                                                 src/statistics.jl:13-14 */
gpf_status gpf_mean(gpf_handle h, int32_t column, double* out);
/* var(state, addr)  (population form)                               src/statistics.jl:48-50 */
gpf_status gpf_var(gpf_handle h, int32_t column, double* out);

/* ---- measurement hooks (bench.py) ----------------------------------------------------------- */
/* kernel ids for gpf_kernel_time */
typedef enum { GPF_K_STEP = 0, GPF_K_MAX = 1, GPF_K_SCAN = 2, GPF_K_SEARCH = 3, GPF_K_GATHER = 4,
               GPF_K_MOVE = 5, GPF_K_COUNT = 6 } gpf_kernel_id;
/* enable!=0: bracket every launch of kernel `id` with hipEvents on the handle's stream */
gpf_status gpf_kernel_timing(gpf_handle h, int32_t id, int32_t enable);
/* synchronises; total elapsed milliseconds and number of launches since timing was enabled */
gpf_status gpf_kernel_time(gpf_handle h, int32_t id, double* total_ms, int64_t* launches);

/* device-side math spec, for the bitwise host/device parity tests (tests/test_math_parity.py):
 * which: 0 exp, 1 log, 2 sincos2pi (out=sin,out2=cos), 3 atan2(a,b), 4 sqrt, 5 a/b,
 *        6 normal2 from counters (a=gid as double, b=blk as double; seed/epoch/tag via the handle),
 *        7 the spacing logarithm of GPF_RESAMPLE_MULTINOMIAL_SORTED: -log((k + 1/2) 2^-52), k = the top 52 bits of a's BIT PATTERN */
gpf_status gpf_debug_math(gpf_handle h, int32_t which, const double* a, const double* b, int64_t n,
                          double* out, double* out2);

/* ---- sub-state views (src/view.jl:16-48) -----------------------------------------------------
 * state[idxs] / view(state, idxs) for a contiguous range: a handle that ALIASES particles [start, start+count) of
 * `parent` (their rows, log-weights and parents) and owns only its scratch.  Every gpf_* operation works on it with the
 * sub-state semantics of the reference: pf_update!/pf_rejuvenate! touch only the range (update_refs! copies back,
 * src/utils.jl:17-20); pf_resample! resamples inside the range with LOCAL ancestor indices, leaves the log-ML estimate
 * alone and resets the weights to the block's average so its total mass is preserved (src/resample.jl:185-187,205-218);
 * gpf_log_ml_estimate = source.log_ml_est + logsumexp(view) - log n (src/utils.jl:174-178).  RNG counters keep the
 * global particle ids.  Destroy with gpf_destroy; a view becomes stale when the parent is resized or re-initialised.
 * A view of a SHARD handle is allowed: resampling it is the communication-free local ("island") resample of that shard.
 * A filter with a trajectory store (gpf_history_enable) has no views (GPF_ERR_STATE): the store keeps one ancestor map and one
 * set of columns per time step for the WHOLE filter. */
gpf_status gpf_view_create(gpf_handle parent, int64_t start, int64_t count, gpf_handle* out);
/* state[start:step:stop] (src/view.jl:35-48 takes any AbstractVector; the reference's tests use strided ranges, state[k:5:100] and
 * state[k:2:100], test/initialize.jl:60,85, test/update.jl:33,61, test/resize.jl:138): the view's particle i is particle
 * start + i * step of `parent` (0-based start, count particles).  Same semantics as gpf_view_create; per-particle RNG counters
 * stay the parent's particle ids, the view's resample stream is indexed by the slot ids start, start + 1, ... whatever the step.
 * A strided view works on a compact copy (gathered on entry, scattered back by every mutating call: update_refs! for sub-states
 * copies back as well, src/utils.jl:17-20). */
gpf_status gpf_view_create_strided(gpf_handle parent, int64_t start, int64_t step, int64_t count, gpf_handle* out);
/* state[idxs] / view(state, idxs) for ANY vector of distinct indices (src/view.jl:35-48: `idxs::AbstractVector`): index = HOST array of
 * count 0-based particle indices of `parent`, in the order the view presents them.  Same semantics and the same compact-copy mechanism as
 * the strided view; a particle keeps its own id as its RNG counter, the view's resample stream is indexed by the slot ids index[0],
 * index[0] + 1, ...  Repeated indices are refused (GPF_ERR_INVALID_ARGUMENT): a particle written through two slots has no defined value. */
gpf_status gpf_view_create_indexed(gpf_handle parent, const int64_t* index, int64_t count, gpf_handle* out);

/* ---- resize family (src/resize.jl) --------------------------------------------------
 * The handle stays valid; its per-particle buffers are reallocated for the new count.  Unsharded filters only. */
gpf_status gpf_n_particles(gpf_handle h, int64_t* out);
/* pf_resize!(state, n_particles, method; priority_fn, check)          src/resize.jl:16-124
 * method = GPF_RESAMPLE_MULTINOMIAL | GPF_RESAMPLE_RESIDUAL | GPF_RESAMPLE_OPTIMAL;
 * priority_alpha / check / invalid as in gpf_resample.
 * GPF_RESAMPLE_OPTIMAL = pf_optimal_resize! (src/resize.jl:149-219, Fearnhead & Clifford): needs n_particles <= current
 * count, ignores priority_alpha (the reference takes no priority_fn there); particles with c w_i >= 1 are kept with
 * their weights, the rest are resampled by systematic sampling (one uniform) and share logsumexp - log c; the
 * log-ML estimate is not touched. */
gpf_status gpf_resize(gpf_handle h, int64_t n_particles, int32_t method, double priority_alpha, int32_t check, int32_t* invalid);
/* pf_replicate!(state, n_replicates; layout)                           src/resize.jl:236-244 */
gpf_status gpf_replicate(gpf_handle h, int32_t n_replicates, int32_t interleaved);
/* pf_dereplicate!(state, n_replicates; layout, method)                 src/resize.jl:267-297  (sample=0 :keepfirst, 1 :sample) */
gpf_status gpf_dereplicate(gpf_handle h, int32_t n_replicates, int32_t interleaved, int32_t sample);

/* ---- trajectory store ------------------------------------------------------------------
 * Gen traces are persistent, so the reference can ask for a PAST choice of every surviving particle:
 * mean(state, 5 => :moving) (reference README.md:97-104, src/statistics.jl:13-14 with a past address).
 * With the store enabled the handle keeps, per time step, the step's latent columns and the composed ancestor map
 * of the resamples of that step (8d + 4 bytes per particle and step); queries follow the ancestry of each
 * current particle.  `step` is 1-based (step 1 = gpf_initialize).  Unsharded filters only. */
gpf_status gpf_history_enable(gpf_handle h, int32_t max_steps);          /* before gpf_initialize */
gpf_status gpf_history_steps(gpf_handle h, int32_t* n_steps);
gpf_status gpf_history_column(gpf_handle h, int32_t step, int32_t column, double* out, int64_t n);   /* trace[step => column] per particle */
gpf_status gpf_history_mean(gpf_handle h, int32_t step, int32_t column, double* out);               /* mean(state, step => column) */
gpf_status gpf_history_var(gpf_handle h, int32_t step, int32_t column, double* out);                /* var(state, step => column)  */
/* proportionmap(state, addr)[value]  (src/statistics.jl:91-101): normalised weight of the particles whose column == value;
 * step = 0 -> current step's column, step >= 1 -> past choice (trajectory store) */
gpf_status gpf_proportion(gpf_handle h, int32_t step, int32_t column, double value, double* out);

/* ---- shard-level building blocks (multi-GPU) ------------------------------------------------------
 * A filter sharded over G GPUs is G handles created with the same seed / n_global and contiguous
 * [gid0, gid0 + n_particles) ranges.  gpf_initialize / gpf_update / gpf_rejuvenate work per shard as they
 * are (RNG counters use global particle ids).  Resampling needs one exchange; it is composed from the
 * phases below by the host (sharded.py with torch.distributed = RCCL; a Julia host would use the same
 * calls with MPI.jl/NCCL.jl).  All pointer arguments are DEVICE pointers owned by the caller; every call
 * is asynchronous on the handle's stream.  The reference has no distributed path; these
 * restate src/resample.jl:48-120,143-175 with a global CDF.  Supported here: priority_fn = nothing and
 * sort_particles = false (sort_particles = true: gpf_shard_resample_sorted, the replicated plan).
 *
 *   phase 1  gpf_shard_weight_max     out2 = {local max, local flags & (NaN|+Inf)} as two doubles
 *            host: all-gather -> mf_all[G][2]
 *   phase 2  gpf_shard_weight_scan    in: mf_all (combined in the kernel); local fixed-point CDF; out5 = {S_local, Ql0..3}
 *                                     (the limbs of sum q^2 only when want_q != 0: the ESS needs them, a resample does not)
 *            host: all-gather -> tot_all[G][5]
 *   phase 2b gpf_shard_residual_scan  (residual only) in: tot_all; out2 = {Ctot_local, Rs_local}
 *            host: all-gather -> cr_all[G][2]
 *   phase 3  gpf_shard_push_count     RNG counters are keyed by the GLOBAL slot id, so every shard can work out by itself which
 *                                     output slots draw from ITS part of the global CDF (owner = first shard whose inclusive
 *                                     total exceeds the target): no request message exists.  Multinomial / residual: one pass
 *                                     over the targets of all output slots (residual: the deterministic head in closed form);
 *                                     stratified: the served slots are ONE contiguous range, found in closed form.  The phase
 *                                     also yields the number of entries this shard will send to and receive from each shard
 *                                     (counters cleared by gpf_shard_weight_scan).
 *            host: gpf_shard_counts (the one host sync of a resample: the all-to-all split sizes)
 *   phase 4  gpf_shard_push           ancestor lookup + row gather for every slot this shard owns the target of;
 *                                     packed_out[sum(sent)][W+1] = row | (slot inside its shard) << 32 | global ancestor id,
 *                                     grouped by destination shard (any order inside a group: every entry names its slot).
 *                                     Packs at most `capacity` entries; may be enqueued before gpf_shard_counts (it publishes
 *                                     the counts to pinned host memory when it starts) and called again with a larger buffer
 *                                     if the counts say it overflowed.
 *            host: ONE all-to-all of packed rows (8W+8 bytes per slot that changes shard)
 *   phase 5  gpf_shard_commit         the m = n_particles received entries become the new population: parents, log-weights = 0,
 *                                     log-ML estimate += logsumexp - log N (from mf_all, tot_all).  DEFERRED: the next gpf_update
 *                                     propagates the entries straight out of `packed` into their slots, any other call scatters
 *                                     them first; packed / mf_all / tot_all must stay alive and unchanged until the next
 *                                     gpf_shard_commit on this handle.
 * me = this shard's index; bounds = HOST int64[G+1], first global slot of every shard (bounds[G] = n_global).  G <= 64.
 */
gpf_status gpf_shard_weight_max(gpf_handle h, double* out2);
gpf_status gpf_shard_weight_scan(gpf_handle h, const double* mf_all, int32_t G, int32_t want_q, int64_t* out5);
/* safe_softmax's validity flags of the GLOBAL weights (bit 0 NaN, bit 1 +Inf, bit 2 all -Inf), published to pinned host memory by
 * the last gpf_shard_weight_scan when it starts: the host polls, the stream is not synchronised (check = true / :warn) */
gpf_status gpf_shard_flags(gpf_handle h, int32_t* flags_out);
gpf_status gpf_shard_residual_scan(gpf_handle h, const int64_t* tot_all, int32_t G, int64_t* out2);
gpf_status gpf_shard_push_count(gpf_handle h, int32_t method, const int64_t* tot_all, const int64_t* cr_all, int32_t G, int32_t me,
                                const int64_t* bounds);
/* the counts of the last gpf_shard_push_count as HOST int64[2G] (sent to each shard | received from each shard); synchronises */
gpf_status gpf_shard_counts(gpf_handle h, int32_t G, int64_t* host_counts);
gpf_status gpf_shard_push(gpf_handle h, int32_t method, const int64_t* tot_all, const int64_t* cr_all, int32_t G, int32_t me,
                          const int64_t* bounds, int64_t capacity, double* packed_out);
gpf_status gpf_shard_commit(gpf_handle h, const double* packed, int64_t m, const double* mf_all, const int64_t* tot_all, int32_t G);
/* running log_ml_est of this shard (identical on all shards) */
gpf_status gpf_shard_lml_est(gpf_handle h, double* out);

/* ---- host-side scalar spec (no GPU needed): the same deterministic functions the kernels use ---------- */
int32_t gpf_host_fix_K(int64_t n_global);
/* GPF_RESAMPLE_MULTINOMIAL_SORTED: fixed-point scale of the tile totals; the kernels' reciprocal divisions
 * floor(P 2^64 / den) (P < den < 2^63) and floor(p W / den) (p < den); one tile's gamma variate */
int32_t gpf_host_gamma_E(int64_t n_tiles);
uint64_t gpf_host_div128(uint64_t P, uint64_t den);
uint64_t gpf_host_muldiv128(uint64_t p, uint64_t W, uint64_t den);
uint64_t gpf_host_gamma_tile(uint64_t seed, uint32_t gid, uint32_t epoch, int64_t shape, int32_t Eg);
double  gpf_host_log(double x);
double  gpf_host_lse(double m, uint64_t S, int32_t K, int32_t flags);     /* m + log(S 2^-K) */
double  gpf_host_ess(uint64_t S, uint64_t Q_hi, uint64_t Q_lo);           /* S^2 / Q */
/* ---- the sharded resample behind ONE call (multi-GPU hosts need nothing but these five entry points) -----------------
 * A filter whose handle was created with n_global > n_particles is one shard of a filter spread over `world` processes, one
 * GPU each.  gpf_comm_create gives the handle an RCCL communicator (librccl is loaded with dlopen the first time; nothing is
 * linked against it); gpf_shard_resample then runs the whole global resample of DESIGN.md §6 -- weight maximum, all-gather,
 * fixed-point scan under the global maximum, all-gather of the shard totals (+ the residual counts), push count / push,
 * ONE variable-size exchange of packed rows (grouped ncclSend / ncclRecv: point-to-point pairs over xGMI), deferred commit --
 * on the handle's stream, every collective issued by the library.  It stands in for pf_resample!(state, method; check)
 * (src/resample.jl:19-30) on a sharded state (priority_fn = w -> alpha w: gpf_shard_resample_tempered; sort_particles = true: gpf_shard_resample_sorted).
 *   gpf_comm_unique_id: 128-byte id made by ONE process (ncclGetUniqueId) and handed to all others by the host's own means
 *   (MPI, Distributed.jl, torch.distributed, a file);  rank r of `world` owns the global range [gid0, gid0 + n) its
 *   gpf_config names; ranks are ordered by gid0.
 *   gpf_shard_effective_sample_size / gpf_shard_log_ml_estimate: the getters of src/utils.jl:163-186 over ALL shards
 *   (two all-gathers each; every rank gets the same value). */
gpf_status gpf_comm_unique_id(void* id128);
gpf_status gpf_comm_create(gpf_handle h, const void* id128, int32_t rank, int32_t world);
gpf_status gpf_comm_destroy(gpf_handle h);
/* How the three small summaries of a sharded resample -- (max, flags), {S, sum q^2 limbs}, the residual counts; 16-40 bytes per rank
 *  -- travel: *mailbox = 1: the producing kernel stores them straight into every peer's mailbox (device
 * memory mapped with hipIpc at gpf_comm_create, xGMI peer writes) and the consuming kernel waits for them -- no collective, no
 * launch, no host; 0: RCCL all-gathers (hipIpc mapping not possible on this system, the self-test failed, or GPF_SHARD_SUMMARY=rccl in the environment).
 * Self-test: gpf_comm_create tries the mapped mailboxes out (four dependent rounds between all ranks) and the receive windows below (one entry to and
 * from every peer) before enabling them; one rank whose test fails and every rank keeps RCCL for that transport (GPF_SHARD_SELFTEST=0: no test).
 * Every rank of a communicator is in the same mode.  How the ROWS travel: gpf_comm_set_exchange below. */
gpf_status gpf_comm_summary_mode(gpf_handle h, int32_t* mailbox);
/* exchange volume of this handle's gpf_shard_resample calls so far (what a scaling run compares with the worksheet of DESIGN.md 6.7):
 * out4 = {calls, entries sent to OTHER ranks, entries received from other ranks, bytes of one exchanged entry in the latest call};
 * reset = 1 clears the counters after reading */
gpf_status gpf_comm_traffic(gpf_handle h, int64_t* out4, int32_t reset);

/* The exchange plan of the i.i.d. resamplers (multinomial, and residual's i.i.d. tail) across shards; both give the same bits
 * (Random.rand(Categorical(weights), n), src/resample.jl:59,108 -- every slot's uniform is keyed by its GLOBAL slot id):
 *   GPF_SHARD_PLAN_PUSH (default): every shard evaluates the targets of ALL n_global slots, looks up the ones that fall in its own range
 *     and pushes [row | slot | ancestor] to the slot's shard -- one row exchange, no request message, O(n_global) ALU work per shard;
 *   GPF_SHARD_PLAN_PULL: every shard evaluates only its own n slots and sends each target to its owner, the owners answer with the
 *     rows -- O(n) work per shard, two exchanges and one more host wait.
 * Stratified resampling plans its exchange in closed form and ignores the setting.  Every rank of a communicator must use the same
 * plan.  gpf_comm_create takes the initial plan from the environment (GPF_SHARD_PLAN=push|pull). */
typedef enum { GPF_SHARD_PLAN_PUSH = 0, GPF_SHARD_PLAN_PULL = 1 } gpf_shard_plan;
gpf_status gpf_comm_set_plan(gpf_handle h, int32_t plan);
gpf_status gpf_comm_plan(gpf_handle h, int32_t* plan);
/* How the ROWS of the resamplers with ascending targets travel across shards (stratified with sort_particles = false, src/resample.jl:143-175, and the
 * opt-in sorted multinomial): their exchange is boundary slabs -- a few thousand rows per shard boundary --, i.e. pure latency.
 *   GPF_SHARD_EXCHANGE_P2P (the default wherever the shard mailboxes are up): every rank exports a slot-addressed receive window (one entry per local
 *     slot, hipIpc-mapped by its peers); the merge kernel of the shard that SERVES a slot stores [row | ancestor | seal] straight into the window of the
 *     rank that HOLDS it (xGMI peer stores) and that rank's next gpf_update reads it there, in the same launch as its own slots.  No split sizes for the
 *     host to wait for, no ncclGroup, no send / receive buffers, no capacity to overflow; gpf_shard_resample returns when its kernels are enqueued.
 *   GPF_SHARD_EXCHANGE_RCCL: packed entries, one host wait for the split sizes, grouped ncclSend / ncclRecv (what the i.i.d. resamplers -- a
 *     bandwidth-bound exchange of (G-1)/G of all rows -- and tempered resamples always use).
 * The same bits either way.  Every rank of a communicator must use the same mode.  Environment at gpf_comm_create: GPF_SHARD_EXCHANGE=p2p|p2p_all|rccl. */
/* What the transports under a sharded resample cost on THIS machine -- a scaling run prints it beside its step times (nobody can attach a profiler to it):
 *   out4[0] us per grouped ncclSend / ncclRecv exchange of `entries` packed entries ((W + 1) doubles each) with EVERY peer, mean of `reps` (2 untimed first)
 *   out4[1] the per-link rate of that exchange in GB/s: bytes one rank put on ONE link / out4[0]   (the scaling worksheet of DESIGN.md 6.7 assumes 76)
 *   out4[2] us per mailbox round: every rank stores its entry into every peer's mailbox and waits for all of theirs (`reps` dependent rounds in ONE launch,
 *           the launch's floor out4[3] taken off) -- what each of the 2 - 3 summary rounds of a sharded resample costs between real GPUs
 *   out4[3] us of that launch with no rounds in it
 * Collective: every rank calls it with the same arguments, between resamples.  0 where there is nothing to measure (one rank; no mailboxes). */
gpf_status gpf_comm_calibrate(gpf_handle h, int64_t entries, int32_t reps, double* out4);
typedef enum { GPF_SHARD_EXCHANGE_RCCL = 0, GPF_SHARD_EXCHANGE_P2P = 1,
               /* opt-in: the i.i.d. resamplers' rows (:multinomial, :residual under the push plan; src/resample.jl:59,108) through the windows too -- every entry
                * names its slot, so nothing else changes: no host wait, no ncclGroup for them either.  Their exchange is bandwidth-bound ((G-1)/G of all rows
                * as scattered 8 (W + 2)-byte peer stores against RCCL's bulk copies): which one wins is for a multi-GPU run to say (bench.py --gpus N times both). */
               GPF_SHARD_EXCHANGE_P2P_ALL = 2 } gpf_shard_exchange;
gpf_status gpf_comm_set_exchange(gpf_handle h, int32_t mode);
gpf_status gpf_comm_exchange(gpf_handle h, int32_t* mode);
/* Where the time of a sharded step goes (what a multi-GPU run cannot attach a profiler to): with phase timing on, gpf_shard_resample and the gpf_update that
 * commits it record events on the handle's stream at their phase boundaries.  gpf_phase_times (synchronises): us6 = total microseconds of
 *   [0] summaries   weight maximum -> (max, flags) round -> fixed-point scan -> {S} round (-> residual scans + their round), waits for the peers included
 *   [1] plan        push count: the own-slot search and the pass over the other shards' slots (i.i.d. methods) / the closed-form plan (ascending targets)
 *   [2] pack        look-ups + packing of the served slots (i.i.d.) / merge + own ancestors + window stores or packing (ascending targets)
 *   [3] host wait   WALL CLOCK the host spent blocked on the exchange's split sizes (overlaps [2] on the GPU; 0 for the window exchange)
 *   [4] exchange    grouped ncclSend / ncclRecv + the self copy (0 for the window exchange: its rows travel inside [2])
 *   [5] commit      from the end of the exchange to the end of the next gpf_update's propagate (the host's way back included)
 * over *resamples calls; [0], [1], [2], [4], [5] on the GPU's timeline, idle gaps included.  enable = 1 clears the record. */
enum { GPF_PHASE_SUMMARIES = 0, GPF_PHASE_PLAN = 1, GPF_PHASE_PACK = 2, GPF_PHASE_HOST_WAIT = 3, GPF_PHASE_EXCHANGE = 4, GPF_PHASE_COMMIT = 5, GPF_PHASE_COUNT = 6 };
gpf_status gpf_phase_timing(gpf_handle h, int32_t enable);
gpf_status gpf_phase_times(gpf_handle h, double* us6, int64_t* resamples);
gpf_status gpf_shard_resample(gpf_handle h, int32_t method, int32_t check, int32_t* invalid);
/* pf_resample!(state, method; priority_fn = w -> priority_alpha * w, check) on a sharded state (src/resample.jl:51-52,57,198-200; the
 * tempering family of test/resample.jl:15): ancestors from the CDF of the priorities over ALL shards, log_ml_est from the raw
 * weights, new log-weights log_ws + (log N - logsumexp(log_ws)), log_ws = lw[a] - lp[a], with the logsumexp over all shards.
 * Three summary rounds instead of one and one more double (log_ws) per exchanged entry; the result is bit-identical to
 * gpf_resample(h, method, priority_alpha, ...) on the unsharded filter.  Not together with sort_particles = true. */
gpf_status gpf_shard_resample_tempered(gpf_handle h, int32_t method, double priority_alpha, int32_t check, int32_t* invalid);
/* pf_resample!(state, :stratified; sort_particles = true, check) -- the reference's DEFAULT form of the stratified resampler (src/resample.jl:143-175: strata over
 * the particles in descending weight order, :145,156-157) -- on a sharded state, called on every rank like gpf_shard_resample; the same ancestors as
 * gpf_resample(h, GPF_RESAMPLE_STRATIFIED, NaN, 1, ...) on the unsharded filter, for any number of shards.  A global sort has no shard-local form: every rank
 * gathers ALL log-weights (one all-gather of 8 bytes per global particle) and runs the unsharded sort + scan + search on them itself (a planner filter of
 * n_global particles on every rank, created at the first call; about 130 bytes of HBM per global particle), then serves the rows its particles are ancestors
 * of as packed entries through the grouped point-to-point exchange.  The cost grows with n_global, not with the shard: across shards
 * gpf_shard_resample(GPF_RESAMPLE_STRATIFIED) -- sort_particles = false -- stays the fast form.  No priority_fn. */
gpf_status gpf_shard_resample_sorted(gpf_handle h, int32_t check, int32_t* invalid);
/* ... and its two phases for a host that brings its own collectives (the python engine of sharded.py; cf. gpf_shard_push_count / gpf_shard_push): after the
 * summary phases (gpf_shard_weight_max, gpf_shard_weight_scan), with lw_all = the log-weights of ALL shards in global order on this device (the host's
 * all-gather): gpf_shard_sorted_count runs the planner and leaves the exchange counts for gpf_shard_counts, gpf_shard_sorted_push packs this shard's
 * entries [row | slot << 32 | ancestor id], grouped by destination, at most `capacity` of them; then the exchange and gpf_shard_commit as ever. */
gpf_status gpf_shard_sorted_count(gpf_handle h, const double* lw_all, int32_t G, int32_t me);
gpf_status gpf_shard_sorted_push(gpf_handle h, int32_t G, int32_t me, int64_t capacity, double* packed_out);
/* gpf_step_ess on a sharded filter -- one iteration of the README loop (README.md:66-77), called on every rank like gpf_shard_resample:
 *     if effective_sample_size(state) < ess_frac * N_global;  pf_resample!(state, method);  pf_rejuvenate!(state, ...; method);  end;  pf_update!(state, ...)
 * with the GLOBAL effective sample size; the same results as the separate calls (gpf_shard_effective_sample_size, gpf_shard_resample,
 * gpf_rejuvenate, gpf_update) on every rank.  With the shard mailboxes up, the summary is ONE reduction launch that exchanges the shard totals
 * itself and leaves the verdict on the device, the propagate runs speculatively behind it.  rejuvenate_method < 0: none. */
gpf_status gpf_shard_step_ess(gpf_handle h, const double* obs, int32_t n_obs, double ess_frac, int32_t method, int32_t check,
                              int32_t rejuvenate_method, int32_t n_iters, int32_t* resampled, int32_t* invalid);
gpf_status gpf_shard_effective_sample_size(gpf_handle h, double* out);
gpf_status gpf_shard_log_ml_estimate(gpf_handle h, double* out);

/* test hook: copy one level of the weight CDF left by the last scan of channel 0 to the host.
 * which: 0 cdf (u64 per padded cell), 1 per-16 prefixes (u64), 2 per-256 prefixes (u64), 3 4-byte keys (u32 per 32 cells),
 * 4 16-bit in-group offsets (u16 per padded cell), 5 coarse offset rows (u16).  *n_bytes: capacity in, bytes written out. */
gpf_status gpf_debug_levels(gpf_handle h, int32_t which, void* out, int64_t* n_bytes);

/* host twin of gpf_debug_math (which = 0..5, 7) */
void    gpf_host_math(int32_t which, const double* a, const double* b, int64_t n, double* out, double* out2);

/* ---- checkpoint / resume ----------------------------------------------------------------------
 * No reference counterpart (a Gen ParticleFilterState is an ordinary Julia object: `serialize` does it there).  The whole state of a filter (or of one
 * shard) as ONE host blob: population, log-weights, parents, log-ML estimate, RNG epoch, the latest observation and strata.  gpf_checkpoint_save first
 * turns everything deferred (lazy move, lazy search, un-gathered resample, un-scattered sharded commit) into state; gpf_checkpoint_load takes a handle
 * created with the SAME gpf_config (model, parameters, particle counts, gid0, seed, keep_prev -- else GPF_ERR_INVALID_ARGUMENT) and continues bit for bit
 * where the saved filter stood.  Not part of a blob: the trajectory store (a filter that records one refuses to load), per-block observations, views. */
gpf_status gpf_checkpoint_size(gpf_handle h, int64_t* bytes);
gpf_status gpf_checkpoint_save(gpf_handle h, void* out, int64_t bytes);
gpf_status gpf_checkpoint_load(gpf_handle h, const void* in, int64_t bytes);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* GPF_H */
