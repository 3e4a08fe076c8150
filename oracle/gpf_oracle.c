/*
 * gpf_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU oracle for the particle-filter hot path.
 *
 * Array-level primitives in plain C that restate, for struct-of-rows Float64 particle state,
 * the algorithms of GenParticleFilters.jl (reference @ v0.2.3, paths relative to /root/reference):
 *   src/initialize.jl:31-44   pf_initialize            -> o_init
 *   src/update.jl:12-25       pf_update!               -> o_step
 *   src/rejuvenate.jl:40-90   pf_move_accept!/reweight -> o_move
 *   src/utils.jl:100-140      lognorm/softmax/safe_softmax  -> o_max_flags, o_fixq, o_scan
 *   src/utils.jl:163-164      effective_sample_size    -> o_ess_from
 *   src/resample.jl:48-175    three resamplers         -> o_targets_*, o_upper_bound, o_residual_split
 *   src/resample.jl:178-202   update_lml_est!/update_weights! -> o_lse_from (+ python composition)
 *   src/statistics.jl:13-14,48-50  mean / var          -> o_wsum
 * The composition of these primitives into pf_* operations is in oracle/oracle.py, which cites
 * the reference line for every statement.
 *
 * PARITY STATUS: the reference cannot be executed in the build container (Julia absent) and its
 * tests hold no golden vectors or seeds, so for RANDOM STREAMS (ancestor indices,
 * sampled states) parity is UNPINNED: this oracle is the definition.  Deterministic arithmetic is
 * pinned against the invariants the reference's tests assert (tests/test_oracle_reference_invariants.py)
 * and against oracle/ref_literal.c, a line-by-line Float64 restatement of src/resample.jl and
 * src/utils.jl driven by the same indexed uniforms.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * All integer weight arithmetic is exact and order-independent: the same inputs
 * give the same ancestors on 1 thread, on the GPU, or sharded over G GPUs.
 */
#include "gpf_oracle_math.h"
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define O_EXPORT __attribute__((visibility("default")))

enum { O_MODEL_LGSSM2 = 1, O_MODEL_BEARINGS4 = 2, O_MODEL_SV1 = 3, O_MODEL_OBJECT_MOTION = 4, O_MODEL_LINE = 5 };
enum { O_FLAG_NAN = 1, O_FLAG_POSINF = 2, O_FLAG_ALL_NEGINF = 4 };

/* Threads for the per-particle loops (counter-based RNG: results do not depend on the thread count).
 * 1 = the reference's execution model (single-threaded Julia); >1 only for the "all host cores" CPU baseline. */
O_EXPORT int o_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n; return 1;
#endif
}

/* ------------------------------------------------------------------ scalar helpers (exported) */
O_EXPORT double o_log_d(double x) { return o_log(x); }
O_EXPORT double o_exp_d(double x) { return o_exp(x); }
O_EXPORT uint64_t o_exp_fix_d(double d, int K) { return o_exp_fix(d, K); }
O_EXPORT double o_atan2_d(double y, double x) { return o_atan2(y, x); }
O_EXPORT void o_sincos2pi_d(double u, double *s, double *c) { o_sincos2pi(u, s, c); }
O_EXPORT void o_philox_d(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                         uint32_t *out)
{
    o_philox_t r = o_philox4x32_10(c0, c1, c2, c3, k0, k1);
    for (int i = 0; i < 4; ++i) out[i] = r.v[i];
}
O_EXPORT void o_normal2_d(uint64_t seed, uint32_t gid, uint32_t blk, uint32_t epoch, uint32_t tag,
                          double *z0, double *z1)
{
    o_normal2(o_rng(seed, gid, blk, epoch, tag), z0, z1);
}
O_EXPORT double o_u52_d(uint64_t seed, uint32_t gid, uint32_t blk, uint32_t epoch, uint32_t tag)
{
    o_philox_t b = o_rng(seed, gid, blk, epoch, tag);
    return o_u52(b.v[0], b.v[1]);
}
/* the resample uniform of output slot `slot` as a 52-bit Float64 (top 52 of the 64 bits the spec uses) */
O_EXPORT double o_resample_u52_d(uint64_t seed, uint32_t slot, uint32_t epoch)
{
    const uint64_t U = o_resample_u64(seed, slot, epoch);
    return o_u52((uint32_t)(U >> 32), (uint32_t)U);
}

/* vectorised math, for the bitwise product-vs-oracle math tests */
O_EXPORT void o_math_vec(int which, const double *a, const double *b, int64_t n, double *out, double *out2)
{
    for (int64_t i = 0; i < n; ++i) {
        switch (which) {
            case 0: out[i] = o_exp(a[i]); break;
            case 1: out[i] = o_log(a[i]); break;
            case 2: o_sincos2pi(a[i], &out[i], &out2[i]); break;
            case 3: out[i] = o_atan2(a[i], b[i]); break;
            case 4: out[i] = sqrt(a[i]); break;
            case 5: out[i] = a[i] / b[i]; break;
            case 7: out[i] = o_neglog_u52(o_d2u(a[i])); break;            /* the argument's BITS are the slot's 64-bit uniform */
            default: out[i] = 0.0;
        }
    }
}

/* K = min(52, 62 - ceil(log2 N)): N * 2^K <= 2^62, so any sum of N fixed-point weights fits in
 * 62 bits (the top two bits of a u64 stay free for the device scan's status field). */
O_EXPORT int o_fix_K(int64_t n_global)
{
    int cl = 0;
    while (((int64_t)1 << cl) < n_global) ++cl;
    int K = 62 - cl;
    return K > 52 ? 52 : K;
}

/* logsumexp from the exact integer sum: m + log(S * 2^-K)  (resample.jl:180, utils.jl:100) */
O_EXPORT double o_lse_from(double m, uint64_t S, int K, int flags)
{
    if (flags & (O_FLAG_NAN | O_FLAG_POSINF)) return NAN;
    if (flags & O_FLAG_ALL_NEGINF) return -INFINITY;
    double Sd = (double)S;                     /* correctly rounded u64 -> f64 */
    return m + o_log(Sd * o_pow2(-K));         /* exact scaling */
}
/* ESS = (sum w)^2 / sum w^2 = S^2 / Q  (utils.jl:163-164; test/utils.jl:10) */
O_EXPORT double o_ess_from(uint64_t S, uint64_t Qhi, uint64_t Qlo)
{
    double Sd = (double)S;
    double Qd = (double)Qhi * 0x1p64 + (double)Qlo;
    return (Sd * Sd) / Qd;
}

/* ------------------------------------------------------------------ models */
static int model_dim(int model)
{
    switch (model) { case O_MODEL_LGSSM2: return 2; case O_MODEL_BEARINGS4: return 4;
                     case O_MODEL_SV1: return 1; case O_MODEL_OBJECT_MOTION: return 2; case O_MODEL_LINE: return 2; }
    return 0;
}
static int model_nblk(int model)
{
    switch (model) { case O_MODEL_LGSSM2: return 1; case O_MODEL_BEARINGS4: return 2;
                     case O_MODEL_SV1: return 1; case O_MODEL_OBJECT_MOTION: return 2; case O_MODEL_LINE: return 1; }
    return 0;
}
O_EXPORT int o_model_dim(int model) { return model_dim(model); }
O_EXPORT int o_model_nblk(int model) { return model_nblk(model); }

/* sample x_t ~ p(. | x_{t-1}) (first=0) or x_1 ~ prior (first=1); RNG blocks blk0.. of (gid,epoch,tag).
 * This is what Gen's generate/update do for the unconstrained latent choices of the step
 * (initialize.jl:40, update.jl:17). */
static void model_sample(int model, const double *P, int first, const double *xp, const double *obs,
                         uint64_t seed, uint32_t gid, uint32_t blk0, uint32_t epoch, uint32_t tag,
                         double *xn)
{
    double z0, z1, z2, z3;
    switch (model) {
    case O_MODEL_LGSSM2: {
        o_normal2(o_rng(seed, gid, blk0, epoch, tag), &z0, &z1);
        if (first) { xn[0] = P[5] * z0; xn[1] = P[5] * z1; }
        else {
            double t0 = P[0] * xp[0] + P[1] * xp[1];
            double t1 = P[2] * xp[0] + P[3] * xp[1];
            xn[0] = t0 + P[4] * z0;
            xn[1] = t1 + P[4] * z1;
        }
    } break;
    case O_MODEL_BEARINGS4: {
        o_normal2(o_rng(seed, gid, blk0, epoch, tag), &z0, &z1);
        o_normal2(o_rng(seed, gid, blk0 + 1, epoch, tag), &z2, &z3);
        if (first) {
            xn[0] = P[0] + P[4] * z0; xn[1] = P[1] + P[5] * z1;
            xn[2] = P[2] + P[6] * z2; xn[3] = P[3] + P[7] * z3;
        } else {
            xn[0] = (xp[0] + xp[2]) + P[8] * z0;
            xn[1] = (xp[1] + xp[3]) + P[8] * z1;
            xn[2] = xp[2] + P[9] * z2;
            xn[3] = xp[3] + P[9] * z3;
        }
    } break;
    case O_MODEL_SV1: {
        o_normal2(o_rng(seed, gid, blk0, epoch, tag), &z0, &z1);
        if (first) xn[0] = P[0] + P[3] * z0;
        else       xn[0] = (P[0] + P[1] * (xp[0] - P[0])) + P[2] * z0;
    } break;
    case O_MODEL_OBJECT_MOTION: {
        /* README.md:43-55: moving ~ bernoulli(moving ? 0.75 : 0.25); y ~ normal(y + vel, 0.01) */
        o_philox_t b = o_rng(seed, gid, blk0, epoch, tag);
        double u = o_u52(b.v[0], b.v[1]);
        o_normal2(o_rng(seed, gid, blk0 + 1, epoch, tag), &z0, &z1);
        double pm = first ? 0.0 : xp[0], py = first ? 0.0 : xp[1];
        double p = (pm != 0.0) ? P[0] : P[1];
        double mv = (u < p) ? 1.0 : 0.0;
        double vel = (mv != 0.0) ? obs[1] : 0.0;
        xn[0] = mv;
        xn[1] = (py + vel) + P[2] * z0;
    } break;
    case O_MODEL_LINE: {
        /* reference test/runtests.jl:3-16: slope ~ uniform_discrete(-2, 2) once (line_model :13); per step (line_step :3-8)
         * outlier ~ bernoulli(0.1).  obs = [y_t, x_t]; x_t = 0 is model args (0,): no step yet, no outlier choice */
        o_philox_t b = o_rng(seed, gid, blk0, epoch, tag);
        xn[0] = first ? P[8] + (double)o_mulhi64(o_u64(b.v[0], b.v[1]), (uint64_t)P[9]) : xp[0];
        xn[1] = (obs[1] != 0.0 && o_u52(b.v[2], b.v[3]) < P[0]) ? 1.0 : 0.0;
    } break;
    }
}

/* log p(y_t | x_t): the weight increment Gen returns for the newly constrained observation */
static double model_loglik(int model, const double *P, const double *x, const double *obs)
{
    switch (model) {
    case O_MODEL_LGSSM2: {
        double z0 = (obs[0] - x[0]) * P[6], z1 = (obs[1] - x[1]) * P[6];
        return -0.5 * (z0 * z0 + z1 * z1) - P[7];
    }
    case O_MODEL_BEARINGS4: {
        const double PI = 3.14159265358979311600e+00, TWOPI = 6.28318530717958623200e+00;
        double b = o_atan2(x[1], x[0]);
        double r = obs[0] - b;
        if (r > PI) r -= TWOPI; else if (r <= -PI) r += TWOPI;
        double z = r * P[10];
        return -0.5 * (z * z) - P[11];
    }
    case O_MODEL_SV1: {
        double y = obs[0];
        return (-0.5 * ((y * y) * o_exp(-x[0])) - 0.5 * x[0]) - P[4];
    }
    case O_MODEL_OBJECT_MOTION: {
        double z = (obs[0] - x[1]) * P[3];
        return -0.5 * (z * z) - P[4];
    }
    case O_MODEL_LINE: {
        /* y ~ normal(x * slope, outlier ? 10. : 1.)  (test/runtests.jl:6); nothing observed at model args (0,) */
        if (obs[1] == 0.0) return 0.0;
        int out = x[1] != 0.0;
        double z = (obs[0] - obs[1] * x[0]) * (out ? P[2] : P[1]);
        return -0.5 * (z * z) - (out ? P[4] : P[3]);
    }
    }
    return 0.0;
}

/* custom-proposal initialize / update (initialize.jl:46-62; update.jl:79-96 + translate.jl:86-105): native locally
 * optimal proposal of the linear-Gaussian model; returns log_weight = model_score_diff - fwd_proposal_score */
static double model_propose(int model, const double *P, int first, const double *xp, const double *obs,
                            uint64_t seed, uint32_t gid, uint32_t blk0, uint32_t epoch, uint32_t tag, double *xn)
{
    if (model == O_MODEL_LINE) {
        /* the reference tests' proposals as one native proposal: slope ~ uniform_discrete(0, 0) at the first step
         * (test/initialize.jl:16-17), outlier ~ bernoulli(0.0) at every step (test/initialize.jl:18-19, test/update.jl:42-43);
         * both deterministic (proposal score 0): weight = model score of the proposed choices + log p(y | .) */
        (void)seed; (void)gid; (void)blk0; (void)epoch; (void)tag;
        xn[0] = first ? 0.0 : xp[0];
        xn[1] = 0.0;
        double w = first ? P[7] : 0.0;                                  /* log(1/5): test/initialize.jl:21 */
        if (obs[1] != 0.0) w = (w + P[6]) + model_loglik(model, P, xn, obs);
        return w;
    }
    if (model != O_MODEL_LGSSM2) return NAN;
    double z0, z1;
    o_normal2(o_rng(seed, gid, blk0, epoch, tag), &z0, &z1);
    double mu0 = first ? 0.0 : P[0] * xp[0] + P[1] * xp[1];
    double mu1 = first ? 0.0 : P[2] * xp[0] + P[3] * xp[1];
    int o = first ? 14 : 8;
    double m0 = mu0 + P[o] * (obs[0] - mu0), m1 = mu1 + P[o] * (obs[1] - mu1);
    xn[0] = m0 + P[o + 1] * z0;
    xn[1] = m1 + P[o + 1] * z1;
    double a0 = (xn[0] - mu0) * P[o + 4], a1 = (xn[1] - mu1) * P[o + 4];
    double lt = -0.5 * (a0 * a0 + a1 * a1) - P[o + 5];                 /* log p(x | x_prev) */
    double b0 = (xn[0] - m0) * P[o + 2], b1 = (xn[1] - m1) * P[o + 2];
    double lq = -0.5 * (b0 * b0 + b1 * b1) - P[o + 3];                 /* log q(x) */
    return (lt + model_loglik(model, P, xn, obs)) - lq;
}
/* log p(y_t | x_t) of given particle rows: the deterministic part of a step, for the reference-twin fixtures
 * (julia/reference_twins.jl evaluates the same quantity with Gen.project on the @gen twin) */
O_EXPORT void o_loglik_rows(int model, const double *P, const double *rows, int W, int64_t n, const double *obs, double *out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = model_loglik(model, P, rows + i * W, obs);
}
/* Strided sub-state views (reference src/view.jl:35-48 with idxs = start:step:stop, e.g. state[k:5:100] in test/update.jl:33):
 * local particle i of the view is particle start + i*step of the source, and its RNG counter stays that GLOBAL id.  The stride
 * is a property of the call (set by the Python composition around a view's per-particle call, 1 otherwise). */
static int64_t g_gid_stride = 1;
O_EXPORT void o_set_gid_stride(int64_t stride) { g_gid_stride = stride; }
/* Views over an arbitrary index vector (src/view.jl:35-48, state[idxs]): local particle i is particle idxs[i] of the source; the map
 * holds idxs[i] - idxs[0] (gid0 = the first index), NULL otherwise. */
static const int32_t *g_gid_map = NULL;
O_EXPORT void o_set_gid_map(const int32_t *map) { g_gid_map = map; }
#define O_GID(gid0, i) ((uint32_t)(g_gid_map ? (gid0) + (int64_t)g_gid_map[i] : (gid0) + (int64_t)(i) * g_gid_stride))

O_EXPORT void o_init_proposal(int model, const double *P, uint64_t seed, uint32_t epoch, int64_t gid0, int64_t n,
                              int W, const double *obs, double *rows, double *lw)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double *r = rows + i * W;
        for (int k = 0; k < W; ++k) r[k] = 0.0;
        lw[i] = model_propose(model, P, 1, NULL, obs, seed, O_GID(gid0, i), 0, epoch, O_TAG_INIT, r);   /* initialize.jl:58 */
    }
}
O_EXPORT void o_step_proposal(int model, const double *P, uint64_t seed, uint32_t epoch, int64_t gid0, int64_t n,
                              int W, int keep_prev, const double *obs, const double *rows_in, double *rows_out, double *lw)
{
    int d = model_dim(model);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const double *ri = rows_in + i * W;
        double *ro = rows_out + i * W;
        double xn[4], xp[4];
        double w = model_propose(model, P, 0, ri, obs, seed, O_GID(gid0, i), 0, epoch, O_TAG_UPDATE, xn);
        for (int k = 0; k < d; ++k) xp[k] = ri[k];
        for (int k = 0; k < W; ++k) ro[k] = 0.0;
        for (int k = 0; k < d; ++k) ro[k] = xn[k];
        if (keep_prev) for (int k = 0; k < d; ++k) ro[d + k] = xp[k];
        lw[i] = lw[i] + w;                                              /* translate.jl:103, update.jl:40 */
    }
}

/* stratified initialisation / update (initialize.jl:92-109, update.jl:193-210): the model's discrete latent is
 * constrained to the stratum's value (merge(stratum, observations)); returns log p(latent = value | parents), the part
 * of the weight increment Gen adds for the constrained latent choice.  Only the object_motion model has one. */
static double model_sample_stratum(int model, const double *P, int first, const double *xp, const double *obs, double value,
                                   uint64_t seed, uint32_t gid, uint32_t blk0, uint32_t epoch, uint32_t tag, double *xn)
{
    double z0, z1;
    if (model == O_MODEL_LINE) {
        /* strata over `slope` at the first step (test/initialize.jl:39-64), over the step's `outlier` afterwards
         * (test/update.jl:13-40); the other choice is sampled as usual */
        o_philox_t b = o_rng(seed, gid, blk0, epoch, tag);
        if (first) {
            xn[0] = value;
            xn[1] = (obs[1] != 0.0 && o_u52(b.v[2], b.v[3]) < P[0]) ? 1.0 : 0.0;
            return P[7];
        }
        xn[0] = xp[0];
        xn[1] = value != 0.0 ? 1.0 : 0.0;
        return value != 0.0 ? P[5] : P[6];
    }
    o_normal2(o_rng(seed, gid, blk0 + 1, epoch, tag), &z0, &z1);                        /* O_MODEL_OBJECT_MOTION, README.md:43-55 */
    double pm = first ? 0.0 : xp[0], py = first ? 0.0 : xp[1];
    double mv = (value != 0.0) ? 1.0 : 0.0;
    double lp = (pm != 0.0) ? ((mv != 0.0) ? P[5] : P[6]) : ((mv != 0.0) ? P[7] : P[8]);
    double vel = (mv != 0.0) ? obs[1] : 0.0;
    xn[0] = mv;
    xn[1] = (py + vel) + P[2] * z0;
    return lp;
}
/* stratified_map! (utils.jl:29-55): block size B = n div K; i < K B -> stratum i div B (:contiguous) or i mod K
 * (:interleaved); the remaining particles draw a stratum uniformly (sample(strata, n_remaining), utils.jl:47) from one
 * more Philox block of the particle (block index = the model's block count) */
static int stratum_of(int model, int K, int interleaved, uint64_t seed, uint32_t epoch, int64_t gid0, int64_t i, int64_t n, uint32_t tag)
{
    int64_t B = n / K;
    if (i < (int64_t)K * B) return (int)(interleaved ? i % K : i / B);
    o_philox_t b = o_rng(seed, O_GID(gid0, i), (uint32_t)model_nblk(model), epoch, tag);
    return (int)o_mulhi64(((uint64_t)b.v[0] << 32) | b.v[1], (uint64_t)K);
}
O_EXPORT void o_init_strata(int model, const double *P, uint64_t seed, uint32_t epoch, int64_t gid0, int64_t n, int W,
                            const double *obs, const double *values, int K, int interleaved, double logK, double *rows, double *lw)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double *r = rows + i * W;
        for (int k = 0; k < W; ++k) r[k] = 0.0;
        double v = values[stratum_of(model, K, interleaved, seed, epoch, gid0, i, n, O_TAG_INIT)];
        double lp = model_sample_stratum(model, P, 1, NULL, obs, v, seed, O_GID(gid0, i), 0, epoch, O_TAG_INIT, r);
        lw[i] = (lp + model_loglik(model, P, r, obs)) + logK;           /* initialize.jl:103-104 */
    }
}
/* pf_initialize(model, args, obs, strata, proposal, proposal_args, n) (initialize.jl:111-129) for line_model as test/initialize.jl:66-90
 * uses it: strata over slope, outlier_propose = bernoulli(0.0): log_weights[i] = model_weight - prop_weight (= 0) + log(n_strata) */
O_EXPORT void o_init_strata_proposal(int model, const double *P, uint64_t seed, uint32_t epoch, int64_t gid0, int64_t n, int W,
                                     const double *obs, const double *values, int K, int interleaved, double logK, double *rows, double *lw)
{
    for (int64_t i = 0; i < n; ++i) {
        double *r = rows + i * W;
        for (int k = 0; k < W; ++k) r[k] = 0.0;
        double v = values[stratum_of(model, K, interleaved, seed, epoch, gid0, i, n, O_TAG_INIT)];
        r[0] = v; r[1] = 0.0;                                           /* :122-123: stratum + proposed outlier = false */
        double w = P[7];                                                /* log p(slope = v) = log(1/5) */
        if (obs[1] != 0.0) w = (w + P[6]) + model_loglik(model, P, r, obs);   /* generate(model, args, merge(stratum, obs, prop_choices)) :124 */
        lw[i] = w + logK;                                               /* :125 */
    }
}
O_EXPORT void o_step_strata(int model, const double *P, uint64_t seed, uint32_t epoch, int64_t gid0, int64_t n, int W, int keep_prev,
                            const double *obs, const double *values, int K, int interleaved, double logK,
                            const double *rows_in, double *rows_out, double *lw)
{
    int d = model_dim(model);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const double *ri = rows_in + i * W;
        double *ro = rows_out + i * W;
        double xn[4], xp[4];
        double v = values[stratum_of(model, K, interleaved, seed, epoch, gid0, i, n, O_TAG_UPDATE)];
        double lp = model_sample_stratum(model, P, 0, ri, obs, v, seed, O_GID(gid0, i), 0, epoch, O_TAG_UPDATE, xn);
        for (int k = 0; k < d; ++k) xp[k] = ri[k];
        for (int k = 0; k < W; ++k) ro[k] = 0.0;
        for (int k = 0; k < d; ++k) ro[k] = xn[k];
        if (keep_prev) for (int k = 0; k < d; ++k) ro[d + k] = xp[k];
        lw[i] = lw[i] + ((lp + model_loglik(model, P, xn, obs)) + logK); /* update.jl:201-206 */
    }
}

/* pf_initialize default proposal, initialize.jl:39-41: x ~ prior, log_weights[i] = log p(y1|x) */
O_EXPORT void o_init(int model, const double *P, uint64_t seed, uint32_t epoch, int64_t gid0, int64_t n,
                     int W, const double *obs, double *rows, double *lw)
{
    int d = model_dim(model);
    #pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double *r = rows + i * W;
        for (int k = 0; k < W; ++k) r[k] = 0.0;
        model_sample(model, P, 1, NULL, obs, seed, O_GID(gid0, i), 0, epoch, O_TAG_INIT, r);
        lw[i] = model_loglik(model, P, r, obs);
        (void)d;
    }
}

/* pf_update! default proposal, update.jl:15-22: x_t ~ p(.|x_{t-1}); log_weights[i] += increment.
 * keep_prev: the row also carries x_{t-1} in columns d..2d-1. */
O_EXPORT void o_step(int model, const double *P, uint64_t seed, uint32_t epoch, int64_t gid0, int64_t n,
                     int W, int keep_prev, const double *obs, const double *rows_in, double *rows_out,
                     double *lw)
{
    int d = model_dim(model);
    #pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const double *ri = rows_in + i * W;
        double *ro = rows_out + i * W;
        double xn[4];
        model_sample(model, P, 0, ri, obs, seed, O_GID(gid0, i), 0, epoch, O_TAG_UPDATE, xn);
        double ll = model_loglik(model, P, xn, obs);
        double xp[4];
        for (int k = 0; k < d; ++k) xp[k] = ri[k];
        for (int k = 0; k < W; ++k) ro[k] = 0.0;
        for (int k = 0; k < d; ++k) ro[k] = xn[k];
        if (keep_prev) for (int k = 0; k < d; ++k) ro[d + k] = xp[k];
        lw[i] = lw[i] + ll;
    }
}

/* pf_move_accept! (rejuvenate.jl:40-53) with kern = Gen.mh(trace, select(current step latent)):
 *   regenerate x_t from p(.|x_{t-1}) (or the prior when no previous step exists), accept iff
 *   log(rand()) < weight, weight = log p(y_t|x*) - log p(y_t|x)   [Gen semantics, SURVEY App. B]
 * pf_move_reweight! (rejuvenate.jl:74-90) with kern = move_reweight(trace, selection) (:125-132):
 *   always move; log_weights[i] += sum of rel_weight.
 * reweight = 0 -> move-accept, 1 -> move-reweight.  Returns number of accepted moves. */
O_EXPORT uint64_t o_move(int model, const double *P, uint64_t seed, uint32_t epoch, int64_t gid0, int64_t n,
                         int W, int has_prev, const double *obs, int n_iters, int reweight,
                         const double *rows_in, double *rows_out, double *lw)
{
    int d = model_dim(model), nb = model_nblk(model);
    uint64_t nacc = 0;
    #pragma omp parallel for schedule(static) reduction(+:nacc)
    for (int64_t i = 0; i < n; ++i) {
        const double *ri = rows_in + i * W;
        double *ro = rows_out + i * W;
        double x[4], xs[4];
        const double *xp = ri + d;
        for (int k = 0; k < d; ++k) x[k] = ri[k];
        double llx = model_loglik(model, P, x, obs);
        double wsum = 0.0;
        for (int it = 0; it < n_iters; ++it) {
            if (reweight) {
                uint32_t blk0 = (uint32_t)(it * nb);
                model_sample(model, P, !has_prev, xp, obs, seed, O_GID(gid0, i), blk0, epoch,
                             O_TAG_REWEIGHT, xs);
                double lls = model_loglik(model, P, xs, obs);
                wsum = wsum + (lls - llx);                 /* rejuvenate.jl:82 */
                for (int k = 0; k < d; ++k) x[k] = xs[k];
                llx = lls;
                nacc++;
            } else {
                uint32_t blk0 = (uint32_t)(it * (nb + 1));
                model_sample(model, P, !has_prev, xp, obs, seed, O_GID(gid0, i), blk0, epoch,
                             O_TAG_MOVE, xs);
                double lls = model_loglik(model, P, xs, obs);
                o_philox_t b = o_rng(seed, O_GID(gid0, i), blk0 + (uint32_t)nb, epoch, O_TAG_MOVE);
                double lu = o_log(o_u52(b.v[0], b.v[1]));
                if (lu < lls - llx) {                     /* Gen.mh: log(rand()) < weight */
                    for (int k = 0; k < d; ++k) x[k] = xs[k];
                    llx = lls;
                    nacc++;
                }
            }
        }
        for (int k = 0; k < W; ++k) ro[k] = ri[k];
        for (int k = 0; k < d; ++k) ro[k] = x[k];
        if (reweight) lw[i] = lw[i] + wsum;               /* rejuvenate.jl:86 */
    }
    return nacc;
}

/* move_reweight(trace, proposal, proposal_args) (rejuvenate.jl:134-148) with a native proposal of the current step's latent:
 *   fwd_choices, fwd_score = propose(proposal, (trace, args...))           :140-141
 *   new_trace, weight, _, discard = update(trace, ..., fwd_choices)        :142-143   weight = model score(new) - model score(old)
 *   bwd_score = assess(proposal, (new_trace, args...), discard)            :144-145
 *   rel_weight = weight - fwd_score + bwd_score                            :147
 * LGSSM2: the locally optimal proposal q = p(x_t | x_{t-1}, y_t): rel_weight = W(x') - W(x), W = (log p(x|x_prev) + log p(y|x)) - log q(x)
 * LINE:   outlier ~ bernoulli(Q[0]) for the current step (test/rejuvenate.jl:19-27), Q = {q, log q, log(1 - q)} */
static double lgssm2_proposal_weight(const double *P, int first, const double *xp, const double *obs, const double *x)
{
    double mu0 = first ? 0.0 : P[0] * xp[0] + P[1] * xp[1];
    double mu1 = first ? 0.0 : P[2] * xp[0] + P[3] * xp[1];
    int o = first ? 14 : 8;
    double m0 = mu0 + P[o] * (obs[0] - mu0), m1 = mu1 + P[o] * (obs[1] - mu1);
    double a0 = (x[0] - mu0) * P[o + 4], a1 = (x[1] - mu1) * P[o + 4];
    double lt = -0.5 * (a0 * a0 + a1 * a1) - P[o + 5];
    double b0 = (x[0] - m0) * P[o + 2], b1 = (x[1] - m1) * P[o + 2];
    double lq = -0.5 * (b0 * b0 + b1 * b1) - P[o + 3];
    return (lt + model_loglik(O_MODEL_LGSSM2, P, x, obs)) - lq;
}
static double model_move_propose(int model, const double *P, const double *Q, int first, const double *xp, const double *x,
                                 const double *obs, uint64_t seed, uint32_t gid, uint32_t blk0, uint32_t epoch, uint32_t tag, double *xn)
{
    if (model == O_MODEL_LINE) {
        o_philox_t b = o_rng(seed, gid, blk0, epoch, tag);
        xn[0] = x[0];
        xn[1] = (obs[1] != 0.0 && o_u52(b.v[2], b.v[3]) < Q[0]) ? 1.0 : 0.0;
        if (obs[1] == 0.0) return 0.0;
        int on = xn[1] != 0.0, oo = x[1] != 0.0;
        double wn = (on ? P[5] : P[6]) + model_loglik(model, P, xn, obs), wo = (oo ? P[5] : P[6]) + model_loglik(model, P, x, obs);
        return ((wn - wo) - (on ? Q[1] : Q[2])) + (oo ? Q[1] : Q[2]);        /* :147 */
    }
    if (model != O_MODEL_LGSSM2) return NAN;
    double wn = model_propose(model, P, first, xp, obs, seed, gid, blk0, epoch, tag, xn);
    return wn - lgssm2_proposal_weight(P, first, xp, obs, x);
}
/* pf_move_reweight! (rejuvenate.jl:74-90) with the proposal variant of the kernel; every particle moves, log_weights[i] += sum rel_weight */
O_EXPORT void o_move_proposal(int model, const double *P, const double *Q, uint64_t seed, uint32_t epoch, int64_t gid0, int64_t n,
                              int W, int has_prev, const double *obs, int n_iters, const double *rows_in, double *rows_out, double *lw)
{
    int d = model_dim(model), nb = model_nblk(model);
    #pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const double *ri = rows_in + i * W;
        double *ro = rows_out + i * W;
        double x[4], xs[4];
        const double *xp = ri + d;
        for (int k = 0; k < d; ++k) x[k] = ri[k];
        double wsum = 0.0;
        for (int it = 0; it < n_iters; ++it) {
            double rw = model_move_propose(model, P, Q, !has_prev, xp, x, obs, seed, O_GID(gid0, i), (uint32_t)(it * nb), epoch, O_TAG_REWEIGHT, xs);
            wsum = wsum + rw;                             /* rejuvenate.jl:86 */
            for (int k = 0; k < d; ++k) x[k] = xs[k];
        }
        for (int k = 0; k < W; ++k) ro[k] = ri[k];
        for (int k = 0; k < d; ++k) ro[k] = x[k];
        lw[i] = lw[i] + wsum;
    }
}

/* pf_move_accept! (rejuvenate.jl:40-53) with Gen.mh(trace, proposal, proposal_args): propose -> update -> assess the reverse move;
 * alpha = weight - fwd_score + bwd_score (the quantity move_reweight calls rel_weight, :146); accept iff log(rand()) < alpha; the
 * weights are untouched.  RNG: proposal blocks blk0 .. blk0 + nb - 1, the uniform in block blk0 + nb, tag MOVE (as o_move's selection variant). */
O_EXPORT uint64_t o_move_proposal_accept(int model, const double *P, const double *Q, uint64_t seed, uint32_t epoch, int64_t gid0, int64_t n,
                                         int W, int has_prev, const double *obs, int n_iters, const double *rows_in, double *rows_out)
{
    int d = model_dim(model), nb = model_nblk(model);
    uint64_t acc = 0;
    #pragma omp parallel for schedule(static) reduction(+:acc)
    for (int64_t i = 0; i < n; ++i) {
        const double *ri = rows_in + i * W;
        double *ro = rows_out + i * W;
        double x[4], xs[4];
        const double *xp = ri + d;
        for (int k = 0; k < d; ++k) x[k] = ri[k];
        for (int it = 0; it < n_iters; ++it) {
            uint32_t blk0 = (uint32_t)(it * (nb + 1));
            double alpha = model_move_propose(model, P, Q, !has_prev, xp, x, obs, seed, O_GID(gid0, i), blk0, epoch, O_TAG_MOVE, xs);
            o_philox_t b = o_rng(seed, O_GID(gid0, i), blk0 + (uint32_t)nb, epoch, O_TAG_MOVE);
            if (o_log(o_u52(b.v[0], b.v[1])) < alpha) { for (int k = 0; k < d; ++k) x[k] = xs[k]; ++acc; }
        }
        for (int k = 0; k < W; ++k) ro[k] = ri[k];
        for (int k = 0; k < d; ++k) ro[k] = x[k];
    }
    return acc;
}

/* ------------------------------------------------------------------ weight normalisation */
/* maximum + validity flags of safe_softmax (utils.jl:119-126): any NaN; all == -Inf; (+Inf present
 * makes exp.(vs .- max) contain NaN -> "total weight is NaN" branch, utils.jl:134-137) */
O_EXPORT void o_max_flags(const double *lp, int64_t n, double *m_out, int *flags_out)
{
    double m = -INFINITY; int nan = 0, pinf = 0;
    for (int64_t i = 0; i < n; ++i) {
        double v = lp[i];
        if (v != v) nan = 1;
        else { if (v > m) m = v; if (v == INFINITY) pinf = 1; }
    }
    int f = 0;
    if (nan) f |= O_FLAG_NAN;
    if (pinf) f |= O_FLAG_POSINF;
    if (!nan && m == -INFINITY) f |= O_FLAG_ALL_NEGINF;
    *m_out = m; *flags_out = f;
}

/* ws = exp.(vs .- maximum(vs)) (utils.jl:128) in K-bit fixed point; uniform fallback q=1 when
 * invalid-but-continuable (utils.jl:123-126,130-133) */
O_EXPORT void o_fixq(const double *lp, int64_t n, double m, int K, int uniform, uint64_t *q)
{
    #pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) q[i] = uniform ? 1u : o_exp_fix(lp[i] - m, K);
}

/* inclusive prefix sum (the CDF every resampler walks: resample.jl:59,113,163-166), exact in u64;
 * also sum of squares in u128 for the ESS */
O_EXPORT uint64_t o_scan(const uint64_t *q, int64_t n, uint64_t *cdf, uint64_t *Qhi, uint64_t *Qlo)
{
    uint64_t s = 0; o_u128 Q = 0;
    for (int64_t i = 0; i < n; ++i) { s += q[i]; if (cdf) cdf[i] = s; Q += (o_u128)q[i] * q[i]; }
    if (Qhi) *Qhi = (uint64_t)(Q >> 64);
    if (Qlo) *Qlo = (uint64_t)Q;
    return s;
}

/* ------------------------------------------------------------------ ancestor targets */
/* multinomial (resample.jl:59): slot j draws T = floor(U_j * S / 2^64), U_j the 64-bit uniform of
 * slot j of the resample stream (o_resample_u64: block j >> 1, word pair j & 1); ancestor = the particle whose CDF cell [cdf[a-1], cdf[a]) holds T */
O_EXPORT void o_targets_multinomial(uint64_t seed, uint32_t epoch, int64_t j0, int64_t n, uint64_t S,
                                    uint64_t *T)
{
    #pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < n; ++j) {
        T[j] = o_mulhi64(o_resample_u64(seed, (uint32_t)(j0 + j), epoch), S);
    }
}
/* stratified (resample.jl:159-167): stratum j = [L_j, L_{j+1}), L_j = floor(j*S/N) computed exactly as
 * j*B + floor(j*rem/N) with S = N*B + rem; T = L_j + floor(U_j * len_j / 2^64).
 * This is u = rand()*step + lower (resample.jl:162) on the integer grid of the fixed-point CDF. */
O_EXPORT void o_targets_stratified(uint64_t seed, uint32_t epoch, int64_t j0, int64_t n, int64_t N,
                                   uint64_t S, uint64_t *T)
{
    uint64_t B = S / (uint64_t)N, rem = S % (uint64_t)N;
    #pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < n; ++j) {
        uint64_t jg = (uint64_t)(j0 + j);
        uint64_t L0 = jg * B + (jg * rem) / (uint64_t)N;
        uint64_t L1 = (jg + 1) * B + ((jg + 1) * rem) / (uint64_t)N;
        T[j] = L0 + o_mulhi64(o_resample_u64(seed, (uint32_t)jg, epoch), L1 - L0);
    }
}
/* "multinomial_sorted" -- an OPT-IN variant of the multinomial resampler (resample.jl:59: N i.i.d. categorical draws) whose ancestors
 * come out in NON-DECREASING order: the N uniforms are drawn already sorted.  NOT the reference's slot order (an i.i.d. sequence); the
 * multiset of ancestors has the same law (offspring counts ~ Multinomial(N, w)), which is all a filter step that treats the particles
 * exchangeably can see.  Construction: uniform spacings.  With e_0..e_N i.i.d. Exp(1) and P_j = e_0 + ... + e_j, the
 * ratios P_j / P_N (j < N) are the order statistics of N uniforms.  The sum over a TILE of O_SP_TILE consecutive slots is Gamma(tile
 * size) and independent of the tile's normalised partial sums, so the tile totals are drawn DIRECTLY (one Marsaglia-Tsang gamma variate
 * per tile: no pass over all spacings is needed to place a tile, on one GPU or across shards) and the spacings only place the slots
 * inside their tile:
 *     G_t   = trunc(gamma(c_t [+ 1 for the last tile: the (N+1)-th spacing]) 2^Eg),  Gtot = sum G_t + 1
 *     Vlo_t = floor((G_0 + ... + G_{t-1}) 2^64 / Gtot)
 *     e_i   = trunc(neglog(u_i) 2^44),  u_i the 52-bit uniform of resample slot j0 + i, neglog = -log to ~6e-12 (o_neglog_u52);  p_j = sum of the e_i of the tile up to slot j;
 *     s_t   = the tile's sum + 1 (the last tile: + e_N)
 *     Tlo_t = floor(Vlo_t S / 2^64)  (the multinomial target formula on the tile's first uniform),  Tw_t = Tlo_{t+1} - Tlo_t
 *     T_j   = Tlo_t + min(Tw_t, trunc(fl(fl(p_j) fl(1 / fl(s_t))) fl(Tw_t)))      (fl: round to Float64)  */
#define O_SP_TILE 2048
#define O_SP_E 44
O_EXPORT int32_t o_gamma_E(int64_t ntl)
{
    int c = 0;
    while (((int64_t)1 << c) < ntl) ++c;
    return 50 - c > 48 ? 48 : 50 - c;                 /* a tile total is < 2^12 (see o_gamma_tile): the sum of ntl of them < 2^62 */
}
/* Gamma(shape, 1), shape >= 1 an integer, by Marsaglia & Tsang (2000): d = shape - 1/3, c = 1 / sqrt(9 d); x ~ N(0,1), v = (1 + c x)^3,
 * accept when v > 0 and log u < x^2 / 2 + d - d v + d log v.  Attempt k reads Philox blocks 1 + 2k (the normal) and 2 + 2k (the uniform)
 * of counter `gid` on the resample stream (block 0 belongs to the slots' own uniforms); 8 attempts, then d (never in practice: the
 * acceptance rate is > 0.95 for shape >= 1 and > 0.999 for a full tile).  |x| <= 8.6 bounds the variate below 1.2 shape + 60 < 2^12. */
static uint64_t o_gamma_tile(uint64_t seed, uint32_t gid, uint32_t epoch, int64_t shape, int Eg)
{
    const double d = (double)shape - 1.0 / 3.0;
    const double c = 1.0 / sqrt(9.0 * d);
    const double sc = o_u2d((uint64_t)(Eg + 1023) << 52);
    for (int k = 0; k < 8; ++k) {
        double x, x1;
        o_normal2(o_rng(seed, gid, (uint32_t)(1 + 2 * k), epoch, O_TAG_RESAMPLE), &x, &x1);
        const o_philox_t b = o_rng(seed, gid, (uint32_t)(2 + 2 * k), epoch, O_TAG_RESAMPLE);
        const double u = o_u52(b.v[0], b.v[1]);
        const double v1 = 1.0 + c * x;
        if (!(v1 > 0.0)) continue;
        const double v = (v1 * v1) * v1;
        if (o_log(u) < ((0.5 * (x * x) + d) - d * v) + d * o_log(v)) return (uint64_t)((d * v) * sc);
    }
    return (uint64_t)(d * sc);
}
static inline uint64_t o_spacing(uint64_t seed, uint32_t slot, uint32_t epoch)
{
    const uint64_t U = o_resample_u64(seed, slot, epoch);
    return (uint64_t)(o_neglog_u52(U) * o_u2d((uint64_t)(O_SP_E + 1023) << 52));      /* exact scaling by 2^E; truncation = floor */
}
O_EXPORT uint64_t o_spacing_d(uint64_t seed, uint32_t slot, uint32_t epoch) { return o_spacing(seed, slot, epoch); }
O_EXPORT uint64_t o_gamma_tile_d(uint64_t seed, uint32_t gid, uint32_t epoch, int64_t shape, int32_t Eg) { return o_gamma_tile(seed, gid, epoch, shape, Eg); }
O_EXPORT void o_targets_sorted(uint64_t seed, uint32_t epoch, int64_t j0, int64_t n, uint64_t S, uint64_t *T)
{
    const int64_t ntl = (n + O_SP_TILE - 1) / O_SP_TILE;
    const int Eg = o_gamma_E(ntl);
    uint64_t *G = (uint64_t *)malloc((size_t)(ntl + 1) * sizeof(uint64_t));      /* exclusive prefix of the tile totals */
    uint64_t acc = 0;
    for (int64_t t = 0; t < ntl; ++t) {
        const int64_t first = t * O_SP_TILE, cnt = (first + O_SP_TILE <= n ? O_SP_TILE : n - first);
        G[t] = acc;
        acc += o_gamma_tile(seed, (uint32_t)(j0 + first), epoch, cnt + (t == ntl - 1 ? 1 : 0), Eg);
    }
    G[ntl] = acc;
    const uint64_t Gtot = acc + 1;
    #pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < ntl; ++t) {
        const int64_t first = t * O_SP_TILE, cnt = (first + O_SP_TILE <= n ? O_SP_TILE : n - first);
        const uint64_t Vlo = (uint64_t)((((o_u128)G[t]) << 64) / Gtot), Vhi = (uint64_t)((((o_u128)G[t + 1]) << 64) / Gtot);
        uint64_t s = 0;
        for (int64_t k = 0; k < cnt; ++k) { s += o_spacing(seed, (uint32_t)(j0 + first + k), epoch); T[first + k] = s; }   /* p_j for now */
        s += 1;                                                   /* p_j < s_t strictly (and s_t > 0) */
        if (t == ntl - 1) s += o_spacing(seed, (uint32_t)(j0 + n), epoch);       /* the (N+1)-th spacing */
        /* the tile covers the targets [Tlo, Thi]; a slot sits at the fraction p_j / s_t of it -- in Float64 (one rounding each for the
         * conversions, the product with 1 / s_t and the product with the width: monotone in p_j; the integer division this replaces
         * cost the GPU ~36 32-bit multiplications per slot) */
        const uint64_t Tlo = o_mulhi64(Vlo, S), Tw = o_mulhi64(Vhi, S) - Tlo;
        const double inv_s = 1.0 / (double)s, dTw = (double)Tw;
        for (int64_t k = 0; k < cnt; ++k) {
            uint64_t tt = (uint64_t)(((double)T[first + k] * inv_s) * dTw);
            if (tt > Tw) tt = Tw;
            T[first + k] = Tlo + tt;
        }
    }
    free(G);
}
/* sub-state views: strata are LOCAL to the view (index j of n), the RNG counter keeps the global particle id gid0 + j */
O_EXPORT void o_targets_stratified_view(uint64_t seed, uint32_t epoch, int64_t gid0, int64_t n, uint64_t S, uint64_t *T)
{
    uint64_t B = S / (uint64_t)n, rem = S % (uint64_t)n;
    for (int64_t j = 0; j < n; ++j) {
        uint64_t jl = (uint64_t)j;
        uint64_t L0 = jl * B + (jl * rem) / (uint64_t)n;
        uint64_t L1 = (jl + 1) * B + ((jl + 1) * rem) / (uint64_t)n;
        T[j] = L0 + o_mulhi64(o_resample_u64(seed, (uint32_t)(gid0 + j), epoch), L1 - L0);
    }
}
/* first index a with cdf[a] > T  (== the while loop of resample.jl:163-166 / inverse-CDF categorical) */
O_EXPORT void o_upper_bound(const uint64_t *cdf, int64_t n, const uint64_t *T, int64_t m, int64_t *idx)
{
    #pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < m; ++j) {
        int64_t lo = 0, hi = n;
        uint64_t t = T[j];
        while (lo < hi) { int64_t mid = lo + (hi - lo) / 2; if (cdf[mid] > t) hi = mid; else lo = mid + 1; }
        idx[j] = lo < n ? lo : n - 1;
    }
}

/* residual (resample.jl:96-115): n_copies = floor(N*w_i) = (N*q_i) div S exactly;
 * residual weight N*w_i - floor(N*w_i) = ((N*q_i) mod S)/S, kept as ((N*q_i) mod S) >> sh with
 * sh = max(0, bitlen(S) + ceil_log2(N) - 62) so the residual CDF also fits 62 bits. */
O_EXPORT int o_residual_shift(uint64_t S, int64_t N)
{
    int bl = 0; while (bl < 64 && (S >> bl) != 0) ++bl;
    int cl = 0; while (((int64_t)1 << cl) < N) ++cl;
    int sh = bl + cl - 62;
    return sh > 0 ? sh : 0;
}
O_EXPORT void o_residual_split(const uint64_t *q, int64_t n, int64_t N, uint64_t S, int sh,
                               uint64_t *c, uint64_t *r)
{
    #pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        uint64_t nq = (uint64_t)N * q[i];
        c[i] = nq / S;
        r[i] = (nq % S) >> sh;
    }
}

/* ------------------------------------------------------------------ gather, sort, statistics */
O_EXPORT void o_gather_rows(const double *rows, int W, const int64_t *idx, int64_t n, double *out)
{
    #pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < n; ++j)
        for (int k = 0; k < W; ++k) out[j * W + k] = rows[idx[j] * W + k];
}

/* order-preserving key: Julia isless total order on floats (-0.0 < 0.0, NaN last) */
static uint64_t f64_key(double x)
{
    uint64_t u = o_d2u(x);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
typedef struct { uint64_t key; int64_t idx; } o_kv;
static int kv_cmp_desc(const void *a, const void *b)
{
    const o_kv *x = a, *y = b;
    if (x->key != y->key) return x->key > y->key ? -1 : 1;     /* descending by value */
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);    /* stable: ties keep index order */
}
/* sortperm(log_priorities, rev=true), resample.jl:156-157 (stable) */
O_EXPORT void o_argsort_desc(const double *lp, int64_t n, int64_t *order)
{
    o_kv *kv = (o_kv *)malloc((size_t)n * sizeof(o_kv));
    for (int64_t i = 0; i < n; ++i) { kv[i].key = f64_key(lp[i]); kv[i].idx = i; }
    qsort(kv, (size_t)n, sizeof(o_kv), kv_cmp_desc);
    for (int64_t i = 0; i < n; ++i) order[i] = kv[i].idx;
    free(kv);
}

/* The order of a Float64 sum is part of the spec: the terms are added by the perfect binary tree over their
 * indices (neighbours first), in chunks of 2048 terms (missing terms are +0.0) whose partials are summed by the same tree again.
 * t is destroyed; n >= 1. */
#define O_TREE_CHUNK 2048
static double o_tree_sum(double *t, int64_t n)
{
    double buf[O_TREE_CHUNK];
    while (1) {
        int64_t nb = (n + O_TREE_CHUNK - 1) / O_TREE_CHUNK;
        for (int64_t b = 0; b < nb; ++b) {
            for (int j = 0; j < O_TREE_CHUNK; ++j) buf[j] = b * O_TREE_CHUNK + j < n ? t[b * O_TREE_CHUNK + j] : 0.0;
            for (int w = 1; w < O_TREE_CHUNK; w *= 2)
                for (int i = 0; i < O_TREE_CHUNK; i += 2 * w) buf[i] = buf[i] + buf[i + w];
            t[b] = buf[0];
        }
        if (nb == 1) return t[0];
        n = nb;
    }
}
/* sum_i q_i * f(x_i[col]) / S with f = identity (pw=1), square of (x - c) (pw=2) or [x == c] (pw=3):
 * statistics.jl:13-14 (mean), :48-50 (var), :91-101 (proportionmap); the reference accumulates sequentially, the spec sums the
 * same terms by o_tree_sum (a parallel machine cannot add one after the other; the tree also has the smaller error bound) */
O_EXPORT double o_wsum(const uint64_t *q, uint64_t S, const double *rows, int W, int col, int64_t n,
                       int pw, double c)
{
    double Sd = (double)S;
    double *t = (double *)malloc((size_t)(n > 0 ? n : 1) * sizeof(double));
    for (int64_t i = 0; i < n; ++i) {
        double w = (double)q[i] / Sd;
        double v = rows[i * W + col];
        if (pw == 2) { v = v - c; v = v * v; }
        if (pw == 3) v = (v == c) ? 1.0 : 0.0;
        t[i] = w * v;
    }
    double r = n > 0 ? o_tree_sum(t, n) : 0.0;
    free(t);
    return r;
}

/* synthetic data generator shared by tests/bench (DATA stream, tag 7): y = truth + noise for the SSMs */
O_EXPORT void o_normals(uint64_t seed, uint32_t epoch, int64_t n, double *out)
{
    for (int64_t i = 0; i < n; i += 2) {
        double z0, z1;
        o_normal2(o_rng(seed, (uint32_t)(i / 2), 0, epoch, O_TAG_DATA), &z0, &z1);
        out[i] = z0; if (i + 1 < n) out[i + 1] = z1;
    }
}

/* ------------------------------------------------------------------ resize family (src/resize.jl) */
/* pf_dereplicate! method=:sample (resize.jl:281-293): per block of k replicates one categorical draw from the
 * block's softmax (K_b-bit fixed point, K_b = fix_K(k)); new weight = logsumexp(block) - log(k) */
O_EXPORT void o_dereplicate_sample(const double *lw, int64_t n_new, int64_t n_old, int k, int interleaved,
                                   uint64_t seed, uint32_t epoch, int64_t *anc, double *lw_out)
{
    (void)n_old;
    const int Kb = o_fix_K(k);
    const double logk = o_log((double)k);
    const int64_t stride = interleaved ? n_new : 1;
    for (int64_t j = 0; j < n_new; ++j) {
        const int64_t first = interleaved ? j : j * k;
        double m = -INFINITY; int nan = 0;
        for (int e = 0; e < k; ++e) { double v = lw[first + e * stride]; if (v != v) nan = 1; else if (v > m) m = v; }
        int uniform = !nan && m == -INFINITY;
        uint64_t S = 0;
        for (int e = 0; e < k; ++e) S += uniform ? 1 : o_exp_fix(lw[first + e * stride] - m, Kb);
        o_philox_t b = o_rng(seed, (uint32_t)j, 0, epoch, O_TAG_RESAMPLE);
        uint64_t T = o_mulhi64(o_u64(b.v[0], b.v[1]), S), acc = 0;
        int pick = k - 1;
        for (int e = 0; e < k; ++e) {
            acc += uniform ? 1 : o_exp_fix(lw[first + e * stride] - m, Kb);
            if (acc > T) { pick = e; break; }
        }
        anc[j] = first + pick * stride;
        lw_out[j] = o_lse_from(m, S, Kb, nan ? O_FLAG_NAN : (uniform ? O_FLAG_ALL_NEGINF : 0)) - logk;
    }
}
