// gpf_models.hpp -- native state-space models: the per-particle work that the reference hands to
// Gen's generate/update/regenerate (src/initialize.jl:40, src/update.jl:17, src/rejuvenate.jl:46,81).
//
// A model is a pair of device functions over a d-column Float64 state:
//   sample(P, first, xprev, obs, rng-counter) -> x      the model's internal proposal for the new
//                                                       latent choices (ancestral sampling)
//   loglik(P, x, obs) -> log p(y_t | x_t)               the weight increment of the constrained obs
// Parameter vectors P (incl. derived constants such as log sigma) are built on the host
// (models.py) so the CPU oracle and the kernels receive bit-identical inputs.
// Operation order inside each expression is part of the spec.
#pragma once
#include "gpf_math.hpp"

namespace gpf {

enum : int { MODEL_LGSSM2 = 1, MODEL_BEARINGS4 = 2, MODEL_SV1 = 3, MODEL_OBJECT_MOTION = 4, MODEL_LINE = 5 };

constexpr int MAX_PARAMS = 24;
constexpr int MAX_OBS = 4;
constexpr int MAX_DIM = 4;
constexpr int MAX_STRATA = 8;

struct ModelArgs {          // passed by value in the kernarg segment: no device copy per step
    double P[MAX_PARAMS];
    double obs[MAX_OBS];
    // stratified initialisation / update (reference src/initialize.jl:92-109, src/update.jl:193-210, src/utils.jl:29-55)
    double strata[MAX_STRATA];   // value of the model's discrete latent in each stratum
    double logK;                 // log(n_strata)
    int32_t n_strata, interleaved;
    // strided sub-state views (reference src/view.jl:35-48 with idxs = start:step:stop): local particle i is particle
    // gid0 + i * gstride of the filter and keeps THAT id as its RNG counter (1 everywhere else)
    int32_t gstride, pad_;
    // views over an arbitrary index vector (state[idxs], src/view.jl:35-48): local particle i is particle gid0 + gid_map[i] (gid_map[i] =
    // idxs[i] - idxs[0], device array), nullptr everywhere else
    const int32_t* gid_map;
    double q[4];                 // parameters of a native MOVE proposal (gpf_rejuvenate_proposal), e.g. {p, log p, log(1 - p)}
    // block-wise operations (many small filters in one state, gpf_update_blocks & co.): particle i belongs to block i / blk_size and
    // sees that block's observation blk_obs[block][MAX_OBS]; blk_mask (rejuvenation): bit 0 of word [block] = the block takes part
    const double* blk_obs; const int32_t* blk_mask; int32_t blk_size, pad2_;
    // gpf_update_blocks_proposal: word [block] != 0 = the block's particles are extended with the model's native proposal (MODE 4 of k_step)
    const int32_t* blk_prop;
};
// the observation particle i conditions on
template <bool BLK>
__device__ __forceinline__ const double* obs_of(const ModelArgs& a, int64_t i)
{
    if constexpr (BLK) return a.blk_obs + (size_t)((uint32_t)i / (uint32_t)a.blk_size) * MAX_OBS;
    else return a.obs;
}

// the RNG counter of local particle i: its id in the whole filter
__device__ __forceinline__ uint32_t particle_gid(const ModelArgs& a, int64_t gid0, int64_t i)
{
    return a.gid_map ? (uint32_t)(gid0 + (int64_t)a.gid_map[i]) : (uint32_t)(gid0 + i * a.gstride);
}

template <int M> struct Model;

// 2-D linear-Gaussian SSM: x' = A x + sq z, y = x + sr e   (BASELINE configs 2, 3)
// P = [a11 a12 a21 a22 | sq | s0 | 1/sr | 2(log sr + log(2 pi)/2) |
//      locally optimal proposal, transition: gain sv 1/sv 2(log sv + ..) 1/sq 2(log sq + ..) | initial: gain0 sv0 1/sv0 2(log sv0 + ..) 1/s0 2(log s0 + ..)]
template <> struct Model<MODEL_LGSSM2> {
    static constexpr bool HAS_STRATA = false;
    static constexpr bool HAS_STRATA_PROPOSAL = false;
    static constexpr int D = 2, NBLK = 1;
    static constexpr bool HAS_PROPOSAL = true;
    // custom-proposal update (reference src/update.jl:79-96, src/translate.jl:86-105 without transform;
    // src/initialize.jl:46-62): x ~ q(. | x_{t-1}, y_t), the locally optimal proposal of the linear-Gaussian model;
    // returns log_weight = model_score_diff - fwd_proposal_score = [log p(x|x_{t-1}) + log p(y|x)] - log q(x)
    static GPF_HD double propose(const double* P, bool first, const double* xp, const double* obs, uint64_t seed,
                                 uint32_t gid, uint32_t blk0, uint32_t epoch, uint32_t tag, double* xn)
    {
        double z0, z1;
        normal2(rng(seed, gid, blk0, epoch, tag), z0, z1);
        const double mu0 = first ? 0.0 : P[0] * xp[0] + P[1] * xp[1];
        const double mu1 = first ? 0.0 : P[2] * xp[0] + P[3] * xp[1];
        const int o = first ? 14 : 8;                  // gain, sv, 1/sv, cq, 1/s_prior, c_prior
        const double m0 = mu0 + P[o] * (obs[0] - mu0), m1 = mu1 + P[o] * (obs[1] - mu1);
        xn[0] = m0 + P[o + 1] * z0;
        xn[1] = m1 + P[o + 1] * z1;
        return proposal_weight(P, first, xp, obs, xn);
    }
    // [log p(x | x_{t-1}) + log p(y | x)] - log q(x | x_{t-1}, y) for a GIVEN x (the weight of `propose`; the two scores of a move)
    static GPF_HD double proposal_weight(const double* P, bool first, const double* xp, const double* obs, const double* x)
    {
        const double mu0 = first ? 0.0 : P[0] * xp[0] + P[1] * xp[1];
        const double mu1 = first ? 0.0 : P[2] * xp[0] + P[3] * xp[1];
        const int o = first ? 14 : 8;
        const double m0 = mu0 + P[o] * (obs[0] - mu0), m1 = mu1 + P[o] * (obs[1] - mu1);
        const double a0 = (x[0] - mu0) * P[o + 4], a1 = (x[1] - mu1) * P[o + 4];
        const double lt = -0.5 * (a0 * a0 + a1 * a1) - P[o + 5];
        const double b0 = (x[0] - m0) * P[o + 2], b1 = (x[1] - m1) * P[o + 2];
        const double lq = -0.5 * (b0 * b0 + b1 * b1) - P[o + 3];
        return (lt + loglik(P, x, obs)) - lq;
    }
    // move_reweight(trace, proposal, proposal_args) (src/rejuvenate.jl:134-148) with the locally optimal proposal of x_t:
    // fwd_choices ~ q(. | x_{t-1}, y_t); update -> weight = model score(new) - model score(old); rel_weight = weight - fwd_score +
    // bwd_score = W(x') - W(x) with W = proposal_weight.  (q is the exact conditional here, so the relative weight is 0 up to
    // rounding: a Gibbs move on x_t.)
    static constexpr bool HAS_MOVE_PROPOSAL = true;
    static GPF_HD double move_propose(const double* P, const double*, bool first, const double* xp, const double* x, const double* obs,
                                      uint64_t seed, uint32_t gid, uint32_t blk0, uint32_t epoch, uint32_t tag, double* xn)
    {
        const double wn = propose(P, first, xp, obs, seed, gid, blk0, epoch, tag, xn);
        return wn - proposal_weight(P, first, xp, obs, x);
    }
    static GPF_HD void sample(const double* P, bool first, const double* xp, const double*, uint64_t seed,
                              uint32_t gid, uint32_t blk0, uint32_t epoch, uint32_t tag, double* xn)
    {
        double z0, z1;
        normal2(rng(seed, gid, blk0, epoch, tag), z0, z1);
        if (first) { xn[0] = P[5] * z0; xn[1] = P[5] * z1; }
        else {
            const double t0 = P[0] * xp[0] + P[1] * xp[1];
            const double t1 = P[2] * xp[0] + P[3] * xp[1];
            xn[0] = t0 + P[4] * z0;
            xn[1] = t1 + P[4] * z1;
        }
    }
    static GPF_HD double loglik(const double* P, const double* x, const double* obs)
    {
        const double z0 = (obs[0] - x[0]) * P[6], z1 = (obs[1] - x[1]) * P[6];
        return -0.5 * (z0 * z0 + z1 * z1) - P[7];
    }
};

// bearings-only tracking, x = (px, py, vx, vy)   (BASELINE config 4)
// P = [mu0..3 | s0..3 | sp | sv | 1/sb | log sb + log(2 pi)/2]
template <> struct Model<MODEL_BEARINGS4> {
    static constexpr bool HAS_STRATA = false;
    static constexpr bool HAS_STRATA_PROPOSAL = false;
    static constexpr int D = 4, NBLK = 2;
    static constexpr bool HAS_PROPOSAL = false;
    static constexpr bool HAS_MOVE_PROPOSAL = false;
    static GPF_HD void sample(const double* P, bool first, const double* xp, const double*, uint64_t seed,
                              uint32_t gid, uint32_t blk0, uint32_t epoch, uint32_t tag, double* xn)
    {
        double z0, z1, z2, z3;
        normal2(rng(seed, gid, blk0, epoch, tag), z0, z1);
        normal2(rng(seed, gid, blk0 + 1, epoch, tag), z2, z3);
        if (first) {
            xn[0] = P[0] + P[4] * z0; xn[1] = P[1] + P[5] * z1;
            xn[2] = P[2] + P[6] * z2; xn[3] = P[3] + P[7] * z3;
        } else {
            xn[0] = (xp[0] + xp[2]) + P[8] * z0;
            xn[1] = (xp[1] + xp[3]) + P[8] * z1;
            xn[2] = xp[2] + P[9] * z2;
            xn[3] = xp[3] + P[9] * z3;
        }
    }
    static GPF_HD double loglik(const double* P, const double* x, const double* obs)
    {
        constexpr double PI = 3.14159265358979311600e+00, TWOPI = 6.28318530717958623200e+00;
        const double b = atan2_(x[1], x[0]);
        double r = obs[0] - b;
        if (r > PI) r -= TWOPI; else if (r <= -PI) r += TWOPI;
        const double z = r * P[10];
        return -0.5 * (z * z) - P[11];
    }
};

// stochastic volatility, x = h   (BASELINE config 5)
// P = [mu | phi | sigma | sigma/sqrt(1-phi^2) | log(2 pi)/2]
template <> struct Model<MODEL_SV1> {
    static constexpr bool HAS_STRATA = false;
    static constexpr bool HAS_STRATA_PROPOSAL = false;
    static constexpr int D = 1, NBLK = 1;
    static constexpr bool HAS_PROPOSAL = false;
    static constexpr bool HAS_MOVE_PROPOSAL = false;
    static GPF_HD void sample(const double* P, bool first, const double* xp, const double*, uint64_t seed,
                              uint32_t gid, uint32_t blk0, uint32_t epoch, uint32_t tag, double* xn)
    {
        double z0, z1;
        normal2(rng(seed, gid, blk0, epoch, tag), z0, z1);
        if (first) xn[0] = P[0] + P[3] * z0;
        else       xn[0] = (P[0] + P[1] * (xp[0] - P[0])) + P[2] * z0;
    }
    static GPF_HD double loglik(const double* P, const double* x, const double* obs)
    {
        const double y = obs[0];
        return (-0.5 * ((y * y) * exp_(-x[0])) - 0.5 * x[0]) - P[4];
    }
};

// README object_motion (reference README.md:43-55; BASELINE config 1), x = (moving, y)
// P = [p(moving|moving) | p(moving|still) | sigma_y | 1/sigma_obs | log sigma_obs + log(2 pi)/2 |
//      log p(moving|moving) | log(1 - p(moving|moving)) | log p(moving|still) | log(1 - p(moving|still))]
// obs = [y_obs, sin(t)]
template <> struct Model<MODEL_OBJECT_MOTION> {
    static constexpr int D = 2, NBLK = 2;
    static constexpr bool HAS_PROPOSAL = false;
    static constexpr bool HAS_MOVE_PROPOSAL = false;
    static constexpr bool HAS_STRATA = true;
    static constexpr bool HAS_STRATA_PROPOSAL = false;
    // stratified generate / update: `moving` is constrained to the stratum's value (merge(stratum, observations),
    // initialize.jl:102, update.jl:200); y is sampled as usual.  Returns log p(moving = value | moving_{t-1}), the part of
    // the weight increment Gen adds for the constrained latent choice.
    static GPF_HD double sample_stratum(const double* P, bool first, const double* xp, const double* obs, double value,
                                        uint64_t seed, uint32_t gid, uint32_t blk0, uint32_t epoch, uint32_t tag, double* xn)
    {
        double z0, z1;
        normal2(rng(seed, gid, blk0 + 1, epoch, tag), z0, z1);
        const double pm = first ? 0.0 : xp[0], py = first ? 0.0 : xp[1];
        const double mv = (value != 0.0) ? 1.0 : 0.0;
        const double lp = (pm != 0.0) ? ((mv != 0.0) ? P[5] : P[6]) : ((mv != 0.0) ? P[7] : P[8]);
        const double vel = (mv != 0.0) ? obs[1] : 0.0;
        xn[0] = mv;
        xn[1] = (py + vel) + P[2] * z0;
        return lp;
    }
    static GPF_HD void sample(const double* P, bool first, const double* xp, const double* obs, uint64_t seed,
                              uint32_t gid, uint32_t blk0, uint32_t epoch, uint32_t tag, double* xn)
    {
        const Philox b = rng(seed, gid, blk0, epoch, tag);
        const double u = u52(b.w0, b.w1);
        double z0, z1;
        normal2(rng(seed, gid, blk0 + 1, epoch, tag), z0, z1);
        const double pm = first ? 0.0 : xp[0], py = first ? 0.0 : xp[1];
        const double p = (pm != 0.0) ? P[0] : P[1];
        const double mv = (u < p) ? 1.0 : 0.0;
        const double vel = (mv != 0.0) ? obs[1] : 0.0;
        xn[0] = mv;
        xn[1] = (py + vel) + P[2] * z0;
    }
    static GPF_HD double loglik(const double* P, const double* x, const double* obs)
    {
        const double z = (obs[0] - x[1]) * P[3];
        return -0.5 * (z * z) - P[4];
    }
};

// line_model, the fixture model of the reference's own tests (reference test/runtests.jl:3-16):
//     slope ~ uniform_discrete(-2, 2);   step t:  x = t;  outlier ~ bernoulli(0.1);  y ~ normal(x * slope, outlier ? 10 : 1)
// Particle row = (slope, outlier of the current step).  obs = [y_t, x_t]; x_t = 0 stands for model args (0,): no step has
// happened yet (pf_initialize(line_model, (0,), choicemap(), n), test/initialize.jl:4), nothing is observed, the weight is 0.
// P = [p_out | 1/s_in | 1/s_out | log s_in + log(2 pi)/2 | log s_out + log(2 pi)/2 | log p_out | log(1 - p_out) |
//      -log(n_slopes) | lowest slope | n_slopes]
// One Philox block per particle and step: words (0,1) the slope (first step only), words (2,3) the outlier.
template <> struct Model<MODEL_LINE> {
    static constexpr int D = 2, NBLK = 1;
    static constexpr bool HAS_PROPOSAL = true;
    static constexpr bool HAS_STRATA = true;
    static GPF_HD void sample(const double* P, bool first, const double* xp, const double* obs, uint64_t seed,
                              uint32_t gid, uint32_t blk0, uint32_t epoch, uint32_t tag, double* xn)
    {
        const Philox b = rng(seed, gid, blk0, epoch, tag);
        xn[0] = first ? P[8] + (double)mulhi64(u64(b.w0, b.w1), (uint64_t)P[9]) : xp[0];      // uniform_discrete(lo, lo + n - 1)
        xn[1] = (obs[1] != 0.0 && u52(b.w2, b.w3) < P[0]) ? 1.0 : 0.0;                         // bernoulli(p_out)
    }
    static GPF_HD double loglik(const double* P, const double* x, const double* obs)
    {
        if (obs[1] == 0.0) return 0.0;                                                         // model args (0,): no observation
        const bool out = x[1] != 0.0;
        const double z = (obs[0] - obs[1] * x[0]) * (out ? P[2] : P[1]);
        return -0.5 * (z * z) - (out ? P[4] : P[3]);
    }
    // The custom proposals of the reference's tests as ONE native proposal (test/initialize.jl:16-19, test/update.jl:42-43):
    // slope ~ uniform_discrete(0, 0) at the first step, outlier ~ bernoulli(0.0) at every step.  Both are deterministic
    // (proposal score 0), so log_weight = model score of the proposed choices + log p(y | x)   (initialize.jl:58, translate.jl:103)
    static GPF_HD double propose(const double* P, bool first, const double* xp, const double* obs, uint64_t, uint32_t,
                                 uint32_t, uint32_t, uint32_t, double* xn)
    {
        xn[0] = first ? 0.0 : xp[0];
        xn[1] = 0.0;
        double w = first ? P[7] : 0.0;                                                         // log p(slope = 0) = log(1/5), test/initialize.jl:21
        if (obs[1] != 0.0) w = (w + P[6]) + loglik(P, xn, obs);                                // log p(outlier = false) + log p(y | .)
        return w;
    }
    // move_reweight(trace, outlier_propose, (idx,)) with outlier_propose = {:line => idx => :outlier} ~ bernoulli(q) for the current
    // step (src/rejuvenate.jl:134-148; the reference's test uses q = 0.9, test/rejuvenate.jl:19-27):
    //   weight = [log p(out') + log p(y | slope, out')] - [log p(out) + log p(y | slope, out)]   (update with the proposed choice)
    //   rel_weight = weight - log q(out') + log q(out)                                            (:146)
    // Q = {q, log q, log(1 - q)}
    static constexpr bool HAS_MOVE_PROPOSAL = true;
    static GPF_HD double move_propose(const double* P, const double* Q, bool, const double*, const double* x, const double* obs,
                                      uint64_t seed, uint32_t gid, uint32_t blk0, uint32_t epoch, uint32_t tag, double* xn)
    {
        const Philox b = rng(seed, gid, blk0, epoch, tag);
        xn[0] = x[0];
        xn[1] = (obs[1] != 0.0 && u52(b.w2, b.w3) < Q[0]) ? 1.0 : 0.0;
        if (obs[1] == 0.0) return 0.0;                                                         // model args (0,): no outlier choice to move
        const bool on = xn[1] != 0.0, oo = x[1] != 0.0;
        const double wn = (on ? P[5] : P[6]) + loglik(P, xn, obs), wo = (oo ? P[5] : P[6]) + loglik(P, x, obs);
        return ((wn - wo) - (on ? Q[1] : Q[2])) + (oo ? Q[1] : Q[2]);
    }
    // pf_initialize(model, args, obs, strata, proposal, proposal_args, n) (src/initialize.jl:111-129) as the reference's test uses it
    // (test/initialize.jl:66-90): strata over `slope`, outlier_propose = bernoulli(0.0) for the step's outlier:
    //   model_weight - prop_weight = [log p(slope) + log p(outlier = false) + log p(y | .)] - 0      (the caller adds log n_strata)
    static constexpr bool HAS_STRATA_PROPOSAL = true;
    static GPF_HD double propose_stratum(const double* P, const double* obs, double value, double* xn)
    {
        xn[0] = value;
        xn[1] = 0.0;
        double w = P[7];                                                                       // log p(slope = value) = log(1/5)
        if (obs[1] != 0.0) w = (w + P[6]) + loglik(P, xn, obs);
        return w;
    }
    // stratified initialise (strata over `slope`, test/initialize.jl:39-64) / update (strata over the step's `outlier`,
    // test/update.jl:13-40): the stratified choice is constrained, the other one is sampled as usual; returns the log
    // probability of the constrained choice
    static GPF_HD double sample_stratum(const double* P, bool first, const double* xp, const double* obs, double value,
                                        uint64_t seed, uint32_t gid, uint32_t blk0, uint32_t epoch, uint32_t tag, double* xn)
    {
        const Philox b = rng(seed, gid, blk0, epoch, tag);
        if (first) {
            xn[0] = value;
            xn[1] = (obs[1] != 0.0 && u52(b.w2, b.w3) < P[0]) ? 1.0 : 0.0;
            return P[7];
        }
        xn[0] = xp[0];
        xn[1] = value != 0.0 ? 1.0 : 0.0;
        return value != 0.0 ? P[5] : P[6];
    }
};

// length of the per-step data vector (observations, plus covariates such as sin(t) for object_motion)
inline int model_obs_dim(int m)
{
    switch (m) {
        case MODEL_LGSSM2: return 2; case MODEL_BEARINGS4: return 1;
        case MODEL_SV1: return 1; case MODEL_OBJECT_MOTION: return 2; case MODEL_LINE: return 2;
    }
    return 0;
}

inline int model_dim(int m)
{
    switch (m) {
        case MODEL_LGSSM2: return 2; case MODEL_BEARINGS4: return 4;
        case MODEL_SV1: return 1; case MODEL_OBJECT_MOTION: return 2; case MODEL_LINE: return 2;
    }
    return 0;
}

} // namespace gpf
