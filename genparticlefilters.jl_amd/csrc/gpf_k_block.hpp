// K11: block-wise resampling -- many small filters in one launch  (part of gpf_kernels.hpp; include that header, not this file)
#pragma once

namespace gpf {
// ----------------------------------------------------------------------------- K11: one workgroup = one block of particles
// The reference runs many small filters in one state as sub-states: `for b in blocks; pf_resample!(state[b], method); end`
// (src/view.jl:16-48, the block-wise resampling of test/resample.jl:130-162; README.md:60-79 and every reference test run
// N = 100).  Through the single-filter path that is one view and three to five launches per block, ~18 us each time however small the
// block.  Here ONE launch does the loop: workgroup b owns the particles [b nb, min((b + 1) nb, n)), nb <= 2048, and runs the
// whole resample of that sub-state out of LDS -- maximum and validity flags (safe_softmax, utils.jl:117-140), K-bit fixed-point
// weights (K from the BLOCK's particle count, like a view), their exact CDF, optionally the stable descending order
// (sort_particles, resample.jl:156-157: bitonic network over (key, index) pairs), the ancestors of the block's slots by
// binary search (multinomial :59, residual :96-115, stratified :159-168), the row gather (:60) and the sub-state weight update
// (every particle gets logsumexp(block weights) - log(block size), :205-211).  RNG counters are the slots' global ids and the
// call's epoch, so block b's result is bit-identical to pf_resample!(state[b]) through a view, and to the oracle's sub-state
// resample.  ess_frac >= 0: a block resamples only if its effective sample size is below ess_frac x (block size) -- the
// `if effective_sample_size(state) < N / 2` of the README loop, decided per block on the device (no host round trip).
// Teams: a block of up to 512 particles is the work of ONE WAVE (TEAM = 64, 2 or 8 consecutive particles per lane, four blocks per
// workgroup, no workgroup barrier anywhere: LDS operations of one wave execute in order), a larger one of the whole workgroup
// (TEAM = 256, 8 per lane).  10^4 blocks of 100 particles: 105 us with a workgroup per block, profiles/r03_small_filters.txt for the rest.
constexpr int BLK_MAX = 2048;                      // particles per block
struct BlockArgs {
    const double* rows_in; double* rows_out;       // [n][W]
    double* lw;                                    // [n] log-weights, rewritten for the blocks that resample
    int32_t* anc;                                  // [n] parents, LOCAL to the block (0-based), rewritten for the blocks that resample
    int64_t n, nb;                                 // particles, particles per block
    int64_t nblocks;
    int64_t gid0;                                  // global id of particle 0 (RNG counters)
    uint64_t seed; uint32_t epoch;
    int sorted;                                    // stratified: sort_particles
    double alpha;                                  // PRIO kernels: priority_fn = w -> alpha w (resample.jl:51-52)
    double ess_frac;                               // < 0: every block resamples
    int check_true;                                // check = true: blocks with invalid (all -Inf) weights are left alone as well
    int32_t* resampled;                            // [n_blocks] bit 0: the block resampled; bits 8..: its validity flags (no shared counter:
                                                   // 10^4 same-address atomics would cost ~100 us; k_block_summary folds the words on request)
};
// OR of the blocks' flags and the number of blocks that resampled -> out2 = {flags, count}
static __global__ __launch_bounds__(BLOCK) void k_block_summary(const int32_t* __restrict__ words, int64_t nblocks, int32_t* __restrict__ out2)
{
    int f = 0; unsigned c = 0;
    for (int64_t i = threadIdx.x; i < nblocks; i += BLOCK) { const int w = words[i]; f |= w >> 8; c += (unsigned)(w & 1); }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { f |= __shfl_xor(f, s, WAVE); c += (unsigned)__shfl_xor((int)c, s, WAVE); }
    __shared__ int s_f[NWAVES]; __shared__ unsigned s_c[NWAVES];
    if (lane_id() == 0) { s_f[wave_id()] = f; s_c[wave_id()] = c; }
    __syncthreads();
    if (threadIdx.x == 0) { for (int w = 1; w < NWAVES; ++w) { f |= s_f[w]; c += s_c[w]; } out2[0] = f; out2[1] = (int32_t)c; }
}

template <int TEAM>
__device__ __forceinline__ void team_sync()
{
    if (TEAM == BLOCK) __syncthreads();
    else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
}
// sums of four words over the team; s_x: [NWAVES][4] words of LDS (TEAM = BLOCK only)
template <int TEAM>
__device__ __forceinline__ void team_sum4(uint64_t (&v)[4], uint64_t (*s_x)[4])
{
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = wave_sum_u64(v[c]);
    if (TEAM == BLOCK) {
        __syncthreads();
        if (lane_id() == 0) { for (int c = 0; c < 4; ++c) s_x[wave_id()][c] = v[c]; }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c) { uint64_t t = 0; for (int w = 0; w < NWAVES; ++w) t += s_x[w][c]; v[c] = t; }
    }
}
// team-wide inclusive scan of ITEMS consecutive values per lane; returns the total.  s_x: LDS scratch as above
template <int TEAM, int ITEMS>
__device__ __forceinline__ uint64_t team_scan_incl(uint64_t (&v)[ITEMS], uint64_t (*s_x)[4])
{
#pragma unroll
    for (int k = 1; k < ITEMS; ++k) v[k] += v[k - 1];
    const uint64_t inc = wave_scan_u64(v[ITEMS - 1]);              // inclusive over the lanes' totals
    uint64_t pre = inc - v[ITEMS - 1], tot = shfl_u64(inc, WAVE - 1);
    if (TEAM == BLOCK) {
        __syncthreads();                                           // s_x may still be read from a previous call
        if (lane_id() == WAVE - 1) s_x[wave_id()][0] = inc;
        __syncthreads();
        tot = 0;
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) { const uint64_t t = s_x[w][0]; if (w < wave_id()) pre += t; tot += t; }
    }
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) v[k] += pre;
    return tot;
}
// first index a in [0, cnt) with cdf[a] > T, clamped to cnt - 1 (the while loop of resample.jl:163-166 / inverse-CDF categorical)
__device__ __forceinline__ int lds_upper_bound(const uint64_t* cdf, int cnt, uint64_t T)
{
    int lo = 0, len = cnt;
    while (len > 0) { const int half = len >> 1; if (cdf[lo + half] <= T) { lo += half + 1; len -= half + 1; } else len = half; }
    return lo < cnt ? lo : cnt - 1;
}

// maximum and validity flags of the team's values (safe_softmax, utils.jl:119-126); s_m / s_f: NWAVES words each (TEAM = BLOCK only)
template <int TEAM, int ITEMS>
__device__ __forceinline__ void team_max_flags(const double (&v)[ITEMS], int tl, int cnt, double* s_m, int* s_f, double& m_out, int& f_out)
{
    double m = -__builtin_huge_val(); int f = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k)
        if (ITEMS * tl + k < cnt) { const double x = v[k]; if (x != x) f |= FLAG_NAN; else { m = x > m ? x : m; if (x == __builtin_huge_val()) f |= FLAG_POSINF; } }
    m = wave_max_f64(m);
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) f |= __shfl_xor(f, s, WAVE);
    if (TEAM == BLOCK) {
        __syncthreads();                                           // (s_m / s_f may still be read from a previous call)
        if (lane_id() == 0) { s_m[wave_id()] = m; s_f[wave_id()] = f; }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) { m = s_m[w] > m ? s_m[w] : m; f |= s_f[w]; }
    }
    if (!(f & FLAG_NAN) && m == -__builtin_huge_val()) f |= FLAG_ALL_NEGINF;
    m_out = m; f_out = f;
}

// PRIO: priority_fn = w -> alpha w (resample.jl:51-52): ancestors from the priorities' CDF, the ESS gate on the raw weights, new weights
// log_ws + (logsumexp(block weights) - logsumexp(log_ws)) with log_ws = lw[a] - lp[a] (the sub-state form, resample.jl:213-216)
template <int METHOD, int W, int TEAM, int ITEMS, bool PRIO = false>     // METHOD 0 multinomial, 1 residual, 2 stratified
__global__ __launch_bounds__(BLOCK) void k_block_resample(BlockArgs a)
{
    constexpr int TEAMS = BLOCK / TEAM, CAP = TEAM * ITEMS;        // blocks per workgroup, particles a team holds
    static_assert(TEAM == WAVE || TEAM == BLOCK, "a wave or the workgroup");
    __shared__ uint64_t s_cdf_[BLOCK * ITEMS];                     // the CDF that is sampled (weights; residual: residual weights)
    __shared__ uint64_t s_aux_[METHOD == 0 ? 1 : BLOCK * ITEMS];   // residual: copy-count CDF; stratified: sort keys
    __shared__ uint16_t s_idx_[METHOD == 2 ? BLOCK * ITEMS : 1];   // stratified, sorted: the order
    __shared__ uint64_t s_x[NWAVES][4];
    __shared__ double s_m[NWAVES];
    __shared__ int s_f[NWAVES];
    __shared__ double s_lw_[PRIO ? BLOCK * ITEMS : 1];             // PRIO: the block's incoming log-weights, then log_ws of its slots
    __shared__ double s_lws_[PRIO ? BLOCK * ITEMS : 1];
    const int tm = (int)threadIdx.x / TEAM, tl = (int)threadIdx.x % TEAM, lane = lane_id(), wv = wave_id();
    const int64_t blk = (int64_t)blockIdx.x * TEAMS + tm;
    if (TEAM != BLOCK && blk >= a.nblocks) return;                 // (an idle wave: the wave-team path has no workgroup barrier)
    double* const s_lw = s_lw_ + (PRIO ? tm * CAP : 0);
    double* const s_lws = s_lws_ + (PRIO ? tm * CAP : 0);
    (void)lane; (void)wv;
    uint64_t* const s_cdf = s_cdf_ + tm * CAP;
    uint64_t* const s_aux = s_aux_ + (METHOD == 0 ? 0 : tm * CAP);
    uint16_t* const s_idx = s_idx_ + (METHOD == 2 ? tm * CAP : 0);
    const int64_t b0 = blk * a.nb;
    const int cnt = (int)(a.n - b0 < a.nb ? a.n - b0 : a.nb);      // particles of this block
    const int K = fix_K(cnt);
    // ---- the raw weights: maximum + flags (utils.jl:119-126), fixed-point weights, their sum (and sum of squares for the ESS gate)
    double lwv[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) { const int i = ITEMS * tl + k; lwv[k] = i < cnt ? a.lw[b0 + i] : -__builtin_huge_val(); }
    double m_r; int f_r;
    team_max_flags<TEAM, ITEMS>(lwv, tl, cnt, s_m, s_f, m_r, f_r);
    uint64_t q[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) q[k] = ITEMS * tl + k < cnt ? ((f_r & FLAG_ALL_NEGINF) ? 1ull : exp_fix(lwv[k] - m_r, K)) : 0ull;
    bool gate = true;                                              // the block passes the ESS test (or there is none)
    uint64_t S_r = 0;
    if (PRIO || a.ess_frac >= 0.0) {                               // (team-uniform)
        unsigned __int128 Q = 0; uint64_t sl = 0;
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) { Q += (unsigned __int128)q[k] * q[k]; sl += q[k]; }
        // 128-bit team sum by limbs: the low word in two 32-bit halves (their carries just add up)
        uint64_t v4[4] = {(uint64_t)Q & 0xffffffffull, (uint64_t)Q >> 32, (uint64_t)(Q >> 64), sl};
        team_sum4<TEAM>(v4, s_x);
        S_r = v4[3];
        if (a.ess_frac >= 0.0) {
            const unsigned __int128 Qt = ((unsigned __int128)v4[2] << 64) + ((unsigned __int128)v4[1] << 32) + v4[0];
            const double ess = ess_from(S_r, (uint64_t)(Qt >> 64), (uint64_t)Qt);
            // `effective_sample_size(state) < N / 2`, README.md:72; invalid weights: the reference's ESS is NaN and the comparison false
            gate = f_r == 0 && ess < a.ess_frac * (double)cnt;
        }
    }
    // ---- what the resampler samples from: the raw weights, or the priorities alpha * lw (their own maximum, flags, fixed-point weights)
    double m = m_r; int f = f_r;
    if (PRIO) {
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) { const int i = ITEMS * tl + k; if (i < cnt) s_lw[i] = lwv[k]; lwv[k] = i < cnt ? a.alpha * lwv[k] : -__builtin_huge_val(); }
        team_max_flags<TEAM, ITEMS>(lwv, tl, cnt, s_m, s_f, m, f);
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) q[k] = ITEMS * tl + k < cnt ? ((f & FLAG_ALL_NEGINF) ? 1ull : exp_fix(lwv[k] - m, K)) : 0ull;
    }
    // NaN / +Inf: Categorical rejects them in the reference; with check = true any invalid block: left as it stands
    const bool skip = (f & (FLAG_NAN | FLAG_POSINF)) != 0 || (a.check_true && f != 0);
    const bool uniform = (f & FLAG_ALL_NEGINF) != 0;
    const bool go = gate && !skip;
    // (a block that does not pass its ESS test never gets as far as safe_softmax: nothing to report for it)
    if (tl == 0) a.resampled[blk] = (go ? 1 : 0) | ((gate ? f : 0) << 8);
    if (!go) {
        // this block keeps its particles: rows move to the other buffer unchanged, weights and parents stay
        for (int t = tl; t < cnt * (W / 2); t += TEAM)
            reinterpret_cast<double2*>(a.rows_out + b0 * W)[t] = reinterpret_cast<const double2*>(a.rows_in + b0 * W)[t];
        return;
    }
    // ---- sort_particles (stratified, resample.jl:156-157): order = sortperm(log_priorities, rev = true), stable.  Bitonic network over
    //      (key, index) pairs -- the index makes every pair distinct, so the network's order IS the stable order
    if (METHOD == 2 && a.sorted) {
        int p2 = 2;                                                // the network's size: the next power of two >= cnt
        while (p2 < cnt) p2 <<= 1;
        for (int i = tl; i < p2; i += TEAM) { s_aux[i] = i < cnt ? sort_key_desc(PRIO ? a.alpha * a.lw[b0 + i] : a.lw[b0 + i]) : ~0ull; s_idx[i] = (uint16_t)i; }
        team_sync<TEAM>();
        for (int size = 2; size <= p2; size <<= 1) {
            for (int stride = size >> 1; stride >= 1; stride >>= 1) {
                for (int t = tl; t < p2 / 2; t += TEAM) {
                    const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                    const bool up = (lo & size) == 0;
                    const uint64_t ka = s_aux[lo], kb = s_aux[hi];
                    const uint16_t ia = s_idx[lo], ib = s_idx[hi];
                    const bool gt = ka > kb || (ka == kb && ia > ib);
                    if (gt == up) { s_aux[lo] = kb; s_aux[hi] = ka; s_idx[lo] = ib; s_idx[hi] = ia; }
                }
                team_sync<TEAM>();
            }
        }
        // the weights in sorted order (the key map is a bijection: no second read of lw)
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const int i = ITEMS * tl + k;
            q[k] = i < cnt ? (uniform ? 1ull : exp_fix(sort_key_value(s_aux[i]) - m, K)) : 0ull;
        }
        team_sync<TEAM>();
    }
    // ---- the CDF(s)
    uint64_t S, Ctot = 0, Rs = 0;
    if (METHOD == 1) {
        // residual (resample.jl:96-115): copies c_i = floor(N w_i) = (N q_i) div S, residual weight ((N q_i) mod S) >> sh
        uint64_t sq[ITEMS];
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) sq[k] = q[k];
        S = team_scan_incl<TEAM, ITEMS>(sq, s_x);
        const int sh = residual_shift(S, cnt);
        uint64_t c[ITEMS], r[ITEMS];
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            // (N q) div S and mod S with a quotient <= N <= 2048: the double estimate is off by at most one, two exact corrections
            // instead of a 64-bit division
            const uint64_t nq = (uint64_t)cnt * q[k];
            uint64_t cq = (uint64_t)((double)nq / (double)S), prod = cq * S;
            if (prod > nq) { cq -= 1; prod -= S; }
            uint64_t rq = nq - prod;
            if (rq >= S) { cq += 1; rq -= S; }
            c[k] = cq; r[k] = rq >> sh;
        }
        Ctot = team_scan_incl<TEAM, ITEMS>(c, s_x);
        Rs = team_scan_incl<TEAM, ITEMS>(r, s_x);
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) { s_aux[ITEMS * tl + k] = c[k]; s_cdf[ITEMS * tl + k] = r[k]; }
    } else {
        S = team_scan_incl<TEAM, ITEMS>(q, s_x);
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) s_cdf[ITEMS * tl + k] = q[k];
    }
    team_sync<TEAM>();
    // ---- ancestors, gather, sub-state weights (resample.jl:205-211: every particle carries the block's average weight)
    const double new_lw = PRIO ? 0.0 : lse_from(m, S, K, f) - log_((double)cnt);
    const uint64_t sB = METHOD == 2 ? S / (uint64_t)cnt : 0, srem = METHOD == 2 ? S % (uint64_t)cnt : 0;
    for (int j = tl; j < cnt; j += TEAM) {                         // consecutive lanes, consecutive slots: coalesced stores
        const uint32_t slot = (uint32_t)(a.gid0 + b0 + j);
        int anc;
        if (METHOD == 0) anc = lds_upper_bound(s_cdf, cnt, mulhi64(resample_u64(a.seed, slot, a.epoch), S));          // :59
        else if (METHOD == 1) {
            if ((uint64_t)j < Ctot) anc = lds_upper_bound(s_aux, cnt, (uint64_t)j);                                 // :101 (no draw)
            else anc = lds_upper_bound(s_cdf, cnt, mulhi64(resample_u64(a.seed, slot, a.epoch), Rs));               // :113
        } else {
            const uint64_t jl = (uint64_t)j;                                                               // strata are local to the block
            // (j rem < 2048^2: 32-bit divisions)
            const uint64_t L0 = jl * sB + (uint32_t)(jl * srem) / (uint32_t)cnt, L1 = (jl + 1) * sB + (uint32_t)((jl + 1) * srem) / (uint32_t)cnt;
            anc = lds_upper_bound(s_cdf, cnt, L0 + mulhi64(resample_u64(a.seed, slot, a.epoch), L1 - L0));        // :162-166
            if (a.sorted) anc = (int)s_idx[anc];                                                           // :168
        }
        const double2* src = reinterpret_cast<const double2*>(a.rows_in + (b0 + anc) * W);
        double2* dst = reinterpret_cast<double2*>(a.rows_out + (b0 + j) * W);
#pragma unroll
        for (int c = 0; c < W / 2; ++c) dst[c] = src[c];
        a.anc[b0 + j] = anc;
        if (PRIO) { const double w0 = s_lw[anc]; s_lws[j] = w0 - a.alpha * w0; }               // log_ws = lw[a] - lp[a]  (:213)
        else a.lw[b0 + j] = new_lw;
    }
    if (PRIO) {
        // lw = log_ws + (logsumexp(block's incoming weights) - logsumexp(log_ws))   (resample.jl:213-216)
        team_sync<TEAM>();
        double wv2[ITEMS];
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) { const int i = ITEMS * tl + k; wv2[k] = i < cnt ? s_lws[i] : -__builtin_huge_val(); }
        double m2; int f2;
        team_max_flags<TEAM, ITEMS>(wv2, tl, cnt, s_m, s_f, m2, f2);
        uint64_t v4[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) v4[3] += ITEMS * tl + k < cnt ? ((f2 & FLAG_ALL_NEGINF) ? 1ull : exp_fix(wv2[k] - m2, K)) : 0ull;
        team_sum4<TEAM>(v4, s_x);
        const double off = lse_from(m_r, S_r, K, f_r) - lse_from(m2, v4[3], K, f2);
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) { const int i = ITEMS * tl + k; if (i < cnt) a.lw[b0 + i] = wv2[k] + off; }
    }
}

// per-block effective sample size and log-ML estimate of a sub-state (utils.jl:163-178): ess[b], lml[b] = (lml_est + logsumexp(block)) - log(block size);
// teams as in k_block_resample
template <int TEAM, int ITEMS>
__global__ __launch_bounds__(BLOCK) void k_block_stats(const double* __restrict__ lw, int64_t n, int64_t nb, int64_t nblocks, const double* lml_est,
                                                       double* __restrict__ ess_out, double* __restrict__ lml_out)
{
    constexpr int TEAMS = BLOCK / TEAM;
    __shared__ double s_m[NWAVES];
    __shared__ int s_f[NWAVES];
    __shared__ uint64_t s_x[NWAVES][4];
    const int tm = (int)threadIdx.x / TEAM, tl = (int)threadIdx.x % TEAM, lane = lane_id(), wv = wave_id();
    const int64_t blk = (int64_t)blockIdx.x * TEAMS + tm;
    if (TEAM != BLOCK && blk >= nblocks) return;
    const int64_t b0 = blk * nb;
    const int cnt = (int)(n - b0 < nb ? n - b0 : nb);
    const int K = fix_K(cnt);
    double lwv[ITEMS];
    double m = -__builtin_huge_val(); int f = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const int i = ITEMS * tl + k;
        lwv[k] = i < cnt ? lw[b0 + i] : -__builtin_huge_val();
        if (i < cnt) { const double v = lwv[k]; if (v != v) f |= FLAG_NAN; else { m = v > m ? v : m; if (v == __builtin_huge_val()) f |= FLAG_POSINF; } }
    }
    m = wave_max_f64(m);
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) f |= __shfl_xor(f, s, WAVE);
    if (TEAM == BLOCK) {
        if (lane == 0) { s_m[wv] = m; s_f[wv] = f; }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) { m = s_m[w] > m ? s_m[w] : m; f |= s_f[w]; }
    }
    if (!(f & FLAG_NAN) && m == -__builtin_huge_val()) f |= FLAG_ALL_NEGINF;
    const bool uniform = (f & FLAG_ALL_NEGINF) != 0;
    unsigned __int128 Q = 0; uint64_t sl = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const uint64_t q = ITEMS * tl + k < cnt ? (uniform ? 1ull : exp_fix(lwv[k] - m, K)) : 0ull;
        Q += (unsigned __int128)q * q; sl += q;
    }
    uint64_t v4[4] = {(uint64_t)Q & 0xffffffffull, (uint64_t)Q >> 32, (uint64_t)(Q >> 64), sl};
    team_sum4<TEAM>(v4, s_x);
    if (tl == 0) {
        const unsigned __int128 Qt = ((unsigned __int128)v4[2] << 64) + ((unsigned __int128)v4[1] << 32) + v4[0];
        ess_out[blk] = f ? __builtin_nan("") : ess_from(v4[3], (uint64_t)(Qt >> 64), (uint64_t)Qt);     // (lognorm of invalid weights is NaN)
        lml_out[blk] = (*lml_est + lse_from(m, v4[3], K, f)) - log_((double)cnt);                       // utils.jl:174-178, left to right
    }
}

// the blocks' observation vectors from a pinned host buffer into device memory, by a KERNEL (coalesced reads over PCIe) rather than a
// hipMemcpyAsync: the copy stays on the compute queue (an SDMA copy costs a cross-queue dependency of ~10-20 us in front of the step
// kernel that reads it).  The last workgroup publishes `ticket` to pinned memory: the host may then refill that staging buffer.
static __global__ __launch_bounds__(BLOCK) void k_stage_obs(const double* __restrict__ src_host, double* __restrict__ dst, int64_t n_words,
                                                     unsigned int* counter, int64_t* host_done, int64_t ticket)
{
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n_words; i += (int64_t)gridDim.x * BLOCK) dst[i] = src_host[i];
    __syncthreads();
    // (the ticket says "the staging buffer has been READ": every load of this workgroup has returned -- its value went into a store that has
    //  been issued -- before the barrier; nothing the host reads is published, so no fence: an agent / system release is an L2 write-back)
    if (threadIdx.x == 0) {
        if (__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
            __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(host_done, ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

} // namespace gpf
