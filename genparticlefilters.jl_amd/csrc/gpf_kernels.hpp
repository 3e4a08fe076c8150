// gpf_kernels.hpp -- gfx950 kernels of the particle-filter hot path.
//
// Layout in HBM (per handle / shard of N particles):
//   rows[2]  N x W Float64 "particle rows" (W = d or 2d, even), ping-pong    (Gen traces, flattened)
//   lw       N Float64 log-weights                                           (state.log_weights)
//   cdf[3]   N u64 inclusive fixed-point prefix sums (weights / residual counts / residual weights)
//   anc      N i32 ancestor of every output slot                              (state.parents)
// Rows instead of one array per column: the resample gather is random-access BY PARTICLE, and a
// 16..64-byte row is fetched with one or a few 16-byte lane loads from one cache line, where a
// column layout touches W different lines per particle.
//
// All kernels are wave64 / 256-thread workgroups; none uses MFMA (nothing here is a contraction).
#pragma once
#include "gpf_models.hpp"

namespace gpf {

constexpr int BLOCK = 256;
constexpr int WAVE = 64;
constexpr int NWAVES = BLOCK / WAVE;
constexpr int SCAN_ITEMS = 8;
constexpr int TILE = BLOCK * SCAN_ITEMS;          // 2048 weights per scan tile
constexpr int MAX_PARTIALS = 1024;                // partial (max, flags) slots of the reduce kernels
constexpr int LDS_TILE_TABLE = 8192;              // tile-prefix entries kept in LDS by the search kernel (64 KiB)

// ----------------------------------------------------------------------------- device scalars
struct WSum {                  // summary of one weight vector
    double   m;                // maximum
    int32_t  flags;            // FLAG_NAN | FLAG_POSINF | FLAG_ALL_NEGINF  (safe_softmax, utils.jl:119-137)
    int32_t  pad;
    uint64_t S;                // sum of fixed-point weights
    uint64_t Ql[4];            // 32-bit limbs sums of sum q^2 (un-normalised)
};
struct Scalars {
    WSum     prio;             // weights the resampler samples from (log_priorities)
    WSum     raw;              // state.log_weights (log-ML estimate, ESS)
    WSum     post;             // log_ws after a prioritised resample (update_weights!, resample.jl:198-200)
    double   lml_est;          // state.log_ml_est
    uint64_t Ctot;             // residual: number of deterministic copies (n_resampled)
    uint64_t Rs;               // residual: sum of residual weights
    uint64_t n_accept;         // accepted MH moves of the last gpf_rejuvenate
    int32_t  sh;               // residual shift
    int32_t  pad;
};

// how the resampler sees the weights: log_priorities = priority_fn.(log_weights) (resample.jl:51-52)
struct PrioView {
    const double* lw;          // state.log_weights
    const double* lp;          // explicit priorities (mode 2) or nullptr
    double alpha;              // mode 1: lp_i = alpha * lw_i
    int mode;                  // 0 none, 1 alpha, 2 explicit
    __device__ __forceinline__ double at(int64_t i) const
    {
        return mode == 0 ? lw[i] : (mode == 1 ? alpha * lw[i] : lp[i]);
    }
};

// ----------------------------------------------------------------------------- wave helpers
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (WAVE - 1)); }
__device__ __forceinline__ int wave_id() { return (int)(threadIdx.x >> 6); }

__device__ __forceinline__ uint64_t shfl_up_u64(uint64_t v, int d)
{
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_up(lo, d, WAVE); hi = __shfl_up(hi, d, WAVE);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m)
{
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_xor(lo, m, WAVE); hi = __shfl_xor(hi, m, WAVE);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src)
{
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl(lo, src, WAVE); hi = __shfl(hi, src, WAVE);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_u64(v, m);
    return v;
}
__device__ __forceinline__ double wave_max_f64(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const double o = u2d(shfl_xor_u64(d2u(v), m));
        v = o > v ? o : v;
    }
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += u2d(shfl_xor_u64(d2u(v), m));
    return v;
}
// inclusive scan across the 64 lanes
__device__ __forceinline__ uint64_t wave_scan_u64(uint64_t v)
{
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const uint64_t o = shfl_up_u64(v, d);
        if (l >= d) v += o;
    }
    return v;
}

// ----------------------------------------------------------------------------- K1/K2: init & step
// pf_initialize (initialize.jl:39-41) / pf_update! (update.jl:15-22): one lane per particle, row in,
// row out, lw += log p(y|x).  Counter-based RNG: no RNG state in memory.
template <int M>
__global__ __launch_bounds__(BLOCK) void k_init(ModelArgs a, uint64_t seed, uint32_t epoch, int64_t gid0,
                                                int64_t n, int W, double* __restrict__ rows,
                                                double* __restrict__ lw)
{
    using Mo = Model<M>;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        double x[MAX_DIM];
        Mo::sample(a.P, true, nullptr, a.obs, seed, (uint32_t)(gid0 + i), 0, epoch, TAG_INIT, x);
        double* r = rows + i * W;
#pragma unroll
        for (int k = 0; k < Mo::D; ++k) r[k] = x[k];
        for (int k = Mo::D; k < W; ++k) r[k] = 0.0;
        lw[i] = Mo::loglik(a.P, x, a.obs);
    }
}

template <int M, int W, bool KEEP_PREV>
__global__ __launch_bounds__(BLOCK) void k_step(ModelArgs a, uint64_t seed, uint32_t epoch, int64_t gid0,
                                                int64_t n, const double* __restrict__ rows_in,
                                                double* __restrict__ rows_out, double* __restrict__ lw)
{
    using Mo = Model<M>;
    constexpr int D = Mo::D;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        double r[W];
        const double2* src = reinterpret_cast<const double2*>(rows_in + i * W);
#pragma unroll
        for (int c = 0; c < (D + 1) / 2; ++c) { const double2 v = src[c]; r[2 * c] = v.x; r[2 * c + 1] = v.y; }
        double xn[MAX_DIM];
        Mo::sample(a.P, false, r, a.obs, seed, (uint32_t)(gid0 + i), 0, epoch, TAG_UPDATE, xn);
        const double ll = Mo::loglik(a.P, xn, a.obs);
        double o[W];
#pragma unroll
        for (int k = 0; k < W; ++k) o[k] = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) o[k] = xn[k];
        if (KEEP_PREV) {
#pragma unroll
            for (int k = 0; k < D; ++k) o[D + k] = r[k];
        }
        double2* dst = reinterpret_cast<double2*>(rows_out + i * W);
#pragma unroll
        for (int c = 0; c < W / 2; ++c) dst[c] = make_double2(o[2 * c], o[2 * c + 1]);
        lw[i] = lw[i] + ll;
    }
}

// K7/K8: pf_move_accept! with Gen.mh on the current step's latent (rejuvenate.jl:40-53) and
// pf_move_reweight! with move_reweight(trace, selection) (rejuvenate.jl:74-90, :125-132)
template <int M, int W, bool REWEIGHT>
__global__ __launch_bounds__(BLOCK) void k_move(ModelArgs a, uint64_t seed, uint32_t epoch, int64_t gid0,
                                                int64_t n, int has_prev, int n_iters,
                                                const double* __restrict__ rows_in,
                                                double* __restrict__ rows_out, double* __restrict__ lw,
                                                unsigned long long* __restrict__ n_accept)
{
    using Mo = Model<M>;
    constexpr int D = Mo::D, NB = Mo::NBLK;
    unsigned long long acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        double r[W];
        const double2* src = reinterpret_cast<const double2*>(rows_in + i * W);
#pragma unroll
        for (int c = 0; c < W / 2; ++c) { const double2 v = src[c]; r[2 * c] = v.x; r[2 * c + 1] = v.y; }
        double x[MAX_DIM], xs[MAX_DIM];
#pragma unroll
        for (int k = 0; k < D; ++k) x[k] = r[k];
        const double* xp = r + D;                    // x_{t-1} (valid when has_prev)
        double llx = Mo::loglik(a.P, x, a.obs);
        double wsum = 0.0;
        const uint32_t gid = (uint32_t)(gid0 + i);
        for (int it = 0; it < n_iters; ++it) {
            if (REWEIGHT) {
                Mo::sample(a.P, !has_prev, xp, a.obs, seed, gid, (uint32_t)(it * NB), epoch, TAG_REWEIGHT, xs);
                const double lls = Mo::loglik(a.P, xs, a.obs);
                wsum = wsum + (lls - llx);
#pragma unroll
                for (int k = 0; k < D; ++k) x[k] = xs[k];
                llx = lls;
                ++acc;
            } else {
                const uint32_t blk0 = (uint32_t)(it * (NB + 1));
                Mo::sample(a.P, !has_prev, xp, a.obs, seed, gid, blk0, epoch, TAG_MOVE, xs);
                const double lls = Mo::loglik(a.P, xs, a.obs);
                const Philox b = rng(seed, gid, blk0 + NB, epoch, TAG_MOVE);
                const double lu = log_(u52(b.w0, b.w1));
                if (lu < lls - llx) {
#pragma unroll
                    for (int k = 0; k < D; ++k) x[k] = xs[k];
                    llx = lls;
                    ++acc;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < D; ++k) r[k] = x[k];
        double2* dst = reinterpret_cast<double2*>(rows_out + i * W);
#pragma unroll
        for (int c = 0; c < W / 2; ++c) dst[c] = make_double2(r[2 * c], r[2 * c + 1]);
        if (REWEIGHT) lw[i] = lw[i] + wsum;
    }
    // one atomic per wave
    unsigned long long t = wave_sum_u64(acc);
    if (lane_id() == 0 && t) atomicAdd(n_accept, t);
}

// ----------------------------------------------------------------------------- K3: max + flags
// maximum(vs), any(isnan), all(== -Inf) of safe_softmax (utils.jl:119-128): per-block partials; the
// consumers (k_scan, k_scalar) fold the <= MAX_PARTIALS partials themselves (no finalize launch).
__global__ __launch_bounds__(BLOCK) void k_max_partial(PrioView pv, int64_t n, double* __restrict__ pmax,
                                                       int32_t* __restrict__ pflags)
{
    double m = -__builtin_huge_val();
    int f = 0;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        const double v = pv.at(i);
        if (v != v) f |= FLAG_NAN;
        else { m = v > m ? v : m; if (v == __builtin_huge_val()) f |= FLAG_POSINF; }
    }
    m = wave_max_f64(m);
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) f |= __shfl_xor(f, s, WAVE);
    __shared__ double sm[NWAVES];
    __shared__ int sf[NWAVES];
    if (lane_id() == 0) { sm[wave_id()] = m; sf[wave_id()] = f; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < NWAVES; ++w) { m = sm[w] > m ? sm[w] : m; f |= sf[w]; }
        pmax[blockIdx.x] = m;
        pflags[blockIdx.x] = f;
    }
}

// fold the partials: every lane of the block ends with (m, flags); needs 2 LDS arrays of NWAVES
__device__ __forceinline__ void fold_partials(const double* __restrict__ pmax, const int32_t* __restrict__ pflags,
                                              int np, double* sm, int* sf, double& m_out, int& f_out)
{
    double m = -__builtin_huge_val();
    int f = 0;
    for (int i = threadIdx.x; i < np; i += BLOCK) { const double v = pmax[i]; m = v > m ? v : m; f |= pflags[i]; }
    m = wave_max_f64(m);
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) f |= __shfl_xor(f, s, WAVE);
    if (lane_id() == 0) { sm[wave_id()] = m; sf[wave_id()] = f; }
    __syncthreads();
    m = sm[0]; f = sf[0];
#pragma unroll
    for (int w = 1; w < NWAVES; ++w) { m = sm[w] > m ? sm[w] : m; f |= sf[w]; }
    if (!(f & FLAG_NAN) && m == -__builtin_huge_val()) f |= FLAG_ALL_NEGINF;
    m_out = m; f_out = f;
    __syncthreads();
}

// ----------------------------------------------------------------------------- K4: fixed-point scan
// Single-pass inclusive prefix sum with decoupled look-back over 2048-element tiles.
// A tile descriptor is ONE naturally aligned 8-byte word {2-bit status | 62-bit value}, written
// and polled with relaxed agent-scope atomics (the data IS the flag: no fence, placement-independent;
// per-XCD L2s are not coherent, so plain loads/stores would not do).
// Deadlock freedom does not rely on dispatch order: the grid is sized to be fully resident and
// block b owns tiles b, b+G, b+2G, ... so a tile only ever waits on tiles of resident blocks.
constexpr uint64_t DESC_AGG = 1ull << 62, DESC_PREFIX = 2ull << 62, DESC_MASK = (1ull << 62) - 1;

__device__ __forceinline__ void desc_store(uint64_t* p, uint64_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t desc_load(const uint64_t* p)
{
    return __hip_atomic_load(const_cast<uint64_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// input functors: q[0..8) of thread t in the tile starting at base (blocked arrangement)
struct InFixQ {                // q_i = trunc(exp(p_i - m) 2^K + 1/2); uniform fallback q_i = 1
    PrioView pv;
    const int32_t* order;      // optional permutation (sort_particles, resample.jl:156-157)
    int K;
    double m; int flags;       // filled in-kernel from the partials
    __device__ __forceinline__ void load(int64_t i0, int64_t n, uint64_t* q) const
    {
        const bool uniform = (flags & FLAG_ALL_NEGINF) != 0, bad = (flags & (FLAG_NAN | FLAG_POSINF)) != 0;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) {
            const int64_t i = i0 + k;
            uint64_t v = 0;
            if (i < n) {
                if (uniform) v = 1;
                else if (!bad) v = exp_fix(pv.at(order ? (int64_t)order[i] : i) - m, K);
            }
            q[k] = v;
        }
    }
};
struct InResidual {            // from the weight CDF: counts (N q_i) div S, or residuals ((N q_i) mod S) >> sh
    const uint64_t* cdf;
    const Scalars* sc;
    const WSum* ws;            // summary of the weights being resampled
    int64_t N;                 // global particle count
    int want_r;
    __device__ __forceinline__ void load(int64_t i0, int64_t n, uint64_t* q) const
    {
        const uint64_t S = ws->S;
        const int sh = sc->sh;
        uint64_t prev = (i0 > 0 && i0 <= n) ? cdf[i0 - 1] : 0;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) {
            const int64_t i = i0 + k;
            uint64_t v = 0;
            if (i < n && S != 0) {
                const uint64_t c = cdf[i];
                const uint64_t nq = (uint64_t)N * (c - prev);
                prev = c;
                v = want_r ? ((nq % S) >> sh) : (nq / S);
            }
            q[k] = v;
        }
    }
};

template <class In, bool WANT_Q>
__global__ __launch_bounds__(BLOCK) void k_scan(In in, int64_t n, int64_t ntiles,
                                                const double* __restrict__ pmax, const int32_t* __restrict__ pflags,
                                                int np, WSum* __restrict__ ws_out,
                                                uint64_t* __restrict__ cdf, uint64_t* __restrict__ desc,
                                                uint64_t* __restrict__ total_out, uint64_t* __restrict__ blockQ)
{
    __shared__ double sm[NWAVES];
    __shared__ int sf[NWAVES];
    __shared__ uint64_t s_wave[NWAVES];
    __shared__ uint64_t s_excl;
    if constexpr (WANT_Q) {
        double m; int f;
        fold_partials(pmax, pflags, np, sm, sf, m, f);
        in.m = m; in.flags = f;
        if (blockIdx.x == 0 && threadIdx.x == 0) { ws_out->m = m; ws_out->flags = f; }
    }
    uint64_t ql[4] = {0, 0, 0, 0};
    const int lane = lane_id(), wv = wave_id();
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t i0 = tile * TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
        uint64_t q[SCAN_ITEMS];
        in.load(i0, n, q);
        uint64_t tsum = 0;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) {
            if constexpr (WANT_Q) {
                const uint64_t lo = q[k] * q[k], hi = __umul64hi(q[k], q[k]);
                ql[0] += lo & 0xffffffffull; ql[1] += lo >> 32; ql[2] += hi & 0xffffffffull; ql[3] += hi >> 32;
            }
            tsum += q[k];
            q[k] = tsum;                               // thread-local inclusive
        }
        const uint64_t winc = wave_scan_u64(tsum);     // inclusive over lanes
        if (lane == WAVE - 1) s_wave[wv] = winc;
        __syncthreads();
        uint64_t wexcl = 0, agg = 0;
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) { if (w < wv) wexcl += s_wave[w]; agg += s_wave[w]; }
        // ---- decoupled look-back, wave 0 ----
        if (wv == 0) {
            uint64_t excl = 0;
            if (tile == 0) {
                if (lane == 0) desc_store(desc, DESC_PREFIX | agg);
            } else {
                if (lane == 0) desc_store(desc + tile, DESC_AGG | agg);
                int64_t base = tile - 1;
                while (true) {
                    const int64_t idx = base - lane;
                    uint64_t d = DESC_PREFIX;          // virtual tile -1: prefix 0
                    if (idx >= 0) {
                        d = desc_load(desc + idx);
                        while ((d >> 62) == 0) { __builtin_amdgcn_s_sleep(1); d = desc_load(desc + idx); }
                    }
                    const unsigned long long pm = __ballot((d >> 62) == 2);
                    const int first = pm ? (int)__builtin_ctzll(pm) : WAVE;   // nearest predecessor holding a prefix
                    excl += wave_sum_u64(lane <= first ? (d & DESC_MASK) : 0);
                    if (pm) break;
                    base -= WAVE;
                }
                if (lane == 0) desc_store(desc + tile, DESC_PREFIX | (excl + agg));
            }
            if (lane == 0) s_excl = excl;
        }
        __syncthreads();
        const uint64_t off = s_excl + wexcl + (winc - tsum);
        if (cdf) {
            if (i0 + SCAN_ITEMS <= n) {
                ulonglong2* dst = reinterpret_cast<ulonglong2*>(cdf + i0);
#pragma unroll
                for (int c = 0; c < SCAN_ITEMS / 2; ++c) dst[c] = make_ulonglong2(off + q[2 * c], off + q[2 * c + 1]);
            } else {
#pragma unroll
                for (int k = 0; k < SCAN_ITEMS; ++k) if (i0 + k < n) cdf[i0 + k] = off + q[k];
            }
        }
        if (tile == ntiles - 1 && threadIdx.x == BLOCK - 1) *total_out = off + tsum;
        __syncthreads();                                // s_wave / s_excl reuse
    }
    if constexpr (WANT_Q) {
        // block partial of the limb sums of sum q^2 (plain stores, folded later by k_scalar)
        __shared__ uint64_t s_q[NWAVES][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) ql[k] = wave_sum_u64(ql[k]);
        if (lane == 0) { for (int k = 0; k < 4; ++k) s_q[wv][k] = ql[k]; }
        __syncthreads();
        if (threadIdx.x < 4) {
            uint64_t t = 0;
            for (int w = 0; w < NWAVES; ++w) t += s_q[w][threadIdx.x];
            blockQ[(int64_t)blockIdx.x * 4 + threadIdx.x] = t;
        }
    }
}

// ----------------------------------------------------------------------------- scalar bookkeeping
// one workgroup; ops on the device scalar block so that no host round trip is needed per step
enum : int { OP_FOLD_Q = 1,          // ws->Ql = sum of block limb partials
             OP_LML_ACCUM = 2,       // update_lml_est!  (resample.jl:178-182): lml += logsumexp(lw) - log N
             OP_RESIDUAL_PREP = 4,   // sh = residual_shift(S, N)
             OP_ZERO_ACCEPT = 8 };
__global__ __launch_bounds__(BLOCK) void k_scalar(int ops, Scalars* sc, WSum* ws, const uint64_t* blockQ, int nblk,
                                                  int K, int64_t n_global, double logN)
{
    if (ops & OP_FOLD_Q) {
        __shared__ uint64_t s_q[NWAVES][4];
        uint64_t ql[4] = {0, 0, 0, 0};
        for (int b = threadIdx.x; b < nblk; b += BLOCK)
            for (int k = 0; k < 4; ++k) ql[k] += blockQ[(int64_t)b * 4 + k];
        for (int k = 0; k < 4; ++k) ql[k] = wave_sum_u64(ql[k]);
        if (lane_id() == 0) for (int k = 0; k < 4; ++k) s_q[wave_id()][k] = ql[k];
        __syncthreads();
        if (threadIdx.x < 4) {
            uint64_t t = 0;
            for (int w = 0; w < NWAVES; ++w) t += s_q[w][threadIdx.x];
            ws->Ql[threadIdx.x] = t;
        }
    }
    if (threadIdx.x == 0) {
        if (ops & OP_LML_ACCUM) sc->lml_est = sc->lml_est + (lse_from(sc->raw.m, sc->raw.S, K, sc->raw.flags) - logN);
        if (ops & OP_RESIDUAL_PREP) sc->sh = residual_shift(ws->S, n_global);
        if (ops & OP_ZERO_ACCEPT) sc->n_accept = 0;
    }
}

// ----------------------------------------------------------------------------- K5: ancestor search
// first index with cdf[i] > T: coarse over the per-tile inclusive prefixes (scan descriptors, in LDS),
// fine inside one 16 KiB tile of the CDF
__device__ __forceinline__ int64_t upper_bound2(const uint64_t* __restrict__ cdf, int64_t n,
                                                const uint64_t* tp, int64_t ntiles, uint64_t T)
{
    int64_t lo = 0, hi = ntiles;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((tp[mid] & DESC_MASK) > T) hi = mid; else lo = mid + 1; }
    if (lo >= ntiles) return n - 1;
    int64_t a = lo * TILE, b = a + TILE < n ? a + TILE : n;
    while (a < b) { const int64_t mid = (a + b) >> 1; if (cdf[mid] > T) b = mid; else a = mid + 1; }
    return a < n ? a : n - 1;
}

struct SearchArgs {
    const uint64_t* cdf;  const uint64_t* desc;  int64_t ntiles;      // weights (or residual weights for the tail)
    const uint64_t* ccdf; const uint64_t* cdesc;                      // residual: copy counts
    const int32_t* order;                                             // sorted stratified
    const Scalars* sc;
    const WSum* ws;                                                   // summary of the sampled weights
    int64_t n, n_global, gid0;
    uint64_t seed; uint32_t epoch;
    int32_t* anc;
};

template <int METHOD>
__global__ __launch_bounds__(BLOCK) void k_search(SearchArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* tp = reinterpret_cast<uint64_t*>(smem);
    uint64_t* ctp = tp + a.ntiles;
    const bool in_lds = (METHOD == 1 ? 2 : 1) * a.ntiles <= LDS_TILE_TABLE;
    if (in_lds) {
        for (int64_t t = threadIdx.x; t < a.ntiles; t += BLOCK) {
            tp[t] = a.desc[t];
            if (METHOD == 1) ctp[t] = a.cdesc[t];
        }
        __syncthreads();
    }
    const uint64_t* tpp = in_lds ? tp : a.desc;
    const uint64_t* ctpp = in_lds ? ctp : a.cdesc;
    const uint64_t S = (METHOD == 1) ? a.sc->Rs : a.ws->S;
    const uint64_t N = (uint64_t)a.n_global;
    for (int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x; j < a.n; j += (int64_t)gridDim.x * BLOCK) {
        const uint64_t jg = (uint64_t)(a.gid0 + j);
        const Philox b = rng(a.seed, (uint32_t)jg, 0, a.epoch, TAG_RESAMPLE);
        const uint64_t U = u64(b.w0, b.w1);
        int64_t idx;
        if (METHOD == 0) {                       // multinomial, resample.jl:59
            idx = upper_bound2(a.cdf, a.n, tpp, a.ntiles, mulhi64(U, S));
        } else if (METHOD == 2) {                // stratified, resample.jl:159-168
            const uint64_t B = S / N, rem = S % N;
            const uint64_t L0 = jg * B + (jg * rem) / N;
            const uint64_t L1 = (jg + 1) * B + ((jg + 1) * rem) / N;
            const int64_t k = upper_bound2(a.cdf, a.n, tpp, a.ntiles, L0 + mulhi64(U, L1 - L0));
            idx = a.order ? (int64_t)a.order[k] : k;
        } else {                                 // residual, resample.jl:96-115
            const uint64_t Ctot = a.sc->Ctot;
            if (jg < Ctot) idx = upper_bound2(a.ccdf, a.n, ctpp, a.ntiles, jg);
            else           idx = upper_bound2(a.cdf, a.n, tpp, a.ntiles, mulhi64(U, S));
        }
        a.anc[j] = (int32_t)idx;
    }
}

// ----------------------------------------------------------------------------- K6: gather + reweight
// new_traces .= view(traces, parents) (resample.jl:60 / :103,114 / :169) as a real row copy, fused
// with update_weights! (resample.jl:190-202): no priorities -> lw = 0; priorities -> log_ws = lw[a] - lp[a].
// One lane per 16-byte row chunk: W/2 consecutive lanes move one row.
template <int W>
__global__ __launch_bounds__(BLOCK) void k_gather(const int32_t* __restrict__ anc, const double* __restrict__ rows_in,
                                                  double* __restrict__ rows_out, PrioView pv,
                                                  double* __restrict__ lw_out, int64_t n)
{
    constexpr int C = W / 2;
    const int64_t total = n * C;
    for (int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x; t < total; t += (int64_t)gridDim.x * BLOCK) {
        const int64_t j = t / C;
        const int c = (int)(t - j * C);
        const int64_t a = anc[j];
        const double2 v = reinterpret_cast<const double2*>(rows_in)[a * C + c];
        reinterpret_cast<double2*>(rows_out)[t] = v;
        if (c == 0) lw_out[j] = pv.mode == 0 ? 0.0 : pv.lw[a] - pv.at(a);
    }
}

// lw = log_ws + (log N - logsumexp(log_ws))   (resample.jl:200)
__global__ __launch_bounds__(BLOCK) void k_apply_post(const Scalars* sc, int K, double logN, const double* __restrict__ lws,
                                                      double* __restrict__ lw, int64_t n)
{
    const double off = logN - lse_from(sc->post.m, sc->post.S, K, sc->post.flags);
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK)
        lw[i] = lws[i] + off;
}

// ----------------------------------------------------------------------------- K9: statistics
// sum_i w_i f(x_i), w_i = q_i / S (statistics.jl:13-14, 48-50); per-block partials in Float64
__global__ __launch_bounds__(BLOCK) void k_wsum(const double* __restrict__ lw, const WSum* ws, int K,
                                                const double* __restrict__ rows, int W, int col, int64_t n,
                                                int pw, const double* center, double* __restrict__ partial)
{
    const double m = ws->m;
    const double Sd = (double)ws->S;
    const bool uniform = (ws->flags & FLAG_ALL_NEGINF) != 0;
    const double c = center ? *center : 0.0;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        const uint64_t q = uniform ? 1 : exp_fix(lw[i] - m, K);
        double v = rows[i * W + col];
        if (pw == 2) { v = v - c; v = v * v; }
        acc += ((double)q / Sd) * v;
    }
    acc = wave_sum_f64(acc);
    __shared__ double s[NWAVES];
    if (lane_id() == 0) s[wave_id()] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int w = 0; w < NWAVES; ++w) t += s[w]; partial[blockIdx.x] = t; }
}
__global__ void k_sum_partials(const double* __restrict__ partial, int np, double* out)
{
    double acc = 0.0;
    for (int i = threadIdx.x; i < np; i += BLOCK) acc += partial[i];
    acc = wave_sum_f64(acc);
    __shared__ double s[NWAVES];
    if (lane_id() == 0) s[wave_id()] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int w = 0; w < NWAVES; ++w) t += s[w]; *out = t; }
}

// ----------------------------------------------------------------------------- small utilities
__global__ void k_iota(int32_t* v, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) v[i] = (int32_t)i;
}
// order-preserving key of Julia's isless on Float64 (-0.0 < 0.0); descending sort = ascending on ~key
__global__ void k_sort_keys(PrioView pv, int64_t n, uint64_t* keys)
{
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        const uint64_t u = d2u(pv.at(i));
        const uint64_t asc = (u >> 63) ? ~u : (u | 0x8000000000000000ull);
        keys[i] = ~asc;                      // ascending radix sort on ~key == descending by value, ties by index
    }
}
__global__ void k_extract_column(const double* __restrict__ rows, int W, int col, int64_t n, double* __restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) out[i] = rows[i * W + col];
}
__global__ void k_parents(const int32_t* __restrict__ anc, int64_t n, int64_t* __restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) out[i] = (int64_t)anc[i] + 1;
}
// get_log_norm_weights / get_norm_weights (utils.jl:100,103-107,148,156)
__global__ void k_norm_weights(const double* __restrict__ lw, const WSum* ws, int K, int64_t n, int want_log,
                               double* __restrict__ out)
{
    const double m = ws->m;
    const double lse = lse_from(m, ws->S, K, ws->flags);
    const double Sd = (double)ws->S;
    const bool uniform = (ws->flags & FLAG_ALL_NEGINF) != 0;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        if (want_log) out[i] = lw[i] - lse;
        else out[i] = (double)(uniform ? 1 : exp_fix(lw[i] - m, K)) / Sd;
    }
}
__global__ void k_debug_math(int which, const double* a, const double* b, int64_t n, uint64_t seed, uint32_t epoch,
                             uint32_t tag, double* out, double* out2)
{
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        switch (which) {
            case 0: out[i] = exp_(a[i]); break;
            case 1: out[i] = log_(a[i]); break;
            case 2: sincos2pi(a[i], out[i], out2[i]); break;
            case 3: out[i] = atan2_(a[i], b[i]); break;
            case 4: out[i] = sqrt_(a[i]); break;
            case 5: out[i] = a[i] / b[i]; break;
            case 6: normal2(rng(seed, (uint32_t)a[i], (uint32_t)b[i], epoch, tag), out[i], out2[i]); break;
            default: out[i] = 0.0;
        }
    }
}

// ----------------------------------------------------------------------------- shard-level kernels (multi-GPU)
// Sharding: GPU g owns the contiguous global particle range [gid0, gid0+n).  The weight
// CDF is global = local inclusive scan + the sum of the lower shards' totals; output slot j (global id)
// draws a target in GLOBAL fixed-point coordinates, the shard that owns that CDF cell looks the ancestor
// up and returns the row.  Integer arithmetic makes the ancestors independent of the number of shards.
constexpr int64_t SPACE_COUNTS = (int64_t)1 << 62;   // residual: target lives in the copy-count CDF

__global__ void k_pack_mflags(const double* __restrict__ pmax, const int32_t* __restrict__ pflags, int np, double* out2)
{
    __shared__ double sm[NWAVES];
    __shared__ int sf[NWAVES];
    double m; int f;
    fold_partials(pmax, pflags, np, sm, sf, m, f);
    if (threadIdx.x == 0) { out2[0] = m; out2[1] = (double)(f & (FLAG_NAN | FLAG_POSINF)); }
}
__global__ void k_unpack_mflags(const double* __restrict__ in2, double* pmax, int32_t* pflags)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) { pmax[0] = in2[0]; pflags[0] = (int32_t)in2[1]; }
}
__global__ void k_export_summary(const WSum* ws, const uint64_t* __restrict__ blockQ, int nblk, int64_t* out5)
{
    // {S_local, Ql0..3}: limb sums folded over the scan blocks (exact integers)
    __shared__ uint64_t s_q[NWAVES][4];
    uint64_t ql[4] = {0, 0, 0, 0};
    for (int b = threadIdx.x; b < nblk; b += BLOCK)
        for (int k = 0; k < 4; ++k) ql[k] += blockQ[(int64_t)b * 4 + k];
    for (int k = 0; k < 4; ++k) ql[k] = wave_sum_u64(ql[k]);
    if (lane_id() == 0) for (int k = 0; k < 4; ++k) s_q[wave_id()][k] = ql[k];
    __syncthreads();
    if (threadIdx.x < 4) {
        uint64_t t = 0;
        for (int w = 0; w < NWAVES; ++w) t += s_q[w][threadIdx.x];
        out5[1 + threadIdx.x] = (int64_t)t;
    }
    if (threadIdx.x == 0) out5[0] = (int64_t)ws->S;
}
// global S (and residual shift) into the device scalar block from the gathered shard totals
__global__ void k_set_global(const int64_t* __restrict__ S_all, int G, int64_t n_global, WSum* ws, Scalars* sc, int64_t* out2)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        uint64_t S = 0;
        for (int g = 0; g < G; ++g) S += (uint64_t)S_all[g];
        ws->S = S;
        sc->sh = residual_shift(S, n_global);
        (void)out2;
    }
}
__global__ void k_export_residual(const Scalars* sc, int64_t* out2)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) { out2[0] = (int64_t)sc->Ctot; out2[1] = (int64_t)sc->Rs; }
}

// targets of this shard's output slots in GLOBAL coordinates (same arithmetic as k_search)
template <int METHOD>
__global__ __launch_bounds__(BLOCK) void k_targets(uint64_t seed, uint32_t epoch, int64_t gid0, int64_t n, int64_t n_global,
                                                   const int64_t* __restrict__ totals, int G, int64_t* __restrict__ T_out)
{
    uint64_t S = 0, Ctot = 0, Rs = 0;
    for (int g = 0; g < G; ++g) {
        S += (uint64_t)totals[g];
        if (METHOD == 1) { Ctot += (uint64_t)totals[G + g]; Rs += (uint64_t)totals[2 * G + g]; }
    }
    const uint64_t N = (uint64_t)n_global;
    for (int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x; j < n; j += (int64_t)gridDim.x * BLOCK) {
        const uint64_t jg = (uint64_t)(gid0 + j);
        const Philox b = rng(seed, (uint32_t)jg, 0, epoch, TAG_RESAMPLE);
        const uint64_t U = u64(b.w0, b.w1);
        int64_t T;
        if (METHOD == 0) T = (int64_t)mulhi64(U, S);
        else if (METHOD == 2) {
            const uint64_t B = S / N, rem = S % N;
            const uint64_t L0 = jg * B + (jg * rem) / N;
            const uint64_t L1 = (jg + 1) * B + ((jg + 1) * rem) / N;
            T = (int64_t)(L0 + mulhi64(U, L1 - L0));
        } else {
            T = jg < Ctot ? ((int64_t)jg | SPACE_COUNTS) : (int64_t)mulhi64(U, Rs);
        }
        T_out[j] = T;
    }
}

// serve requests in LOCAL coordinates: ancestor lookup in this shard's CDF + row gather
template <int W>
__global__ __launch_bounds__(BLOCK) void k_serve(const int64_t* __restrict__ T_local, int64_t m_req,
                                                 const uint64_t* __restrict__ cdf, const uint64_t* __restrict__ desc,
                                                 const uint64_t* __restrict__ ccdf, const uint64_t* __restrict__ cdesc,
                                                 int64_t n, int64_t ntiles, int64_t gid0, const double* __restrict__ rows,
                                                 double* __restrict__ rows_out, int64_t* __restrict__ anc_out)
{
    constexpr int C = W / 2;
    for (int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x; r < m_req; r += (int64_t)gridDim.x * BLOCK) {
        const int64_t t = T_local[r];
        int64_t a;
        if (t & SPACE_COUNTS) a = upper_bound2(ccdf, n, cdesc, ntiles, (uint64_t)(t & ~SPACE_COUNTS));
        else                  a = upper_bound2(cdf, n, desc, ntiles, (uint64_t)t);
        anc_out[r] = gid0 + a;
        const double2* src = reinterpret_cast<const double2*>(rows) + a * C;
        double2* dst = reinterpret_cast<double2*>(rows_out) + r * C;
#pragma unroll
        for (int c = 0; c < C; ++c) dst[c] = src[c];
    }
}

__global__ void k_commit(const int64_t* __restrict__ anc_in, int64_t n, int32_t* __restrict__ anc, double* __restrict__ lw)
{
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        anc[i] = (int32_t)anc_in[i];
        lw[i] = 0.0;                                   // update_weights!, resample.jl:195
    }
}
// update_lml_est! from the gathered global summary: lml += (m + log(S 2^-K)) - log N
__global__ void k_lml_global(const double* __restrict__ m_flags, const int64_t* __restrict__ S_all, int G, int K, double logN,
                             Scalars* sc)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        uint64_t S = 0;
        for (int g = 0; g < G; ++g) S += (uint64_t)S_all[g];
        const double m = m_flags[0];
        int f = (int)m_flags[1];
        if (!(f & FLAG_NAN) && m == -__builtin_huge_val()) f |= FLAG_ALL_NEGINF;
        sc->lml_est = sc->lml_est + (lse_from(m, S, K, f) - logN);
    }
}

} // namespace gpf
