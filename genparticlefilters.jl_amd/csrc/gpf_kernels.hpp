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
constexpr int MAX_PARTIALS = 2048;                // partial (max, flags) slots of the reduce kernels
constexpr int MAX_SHARDS = 64;                    // shards (GPUs) of one filter
constexpr int LDS_TILE_TABLE = 8192;              // tile-prefix entries kept in LDS by the search kernel (64 KiB)

// ----------------------------------------------------------------------------- device scalars
struct WSum {                  // summary of one weight vector
    double   m;                // maximum
    int32_t  flags;            // FLAG_NAN | FLAG_POSINF | FLAG_ALL_NEGINF  (safe_softmax, utils.jl:119-137)
    int32_t  pad;
    uint64_t S;                // sum of fixed-point weights
    uint64_t Ql[4];            // 32-bit limbs sums of sum q^2 (un-normalised)
    // strata of S over the filter's output slots, left by the scan that produced S: S = N sB + srem, sinv = N / S
    uint64_t sB, srem;
    double   sinv;
};
struct Scalars {
    WSum     prio;             // weights the resampler samples from (log_priorities)
    WSum     raw;              // state.log_weights (log-ML estimate, ESS)
    WSum     post;             // log_ws after a prioritised resample (update_weights!, resample.jl:198-200)
    double   lml_est;          // state.log_ml_est
    double   lw_fill;          // log-weight every particle carries after a whole-shard sub-state resample (gpf_resample_local)
    uint64_t Ctot;             // residual: number of deterministic copies (n_resampled)
    uint64_t Rs;               // residual: sum of residual weights
    uint64_t n_accept;         // accepted MH moves of the last gpf_rejuvenate
    int32_t  timeout;          // set if a bounded inter-workgroup spin gave up (never expected)
    int32_t  pad;
    long long opt_d;           // optimal resize: threshold position in the descending order (-1: none)
    uint64_t opt_a, opt_B;     // optimal resize: inverse weight threshold c = a S / B as the exact pair (a, B)
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

// inclusive max-scan of a u32 across the 64 lanes, by DPP (no LDS crossbar round trips): Hillis-Steele inside each row of
// 16, then the row totals travel with row_bcast15 / row_bcast31.  Lanes without a source read the identity 0.
__device__ __forceinline__ uint32_t wave_scan_max_u32(uint32_t v)
{
#define GPF_DPP_MAX(ctrl, rmask) { const uint32_t o_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xF, false); v = o_ > v ? o_ : v; }
    GPF_DPP_MAX(0x111, 0xF)   // row_shr:1
    GPF_DPP_MAX(0x112, 0xF)   // row_shr:2
    GPF_DPP_MAX(0x114, 0xF)   // row_shr:4
    GPF_DPP_MAX(0x118, 0xF)   // row_shr:8
    GPF_DPP_MAX(0x142, 0xA)   // row_bcast:15 -> rows 1, 3
    GPF_DPP_MAX(0x143, 0xC)   // row_bcast:31 -> rows 2, 3
#undef GPF_DPP_MAX
    return v;
}

// order-preserving key of Julia's isless on Float64 (-0.0 < 0.0); descending sort = ascending on ~key (K10), and its inverse
__device__ __forceinline__ uint64_t sort_key_desc(double v)
{
    const uint64_t u = d2u(v);
    const uint64_t asc = (u >> 63) ? ~u : (u | 0x8000000000000000ull);
    return ~asc;                              // ascending radix sort on ~key == descending by value, ties by index
}
__device__ __forceinline__ double sort_key_value(uint64_t key)
{
    const uint64_t asc = ~key;
    return u2d((asc >> 63) ? (asc & 0x7fffffffffffffffull) : ~asc);
}

// ----------------------------------------------------------------------------- K1/K2: init & step
// per-block (max, flags) of the log-weights a kernel has just written: the first pass of safe_softmax
// (utils.jl:119-128) rides on the kernel that produces the weights instead of re-reading them
__device__ __forceinline__ void track_max(double v, double& m, int& f)
{
    if (v != v) f |= FLAG_NAN;
    else { m = v > m ? v : m; if (v == __builtin_huge_val()) f |= FLAG_POSINF; }
}
__device__ __forceinline__ void block_max_store(double m, int f, double* __restrict__ pmax, int32_t* __restrict__ pflags)
{
    m = wave_max_f64(m);
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) f |= __shfl_xor(f, s, WAVE);
    __shared__ double sm_[NWAVES];
    __shared__ int sf_[NWAVES];
    if (lane_id() == 0) { sm_[wave_id()] = m; sf_[wave_id()] = f; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < NWAVES; ++w) { m = sm_[w] > m ? sm_[w] : m; f |= sf_[w]; }
        pmax[blockIdx.x] = m;
        pflags[blockIdx.x] = f;
    }
}

// stratified_map! (utils.jl:29-55): K strata, block size B = n div K; particle i < K B belongs to stratum i div B
// (:contiguous) or i mod K (:interleaved); the n - K B remaining particles draw a stratum uniformly (sample(strata, R)),
// here from one more Philox block of the particle (block index NBLK, behind the model's own blocks)
template <class Mo>
__device__ __forceinline__ int stratum_of(const ModelArgs& a, uint64_t seed, uint32_t epoch, int64_t gid0, int64_t i, int64_t n, uint32_t tag)
{
    const int64_t K = a.n_strata, B = n / K;
    if (i < K * B) return (int)(a.interleaved ? i % K : i / B);
    const Philox b = rng(seed, (uint32_t)(gid0 + i), (uint32_t)Mo::NBLK, epoch, tag);
    return (int)mulhi64(u64(b.w0, b.w1), (uint64_t)K);
}

// pf_initialize (initialize.jl:39-41) / pf_update! (update.jl:15-22): one lane per particle, row in,
// row out, lw += log p(y|x).  Counter-based RNG: no RNG state in memory.
// MODE 0: the model's own sampler; 1: native custom proposal; 2: stratified (the discrete latent constrained per stratum)
template <int M, int MODE = 0>
__global__ __launch_bounds__(BLOCK) void k_init(ModelArgs a, uint64_t seed, uint32_t epoch, int64_t gid0,
                                                int64_t n, int W, double* __restrict__ rows,
                                                double* __restrict__ lw, double* __restrict__ pmax,
                                                int32_t* __restrict__ pflags)
{
    using Mo = Model<M>;
    double bm = -__builtin_huge_val(); int bf = 0;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        double x[MAX_DIM];
        double ll;
        if constexpr (MODE == 1) ll = Mo::propose(a.P, true, nullptr, a.obs, seed, (uint32_t)(gid0 + i), 0, epoch, TAG_INIT, x);
        else if constexpr (MODE == 2) {
            const double v = a.strata[stratum_of<Mo>(a, seed, epoch, gid0, i, n, TAG_INIT)];
            const double lp = Mo::sample_stratum(a.P, true, nullptr, a.obs, v, seed, (uint32_t)(gid0 + i), 0, epoch, TAG_INIT, x);
            ll = (lp + Mo::loglik(a.P, x, a.obs)) + a.logK;                      // initialize.jl:103-104
        } else {
            Mo::sample(a.P, true, nullptr, a.obs, seed, (uint32_t)(gid0 + i), 0, epoch, TAG_INIT, x);
            ll = Mo::loglik(a.P, x, a.obs);
        }
        double* r = rows + i * W;
#pragma unroll
        for (int k = 0; k < Mo::D; ++k) r[k] = x[k];
        for (int k = Mo::D; k < W; ++k) r[k] = 0.0;
        lw[i] = ll;
        track_max(ll, bm, bf);
    }
    block_max_store(bm, bf, pmax, pflags);
}

// GATHER: the preceding pf_resample! left its ancestor vector pending; this kernel reads row anc[i]
// instead of row i (new_traces .= view(traces, parents), resample.jl:60, fused into the propagate) and
// the incoming log-weights are known to be 0 (update_weights!, resample.jl:195): lw = ll, no read.
// PACKED (sharded filters): the preceding resample left the population as the received exchange buffer
// [row | slot << 32 | global ancestor id] (gpf_shard_commit); entry k is propagated straight into its slot, the
// scatter pass (k_commit_packed) and its round trip through HBM disappear, the log-ML update rides along.
struct PackedCommit {
    const double* packed;      // [n][W + 1], or nullptr
    int32_t* anc;              // parents of the committed population
    const double* mf_all; const int64_t* tot_all; int G, K; double logN; Scalars* sc;   // update_lml_est! from the gathered summaries
    const double* lw_fill;     // GATHER after gpf_resample_local: the incoming log-weights are this constant, not 0 (resample.jl:210)
};
template <int M, int W, bool KEEP_PREV, bool GATHER, int MODE = 0, bool PACKED = false>
__global__ __launch_bounds__(BLOCK) void k_step(ModelArgs a, uint64_t seed, uint32_t epoch, int64_t gid0,
                                                int64_t n, const int32_t* __restrict__ anc,
                                                const double* __restrict__ rows_in,
                                                double* __restrict__ rows_out, double* __restrict__ lw,
                                                double* __restrict__ pmax, int32_t* __restrict__ pflags, PackedCommit pc)
{
    using Mo = Model<M>;
    constexpr int D = Mo::D;
    double bm = -__builtin_huge_val(); int bf = 0;
    if constexpr (PACKED) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            uint64_t S = 0;
            double mx = -__builtin_huge_val();
            int f = 0;
            for (int g = 0; g < pc.G; ++g) {
                S += (uint64_t)pc.tot_all[5 * g];
                const double v = pc.mf_all[2 * g]; mx = v > mx ? v : mx; f |= (int)pc.mf_all[2 * g + 1];
            }
            if (!(f & FLAG_NAN) && mx == -__builtin_huge_val()) f |= FLAG_ALL_NEGINF;
            pc.sc->lml_est = pc.sc->lml_est + (lse_from(mx, S, pc.K, f) - pc.logN);
        }
    }
    for (int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x; e < n; e += (int64_t)gridDim.x * BLOCK) {
        int64_t i = e;                                  // the slot this lane fills
        double r[W];
        if constexpr (PACKED) {
            const double* src = pc.packed + e * (W + 1);
#pragma unroll
            for (int c = 0; c < W; ++c) r[c] = src[c];
            const uint64_t meta = d2u(src[W]);
            i = (int64_t)(meta >> 32);
            pc.anc[i] = (int32_t)(meta & 0xffffffffull);
        } else {
        const int64_t srow = GATHER ? (int64_t)anc[i] : i;
        const double2* src = reinterpret_cast<const double2*>(rows_in + srow * W);
#pragma unroll
        for (int c = 0; c < (D + 1) / 2; ++c) { const double2 v = src[c]; r[2 * c] = v.x; r[2 * c + 1] = v.y; }
        }
        double xn[MAX_DIM];
        double ll;
        if constexpr (MODE == 1) ll = Mo::propose(a.P, false, r, a.obs, seed, (uint32_t)(gid0 + i), 0, epoch, TAG_UPDATE, xn);
        else if constexpr (MODE == 2) {
            const double v = a.strata[stratum_of<Mo>(a, seed, epoch, gid0, i, n, TAG_UPDATE)];
            const double lp = Mo::sample_stratum(a.P, false, r, a.obs, v, seed, (uint32_t)(gid0 + i), 0, epoch, TAG_UPDATE, xn);
            ll = (lp + Mo::loglik(a.P, xn, a.obs)) + a.logK;                     // update.jl:201-206
        } else {
            Mo::sample(a.P, false, r, a.obs, seed, (uint32_t)(gid0 + i), 0, epoch, TAG_UPDATE, xn);
            ll = Mo::loglik(a.P, xn, a.obs);
        }
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
        const double nl = (GATHER || PACKED) ? ((GATHER && pc.lw_fill) ? *pc.lw_fill + ll : ll)   // after a resample the incoming
                                             : lw[i] + ll;                                    // log-weights are 0 (or one constant)
        lw[i] = nl;
        track_max(nl, bm, bf);
    }
    block_max_store(bm, bf, pmax, pflags);
}

// K7/K8: pf_move_accept! with Gen.mh on the current step's latent (rejuvenate.jl:40-53) and
// pf_move_reweight! with move_reweight(trace, selection) (rejuvenate.jl:74-90, :125-132)
// GATHER: a pf_resample! left its ancestor vector pending; the move reads row anc[i] (new_traces .= view(traces, parents),
// resample.jl:60, fused) and the incoming log-weights are 0 (resample.jl:195), exactly like k_step<GATHER>.
template <int M, int W, bool REWEIGHT, bool GATHER = false>
__global__ __launch_bounds__(BLOCK) void k_move(ModelArgs a, uint64_t seed, uint32_t epoch, int64_t gid0,
                                                int64_t n, int has_prev, int n_iters, const int32_t* __restrict__ anc,
                                                const double* __restrict__ rows_in,
                                                double* __restrict__ rows_out, double* __restrict__ lw,
                                                unsigned long long* __restrict__ n_accept,
                                                double* __restrict__ pmax, int32_t* __restrict__ pflags)
{
    using Mo = Model<M>;
    constexpr int D = Mo::D, NB = Mo::NBLK;
    unsigned long long acc = 0;
    double bm = -__builtin_huge_val(); int bf = 0;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        double r[W];
        const int64_t srow = GATHER ? (int64_t)anc[i] : i;
        const double2* src = reinterpret_cast<const double2*>(rows_in + srow * W);
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
        if (REWEIGHT) { const double nl = (GATHER ? 0.0 : lw[i]) + wsum; lw[i] = nl; track_max(nl, bm, bf); }
        else if (GATHER) lw[i] = 0.0;
    }
    // one atomic per wave
    unsigned long long t = wave_sum_u64(acc);
    if (lane_id() == 0 && t) atomicAdd(n_accept, t);
    if (REWEIGHT) block_max_store(bm, bf, pmax, pflags);
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
// Single-pass inclusive prefix sum over 2048-element tiles.  Every tile publishes its AGGREGATE at once;
// its exclusive prefix is then ONE round trip: the whole workgroup reads, in parallel, the aggregates of
// all earlier tiles of the same round (<= grid-1 <= 511 words, two per lane) plus the inclusive PREFIX of
// the last tile of the previous round, and block-reduces them.  (A classic decoupled look-back walks 64
// predecessors per dependent L2 round trip; with <= a few thousand tiles the flat read is shorter.)
// A descriptor is ONE naturally aligned 8-byte word {valid bit 63 | 62-bit value}, written and polled
// with relaxed agent-scope atomics (the data IS the flag: no fence, placement-independent; per-XCD L2s
// are not coherent, so plain loads/stores would not do).  Deadlock freedom does not rely on dispatch
// order: the grid is sized to be fully resident and block b owns tiles b, b+G, b+2G, ...  Spins are
// bounded (Scalars::timeout).  Descriptor buffers are double-buffered per scan channel: a launch polls
// buffer `dcur` and zeroes `dnext` for the following launch, so no memset node is needed.
constexpr uint64_t DESC_VALID = 1ull << 63, DESC_MASK = (1ull << 62) - 1;
constexpr unsigned SPIN_LIMIT = 1u << 22;

__device__ __forceinline__ void desc_store(uint64_t* p, uint64_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t desc_load(const uint64_t* p)
{
    return __hip_atomic_load(const_cast<uint64_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t desc_wait(const uint64_t* p, int32_t* timeout)
{
    uint64_t d = desc_load(p);
    unsigned spins = 0;
    while (!(d & DESC_VALID)) {
        __builtin_amdgcn_s_sleep(1);
        d = desc_load(p);
        if (++spins > SPIN_LIMIT) { *timeout = 1; break; }
    }
    return d & DESC_MASK;
}

// input functors: the two fixed-point weights at elements idx, idx+1 (idx even; zero beyond n)
struct InFixQ {                // q_i = trunc(exp(p_i - m) 2^K + 1/2); uniform fallback q_i = 1
    PrioView pv;
    const int32_t* order;      // optional permutation (sort_particles, resample.jl:156-157)
    const uint64_t* sorted_keys;   // with `order`: the sorted keys themselves -- key i IS log_priorities[order[i]] (sort_key_value),
                                   // read in streaming order instead of 8-byte random reads through `order`
    int K;
    double m; int flags;       // filled in-kernel from the partials
    __device__ __forceinline__ uint64_t one(double v, bool uniform, bool bad) const
    {
        return uniform ? 1 : (bad ? 0 : exp_fix(v - m, K));
    }
    // the two log-priorities at idx, idx + 1 (anything beyond n): needs neither the maximum nor the flags, so a scan can
    // have its first tile's loads in flight while it folds the partial maxima
    __device__ __forceinline__ void raw2(int64_t idx, int64_t n, double& v0, double& v1) const
    {
        if (pv.mode == 0 && order == nullptr && idx + 1 < n) {          // 16 B per lane, 1 KiB per wave-instruction
            const double2 v = *reinterpret_cast<const double2*>(pv.lw + idx);
            v0 = v.x; v1 = v.y;
        } else if (sorted_keys && idx + 1 < n) {
            const ulonglong2 k = *reinterpret_cast<const ulonglong2*>(sorted_keys + idx);
            v0 = sort_key_value(k.x); v1 = sort_key_value(k.y);
        } else {
            v0 = idx < n ? pv.at(order ? (int64_t)order[idx] : idx) : 0.0;
            v1 = idx + 1 < n ? pv.at(order ? (int64_t)order[idx + 1] : idx + 1) : 0.0;
        }
    }
    __device__ __forceinline__ void conv2(int64_t idx, int64_t n, double v0, double v1, uint64_t& q0, uint64_t& q1) const
    {
        const bool uniform = (flags & FLAG_ALL_NEGINF) != 0, bad = (flags & (FLAG_NAN | FLAG_POSINF)) != 0;
        q0 = idx < n ? one(v0, uniform, bad) : 0;
        q1 = idx + 1 < n ? one(v1, uniform, bad) : 0;
    }
    __device__ __forceinline__ void load2(int64_t idx, int64_t n, uint64_t& q0, uint64_t& q1) const
    {
        double v0, v1;
        raw2(idx, n, v0, v1);
        conv2(idx, n, v0, v1, q0, q1);
    }
};
// a <= ... products of a 31-bit count and a 62-bit weight need 128 bits:  B <= a * k
__device__ __forceinline__ bool le_mul(uint64_t B, uint64_t a, uint64_t k)
{
    return __umul64hi(a, k) != 0 || B <= a * k;
}
struct InOptimal {             // optimal resize (resize.jl:156-167): keep flags [c w_i >= 1], or the weights of the others
    const double* lw;
    const WSum* ws;            // summary of state.log_weights
    const Scalars* sc;         // (opt_a, opt_B)
    int K;
    int mode;                  // 0: keep flags; 1: q_i of the particles not kept; 2: 1 for every particle not kept
    __device__ __forceinline__ uint64_t one(int64_t i, double m, bool uniform, bool bad, uint64_t a, uint64_t B) const
    {
        const uint64_t q = uniform ? 1 : (bad ? 0 : exp_fix(lw[i] - m, K));
        const bool keep = le_mul(B, a, q);
        return mode == 0 ? (uint64_t)keep : (keep ? 0 : (mode == 2 ? 1 : q));
    }
    __device__ __forceinline__ void load2(int64_t idx, int64_t n, uint64_t& q0, uint64_t& q1) const
    {
        const double m = ws->m;
        const int fl = ws->flags;
        const bool uniform = (fl & FLAG_ALL_NEGINF) != 0, bad = (fl & (FLAG_NAN | FLAG_POSINF)) != 0;
        const uint64_t a = sc->opt_a, B = sc->opt_B;
        q0 = idx < n ? one(idx, m, uniform, bad, a, B) : 0;
        q1 = idx + 1 < n ? one(idx + 1, m, uniform, bad, a, B) : 0;
    }
};

// where a scan writes: the CDF (padded to whole tiles) and its coarser levels, by-products of the same pass
struct ScanOut {
    uint64_t* cdf;             // [ntiles*2048] inclusive prefix of every element (nullptr: totals only)
    uint64_t* t16;             // [ntiles*128]  inclusive prefix at the end of every 16-element group (one 128-B line of cdf)
    uint64_t* t256;            // [ntiles*8]    ... of every 256-element group (one 128-B line of t16)
    uint32_t* k32;             // [ntiles*64]   (prefix at the end of every 32-element group) >> KEY_SHIFT: the 4-byte keys k_search_multi keeps in LDS
    // k_search_multi's two narrow levels below a key group of G = 32 << logg cells (nullptr / -1: not wanted):
    uint16_t* off16;           // [ntiles*2048] every prefix as a 16-bit offset inside its key group (key_quant)
    uint16_t* coarse;          // [ntiles*2048 / CS] the offsets of cells CS-1 (mod CS), CS = G / 8: one 16-byte row per key group
    int logg;
};
constexpr int KEY_SHIFT = 30;  // S < 2^62, so (prefix >> 30) fits 32 bits whatever N is
// A key group spans the prefixes [klo << 30, (khi + 1) << 30) (klo / khi: the 4-byte keys at its two ends).  Inside it a
// prefix -- and a target -- is quantised to 16 bits by ONE shift: x -> (x - (klo << 30)) >> sh, sh = 14 + ceil(log2(khi - klo + 1)).
// The map is monotone and the SAME on the producer (scan) and consumer (search) side, so
//     off(cell) < off(T) => prefix < T,   off(cell) > off(T) => prefix > T,   equal offsets: the exact prefixes decide.
__device__ __forceinline__ int key_quant_shift(uint32_t klo, uint32_t khi)
{
    const uint64_t w = (uint64_t)khi - klo + 1;
    return (KEY_SHIFT - 16) + (w > 1 ? 64 - (int)__builtin_clzll(w - 1) : 0);
}

// Arrangement: wave w of the workgroup owns 512 consecutive elements of the tile as 4 rows of 128;
// lane l holds elements 2l, 2l+1 of each row, so every global access is 16 B per lane, contiguous
// across the wave (1 KiB per wave-instruction), for the loads AND the CDF stores.
struct ScanExtras {            // optional side jobs of a scan launch
    int64_t* zero128;          // clear 2 * MAX_SHARDS exchange counters (sharded resample), or nullptr
    int64_t* host_flags;       // pinned host {flags, ticket}: publish the validity flags of the weights, or nullptr
    int64_t ticket;
    int64_t n_slots;           // > 0: also write ws_out->{sB, srem, sinv}, the strata of the total over n_slots slots
};
constexpr int SCAN_ROWS = 4;
// MODE 0: plain scan of In; 1: fixed-point weights (folds the max partials); 2: as 1, plus sum q^2 for the ESS;
// 3 / 4: as 1 / 2 with the maximum and flags taken from the np gathered (max, flags) pairs of the shards (pmax = mf_all)
template <class In, int MODE>
__global__ __launch_bounds__(BLOCK) void k_scan(In in, int64_t n, int64_t ntiles,
                                                const double* __restrict__ pmax, const int32_t* __restrict__ pflags,
                                                int np, WSum* __restrict__ ws_out, ScanOut out,
                                                uint64_t* __restrict__ dcur, uint64_t* __restrict__ dnext,
                                                uint64_t* __restrict__ total_out, uint64_t* __restrict__ blockQ,
                                                int32_t* __restrict__ timeout, ScanExtras ex)
{
    // (ex.n_slots: the thread that ends up with the total also leaves the stratum width of S over n_slots output slots)
    // sharded resamples: the exchange counters of the push pass that follows are cleared here (no memset node)
    if (ex.zero128 && blockIdx.x == 0 && threadIdx.x < 2 * MAX_SHARDS) ex.zero128[threadIdx.x] = 0;
    __shared__ double sm[NWAVES];
    __shared__ int sf[NWAVES];
    __shared__ uint64_t s_wave[NWAVES];
    __shared__ uint64_t s_red[NWAVES];
    uint64_t* const d_agg = dcur;
    uint64_t* const d_pre = dcur + ntiles;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < 2 * ntiles; i += (int64_t)gridDim.x * BLOCK) dnext[i] = 0;
    constexpr bool WANT_Q = MODE == 2 || MODE == 4;
    // the first tile's log-weights are loaded BEFORE the partial maxima are folded (they need neither m nor the flags)
    double pre[2 * SCAN_ROWS];
    if constexpr (MODE >= 1) {
        const int64_t wb0 = (int64_t)blockIdx.x * TILE + (int64_t)wave_id() * (SCAN_ROWS * 2 * WAVE) + 2 * lane_id();
#pragma unroll
        for (int k = 0; k < SCAN_ROWS; ++k) in.raw2(wb0 + k * 2 * WAVE, n, pre[2 * k], pre[2 * k + 1]);
    }
    if constexpr (MODE >= 1) {
        double m; int f;
        if constexpr (MODE >= 3) {
            m = -__builtin_huge_val(); f = 0;
            for (int g = 0; g < np; ++g) { const double v = pmax[2 * g]; m = v > m ? v : m; f |= (int)pmax[2 * g + 1]; }
            if (!(f & FLAG_NAN) && m == -__builtin_huge_val()) f |= FLAG_ALL_NEGINF;
            (void)sm; (void)sf;
        } else fold_partials(pmax, pflags, np, sm, sf, m, f);
        in.m = m; in.flags = f;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            ws_out->m = m; ws_out->flags = f;
            // check = true / :warn (resample.jl:54-55): the host learns safe_softmax's validity flags NOW, from pinned memory,
            // while this kernel and the ancestor search behind it keep running (no stream synchronisation, no idle gap)
            if (ex.host_flags) {
                __hip_atomic_store(ex.host_flags, (int64_t)f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(ex.host_flags + 1, ex.ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    uint64_t ql[4] = {0, 0, 0, 0};
    const int lane = lane_id(), wv = wave_id();
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t wbase = tile * TILE + (int64_t)wv * (SCAN_ROWS * 2 * WAVE) + 2 * lane;
        uint64_t p[2 * SCAN_ROWS];                     // inclusive prefixes inside the wave's 512-element chunk
        uint64_t cb[SCAN_ROWS];                        // the chunk's total before each row
        uint64_t carry = 0;
#pragma unroll
        for (int k = 0; k < SCAN_ROWS; ++k) {
            uint64_t q0, q1;
            cb[k] = carry;
            if constexpr (MODE >= 1) {
                if (tile == blockIdx.x) in.conv2(wbase + k * 2 * WAVE, n, pre[2 * k], pre[2 * k + 1], q0, q1);
                else in.load2(wbase + k * 2 * WAVE, n, q0, q1);
            } else in.load2(wbase + k * 2 * WAVE, n, q0, q1);
            if constexpr (WANT_Q) {
                uint64_t lo = q0 * q0, hi = __umul64hi(q0, q0);
                ql[0] += lo & 0xffffffffull; ql[1] += lo >> 32; ql[2] += hi & 0xffffffffull; ql[3] += hi >> 32;
                lo = q1 * q1; hi = __umul64hi(q1, q1);
                ql[0] += lo & 0xffffffffull; ql[1] += lo >> 32; ql[2] += hi & 0xffffffffull; ql[3] += hi >> 32;
            }
            const uint64_t pair = q0 + q1;
            const uint64_t inc = wave_scan_u64(pair);
            p[2 * k] = carry + (inc - pair) + q0;
            p[2 * k + 1] = p[2 * k] + q1;
            carry += shfl_u64(inc, WAVE - 1);
        }
        if (lane == 0) s_wave[wv] = carry;             // wave total
        __syncthreads();
        uint64_t wexcl = 0, agg = 0;
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) { if (w < wv) wexcl += s_wave[w]; agg += s_wave[w]; }
        if (threadIdx.x == 0) desc_store(d_agg + tile, DESC_VALID | agg);
        // exclusive prefix of this tile: one parallel read of the round's earlier aggregates
        const int64_t first = (tile / gridDim.x) * gridDim.x;
        uint64_t acc = 0;
        for (int64_t idx = first + threadIdx.x; idx < tile; idx += BLOCK) acc += desc_wait(d_agg + idx, timeout);
        if (first > 0 && threadIdx.x == BLOCK - 1) acc += desc_wait(d_pre + first - 1, timeout);
        acc = wave_sum_u64(acc);
        if (lane == 0) s_red[wv] = acc;
        __syncthreads();
        uint64_t excl = 0;
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) excl += s_red[w];
        if (threadIdx.x == 0) desc_store(d_pre + tile, DESC_VALID | (excl + agg));
        const uint64_t off = excl + wexcl;
        if (out.cdf) {
#pragma unroll
            for (int k = 0; k < SCAN_ROWS; ++k) {
                const int64_t idx = wbase + k * 2 * WAVE;
                const uint64_t v1 = off + p[2 * k + 1];
                *reinterpret_cast<ulonglong2*>(out.cdf + idx) = make_ulonglong2(off + p[2 * k], v1);
                if ((lane & 7) == 7) out.t16[(idx + 1) >> 4] = v1;                     // element idx+1 = 15 (mod 16)
                if ((lane & 15) == 15) out.k32[(idx + 1) >> 5] = (uint32_t)(v1 >> KEY_SHIFT);   // ... = 31 (mod 32)
                if (lane == WAVE - 1 && (k & 1)) out.t256[(idx + 1) >> 8] = v1;         // ... = 255 (mod 256)
                if (out.off16) {                                                        // kernel-uniform
                    // 16-bit offsets inside the key group (16 << logg lanes of this row): klo = key of the previous group
                    const int GL = 16 << out.logg;
                    const uint32_t kv = (uint32_t)(v1 >> KEY_SHIFT);
                    const uint32_t khi = (uint32_t)__shfl((int)kv, lane | (GL - 1), WAVE);
                    const uint32_t kprev = (uint32_t)__shfl((int)kv, ((lane & ~(GL - 1)) - 1) & (WAVE - 1), WAVE);
                    const uint32_t klo = lane < GL ? (uint32_t)((off + cb[k]) >> KEY_SHIFT) : kprev;
                    const int sh = key_quant_shift(klo, khi);
                    const uint64_t kb = (uint64_t)klo << KEY_SHIFT;
                    const uint32_t o0 = (uint32_t)((off + p[2 * k] - kb) >> sh), o1 = (uint32_t)((v1 - kb) >> sh);
                    reinterpret_cast<uint32_t*>(out.off16)[idx >> 1] = o0 | (o1 << 16);
                    // the coarse row: offsets of the cells CS-1 (mod CS), CS = 4 << logg, two per 4-byte store
                    if (out.logg == 0) {
                        const uint32_t part = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)o1, 0x55, 0xF, 0xF, false);    // quad_perm [1,1,1,1]
                        if ((lane & 3) == 3) reinterpret_cast<uint32_t*>(out.coarse)[(idx + 1) >> 3] = part | (o1 << 16);
                    } else {
                        const uint32_t part = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)o1, 0x114, 0xF, 0xF, false);   // row_shr:4
                        if ((lane & 7) == 7) reinterpret_cast<uint32_t*>(out.coarse)[(idx + 1) >> 4] = part | (o1 << 16);
                    }
                }
            }
        }
        if (tile == ntiles - 1 && threadIdx.x == BLOCK - 1) {
            const uint64_t Stot = off + p[2 * SCAN_ROWS - 1];
            *total_out = Stot;
            if constexpr (MODE >= 1) {
                if (ex.n_slots > 0) {                   // one thread per launch: a true 64-bit division is fine here
                    const uint64_t Bq = Stot / (uint64_t)ex.n_slots;
                    ws_out->sB = Bq; ws_out->srem = Stot - Bq * (uint64_t)ex.n_slots;
                    ws_out->sinv = (double)ex.n_slots / (double)Stot;
                }
            }
        }
        __syncthreads();                                // s_wave / s_red reuse
    }
    if constexpr (WANT_Q) {
        // block partial of the limb sums of sum q^2 (plain stores, folded on demand by k_publish_scalars)
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

// Residual resampling needs TWO prefix sums over the same elements: the copy counts c_i = (N q_i) div S and the
// residual weights r_i = ((N q_i) mod S) >> sh (resample.jl:99,109).  One pass computes both: one read of the weight CDF,
// ONE 64-bit division per element (quotient and remainder), two descriptor channels polled in the same round trip.
// Same tile / descriptor protocol as k_scan (channel A = counts, channel B = residual weights).
struct Scan2Chan { ScanOut out; uint64_t* dcur; uint64_t* dnext; uint64_t* total_out; };
__global__ __launch_bounds__(BLOCK) void k_scan_residual2(const uint64_t* __restrict__ cdf, const WSum* ws, int64_t Nslots,
                                                          int64_t n, int64_t ntiles, Scan2Chan A, Scan2Chan B,
                                                          int32_t* __restrict__ timeout)
{
    __shared__ uint64_t s_wave[2][NWAVES];
    __shared__ uint64_t s_red[2][NWAVES];
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < 2 * ntiles; i += (int64_t)gridDim.x * BLOCK) { A.dnext[i] = 0; B.dnext[i] = 0; }
    const uint64_t S = ws->S;
    const int sh = residual_shift(S, Nslots);
    const int lane = lane_id(), wv = wave_id();
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t wbase = tile * TILE + (int64_t)wv * (SCAN_ROWS * 2 * WAVE) + 2 * lane;
        uint64_t pa[2 * SCAN_ROWS], pb[2 * SCAN_ROWS];
        uint64_t ca = 0, cb = 0;
#pragma unroll
        for (int k = 0; k < SCAN_ROWS; ++k) {
            const int64_t idx = wbase + k * 2 * WAVE;
            const ulonglong2 c = *reinterpret_cast<const ulonglong2*>(cdf + idx);      // padded to whole tiles, flat beyond n
            const uint64_t prev = idx > 0 ? cdf[idx - 1] : 0;
            uint64_t a0 = 0, a1 = 0, b0 = 0, b1 = 0;
            if (S != 0) {
                const uint64_t n0 = (uint64_t)Nslots * (c.x - prev), n1 = (uint64_t)Nslots * (c.y - c.x);
                a0 = n0 / S; b0 = (n0 - a0 * S) >> sh;
                a1 = n1 / S; b1 = (n1 - a1 * S) >> sh;
            }
            if (idx >= n) { a0 = 0; b0 = 0; }
            if (idx + 1 >= n) { a1 = 0; b1 = 0; }
            const uint64_t paira = a0 + a1, pairb = b0 + b1;
            const uint64_t inca = wave_scan_u64(paira), incb = wave_scan_u64(pairb);
            pa[2 * k] = ca + (inca - paira) + a0; pa[2 * k + 1] = pa[2 * k] + a1;
            pb[2 * k] = cb + (incb - pairb) + b0; pb[2 * k + 1] = pb[2 * k] + b1;
            ca += shfl_u64(inca, WAVE - 1); cb += shfl_u64(incb, WAVE - 1);
        }
        if (lane == 0) { s_wave[0][wv] = ca; s_wave[1][wv] = cb; }
        __syncthreads();
        uint64_t wexa = 0, agga = 0, wexb = 0, aggb = 0;
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) {
            if (w < wv) { wexa += s_wave[0][w]; wexb += s_wave[1][w]; }
            agga += s_wave[0][w]; aggb += s_wave[1][w];
        }
        if (threadIdx.x == 0) { desc_store(A.dcur + tile, DESC_VALID | agga); desc_store(B.dcur + tile, DESC_VALID | aggb); }
        const int64_t first = (tile / gridDim.x) * gridDim.x;
        uint64_t acca = 0, accb = 0;
        for (int64_t idx = first + threadIdx.x; idx < tile; idx += BLOCK) {
            acca += desc_wait(A.dcur + idx, timeout);
            accb += desc_wait(B.dcur + idx, timeout);
        }
        if (first > 0 && threadIdx.x == BLOCK - 1) {
            acca += desc_wait(A.dcur + ntiles + first - 1, timeout);
            accb += desc_wait(B.dcur + ntiles + first - 1, timeout);
        }
        acca = wave_sum_u64(acca); accb = wave_sum_u64(accb);
        if (lane == 0) { s_red[0][wv] = acca; s_red[1][wv] = accb; }
        __syncthreads();
        uint64_t exa = 0, exb = 0;
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) { exa += s_red[0][w]; exb += s_red[1][w]; }
        if (threadIdx.x == 0) {
            desc_store(A.dcur + ntiles + tile, DESC_VALID | (exa + agga));
            desc_store(B.dcur + ntiles + tile, DESC_VALID | (exb + aggb));
        }
        const uint64_t offa = exa + wexa, offb = exb + wexb;
#pragma unroll
        for (int k = 0; k < SCAN_ROWS; ++k) {
            const int64_t idx = wbase + k * 2 * WAVE;
            const uint64_t va = offa + pa[2 * k + 1], vb = offb + pb[2 * k + 1];
            *reinterpret_cast<ulonglong2*>(A.out.cdf + idx) = make_ulonglong2(offa + pa[2 * k], va);
            *reinterpret_cast<ulonglong2*>(B.out.cdf + idx) = make_ulonglong2(offb + pb[2 * k], vb);
            if ((lane & 7) == 7) { A.out.t16[(idx + 1) >> 4] = va; B.out.t16[(idx + 1) >> 4] = vb; }
            if ((lane & 15) == 15) { A.out.k32[(idx + 1) >> 5] = (uint32_t)(va >> KEY_SHIFT); B.out.k32[(idx + 1) >> 5] = (uint32_t)(vb >> KEY_SHIFT); }
            if (lane == WAVE - 1 && (k & 1)) { A.out.t256[(idx + 1) >> 8] = va; B.out.t256[(idx + 1) >> 8] = vb; }
        }
        if (tile == ntiles - 1 && threadIdx.x == BLOCK - 1) { *A.total_out = offa + pa[2 * SCAN_ROWS - 1]; *B.total_out = offb + pb[2 * SCAN_ROWS - 1]; }
        __syncthreads();                                // s_wave / s_red reuse
    }
}

// the device scalar block -> its pinned host mirror, ticket last: the host polls the ticket instead of synchronising the
// stream (a hipMemcpyAsync + hipStreamSynchronize pair costs ~13 us of wake-up latency per getter; this costs the launch)
// blockQ != nullptr: the limb partials of sum q^2 that the scan blocks left (only the ESS needs them) are folded into sc->raw.Ql
// on the way -- one launch for "fold + publish" (the ESS-triggered loop of BASELINE config 4 asks for the ESS every step).
// Launched with ONE wave.
__global__ void k_publish_scalars(Scalars* sc, Scalars* host, long long* host_ticket, long long ticket,
                                  const uint64_t* __restrict__ blockQ, int nblk)
{
    constexpr int NW = (int)(sizeof(Scalars) / sizeof(unsigned long long));
    static_assert(sizeof(Scalars) % sizeof(unsigned long long) == 0, "Scalars must be a whole number of 8-byte words");
    constexpr int QW = (int)((offsetof(Scalars, raw) + offsetof(WSum, Ql)) / sizeof(unsigned long long));
    uint64_t q[4] = {0, 0, 0, 0};
    const bool fold = blockQ != nullptr;
    if (fold) {
        for (int b = threadIdx.x; b < nblk; b += WAVE)
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] += blockQ[(int64_t)b * 4 + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = wave_sum_u64(q[k]);                 // every lane holds the totals
        if (threadIdx.x < 4) sc->raw.Ql[threadIdx.x] = threadIdx.x == 0 ? q[0] : threadIdx.x == 1 ? q[1] : threadIdx.x == 2 ? q[2] : q[3];
    }
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(sc);
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(host);
    for (int i = threadIdx.x; i < NW; i += blockDim.x) {
        unsigned long long v = src[i];
        if (fold && i >= QW && i < QW + 4) v = i == QW ? q[0] : i == QW + 1 ? q[1] : i == QW + 2 ? q[2] : q[3];   // (not read back: just written)
        __hip_atomic_store(dst + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(host_ticket, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ----------------------------------------------------------------------------- scalar bookkeeping
// ----------------------------------------------------------------------------- K5: ancestor search
// a = first index with cdf[a] > T.  The CDF comes with coarser levels written by the scan (fan-out 16):
// top level (per-256 prefixes, or the prefix of every 2^g-th tile when those do not fit) is binary-searched in LDS, then each
// further level costs ONE 128-byte line: 16 consecutive u64 loaded with 8 independent 16-B loads and
// compared in registers.  Two dependent L2 round trips per slot instead of eleven.
struct CdfLevels {
    const uint64_t* cdf;  const uint64_t* t16;  const uint64_t* t256;  const uint64_t* ttile;   // ttile: descriptor words
    const uint32_t* k32;                                                                        // 4-byte keys per 32 cells (ScanOut::k32)
    const uint16_t* off16; const uint16_t* coarse; int logg;                                    // ScanOut::off16 / coarse / logg
};
// a pointer rebuilt from an integer is generic (flat_load: also counts on lgkmcnt and serialises behind the LDS
// reads); the lines live in global memory, so say so
__device__ __forceinline__ ulonglong2 load_global_16(uint64_t addr, int sub)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned long long __attribute__((ext_vector_type(2))) u64x2;
    const __attribute__((address_space(1))) u64x2* g = reinterpret_cast<const __attribute__((address_space(1))) u64x2*>(addr);
    const u64x2 v = g[sub];
    return make_ulonglong2(v.x, v.y);
#else
    (void)addr; (void)sub;
    return make_ulonglong2(0, 0);
#endif
}

// Number of entries <= T in a 128-byte line (16 u64), for every lane's own (line, T) at once.
// A lane reading its whole line alone costs 8 L1 transactions on 8 different cycles (each 16-B lane access
// to a distinct line is its own tag lookup); here 8 lanes share one line: in round r the 8-lane group g
// serves the slot of lane 8r+g, each lane loads 16 B of it (one line = ONE coalesced transaction), the
// group sums its compare results and hands the count back.  8x fewer L1 transactions per slot.
// sum of an int over each aligned group of 8 lanes, by DPP (no LDS traffic): xor 1, xor 2 inside the quad,
// then the mirrored lane of the other quad
__device__ __forceinline__ int group8_sum(int c)
{
    c += __builtin_amdgcn_update_dpp(0, c, 0xB1, 0xF, 0xF, false);     // quad_perm [1,0,3,2]
    c += __builtin_amdgcn_update_dpp(0, c, 0x4E, 0xF, 0xF, false);     // quad_perm [2,3,0,1]
    c += __builtin_amdgcn_update_dpp(0, c, 0x141, 0xF, 0xF, false);    // row_half_mirror: lane i <- lane 7-i
    return c;
}
// In round r the 8-lane group g serves ITS OWN member 8g+r: the member's (line, T) is broadcast through the wave's
// LDS strip, each lane loads 16 B of the line (one coalesced transaction per line), compares, the group sums.
// two independent slots per lane at once (16 line loads in flight per lane): lds_wave holds 2 x 64 entries
__device__ __forceinline__ void coop_count_le2(const uint64_t* line0, uint64_t T0, const uint64_t* line1, uint64_t T1,
                                               ulonglong2* lds_wave, int& c0, int& c1)
{
    const int lane = lane_id(), sub = lane & 7, gbase = lane & ~7;
    lds_wave[lane] = make_ulonglong2(reinterpret_cast<uint64_t>(line0), T0);
    lds_wave[WAVE + lane] = make_ulonglong2(reinterpret_cast<uint64_t>(line1), T1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    ulonglong2 v[16];
    uint64_t t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const ulonglong2 pt = lds_wave[(r >> 3) * WAVE + gbase + (r & 7)];
        t[r] = pt.y;
        v[r] = load_global_16(pt.x, sub);
    }
    c0 = 0; c1 = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int c = group8_sum((int)(v[r].x <= t[r]) + (int)(v[r].y <= t[r]));
        if (r < 8) c0 = sub == r ? c : c0; else c1 = sub == (r - 8) ? c : c1;
    }
    __builtin_amdgcn_wave_barrier();
}
// per-lane variant for coherent targets (stratified, residual head): neighbouring lanes hit the same lines,
// the loads coalesce by themselves and the cooperation overhead is not worth it
__device__ __forceinline__ int count_le_line(const uint64_t* __restrict__ line, uint64_t T)
{
    const ulonglong2* v = reinterpret_cast<const ulonglong2*>(line);
    ulonglong2 r[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) r[c] = v[c];
    int cnt = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) cnt += (r[c].x <= T) + (r[c].y <= T);
    return cnt;
}

// sharded stratified resampling (k_strat_plan): strata are contiguous in slot order and the shards' CDF ranges are contiguous
// in target order, so the global slots a shard serves are ONE range
struct ShardPlan {
    WSum ws;                                                          // the GLOBAL weight sum and its strata constants
    int64_t first, count;                                             // this shard serves the global slots [first, first + count)
    uint64_t t_off;                                                   // where this shard's CDF starts in the global one
};
struct SearchArgs {
    CdfLevels w;                                                      // weights (or residual weights for the tail)
    CdfLevels c;                                                      // residual: copy counts
    int64_t ntiles;
    const int32_t* order;                                             // sorted stratified
    Scalars* sc;
    const WSum* ws;                                                   // summary of the sampled weights
    const WSum* raw;                                                  // summary of state.log_weights (log-ML estimate)
    const ShardPlan* plan;                                            // k_search_strat on a shard: slots and target offset (ws = &plan->ws)
    int64_t n, n_global, gid0;                                        // n = output slots; n_global = slots of the whole filter
    int64_t n_cells;                                                  // particles the CDF ranges over (== n except when resizing)
    uint64_t seed; uint32_t epoch;
    int K; double logN;
    double invN;                                                      // 1 / n_global (stratified)
    int update_lml;                                                   // 0 for sub-state views (resample.jl:185-187); 2: whole-shard
                                                                      // sub-state, the kept mass goes to sc->lw_fill (resample.jl:210)
    int32_t* anc;
};

// LDS copy of the top level: one pad word per 64 entries.  The branch-free search probes at power-of-two strides;
// unpadded, every probe of the middle steps would land in the same bank (up to 64-way conflicts).
__host__ __device__ __forceinline__ int64_t lds_pad(int64_t i) { return i + (i >> 6); }

// One fat workgroup (1024 threads = 16 waves) per CU: the top level of the CDF is copied into LDS once per CU
// instead of once per 256-thread workgroup.
constexpr int SBLOCK = 1024;
#ifndef SEARCH_WAVES_PER_SIMD
#define SEARCH_WAVES_PER_SIMD 4
#endif
#ifndef SEARCH_BLOCKS_PER_CU
#define SEARCH_BLOCKS_PER_CU 1
#endif
// ---- the search core shared by k_search (single GPU) and k_serve (sharded): top level in LDS + two line levels
struct SearchTop {
    const uint64_t* topw; const uint64_t* topc;      // top level of the weight CDF / of the residual copy-count CDF
    int64_t tn;                                      // entries of the top level
    int steps;                                       // ceil(log2(tn + 1))
    int gshift;                                      // !top256: one top entry = the prefix at the end of 2^gshift tiles
    bool top256;
};
// shape of the LDS top level for a CDF of `ntiles` tiles (nt tables side by side): the per-256 prefixes when they fit, else
// the prefix at the end of every g-th tile with the smallest power of two g that fits (g = 1 up to 16.7 M particles, 8 up
// to 134 M, ...): the table always lives in LDS, whatever N is
__host__ __device__ __forceinline__ void search_top_shape(int64_t ntiles, int nt, bool& top256, int& gshift, int64_t& tn)
{
    top256 = nt * ntiles * 8 <= LDS_TILE_TABLE;
    gshift = 0;
    if (top256) { tn = ntiles * 8; return; }
    while (nt * ((ntiles + ((int64_t)1 << gshift) - 1) >> gshift) > LDS_TILE_TABLE) ++gshift;
    tn = (ntiles + ((int64_t)1 << gshift) - 1) >> gshift;
}
__host__ inline size_t search_lds_bytes(int64_t ntiles, int nt)
{
    bool t256; int gs; int64_t tn;
    search_top_shape(ntiles, nt, t256, gs, tn);
    return (size_t)(nt * (lds_pad(tn) + 1)) * sizeof(uint64_t);
}
// block-collective: copy the top level(s) into LDS (smem: dynamic LDS, (two ? 2 : 1) * (lds_pad(tn) + 1) words)
__device__ __forceinline__ SearchTop search_prologue(const CdfLevels& w, const CdfLevels& c, bool two, int64_t ntiles, uint64_t* smem)
{
    SearchTop st;
    const int nt = two ? 2 : 1;
    search_top_shape(ntiles, nt, st.top256, st.gshift, st.tn);
    uint64_t* tw = smem;
    uint64_t* tc = tw + lds_pad(st.tn);
    if (st.top256) {
        const uint64_t* srcw = w.t256;
        const uint64_t* srcc = c.t256;
        // 16 B per lane (the per-256 level has a multiple of 8 entries)
        for (int64_t t = 2 * (int64_t)threadIdx.x; t < st.tn; t += 2 * (int64_t)blockDim.x) {
            const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(srcw + t);
            tw[lds_pad(t)] = v.x; tw[lds_pad(t + 1)] = v.y;
            if (two) {
                const ulonglong2 x = *reinterpret_cast<const ulonglong2*>(srcc + t);
                tc[lds_pad(t)] = x.x; tc[lds_pad(t + 1)] = x.y;
            }
        }
    } else {
        // prefix at the end of every 2^gshift-th tile, from the tiles' descriptor words (they carry a valid bit)
        const int64_t g = (int64_t)1 << st.gshift;
        for (int64_t t = threadIdx.x; t < st.tn; t += blockDim.x) {
            const int64_t last = ((t + 1) * g < ntiles ? (t + 1) * g : ntiles) - 1;
            tw[lds_pad(t)] = w.ttile[last] & DESC_MASK;
            if (two) tc[lds_pad(t)] = c.ttile[last] & DESC_MASK;
        }
    }
    __syncthreads();
    st.topw = tw;
    st.topc = tc;
    st.steps = 0;
    while (((int64_t)1 << st.steps) <= st.tn) ++st.steps;
    return st;
}
// two slots per lane: idx[u] = first index of L[u] whose prefix exceeds T[u].  WAVE-COLLECTIVE when coop (wave-uniform).
__device__ __forceinline__ void search_pair(const SearchTop& st, const CdfLevels* const L[2], const uint64_t* const top[2],
                                            const uint64_t T[2], bool coop, ulonglong2* lds_wave, int64_t n_cells, int64_t ntiles,
                                            int64_t idx[2])
{
    const int64_t n256 = ntiles * 8, n16 = ntiles * (TILE / 16);
    // top level: branch-free binary search, both slots interleaved; pos = number of entries <= T
    int64_t pos[2] = {0, 0};
    for (int s = st.steps - 1; s >= 0; --s) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t np = pos[u] + ((int64_t)1 << s);
            if (np <= st.tn) {
                const uint64_t v = top[u][lds_pad(np - 1)];
                if (v <= T[u]) pos[u] = np;
            }
        }
    }
    int64_t s256[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (st.top256) s256[u] = pos[u];
        else {
            int64_t tile = pos[u] << st.gshift;
            if (st.gshift) {                                      // inside the group of 2^gshift tiles: their descriptor prefixes
                const int64_t hi = tile + ((int64_t)1 << st.gshift) < ntiles ? tile + ((int64_t)1 << st.gshift) : ntiles;
                int64_t cnt = 0;
                for (int64_t e = tile; e < hi; ++e) cnt += ((L[u]->ttile[e] & DESC_MASK) <= T[u]);
                tile += cnt;
            }
            tile = tile < ntiles ? tile : ntiles - 1;
            const uint64_t* g = L[u]->t256 + tile * 8;             // the tile's 8 per-256 prefixes: 64 B
            int c = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) c += (g[e] <= T[u]);
            s256[u] = tile * 8 + c;
        }
        s256[u] = s256[u] < n256 ? s256[u] : n256 - 1;
    }
    int c0, c1;
    const uint64_t* l0 = L[0]->t16 + s256[0] * 16;
    const uint64_t* l1 = L[1]->t16 + s256[1] * 16;
    if (coop) coop_count_le2(l0, T[0], l1, T[1], lds_wave, c0, c1);
    else { c0 = count_le_line(l0, T[0]); c1 = count_le_line(l1, T[1]); }
    int64_t s16a = s256[0] * 16 + c0, s16b = s256[1] * 16 + c1;
    s16a = s16a < n16 ? s16a : n16 - 1;
    s16b = s16b < n16 ? s16b : n16 - 1;
    l0 = L[0]->cdf + s16a * 16;
    l1 = L[1]->cdf + s16b * 16;
    if (coop) coop_count_le2(l0, T[0], l1, T[1], lds_wave, c0, c1);
    else { c0 = count_le_line(l0, T[0]); c1 = count_le_line(l1, T[1]); }
    idx[0] = s16a * 16 + c0; idx[1] = s16b * 16 + c1;
    idx[0] = idx[0] < n_cells ? idx[0] : n_cells - 1;
    idx[1] = idx[1] < n_cells ? idx[1] : n_cells - 1;
}

// once per resample: update_lml_est! (resample.jl:57,178-182), or for a whole-shard sub-state the log-weight its particles keep
__device__ __forceinline__ void resample_bookkeeping(const SearchArgs& a)
{
    const double v = lse_from(a.raw->m, a.raw->S, a.K, a.raw->flags) - a.logN;
    if (a.update_lml == 2) a.sc->lw_fill = v;                    // resample.jl:210: every log-weight = logsumexp - log n
    else a.sc->lml_est = a.sc->lml_est + v;
}
template <int METHOD>
__global__ __launch_bounds__(SBLOCK, SEARCH_WAVES_PER_SIMD) void k_search(SearchArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const SearchTop st = search_prologue(a.w, a.c, METHOD == 1, a.ntiles, reinterpret_cast<uint64_t*>(smem));
    // update_lml_est! (resample.jl:57,178-182): log_ml_est += logsumexp(log_weights) - log N, once per resample
    if (a.update_lml && blockIdx.x == 0 && threadIdx.x == 0)
        resample_bookkeeping(a);
    const uint64_t S = (METHOD == 1 || METHOD == 3) ? a.sc->Rs : a.ws->S;
    const uint64_t N = (uint64_t)a.n_global;
    // systematic: S = N B + rem, once per workgroup (u64 division is ~100 instructions)
    __shared__ uint64_t s_div[2];
    __shared__ ulonglong2 s_coop[2 * SBLOCK];
    ulonglong2* const lds_wave = s_coop + wave_id() * (2 * WAVE);
    if (METHOD == 3) {
        if (threadIdx.x == 0) { s_div[0] = S / N; s_div[1] = S % N; }
        __syncthreads();
    }
    const uint64_t Ctot = (METHOD == 1) ? a.sc->Ctot : 0;
    // two slots per lane and iteration (independent dependency chains); the loop is wave-uniform
    for (int64_t base = (int64_t)blockIdx.x * 2 * SBLOCK; base < a.n; base += (int64_t)gridDim.x * 2 * SBLOCK) {
        int64_t j[2]; bool act[2], head[2]; uint64_t T[2]; const uint64_t* top[2]; const CdfLevels* L[2];
        // the lane's two CONSECUTIVE slots share one Philox block when their ids form an aligned pair (gfp_math.hpp
        // resample_u64); RNG keyed by the global id; systematic sampling (METHOD 3) draws ONE uniform for all slots
        const uint32_t s0 = (uint32_t)(a.gid0 + base + 2 * (int64_t)threadIdx.x);
        const Philox pb0 = rng(a.seed, METHOD == 3 ? 0u : s0 >> 1, 0, a.epoch, TAG_RESAMPLE);
        const Philox pb1 = (METHOD != 3 && (s0 & 1u)) ? rng(a.seed, (s0 >> 1) + 1u, 0, a.epoch, TAG_RESAMPLE) : pb0;   // kernel-uniform branch
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            j[u] = base + 2 * (int64_t)threadIdx.x + u;
            act[u] = j[u] < a.n;
            const uint64_t jg = (uint64_t)j[u];                        // slot index inside this filter / view (never used on shards)
            const uint64_t U = METHOD == 3 ? u64(pb0.w0, pb0.w1) : resample_pick(u ? pb1 : pb0, s0 + (uint32_t)u);
            head[u] = false; top[u] = st.topw; L[u] = &a.w;
            if (METHOD == 0) T[u] = mulhi64(U, S);                    // multinomial, resample.jl:59
            else if (METHOD == 3) {                                 // systematic: floor((j S + floor(U S)) / n), resize.jl:170-178
                T[u] = jg * s_div[0] + (jg * s_div[1] + mulhi64(U, S)) / N;
            } else {                                                  // residual, resample.jl:96-115
                head[u] = jg < Ctot;
                T[u] = head[u] ? jg : mulhi64(U, S);
                if (head[u]) { top[u] = st.topc; L[u] = &a.c; }
            }
        }
        // coherent targets (stratified; residual waves that are all deterministic copies) read their lines per lane
        const bool coop = METHOD == 0 ? true : (METHOD == 3 ? false : __any(!head[0] || !head[1]) != 0);
        int64_t idx[2];
        search_pair(st, L, top, T, coop, lds_wave, a.n_cells, a.ntiles, idx);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (act[u]) a.anc[j[u]] = (int32_t)idx[u];
        }
    }
}

// ----------------------------------------------------------------------------- K5a: i.i.d. targets, 4-byte keys in LDS
// rand!(Categorical(weights), parents) (resample.jl:59): N independent targets, no locality to exploit.  What a slot
// costs is (a) instructions and (b) bytes fetched from arrays too large for the XCD's 4 MB L2 -- random 128-byte lines
// of the 8 N-byte CDF come over the fabric, and that traffic, not the ALU, bounded the line-counting search.  Here a
// slot touches the CDF itself only when two 16-bit offsets tie (about 1e-3 of the slots):
//   level 1, LDS: the prefix at the end of every G = 32 << LOGG cells as a 4-byte key (prefix >> KEY_SHIFT, ScanOut::k32;
//            122 KB at 10^6 particles, one 1024-thread workgroup per CU), uniform binary search, 32-bit compares;
//            key < (T >> KEY_SHIFT) => prefix <= T, key > => prefix > T, equal keys: the exact prefix decides;
//   level 2, one 16-byte read per lane: the group's coarse row, the 16-bit offsets (key_quant_shift) of every
//            (G/8)-th cell -> which run of CS = G / 8 cells;
//   level 3, one CS*2-byte read per lane: the offsets of that run -> the cell.  Equal offsets: the exact prefixes decide.
// Both offset arrays are written by the scan (ScanOut::off16 / coarse), 2.5 N bytes together: they stay in L2.
constexpr int MULTI_LDS_BUDGET = 160 * 1024 - 2048;            // (the kernels keep up to ~1 KiB of static LDS besides the table)
__host__ __device__ __forceinline__ int64_t multi_groups(int64_t ntiles, int logg) { return (ntiles * (TILE / 32)) >> logg; }
// LDS copy of the keys: one pad word per 32 entries.  The uniform binary search probes at power-of-two strides; unpadded,
// every probe of the middle steps would land in the same bank (64-way conflicts)
__host__ __device__ __forceinline__ uint32_t kpad(uint32_t i) { return i + (i >> 5); }
__host__ inline size_t multi_lds_bytes(int64_t ntiles, int logg) { return (size_t)(kpad((uint32_t)multi_groups(ntiles, logg)) + 1) * sizeof(uint32_t); }
// smallest LOGG whose key table fits (-1: none; the caller falls back to k_search)
__host__ inline int multi_logg(int64_t ntiles)
{
    for (int g = 0; g <= 1; ++g) if (multi_lds_bytes(ntiles, g) <= (size_t)MULTI_LDS_BUDGET) return g;
    return -1;
}
// number of 16-bit halves of x that are < the halves of qq (qq = q | q << 16), as 0/1 per half; and != qq
typedef unsigned short __attribute__((ext_vector_type(2))) u16x2;
__device__ __forceinline__ uint32_t pk_lt(uint32_t x, uint32_t qq)
{
    const u16x2 d = __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, qq), __builtin_bit_cast(u16x2, x));   // > 0 iff x < q
    const u16x2 one = {1, 1};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(d, one));
}
__device__ __forceinline__ uint32_t pk_ne(uint32_t x, uint32_t qq)
{
    const u16x2 one = {1, 1};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, x ^ qq), one));
}

#ifndef GPF_MULTI_NS
#define GPF_MULTI_NS 4
#endif
struct MultiTable { const uint32_t* keys; uint32_t ng, p2; float kscale; };      // the LDS key table of k_search_multi
constexpr uint32_t MULTI_WIN = 512;                // interpolation window of the key search

// the second half of the lookup: pos[u] = the key group that holds T[u] (number of groups that end at or below it); the
// target becomes a 16-bit offset inside the group, then two narrow reads.  key(i) = key of group i (LDS table or global level).
template <int LOGG, int NS, class KeyFn>
__device__ __forceinline__ void multi_inside(KeyFn&& key, uint32_t ng, const CdfLevels& w, int64_t n_cells, const uint64_t (&T)[NS],
                                             const uint32_t (&pos)[NS], uint32_t (&idx)[NS])
{
    constexpr int G = 32 << LOGG, CS = G / 8;
    uint32_t g[NS], qq[NS], run[NS];
    uint4 row[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        g[u] = pos[u] < ng ? pos[u] : ng - 1;
        const uint32_t klo = g[u] ? key(g[u] - 1) : 0u, khi = key(g[u]);
        const uint64_t kb = (uint64_t)klo << KEY_SHIFT;
        const uint64_t d = T[u] > kb ? T[u] - kb : 0;
        uint32_t q = (uint32_t)(d >> key_quant_shift(klo, khi));
        q = q < 65535u ? q : 65535u;
        qq[u] = q | (q << 16);
        row[u] = *reinterpret_cast<const uint4*>(w.coarse + (size_t)g[u] * 8);
    }
    bool tie[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        // run = number of coarse offsets < q (the last one, the group's end, is >= q: T lies in this group)
        uint32_t c = pk_lt(row[u].x, qq[u]) + pk_lt(row[u].y, qq[u]) + pk_lt(row[u].z, qq[u]) + pk_lt(row[u].w, qq[u]);
        c = (c & 0xffffu) + (c >> 16);
        run[u] = c < 8u ? c : 7u;
    }
    uint4 fine[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        const uint16_t* fp = w.off16 + (size_t)(g[u] * (uint32_t)G + run[u] * (uint32_t)CS);
        if (CS == 4) { const uint2 f = *reinterpret_cast<const uint2*>(fp); fine[u] = make_uint4(f.x, f.y, 0xffffffffu, 0xffffffffu); }
        else fine[u] = *reinterpret_cast<const uint4*>(fp);
    }
    bool anytie = false;
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        uint32_t lt = pk_lt(fine[u].x, qq[u]) + pk_lt(fine[u].y, qq[u]);
        uint32_t ne = pk_ne(fine[u].x, qq[u]) + pk_ne(fine[u].y, qq[u]);
        if (CS != 4) {
            lt += pk_lt(fine[u].z, qq[u]) + pk_lt(fine[u].w, qq[u]);
            ne += pk_ne(fine[u].z, qq[u]) + pk_ne(fine[u].w, qq[u]);
        }
        lt = (lt & 0xffffu) + (lt >> 16); ne = (ne & 0xffffu) + (ne >> 16);
        tie[u] = ne != (uint32_t)CS;
        anytie = anytie || tie[u];
        idx[u] = g[u] * (uint32_t)G + run[u] * (uint32_t)CS + lt;
    }
    if (__any(anytie)) {
        // a cell of the run shares the target's offset: the exact prefixes decide.  Every cell before the run is below T
        // (its run's coarse offset is < q); walk from the run's first cell -- equal offsets may continue into later runs
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            if (!tie[u]) continue;
            uint32_t i = g[u] * (uint32_t)G + run[u] * (uint32_t)CS;
            const uint32_t end = g[u] * (uint32_t)G + (uint32_t)G;
            // the run's CS exact prefixes in one round trip; only a run that lies entirely at or below T walks on
            const ulonglong2* cp = reinterpret_cast<const ulonglong2*>(w.cdf + i);
            uint32_t c = 0;
#pragma unroll
            for (int e = 0; e < CS / 2; ++e) { const ulonglong2 v = cp[e]; c += (uint32_t)(v.x <= T[u]) + (uint32_t)(v.y <= T[u]); }
            i += c;
            if (c == (uint32_t)CS) while (i < end && w.cdf[i] <= T[u]) ++i;
            idx[u] = i;
        }
    }
    const uint32_t last = (uint32_t)(n_cells - 1);
#pragma unroll
    for (int u = 0; u < NS; ++u) idx[u] = idx[u] < last ? idx[u] : last;
}

// idx[u] = first cell whose prefix exceeds T[u], for the lane's NS independent targets (wave-collective: the fast paths are
// taken when every lane of the wave can take them).  Levels as described above; LOGG as in the key table.
template <int LOGG, int NS>
__device__ __forceinline__ void multi_lookup(const MultiTable& tb, const CdfLevels& w, int64_t n_cells, const uint64_t (&T)[NS], uint32_t (&idx)[NS])
{
    constexpr int G = 32 << LOGG, CS = G / 8;
    constexpr uint32_t WIN = MULTI_WIN;
    uint32_t t[NS], pos[NS];
    // ---- number of keys < t.  Fast path: the CDF of exchangeable weights is close to linear, so a window of WIN keys
    //      around the interpolated position brackets the answer (checked); else the uniform binary search of the whole table
    bool inwin = tb.ng >= 2 * WIN;
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        t[u] = (uint32_t)(T[u] >> KEY_SHIFT);
        const uint32_t pe = (uint32_t)((float)t[u] * tb.kscale);
        uint32_t lo = pe > WIN / 2 ? pe - WIN / 2 : 0u;
        lo = lo + WIN > tb.ng ? tb.ng - WIN : lo;
        pos[u] = lo;
    }
    if (inwin) {
#pragma unroll
        for (int u = 0; u < NS; ++u)
            inwin = inwin && (pos[u] == 0u || tb.keys[kpad(pos[u] - 1)] < t[u]) && tb.keys[kpad(pos[u] + WIN - 1)] >= t[u];
    }
    if (__all(inwin)) {
#pragma unroll
        for (uint32_t h = WIN / 2; h >= 1; h >>= 1) {
#pragma unroll
            for (int u = 0; u < NS; ++u) pos[u] += tb.keys[kpad(pos[u] + h - 1)] < t[u] ? h : 0u;
        }
    } else {
#pragma unroll
        for (int u = 0; u < NS; ++u) pos[u] = tb.keys[kpad(tb.p2 - 1)] < t[u] ? tb.ng - tb.p2 : 0u;          // uniform binary search: no bounds checks below
        for (uint32_t h = tb.p2 >> 1; h >= 1; h >>= 1) {
#pragma unroll
            for (int u = 0; u < NS; ++u) pos[u] += tb.keys[kpad(pos[u] + h - 1)] < t[u] ? h : 0u;
        }
    }
    bool amb = false;
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        const uint32_t k = tb.keys[kpad(pos[u])];                       // pos <= tb.ng - 1 here
        pos[u] += k < t[u] ? 1u : 0u;                                // pos = number of keys < t: those groups end at or below T
        amb = amb || k == t[u] || (k < t[u] && pos[u] < tb.ng && tb.keys[kpad(pos[u])] == t[u]);
    }
    if (__any(amb)) {
        // equal keys: the exact prefix decides (rare: one key value in 2^32 S / (2^30 groups) per slot)
#pragma unroll
        for (int u = 0; u < NS; ++u)
            while (pos[u] < tb.ng && tb.keys[kpad(pos[u])] == t[u] && w.cdf[(int64_t)pos[u] * G + (G - 1)] <= T[u]) ++pos[u];
    }
    multi_inside<LOGG, NS>([&](uint32_t i) { return tb.keys[kpad(i)]; }, tb.ng, w, n_cells, T, pos, idx);
}

// block-collective: the key table into LDS, 16 B per lane from the scan's key level (every (1 << LOGG)-th key).  The loads are
// issued first, `between()` runs while they are in flight (the caller's first targets), then the table is written.
template <int LOGG, class Between>
__device__ __forceinline__ MultiTable multi_table_load(const CdfLevels& w, int64_t ntiles, uint64_t S, uint32_t* keys, Between&& between)
{
    constexpr int KT = (MULTI_LDS_BUDGET / 4 / (LOGG == 0 ? 4 : 2) + SBLOCK - 1) / SBLOCK;   // 16-byte source loads per lane that cover any table within the budget
    MultiTable tb;
    tb.keys = keys;
    tb.ng = (uint32_t)multi_groups(ntiles, LOGG);                       // >= 64 >> LOGG
    const uint32_t nq = LOGG == 0 ? tb.ng / 4 : tb.ng / 2;
    uint4 kv[KT];
    const uint4* src = reinterpret_cast<const uint4*>(w.k32);
#pragma unroll
    for (int r = 0; r < KT; ++r) { const uint32_t q = threadIdx.x + (uint32_t)r * SBLOCK; if (q < nq) kv[r] = src[q]; }
    tb.p2 = 1;                                                           // largest power of two <= ng
    while (2 * tb.p2 <= tb.ng) tb.p2 *= 2;
    tb.kscale = (float)tb.ng / (float)((S >> KEY_SHIFT) + 1);            // groups per key unit: where a key would sit were the CDF linear
    between();
#pragma unroll
    for (int r = 0; r < KT; ++r) {
        const uint32_t q = threadIdx.x + (uint32_t)r * SBLOCK;
        if (q < nq) {
            if (LOGG == 0) { uint32_t* d = keys + kpad(4 * q); d[0] = kv[r].x; d[1] = kv[r].y; d[2] = kv[r].z; d[3] = kv[r].w; }   // 4 q .. 4 q + 3 share their pad offset
            else { uint32_t* d = keys + kpad(2 * q); d[0] = kv[r].y; d[1] = kv[r].w; }
        }
    }
    __syncthreads();
    return tb;
}

template <int LOGG>
__global__ __launch_bounds__(SBLOCK, 4) void k_search_multi(SearchArgs a)
{
    constexpr int NS = GPF_MULTI_NS;                                     // 2 or 4 slots per lane
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // update_lml_est! (resample.jl:57,178-182): log_ml_est += logsumexp(log_weights) - log N, once per resample
    if (a.update_lml && blockIdx.x == 0 && threadIdx.x == 0)
        resample_bookkeeping(a);
    const uint64_t S = a.ws->S;
    // the lane's NS consecutive slots from slot `base` on (independent chains: the LDS and L2 round trips of one hide
    // behind the others); one Philox block per aligned slot pair (resample_u64), one more block when the run starts odd
    auto targets = [&](int64_t base, uint64_t* T) {
        const uint32_t s0 = (uint32_t)(a.gid0 + base + NS * (int64_t)threadIdx.x), sb = s0 >> 1;
        if (!(s0 & 1u)) {                                                // kernel-uniform
#pragma unroll
            for (int q = 0; q < NS / 2; ++q) {
                const Philox b = rng(a.seed, sb + (uint32_t)q, 0, a.epoch, TAG_RESAMPLE);
                T[2 * q] = mulhi64(u64(b.w0, b.w1), S); T[2 * q + 1] = mulhi64(u64(b.w2, b.w3), S);     // resample.jl:59
            }
        } else {
#pragma unroll
            for (int q = 0; q <= NS / 2; ++q) {
                const Philox b = rng(a.seed, sb + (uint32_t)q, 0, a.epoch, TAG_RESAMPLE);
                if (q > 0) T[2 * q - 1] = mulhi64(u64(b.w0, b.w1), S);
                if (q < NS / 2) T[2 * q] = mulhi64(u64(b.w2, b.w3), S);
            }
        }
    };
    const int64_t stride = (int64_t)gridDim.x * NS * SBLOCK;
    int64_t base = (int64_t)blockIdx.x * NS * SBLOCK;
    uint64_t T[NS];
    const MultiTable tb = multi_table_load<LOGG>(a.w, a.ntiles, S, reinterpret_cast<uint32_t*>(smem), [&]() { targets(base, T); });
    for (; base < a.n; base += stride) {
        const int64_t j0 = base + NS * (int64_t)threadIdx.x;
        uint32_t idx[NS];
        multi_lookup<LOGG, NS>(tb, a.w, a.n_cells, T, idx);
        int32_t* dst = a.anc + j0;
        if (j0 + NS <= a.n && (reinterpret_cast<uintptr_t>(dst) & (4 * NS - 1)) == 0) {
            if (NS == 4) *reinterpret_cast<int4*>(dst) = make_int4((int32_t)idx[0], (int32_t)idx[1], (int32_t)idx[2], (int32_t)idx[3]);
            else *reinterpret_cast<int2*>(dst) = make_int2((int32_t)idx[0], (int32_t)idx[1]);
        } else {
#pragma unroll
            for (int u = 0; u < NS; ++u) if (j0 + u < a.n) dst[u] = (int32_t)idx[u];
        }
        if (base + stride < a.n) targets(base + stride, T);
    }
}

// ----------------------------------------------------------------------------- K5b: stratified search = a streaming merge
// Stratified targets are monotone in the slot index (resample.jl:159-168 walks strata and weights with two pointers).
// A workgroup owns MJB consecutive slots; their targets lie in [L(j0), L(j0 + MJB)), i.e. in ONE contiguous range of CDF
// cells, found with two cooperative 128-ary searches of the per-256 level.  The range is streamed (16 B per lane) and the
// merge runs from the CELL side: cell i resolves every slot with a target below cdf[i], and that count is closed-form --
// the stratum t that contains cdf[i] (one Float64 multiply, off by one at most) plus a look at the targets of the
// neighbouring slots, kept in LDS.  The first slot NOT resolved by cells <= i belongs to a cell >= i + 1: an LDS max
// of (i + 1) at that slot, then ONE inclusive max-scan over the slots yields every ancestor.  No per-slot search, no
// dependent memory round trip per slot: 8 N bytes in, 4 N bytes out.
constexpr int MBLOCK = 256;
#ifndef GPF_MSLOTS
#define GPF_MSLOTS 8
#endif
constexpr int MSLOTS = GPF_MSLOTS;                 // consecutive slots per lane (16-byte ancestor stores)
constexpr int MJB = MBLOCK * MSLOTS;               // slots per workgroup
constexpr int64_t MONO_WIDE = 8 * (int64_t)MJB;    // a cell range wider than this is searched per slot, not streamed

// Block-cooperative: A0 / A1 = number of entries of arr[0..cnt) (ascending) that are <= L0 / <= L1 (L0 <= L1).
// Fast path, ONE global round trip of one coalesced 8-byte load per thread: a 256-entry window around `guess` (for
// exchangeable weights the CDF is close to linear, so the caller's guess is a few entries off at most); accepted only if the
// window brackets both answers.  Otherwise 256-ary rounds over the whole array.  `between()` runs after the window's loads
// have been issued and before their values are needed -- it also produces the two bounds (L0, L1), so that whatever THEY
// wait for (device scalars) and the caller's ALU work hide the round trip.
template <class Between>
__device__ __forceinline__ void block_count_le_pair(const uint64_t* __restrict__ arr, int64_t cnt, int64_t guess,
                                                    int (*s_cnt)[2][NWAVES], int64_t& A0, int64_t& A1, Between&& between)
{
    const int tid = (int)threadIdx.x;
    uint64_t Lq[2];
    int par = 0;
    {
        int64_t w_lo = guess - MBLOCK / 2;
        w_lo = w_lo + MBLOCK > cnt ? cnt - MBLOCK : w_lo;
        w_lo = w_lo < 0 ? 0 : w_lo;
        const int64_t w_hi = w_lo + MBLOCK < cnt ? w_lo + MBLOCK : cnt;
        const uint64_t v = w_lo + tid < w_hi ? arr[w_lo + tid] : ~0ull;
        between(Lq[0], Lq[1]);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = (int)__popcll(__ballot(v <= Lq[q]));
            if (lane_id() == 0) s_cnt[par][q][wave_id()] = c;
        }
        __syncthreads();
        int64_t k0 = 0, k1 = 0;
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) { k0 += s_cnt[par][0][w]; k1 += s_cnt[par][1][w]; }
        par ^= 1;
        if ((k0 > 0 || w_lo == 0) && (k1 < w_hi - w_lo || w_hi == cnt)) { A0 = w_lo + k0; A1 = w_lo + k1; return; }   // block-uniform
    }
    int64_t lo[2] = {0, 0}, hi[2] = {cnt, cnt};    // invariant: lo <= answer <= hi
    while (hi[0] > lo[0] || hi[1] > lo[1]) {       // block-uniform
        int64_t step[2]; uint64_t v[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int64_t len = hi[q] - lo[q];
            step[q] = len <= MBLOCK ? 1 : (len + MBLOCK - 1) / MBLOCK;
            const int64_t p = lo[q] + (int64_t)(tid + 1) * step[q] - 1;
            v[q] = p < hi[q] ? arr[p] : ~0ull;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = (int)__popcll(__ballot(v[q] <= Lq[q]));
            if (lane_id() == 0) s_cnt[par][q][wave_id()] = c;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (hi[q] > lo[q]) {
                int64_t k = 0;
#pragma unroll
                for (int w = 0; w < NWAVES; ++w) k += s_cnt[par][q][w];
                const int64_t nlo = lo[q] + k * step[q];
                const int64_t cap = step[q] == 1 ? nlo : nlo + step[q] - 1;        // the first probe that failed bounds the answer
                hi[q] = cap < hi[q] ? cap : hi[q];
                lo[q] = nlo;
            }
        }
        par ^= 1;
    }
    A0 = lo[0]; A1 = lo[1];
}

// (4 waves per SIMD = 4 workgroups per CU: a 10^6-slot launch is ONE resident round of workgroups)
__global__ __launch_bounds__(MBLOCK, 4) void k_search_strat(SearchArgs a)
{
    static_assert(MSLOTS % 4 == 0, "ancestors leave the lane as 16-byte stores");
    __shared__ __attribute__((aligned(16))) uint64_t s_T[MJB + 4];   // targets of the block's slots (+inf beyond n, and as padding)
    __shared__ __attribute__((aligned(16))) uint32_t s_mark[MJB];
    __shared__ int s_cnt[2][2][NWAVES];
    __shared__ uint32_t s_wmax[NWAVES];
    const int tid = (int)threadIdx.x, lane = lane_id(), wv = wave_id();
    // update_lml_est! (resample.jl:57,178-182): log_ml_est += logsumexp(log_weights) - log N, once per resample
    if (a.update_lml && blockIdx.x == 0 && tid == 0)
        resample_bookkeeping(a);
    const uint64_t N = (uint64_t)a.n_global;
    const double invN = a.invN;
    const int64_t j0 = (int64_t)blockIdx.x * MJB;
    // a shard serves the global slots [first, first + count) out of its own CDF, which starts at t_off in the global one:
    // strata and RNG counters by GLOBAL slot, targets and strata bounds shifted into local coordinates (signed: the first
    // served stratum may start below the shard's range)
    const int64_t n_out = a.plan ? (a.plan->count < a.n ? a.plan->count : a.n) : a.n;
    if (j0 >= n_out) return;                                          // (the grid of a shard is sized for the send buffer)
    const int64_t sbase = a.plan ? a.plan->first : 0;                 // strata: global slot of the launch's slot 0
    const int64_t pbase = a.plan ? a.plan->first : a.gid0;            // RNG counters
    const int64_t t_off = a.plan ? (int64_t)a.plan->t_off : 0;
    // ---- the CDF cells the block's targets can fall into, at per-256 granularity (block_count_le_pair on the per-256
    //      level), with the block's targets computed while the probes are in flight: MSLOTS consecutive slots per lane,
    //      one Philox block per aligned slot pair (gpf_math.hpp resample_u64; one more block when the run starts odd),
    //      strata boundaries by running remainder (no division per slot)
    constexpr int NPB = MSLOTS / 2;
    const uint64_t t0 = (uint64_t)(MSLOTS * tid);
    const uint32_t s0 = (uint32_t)(pbase + j0 + (int64_t)t0), sb = s0 >> 1;
    const bool odd = (s0 & 1u) != 0;                   // kernel-uniform
    const int64_t n256 = a.ntiles * 8;
    int64_t A0, A1;
    uint64_t Lj0, Lj1;                                                 // strata bounds of the block, local, clamped at 0
    int64_t Lj0s;                                                      // ... unclamped
    // (the guess: were the weights equal, slot j0's target would fall into cell j0 n_cells / n_out)
    const int64_t guess = (int64_t)((double)j0 * (a.plan ? (double)a.n_cells / (double)n_out : (double)a.n_cells * invN)) >> 8;
    block_count_le_pair(a.w.t256, n256, guess, s_cnt, A0, A1, [&](uint64_t& L0, uint64_t& L1) {
        // S = N B + rem; stratum j is [L(j), L(j+1)), L(j) = j B + floor(j rem / N)   (DESIGN.md §3.3); B, rem and N / S
        // were left beside S by the scan that produced it
        const uint64_t B = a.ws->sB, rem = a.ws->srem;
        const uint64_t jg0 = (uint64_t)(sbase + j0);
        const uint64_t q0 = div_small(jg0 * rem, N, invN), r0 = jg0 * rem - q0 * N;
        const uint64_t Lg0 = jg0 * B + q0;                                              // global
        Lj0s = (int64_t)Lg0 - t_off;
        Lj0 = Lj0s > 0 ? (uint64_t)Lj0s : 0;
        Lj1 = (uint64_t)(Lj0s + (int64_t)((uint64_t)MJB * B + div_small(r0 + (uint64_t)MJB * rem, N, invN)));
        L0 = Lj0; L1 = Lj1 - 1;
        uint64_t U[MSLOTS];
        if (!odd) {
#pragma unroll
            for (int q = 0; q < NPB; ++q) {
                const Philox b = rng(a.seed, sb + (uint32_t)q, 0, a.epoch, TAG_RESAMPLE);
                U[2 * q] = u64(b.w0, b.w1); U[2 * q + 1] = u64(b.w2, b.w3);
            }
        } else {                                       // the run starts on the odd half of a block: one block more
#pragma unroll
            for (int q = 0; q <= NPB; ++q) {
                const Philox b = rng(a.seed, sb + (uint32_t)q, 0, a.epoch, TAG_RESAMPLE);
                if (q > 0) U[2 * q - 1] = u64(b.w0, b.w1);
                if (q < NPB) U[2 * q] = u64(b.w2, b.w3);
            }
        }
        const uint64_t x = r0 + t0 * rem, qq = div_small(x, N, invN);
        uint64_t rr = x - qq * N;
        int64_t L = Lj0s + (int64_t)(t0 * B + qq);                                      // local: a target of a served slot is >= 0
        uint64_t T[MSLOTS];
#pragma unroll
        for (int k = 0; k < MSLOTS; ++k) {
            const int64_t j = j0 + (int64_t)t0 + k;
            const uint64_t r2 = rr + rem;
            const bool carry = r2 >= N;
            const int64_t Ln = L + (int64_t)B + (carry ? 1 : 0);
            rr = carry ? r2 - N : r2;
            T[k] = j < n_out ? (uint64_t)(L + (int64_t)mulhi64(U[k], (uint64_t)(Ln - L))) : ~0ull;   // resample.jl:162
            L = Ln;
        }
#pragma unroll
        for (int k = 0; k < MSLOTS; k += 2) *reinterpret_cast<ulonglong2*>(s_T + MSLOTS * tid + k) = make_ulonglong2(T[k], T[k + 1]);
        if (tid < 4) s_T[MJB + tid] = ~0ull;
#pragma unroll
        for (int k = 0; k < MSLOTS; k += 4) *reinterpret_cast<uint4*>(s_mark + MSLOTS * tid + k) = make_uint4(0u, 0u, 0u, 0u);
    });
    const int64_t g_lo = A0 < n256 ? A0 : n256 - 1, g_hi = A1 < n256 ? A1 : n256 - 1;
    const int64_t i_start = g_lo * 256, i_end = g_hi * 256 + 256;
    if (tid == 0) s_mark[0] = (uint32_t)i_start;
    __syncthreads();
    uint32_t res[MSLOTS];
    if (i_end - i_start <= MONO_WIDE) {
        // ---- stream the cells; cell i resolves e = #{slots of the block with a target < cdf[i]} slots
        const double inv_step = a.ws->sinv;
        const uint64_t* cbase = a.w.cdf + i_start;
        const uint32_t ncell = (uint32_t)(i_end - i_start), ibase = (uint32_t)i_start + 1u;
        constexpr int CPF = 6;                                            // 16-byte loads in flight per lane: 3072 cells per sweep
        for (uint32_t i0 = 0; i0 < ncell; i0 += 2u * MBLOCK * CPF) {
            ulonglong2 cc[CPF];
#pragma unroll
            for (int r = 0; r < CPF; ++r) {
                const uint32_t i = i0 + 2u * MBLOCK * r + 2u * (uint32_t)tid;
                cc[r] = i < ncell ? *reinterpret_cast<const ulonglong2*>(cbase + i) : make_ulonglong2(~0ull, ~0ull);
            }
#pragma unroll
            for (int r = 0; r < CPF; ++r) {
                const uint32_t i = i0 + 2u * MBLOCK * r + 2u * (uint32_t)tid;
                if (i0 + 2u * MBLOCK * r >= ncell) break;                 // block-uniform
                // cells at or below L(j0) resolve nothing: only the LAST of them (cells ascend) bounds slot 0
                const uint64_t below = __ballot(cc[r].y <= Lj0);
                if (cc[r].y <= Lj0) {
                    if (lane == (int)__popcll(below) - 1) atomicMax(&s_mark[0], ibase + i + 1u);
                    continue;
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const uint64_t c = u ? cc[r].y : cc[r].x;
                    if (c >= Lj1) continue;                               // every slot of the block is resolved by then
                    uint32_t e = 0;
                    if (c > Lj0) {
                        // c lies in stratum t of the block, t within [te - 1, te + 2] (L(j) = L(j0) + t step +- 1, step >= 1)
                        const int te = (int)((double)((int64_t)c - Lj0s) * inv_step);
                        const int b = te > 0 ? (te < MJB ? te - 1 : MJB - 1) : 0;
                        e = (uint32_t)b + (s_T[b] < c) + (s_T[b + 1] < c) + (s_T[b + 2] < c) + (s_T[b + 3] < c);
                    }
                    atomicMax(&s_mark[e], ibase + i + (uint32_t)u);       // slot e belongs to a cell >= i + 1
                }
            }
        }
        __syncthreads();
        // ---- inclusive max-scan over the slots
#pragma unroll
        for (int k = 0; k < MSLOTS; k += 4) {
            const uint4 m = *reinterpret_cast<const uint4*>(s_mark + MSLOTS * tid + k);
            res[k] = m.x; res[k + 1] = m.y; res[k + 2] = m.z; res[k + 3] = m.w;
        }
#pragma unroll
        for (int k = 1; k < MSLOTS; ++k) res[k] = res[k] > res[k - 1] ? res[k] : res[k - 1];
        const uint32_t inc = wave_scan_max_u32(res[MSLOTS - 1]);
        if (lane == WAVE - 1) s_wmax[wv] = inc;
        uint32_t pre = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x138, 0xF, 0xF, false);   // wave_shr:1 (lane 0 reads 0)
        __syncthreads();
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) if (w < wv) pre = s_wmax[w] > pre ? s_wmax[w] : pre;
#pragma unroll
        for (int k = 0; k < MSLOTS; ++k) res[k] = res[k] > pre ? res[k] : pre;
    } else {
        // ---- a few slots over very many cells (e.g. the light tail of a sorted order): per-slot search of the range.
        //      The range's per-256 entries are staged in LDS (over s_mark) and searched there; the two line counts below
        //      them (per-16 level, cells) are dependent global reads, two slots of the lane in flight at a time.  Targets
        //      are read from and results written to the lane's own s_T entries (rolled loop: the streaming path's registers).
        const int64_t n16 = a.ntiles * (TILE / 16);
        constexpr int WIDE_ENTRIES = MJB / 2;                             // u64 entries that fit s_mark
        uint64_t* const s_w = reinterpret_cast<uint64_t*>(s_mark);
        const int64_t nent = g_hi - g_lo;                                 // entries [g_lo, g_hi) decide the group
        const bool staged = nent <= WIDE_ENTRIES;                         // block-uniform
        __syncthreads();                                                  // s_mark[0] above
        if (staged) for (int64_t i = tid; i < nent; i += MBLOCK) s_w[i] = a.w.t256[g_lo + i];
        __syncthreads();
        auto group_of = [&](uint64_t Tk) {
            if (staged) {
                int lo = 0, len = (int)nent;
                while (len > 0) { const int half = len >> 1; if (s_w[lo + half] <= Tk) { lo += half + 1; len -= half + 1; } else len = half; }
                return g_lo + lo;
            }
            int64_t lo = g_lo, hi = g_hi;
            while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (a.w.t256[mid] <= Tk) lo = mid + 1; else hi = mid; }
            return lo;
        };
#pragma unroll 1
        for (int k = 0; k < MSLOTS; k += 2) {
            const uint64_t Ta = s_T[MSLOTS * tid + k], Tb = s_T[MSLOTS * tid + k + 1];
            const int64_t ga = group_of(Ta), gb = group_of(Tb);
            const int ca = count_le_line(a.w.t16 + ga * 16, Ta), cb = count_le_line(a.w.t16 + gb * 16, Tb);
            int64_t sa = ga * 16 + ca, sb2 = gb * 16 + cb;
            sa = sa < n16 ? sa : n16 - 1;
            sb2 = sb2 < n16 ? sb2 : n16 - 1;
            const int da = count_le_line(a.w.cdf + sa * 16, Ta), db = count_le_line(a.w.cdf + sb2 * 16, Tb);
            s_T[MSLOTS * tid + k] = (uint64_t)(sa * 16 + da);
            s_T[MSLOTS * tid + k + 1] = (uint64_t)(sb2 * 16 + db);
        }
#pragma unroll
        for (int k = 0; k < MSLOTS; ++k) res[k] = (uint32_t)s_T[MSLOTS * tid + k];
    }
    // ---- parents[j] = order[i_old]   (resample.jl:168)
    const int64_t jb = j0 + MSLOTS * tid;
    int32_t out[MSLOTS];
    const uint32_t last = (uint32_t)(a.n_cells - 1);
#pragma unroll
    for (int k = 0; k < MSLOTS; ++k) {
        uint32_t idx = res[k] < last ? res[k] : last;
        if (a.order && jb + k < n_out) idx = (uint32_t)a.order[idx];
        out[k] = (int32_t)idx;
    }
    int32_t* dst = a.anc + jb;
    if (jb + MSLOTS <= n_out && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
#pragma unroll
        for (int k = 0; k < MSLOTS; k += 4) *reinterpret_cast<int4*>(dst + k) = make_int4(out[k], out[k + 1], out[k + 2], out[k + 3]);
    } else {
#pragma unroll
        for (int k = 0; k < MSLOTS; ++k) if (jb + k < n_out) dst[k] = out[k];
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
// ----------------------------------------------------------------------------- K10: stable descending sort
// order = sortperm(log_priorities, rev=true) (resample.jl:156-157; stable: ties keep ascending index order).
// Least-significant-digit radix sort of the order-preserving 64-bit key with the particle index as payload: 8 passes of
// 8 bits, each ONE kernel ("onesweep"): a workgroup of 1024 threads takes the next tile by ticket, ranks its 4096 keys by digit (wave-level
// match + per-wave counters in LDS), learns the global offset of each of its 256 digit bins by a decoupled look-back over
// the earlier tiles' descriptors ({valid | count} in one 8-byte word, relaxed agent-scope atomics as in k_scan), reorders
// the tile in LDS so that every digit's run leaves as contiguous stores, and scatters.  The histograms of all eight digits
// come from the key-generation pass.  24 N bytes of traffic per pass.
constexpr int SORT_TILE = 4096;                    // keys per workgroup
constexpr int SORT_BLOCK = 1024, SORT_WAVES = SORT_BLOCK / WAVE;   // many waves with few keys each: the chain ticket -> load ->
constexpr int SORT_ITEMS = SORT_TILE / SORT_BLOCK;                 // rank -> look-back -> scatter is latency, not bandwidth
constexpr int SORT_PASSES = 8, SORT_BINS = 256;
constexpr uint64_t SORT_VALID = 1ull << 62, SORT_VAL = (1ull << 62) - 1;   // descriptor = {valid | count}
// workspace: [8][256] u32 histograms | [8] u32 tile tickets | pad | per pass: [ntiles | ntiles/16 | ntiles/256][256] u64 descriptors
__host__ __device__ __forceinline__ size_t sort_ws_desc_offset() { return (size_t)(SORT_PASSES * SORT_BINS + 64) * sizeof(uint32_t); }
__host__ inline size_t sort_ws_bytes(int64_t n)
{
    const int64_t nt = (n + SORT_TILE - 1) / SORT_TILE;
    return sort_ws_desc_offset() + (size_t)SORT_PASSES * (nt + (nt + 15) / 16 + (nt + 255) / 256) * SORT_BINS * sizeof(uint64_t);
}
// keys of the log-priorities + the histograms of all eight digits in one pass over the weights
__global__ __launch_bounds__(BLOCK) void k_sort_keys_hist(PrioView pv, int64_t n, uint64_t* __restrict__ keys, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t s_h[SORT_PASSES][SORT_BINS];
    for (int i = threadIdx.x; i < SORT_PASSES * SORT_BINS; i += BLOCK) (&s_h[0][0])[i] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        const uint64_t k = sort_key_desc(pv.at(i));
        keys[i] = k;
#pragma unroll
        for (int p = 0; p < SORT_PASSES; ++p) atomicAdd(&s_h[p][(k >> (8 * p)) & 0xff], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SORT_PASSES * SORT_BINS; i += BLOCK) { const uint32_t c = (&s_h[0][0])[i]; if (c) atomicAdd(hist + i, c); }
}

// one digit pass.  vals_in == nullptr: the payload is the element's index (first pass).
__global__ __launch_bounds__(SORT_BLOCK) void k_sort_pass(const uint64_t* __restrict__ keys_in, const int32_t* __restrict__ vals_in,
                                                     uint64_t* __restrict__ keys_out, int32_t* __restrict__ vals_out, int64_t n,
                                                     int pass, const uint32_t* __restrict__ hist, uint32_t* __restrict__ ticket,
                                                     uint64_t* __restrict__ desc, int32_t* __restrict__ timeout)
{
    __shared__ uint32_t s_cnt[SORT_WAVES][SORT_BINS];      // per-wave digit counts, then exclusive offsets of the wave inside the tile's bin
    __shared__ uint32_t s_lstart[SORT_BINS];           // first position of the bin in the tile's sorted order
    __shared__ int64_t s_gbase[SORT_BINS];             // global position of the bin's first element of this tile, minus s_lstart
    __shared__ uint32_t s_scan[SORT_WAVES];
    __shared__ uint64_t s_keys[SORT_TILE];
    __shared__ int32_t s_vals[SORT_TILE];
    __shared__ uint32_t s_tile;
    const int tid = (int)threadIdx.x, lane = lane_id(), wv = wave_id();
    const int shift = 8 * pass;
    if (tid == 0) s_tile = atomicAdd(ticket + pass, 1u);
    for (int i = tid; i < SORT_WAVES * SORT_BINS; i += SORT_BLOCK) (&s_cnt[0][0])[i] = 0;
    // exclusive scan of the digit's histogram: where each bin starts in the output
    const bool binthr = tid < SORT_BINS;               // the first four waves double as "thread = bin"
    const uint32_t hv = binthr ? hist[pass * SORT_BINS + tid] : 0u;
    uint32_t hinc = hv;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) { const uint32_t o = __shfl_up(hinc, d, WAVE); if (lane >= d) hinc += o; }
    if (binthr && lane == WAVE - 1) s_scan[wv] = hinc;
    __syncthreads();
    uint32_t hbase = hinc - hv;
#pragma unroll
    for (int w = 0; w < SORT_BINS / WAVE; ++w) if (w < wv) hbase += s_scan[w];
    const int64_t tile = s_tile;
    const int64_t t0 = tile * SORT_TILE;
    // ---- load (wave-striped: element = t0 + wave * 1024 + item * 64 + lane), rank inside the wave by digit
    uint64_t key[SORT_ITEMS]; int32_t val[SORT_ITEMS]; uint32_t rank[SORT_ITEMS];
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; ++it) {
        const int64_t i = t0 + wv * (WAVE * SORT_ITEMS) + it * WAVE + lane;
        key[it] = i < n ? keys_in[i] : ~0ull;
        val[it] = i < n ? (vals_in ? vals_in[i] : (int32_t)i) : 0;
    }
    const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; ++it) {
        const int64_t i = t0 + wv * (WAVE * SORT_ITEMS) + it * WAVE + lane;
        const bool valid = i < n;
        const uint32_t d = (uint32_t)(key[it] >> shift) & 0xffu;
        uint64_t peers = __ballot(valid);                 // lanes with the same digit (invalid lanes take no part)
#pragma unroll
        for (int b = 0; b < 8; ++b) { const uint64_t m = __ballot((d >> b) & 1u); peers &= ((d >> b) & 1u) ? m : ~m; }
        const uint32_t prev = s_cnt[wv][d];
        rank[it] = prev + (uint32_t)__popcll(peers & lt_mask);
        __builtin_amdgcn_wave_barrier();
        if (valid && (peers & lt_mask) == 0) s_cnt[wv][d] = prev + (uint32_t)__popcll(peers);     // the group's lowest lane
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // ---- per bin (thread = bin): offsets of the waves inside the bin, the tile's count, the bin's start inside the tile
    uint32_t tcnt = 0;
    if (binthr) {
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) { const uint32_t c = s_cnt[w][tid]; s_cnt[w][tid] = tcnt; tcnt += c; }
    }
    uint32_t linc = tcnt;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) { const uint32_t o = __shfl_up(linc, d, WAVE); if (lane >= d) linc += o; }
    __syncthreads();                                        // s_scan reuse
    if (binthr && lane == WAVE - 1) s_scan[wv] = linc;
    // the tile's aggregate is published NOW; the tile is then reordered in LDS (local information only) while the other
    // tiles publish theirs, and only then are the earlier tiles' words read
    {
        const size_t nt_ = gridDim.x, ng_ = (nt_ + 15) / 16, nsg_ = (nt_ + 255) / 256;
        if (binthr) __hip_atomic_store(desc + ((size_t)pass * (nt_ + ng_ + nsg_) + (size_t)tile) * SORT_BINS + tid, SORT_VALID | tcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    uint32_t lstart = linc - tcnt;
#pragma unroll
    for (int w = 0; w < SORT_BINS / WAVE; ++w) if (w < wv) lstart += s_scan[w];
    if (binthr) s_lstart[tid] = lstart;
    __syncthreads();
    // ---- reorder inside the tile: afterwards every digit's run leaves as contiguous stores
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; ++it) {
        const int64_t i = t0 + wv * (WAVE * SORT_ITEMS) + it * WAVE + lane;
        if (i < n) {
            const uint32_t d = (uint32_t)(key[it] >> shift) & 0xffu;
            const uint32_t lp = s_lstart[d] + s_cnt[wv][d] + rank[it];
            s_keys[lp] = key[it]; s_vals[lp] = val[it];
        }
    }
    uint64_t excl = 0;
    // ---- global number of this bin's elements in earlier tiles.  Tiles are taken by ticket, so every earlier tile is running
    //      or done, and the tiles of a launch mostly start TOGETHER: a one-word-per-hop look-back would crawl through a chain
    //      of tiles that are all still looking back themselves.  Three planes of {valid | count} words instead, every read
    //      independent of the others:  AGG[tile] (published right after ranking),  GT[group of 16 tiles] (the group's total,
    //      published by the group's last tile from the 16 aggregates),  PRE[super-group of 256 tiles] (inclusive prefix of
    //      everything up to the super-group's end, published by its last tile).
    //      excl(tile) = PRE[super-group before] + sum of GT of the earlier groups of this super-group + sum of AGG of the earlier
    //      tiles of this group: at most 1 + 15 + 15 words, two or three round trips whatever the number of tiles.
    if (binthr) {
        const size_t nt = gridDim.x, ng = (nt + 15) / 16, nsg = (nt + 255) / 256;
        uint64_t* const agg = desc + ((size_t)pass * (nt + ng + nsg)) * SORT_BINS + tid;      // this pass, this bin
        uint64_t* const gt = agg + nt * SORT_BINS;
        uint64_t* const pre = gt + ng * SORT_BINS;
        auto wait_word = [&](const uint64_t* p) {
            uint64_t v = __hip_atomic_load(const_cast<uint64_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (!(v & SORT_VALID)) {
                __builtin_amdgcn_s_sleep(1);
                v = __hip_atomic_load(const_cast<uint64_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (++spins > SPIN_LIMIT) { *timeout = 1; break; }
            }
            return v & SORT_VAL;
        };
        // sum of words p[0], p[stride], ..., cnt <= 15 of them: all loads first, then the (rare) waits
        auto sum_words = [&](const uint64_t* p, int cnt) {
            uint64_t v[15], acc = 0;
#pragma unroll
            for (int e = 0; e < 15; ++e) v[e] = e < cnt ? __hip_atomic_load(const_cast<uint64_t*>(p + (size_t)e * SORT_BINS), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : SORT_VALID;
#pragma unroll
            for (int e = 0; e < 15; ++e) acc += (v[e] & SORT_VALID) ? (v[e] & SORT_VAL) : wait_word(p + (size_t)e * SORT_BINS);
            return acc;
        };
        const int64_t k = tile & 15, g = tile >> 4, gk = g & 15, sg = tile >> 8;
        const uint64_t in_group = sum_words(agg + (size_t)(tile - k) * SORT_BINS, (int)k);
        if (k == 15) __hip_atomic_store(gt + (size_t)g * SORT_BINS, SORT_VALID | (in_group + tcnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint64_t e_ = in_group + sum_words(gt + (size_t)(g - gk) * SORT_BINS, (int)gk);
        if (sg > 0) e_ += wait_word(pre + (size_t)(sg - 1) * SORT_BINS);
        if ((tile & 255) == 255) __hip_atomic_store(pre + (size_t)sg * SORT_BINS, SORT_VALID | (e_ + tcnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        excl = e_;
    }
    if (binthr) s_gbase[tid] = (int64_t)hbase + (int64_t)excl - (int64_t)lstart;
    __syncthreads();
    const int64_t nvalid = n - t0 < SORT_TILE ? n - t0 : SORT_TILE;
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; ++k) {
        const int lp = k * SORT_BLOCK + tid;
        if (lp < nvalid) {
            const uint64_t kk = s_keys[lp];
            const int64_t g = s_gbase[(uint32_t)(kk >> shift) & 0xffu] + lp;
            keys_out[g] = kk; vals_out[g] = s_vals[lp];
        }
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
// {Ql0..3} -> out5[1..4]: limb sums of sum q^2 folded over the scan blocks (exact integers); out5[0] = S_local is written by the scan
__global__ void k_export_q(const uint64_t* __restrict__ blockQ, int nblk, int64_t* out5)
{
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
}
// global S (and residual shift) into the device scalar block from the gathered shard totals
// tot_all = the gathered {S_local, Ql0..3} of all G shards -> the global S
__global__ void k_set_global(const int64_t* __restrict__ tot_all, int G, WSum* ws)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        uint64_t S = 0;
        for (int g = 0; g < G; ++g) S += (uint64_t)tot_all[5 * g];
        ws->S = S;
    }
}
__global__ void k_export_residual(const Scalars* sc, int64_t* out2)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) { out2[0] = (int64_t)sc->Ctot; out2[1] = (int64_t)sc->Rs; }
}

// ---- sharded resampling, the PUSH exchange.  RNG counters are keyed by the GLOBAL slot id, so every
// shard can evaluate the target of EVERY output slot itself (Philox is pure ALU work): the owner of a target finds
// out on its own which slots draw from it, looks the ancestors up and pushes [row | slot | ancestor id] to the shard
// that holds the slot.  No request message exists.  Pass 1 walks all slots in chunks that never straddle a shard
// boundary, compacts each chunk's hits in LDS and appends them to the staging list of the slot's shard (one global
// atomic per chunk; the order of chunks inside a list is arbitrary, every entry names its slot); it also counts what
// this shard will receive from whom.  Pass 2 looks the staged hits up (same core as k_search) and packs the rows.
constexpr int PUSH_CHUNK = 2048;                  // output slots per chunk
struct PushArgs {
    uint64_t seed; uint32_t epoch;
    int64_t n_global;
    int G, me;
    const int64_t* tot_all;                       // [G][5] gathered {S_local, Ql0..3}
    const int64_t* cr_all;                        // [G][2] gathered residual {Ctot_local, Rs_local}, or nullptr
    int64_t bounds[MAX_SHARDS + 1];               // first global slot of every shard
    int64_t chunk0[MAX_SHARDS + 1];               // first chunk of every shard's slots
    int64_t nchunks;
    ulonglong2* stage;                            // [n_global]: hits for shard g's slots at stage + bounds[g]: {T_local | space << 62, slot inside g}
    int64_t* counts;                              // [2G]: entries sent to each shard | received from each shard
    int64_t* host_counts;                         // pinned host mirror [2 * MAX_SHARDS + 1]: k_push publishes the counts + a ticket
    int64_t ticket;
};
struct PushTables {                               // LDS copy of the per-shard tables
    int64_t w_incl[MAX_SHARDS], c_incl[MAX_SHARDS], bounds[MAX_SHARDS + 1], chunk0[MAX_SHARDS + 1];
};
__device__ __forceinline__ void push_tables(const PushArgs& a, PushTables& t)
{
    // inclusive shard totals of the sampled space (weights, or residual weights) and of the residual copy counts: the
    // first wave, one shard per lane (G <= MAX_SHARDS = 64)
    static_assert(MAX_SHARDS <= WAVE, "one lane per shard");
    if (threadIdx.x < WAVE) {
        const int g = (int)threadIdx.x;
        uint64_t w = g < a.G ? (uint64_t)(a.cr_all ? a.cr_all[2 * g + 1] : a.tot_all[5 * g]) : 0;
        uint64_t c = g < a.G && a.cr_all ? (uint64_t)a.cr_all[2 * g] : 0;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) {
            const uint64_t ow = shfl_up_u64(w, d), oc = shfl_up_u64(c, d);
            if (g >= d) { w += ow; c += oc; }
        }
        if (g < a.G) { t.w_incl[g] = (int64_t)w; t.c_incl[g] = (int64_t)c; }
    }
    for (int g = threadIdx.x; g <= a.G; g += blockDim.x) { t.bounds[g] = a.bounds[g]; t.chunk0[g] = a.chunk0[g]; }
    __syncthreads();
}
// the slots [j0, j1) of chunk c and the shard g that holds them
__device__ __forceinline__ int push_chunk(const PushArgs& a, const PushTables& t, int64_t c, int64_t& j0, int64_t& j1)
{
    int g = 0;
    while (g < a.G - 1 && c >= t.chunk0[g + 1]) ++g;
    j0 = t.bounds[g] + (c - t.chunk0[g]) * PUSH_CHUNK;
    j1 = j0 + PUSH_CHUNK < t.bounds[g + 1] ? j0 + PUSH_CHUNK : t.bounds[g + 1];
    return g;
}
struct PushScal { uint64_t Sw, Ctot; };
template <int METHOD>
__device__ __forceinline__ PushScal push_scalars(const PushArgs& a, const PushTables& t)
{
    PushScal s;
    s.Sw = (uint64_t)t.w_incl[a.G - 1];           // total of the sampled space: weights, or residual weights
    s.Ctot = METHOD == 1 ? (uint64_t)t.c_incl[a.G - 1] : 0;
    return s;
}
// target of global slot jg, same arithmetic as k_search; space 1 = the residual copy-count CDF
template <int METHOD>
__device__ __forceinline__ void push_target(const PushArgs& a, const PushScal& s, uint64_t jg, uint64_t U, uint64_t& T, int& space)
{
    static_assert(METHOD == 0 || METHOD == 1, "stratified shards take k_strat_plan + k_search_strat");
    space = 0;
    if (METHOD == 0) T = mulhi64(U, s.Sw);
    else if (jg < s.Ctot) { space = 1; T = jg; } else T = mulhi64(U, s.Sw);
}
// owner = first shard whose inclusive total exceeds T; T_local in the owner's coordinates
__device__ __forceinline__ int push_owner(const PushTables& t, int G, int space, uint64_t T, uint64_t& T_local)
{
    const int64_t* incl = space ? t.c_incl : t.w_incl;
    int h = 0;
    if (G <= 8) {                                 // one node: branch-free count, the table reads are LDS broadcasts
#pragma unroll
        for (int g = 0; g < 7; ++g) h += (g < G - 1 && (uint64_t)incl[g] <= T) ? 1 : 0;
    } else {
        while (h < G - 1 && (uint64_t)incl[h] <= T) ++h;
    }
    T_local = T - (h ? (uint64_t)incl[h - 1] : 0);
    return h;
}
// pass 1: stage the hits (slots whose target this shard owns), count them per destination, and count who owns the
// targets of this shard's own slots
constexpr int PUSH_SCAN_BLOCK = 512;
template <int METHOD>
__global__ __launch_bounds__(PUSH_SCAN_BLOCK) void k_push_scan(PushArgs a)
{
    constexpr int R = PUSH_CHUNK / PUSH_SCAN_BLOCK, NW = PUSH_SCAN_BLOCK / WAVE;
    __shared__ PushTables t;
    __shared__ unsigned int s_recv[MAX_SHARDS];
    __shared__ unsigned int s_wtot[NW];
    __shared__ unsigned long long s_base;
    if (threadIdx.x < MAX_SHARDS) s_recv[threadIdx.x] = 0;
    push_tables(a, t);
    const PushScal sc = push_scalars<METHOD>(a, t);
    const int lane = lane_id(), wv = (int)threadIdx.x / WAVE;
    unsigned recv_cnt = 0;                        // lane h counts the wave's own-slot targets owned by shard h
    // this shard's range of the sampled space(s); the last shard also takes a target at the very end (push_owner's clamp)
    const uint64_t w_lo = a.me ? (uint64_t)t.w_incl[a.me - 1] : 0, w_hi = a.me == a.G - 1 ? ~0ull : (uint64_t)t.w_incl[a.me];
    const uint64_t c_lo = a.me ? (uint64_t)t.c_incl[a.me - 1] : 0, c_hi = a.me == a.G - 1 ? ~0ull : (uint64_t)t.c_incl[a.me];
    for (int64_t c = blockIdx.x; c < a.nchunks; c += gridDim.x) {
        int64_t j0, j1;
        const int g = push_chunk(a, t, c, j0, j1);
        if (METHOD == 1 && (uint64_t)j1 <= sc.Ctot) {                                 // block-uniform
            // the chunk lies in the residual resampler's deterministic head (resample.jl:96-106): slot j is the j-th copy, its
            // target is j itself in the copy-count space -- no uniform to draw, and the hits are ONE range of slots
            const uint64_t h0 = (uint64_t)j0 > c_lo ? (uint64_t)j0 : c_lo, h1 = (uint64_t)j1 < c_hi ? (uint64_t)j1 : c_hi;
            const unsigned total = h1 > h0 ? (unsigned)(h1 - h0) : 0u;
            if (g == a.me && wv == 0 && lane < a.G) {                                 // who serves this shard's own slots
                const uint64_t q0 = lane ? (uint64_t)t.c_incl[lane - 1] : 0, q1 = lane == a.G - 1 ? ~0ull : (uint64_t)t.c_incl[lane];
                const uint64_t r0 = (uint64_t)j0 > q0 ? (uint64_t)j0 : q0, r1 = (uint64_t)j1 < q1 ? (uint64_t)j1 : q1;
                if (r1 > r0) recv_cnt += (unsigned)(r1 - r0);
            }
            if (total) {                                                              // block-uniform
                if (threadIdx.x == 0) s_base = atomicAdd(reinterpret_cast<unsigned long long*>(a.counts + g), (unsigned long long)total);
                __syncthreads();
                ulonglong2* dst = a.stage + t.bounds[g] + s_base;
                for (unsigned k = threadIdx.x; k < total; k += PUSH_SCAN_BLOCK)
                    dst[k] = make_ulonglong2((h0 + k - c_lo) | (1ull << 62), h0 + k - (uint64_t)t.bounds[g]);
                __syncthreads();                                                      // s_base
            }
            continue;
        }
        uint64_t Tl[R];
        unsigned hits = 0;                                                            // bit r: round r is a hit
        // the lane's R consecutive slots: one Philox block per aligned slot pair (resample_u64), one more when the run starts odd
        uint64_t U[R];
        {
            const uint32_t s0 = (uint32_t)(j0 + (int64_t)threadIdx.x * R), sb = s0 >> 1;
            if (!(s0 & 1u)) {
#pragma unroll
                for (int q = 0; q < R / 2; ++q) {
                    const Philox b = rng(a.seed, sb + (uint32_t)q, 0, a.epoch, TAG_RESAMPLE);
                    U[2 * q] = u64(b.w0, b.w1); U[2 * q + 1] = u64(b.w2, b.w3);
                }
            } else {
#pragma unroll
                for (int q = 0; q <= R / 2; ++q) {
                    const Philox b = rng(a.seed, sb + (uint32_t)q, 0, a.epoch, TAG_RESAMPLE);
                    if (q > 0) U[2 * q - 1] = u64(b.w0, b.w1);
                    if (q < R / 2) U[2 * q] = u64(b.w2, b.w3);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t j = j0 + (int64_t)threadIdx.x * R + r;       // R consecutive slots per lane: staged in slot order
            uint64_t T = 0; int space = 0, h = -1;
            Tl[r] = 0;
            if (j < j1) {
                push_target<METHOD>(a, sc, (uint64_t)j, U[r], T, space);
                if (g == a.me) h = push_owner(t, a.G, space, T, Tl[r]);              // own slots: who serves them (receive counts)
                else {                                                                // other shards' slots: only "is it mine?"
                    const uint64_t lo = space ? c_lo : w_lo, hi = space ? c_hi : w_hi;
                    h = (T >= lo && T < hi) ? a.me : -1;
                    Tl[r] = T - lo;
                }
                Tl[r] |= (uint64_t)space << 62;
            }
            if (g == a.me) {                                                          // block-uniform
                for (int q = 0; q < a.G; ++q) {
                    const unsigned n = (unsigned)__popcll(__ballot(h == q));
                    if (lane == q) recv_cnt += n;
                }
            }
            hits |= (h == a.me ? 1u : 0u) << r;
        }
        // exclusive position of this lane's hits inside the chunk; ONE global atomic per chunk reserves the chunk's range
        const unsigned cnt = (unsigned)__popc(hits);
        unsigned incl = cnt;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) { const unsigned o = __shfl_up(incl, d, WAVE); if (lane >= d) incl += o; }
        if (lane == WAVE - 1) s_wtot[wv] = incl;
        __syncthreads();
        unsigned before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) { const unsigned v = s_wtot[w]; before += w < wv ? v : 0; total += v; }
        if (threadIdx.x == 0 && total)
            s_base = atomicAdd(reinterpret_cast<unsigned long long*>(a.counts + g), (unsigned long long)total);
        __syncthreads();
        if (cnt) {
            ulonglong2* dst = a.stage + t.bounds[g] + s_base + before + (incl - cnt);
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (hits >> r & 1u)
                    *dst++ = make_ulonglong2(Tl[r], (uint64_t)(j0 + (int64_t)threadIdx.x * R + r - t.bounds[g]));
        }
    }
    if (recv_cnt) atomicAdd(&s_recv[lane], recv_cnt);
    __syncthreads();
    if (threadIdx.x < a.G && s_recv[threadIdx.x])
        atomicAdd(reinterpret_cast<unsigned long long*>(a.counts + a.G + threadIdx.x), (unsigned long long)s_recv[threadIdx.x]);
}
// pass 2: every staged hit is looked up in this shard's CDF (same core as k_search) and pushed with its row:
// packed_out[e] = [row (W doubles) | (slot inside its shard) << 32 | global ancestor id], grouped by destination shard
template <int METHOD, int W>
__global__ __launch_bounds__(SBLOCK, SEARCH_WAVES_PER_SIMD) void k_push(PushArgs a, CdfLevels lw_, CdfLevels lc_, int64_t n, int64_t ntiles,
                                                                         int64_t gid0, const double* __restrict__ rows,
                                                                         int64_t capacity, double* __restrict__ packed_out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const SearchTop st = search_prologue(lw_, lc_, METHOD == 1, ntiles, reinterpret_cast<uint64_t*>(smem));
    __shared__ ulonglong2 s_coop[2 * SBLOCK];
    __shared__ int64_t s_off[MAX_SHARDS + 1];     // first entry of every destination in the send buffer
    __shared__ int64_t s_bnd[MAX_SHARDS + 1];
    ulonglong2* const lds_wave = s_coop + wave_id() * (2 * WAVE);
    if (threadIdx.x == 0) {
        int64_t o = 0;
        for (int g = 0; g < a.G; ++g) { s_off[g] = o; o += a.counts[g]; s_bnd[g] = a.bounds[g]; }
        s_off[a.G] = o;
        // the host needs the counts for the all-to-all split sizes: publish them to pinned host memory NOW, so the host
        // reads them while this kernel is still looking ancestors up (system-scope stores, ticket last)
        if (blockIdx.x == 0 && a.host_counts) {
            for (int g = 0; g < 2 * a.G; ++g)
                __hip_atomic_store(a.host_counts + g, a.counts[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(a.host_counts + 2 * MAX_SHARDS, a.ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __syncthreads();
    const int64_t total = s_off[a.G] < capacity ? s_off[a.G] : capacity;
    for (int64_t base = (int64_t)blockIdx.x * 2 * SBLOCK; base < total; base += (int64_t)gridDim.x * 2 * SBLOCK) {
        int64_t e[2]; bool act[2]; uint64_t T[2], slot[2]; const uint64_t* top[2]; const CdfLevels* L[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            e[u] = base + u * SBLOCK + threadIdx.x;
            act[u] = e[u] < total;
            const int64_t ee = act[u] ? e[u] : total - 1;
            int g = 0;
            while (g < a.G - 1 && ee >= s_off[g + 1]) ++g;
            const ulonglong2 q = a.stage[s_bnd[g] + (ee - s_off[g])];
            const bool incounts = (q.x >> 62) != 0;
            T[u] = q.x & DESC_MASK;
            slot[u] = q.y;
            top[u] = incounts ? st.topc : st.topw;
            L[u] = incounts ? &lc_ : &lw_;
        }
        int64_t idx[2];
        search_pair(st, L, top, T, true, lds_wave, n, ntiles, idx);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!act[u]) continue;
            const double* src = rows + idx[u] * W;
            double* dst = packed_out + e[u] * (W + 1);
#pragma unroll
            for (int c = 0; c < W; ++c) dst[c] = src[c];
            dst[W] = u2d((slot[u] << 32) | (uint64_t)(gid0 + idx[u]));
        }
    }
}

// ---- sharded STRATIFIED resampling needs none of the above.  Stratum j is [L(j), L(j+1)) with L ascending in j, and shard h
// owns the targets in [lo_h, lo_(h+1)) (lo = exclusive totals): the slots shard h serves are the contiguous range
// [F[h], F[h+1]), F[h] = first slot whose target is >= lo_h -- the stratum that contains lo_h, or the one after it, decided by
// that one slot's target.  One small workgroup derives F from the gathered totals (every shard the same), the exchange
// counts follow by intersecting slot ranges, and the ancestors of the served slots come from k_search_strat (streaming merge
// over the shard's own CDF) -- no pass over the global slots, no staging list.
__global__ __launch_bounds__(128) void k_strat_plan(PushArgs a, ShardPlan* plan)
{
    __shared__ int64_t F[MAX_SHARDS + 1];
    const int h = (int)threadIdx.x;
    const uint64_t N = (uint64_t)a.n_global;
    uint64_t S = 0, lo = 0, lo_me = 0;
    for (int g = 0; g < a.G; ++g) {
        const uint64_t v = (uint64_t)a.tot_all[5 * g];
        if (g < h) lo += v;
        if (g < a.me) lo_me += v;
        S += v;
    }
    const uint64_t B = S / N, rem = S % N;
    if (h <= a.G) {
        int64_t f;
        if (h == 0) f = 0;
        else if (h == a.G || lo >= S) f = (int64_t)N;
        else {
            auto L = [&](uint64_t j) { return j * B + j * rem / N; };
            uint64_t j = (uint64_t)((double)lo * ((double)N / (double)S));   // the stratum that contains lo: estimate, then exact
            j = j < N ? j : N - 1;
            while (j + 1 < N && L(j + 1) <= lo) ++j;
            while (j > 0 && L(j) > lo) --j;
            const uint64_t L0 = L(j), L1 = L(j + 1);
            const uint64_t T = L0 + mulhi64(resample_u64(a.seed, (uint32_t)j, a.epoch), L1 - L0);             // resample.jl:162
            f = (int64_t)(T >= lo ? j : j + 1);
        }
        F[h] = f;
    }
    __syncthreads();
    if (h < a.G) {
        // sent to shard h: the served slots that lie in h's slot range; received from shard h: h's served slots in this shard's range
        const int64_t s0 = F[a.me] > a.bounds[h] ? F[a.me] : a.bounds[h], s1 = F[a.me + 1] < a.bounds[h + 1] ? F[a.me + 1] : a.bounds[h + 1];
        const int64_t r0 = F[h] > a.bounds[a.me] ? F[h] : a.bounds[a.me], r1 = F[h + 1] < a.bounds[a.me + 1] ? F[h + 1] : a.bounds[a.me + 1];
        const int64_t ns = s1 > s0 ? s1 - s0 : 0, nr = r1 > r0 ? r1 - r0 : 0;
        a.counts[h] = ns; a.counts[a.G + h] = nr;
        if (a.host_counts) {
            __hip_atomic_store(a.host_counts + h, ns, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(a.host_counts + a.G + h, nr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (h == 0) {
        plan->ws.S = S; plan->ws.sB = B; plan->ws.srem = rem; plan->ws.sinv = (double)N / (double)S;
        plan->first = F[a.me]; plan->count = F[a.me + 1] - F[a.me]; plan->t_off = lo_me;
    }
    __syncthreads();
    if (h == 0 && a.host_counts) {
        __threadfence_system();
        __hip_atomic_store(a.host_counts + 2 * MAX_SHARDS, a.ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// packed_out[e] = [row of the ancestor | (slot inside its shard) << 32 | global ancestor id] for the served slots in slot
// order -- which IS grouped by destination shard.  The ancestors ascend: the row reads coalesce.
template <int W>
__global__ __launch_bounds__(BLOCK) void k_push_pack(PushArgs a, const ShardPlan* __restrict__ plan, const int32_t* __restrict__ idx, int64_t gid0,
                                                     const double* __restrict__ rows, int64_t capacity, double* __restrict__ packed_out)
{
    __shared__ int64_t s_bnd[MAX_SHARDS + 1];
    for (int g = threadIdx.x; g <= a.G; g += BLOCK) s_bnd[g] = a.bounds[g];
    __syncthreads();
    const int64_t first = plan->first, total = plan->count < capacity ? plan->count : capacity;
    for (int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (int64_t)gridDim.x * BLOCK) {
        const int64_t jg = first + e;
        int lo = 0, hi = a.G - 1;                                     // the shard that holds slot jg
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_bnd[mid] <= jg) lo = mid; else hi = mid - 1; }
        const int64_t i = idx[e];
        const double2* src = reinterpret_cast<const double2*>(rows + i * W);
        double* dst = packed_out + e * (W + 1);
#pragma unroll
        for (int c = 0; c < W / 2; ++c) { const double2 v = src[c]; dst[2 * c] = v.x; dst[2 * c + 1] = v.y; }
        dst[W] = u2d(((uint64_t)(jg - s_bnd[lo]) << 32) | (uint64_t)(gid0 + i));
    }
}

// k_push for multinomial shards whose CDF carries the offset levels of k_search_multi: the 4-byte key table in LDS, four staged
// hits per lane in flight.  What bounds these kernels is the number of DIVERGENT global loads per entry (each costs the CU's L1
// about four cycles per lane): here the coarse row, the fine run and the particle's row -- the keys never leave LDS.
template <int LOGG, int W>
__global__ __launch_bounds__(SBLOCK, 4) void k_push_multi(PushArgs a, CdfLevels lw_, int64_t n, int64_t ntiles, int64_t gid0,
                                                          const double* __restrict__ rows, int64_t capacity, double* __restrict__ packed_out)
{
    constexpr int NE = GPF_MULTI_NS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int64_t s_off[MAX_SHARDS + 1];     // first entry of every destination in the send buffer
    __shared__ int64_t s_bnd[MAX_SHARDS + 1];
    if (threadIdx.x == 0) {
        int64_t o = 0;
        for (int g = 0; g < a.G; ++g) { s_off[g] = o; o += a.counts[g]; s_bnd[g] = a.bounds[g]; }
        s_off[a.G] = o;
        if (blockIdx.x == 0 && a.host_counts) {   // (as in k_push: the host reads the counts while the look-ups run)
            for (int g = 0; g < 2 * a.G; ++g)
                __hip_atomic_store(a.host_counts + g, a.counts[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(a.host_counts + 2 * MAX_SHARDS, a.ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    const MultiTable tb = multi_table_load<LOGG>(lw_, ntiles, (uint64_t)a.tot_all[5 * a.me], reinterpret_cast<uint32_t*>(smem), [] {});
    const int64_t total = s_off[a.G] < capacity ? s_off[a.G] : capacity;
    for (int64_t base = (int64_t)blockIdx.x * NE * SBLOCK; base < total; base += (int64_t)gridDim.x * NE * SBLOCK) {
        int64_t e[NE]; bool act[NE]; uint64_t T[NE]; uint32_t slot[NE];
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            e[u] = base + u * SBLOCK + threadIdx.x;
            act[u] = e[u] < total;
            const int64_t ee = act[u] ? e[u] : total - 1;
            int g = 0;
            while (g < a.G - 1 && ee >= s_off[g + 1]) ++g;
            const ulonglong2 q = a.stage[s_bnd[g] + (ee - s_off[g])];
            T[u] = q.x & DESC_MASK;
            slot[u] = (uint32_t)q.y;
        }
        uint32_t idx[NE];
        multi_lookup<LOGG, NE>(tb, lw_, n, T, idx);
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            if (!act[u]) continue;
            const double2* src = reinterpret_cast<const double2*>(rows + (int64_t)idx[u] * W);
            double* dst = packed_out + e[u] * (W + 1);
#pragma unroll
            for (int c = 0; c < W / 2; ++c) { const double2 v = src[c]; dst[2 * c] = v.x; dst[2 * c + 1] = v.y; }
            dst[W] = u2d(((uint64_t)slot[u] << 32) | (uint64_t)(gid0 + idx[u]));
        }
    }
}

// install the received population: every entry names its slot
template <int W>
__global__ __launch_bounds__(BLOCK) void k_commit_packed(const double* __restrict__ packed, int64_t m, double* __restrict__ rows_new,
                                                         int32_t* __restrict__ anc, double* __restrict__ lw,
                                                         const double* __restrict__ mf_all, const int64_t* __restrict__ tot_all, int G, int K,
                                                         double logN, Scalars* sc)
{
    // update_lml_est! (resample.jl:178-182) from the gathered global summary: lml += (m + log(S 2^-K)) - log N
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        uint64_t S = 0;
        double mx = -__builtin_huge_val();
        int f = 0;
        for (int g = 0; g < G; ++g) {
            S += (uint64_t)tot_all[5 * g];
            const double v = mf_all[2 * g]; mx = v > mx ? v : mx; f |= (int)mf_all[2 * g + 1];
        }
        if (!(f & FLAG_NAN) && mx == -__builtin_huge_val()) f |= FLAG_ALL_NEGINF;
        sc->lml_est = sc->lml_est + (lse_from(mx, S, K, f) - logN);
    }
    for (int64_t k = (int64_t)blockIdx.x * BLOCK + threadIdx.x; k < m; k += (int64_t)gridDim.x * BLOCK) {
        const double* src = packed + k * (W + 1);
        const uint64_t meta = d2u(src[W]);
        const int64_t j = (int64_t)(meta >> 32);
        double* dst = rows_new + j * W;
#pragma unroll
        for (int c = 0; c < W; ++c) dst[c] = src[c];
        anc[j] = (int32_t)(meta & 0xffffffffull);
        lw[j] = 0.0;                                   // update_weights!, resample.jl:195
    }
}
// ----------------------------------------------------------------------------- resize family (reference src/resize.jl)
// pf_replicate! (resize.jl:236-244): parents = repeat(1:N, inner=k) (contiguous) or repeat(1:N, k) (interleaved);
// pf_dereplicate! :keepfirst (resize.jl:267-280): parents = 1:k:N (contiguous) or 1:N/k (interleaved)
__global__ void k_replicate_anc(int64_t n_new, int64_t n_old, int k, int interleaved, int shrink, int32_t* __restrict__ anc)
{
    for (int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x; j < n_new; j += (int64_t)gridDim.x * BLOCK) {
        int64_t a;
        if (!shrink) a = interleaved ? j % n_old : j / k;
        else         a = interleaved ? j : j * k;
        anc[j] = (int32_t)a;
    }
}
// rows_out[j] = rows_in[anc[j]], lw_out[j] = lw_in[anc[j]]  (traces and weights of the selected parents)
template <int W>
__global__ __launch_bounds__(BLOCK) void k_gather_rows_lw(const int32_t* __restrict__ anc, const double* __restrict__ rows_in,
                                                         const double* __restrict__ lw_in, double* __restrict__ rows_out,
                                                         double* __restrict__ lw_out, int64_t n)
{
    constexpr int C = W / 2;
    const int64_t total = n * C;
    for (int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x; t < total; t += (int64_t)gridDim.x * BLOCK) {
        const int64_t j = t / C;
        const int c = (int)(t - j * C);
        const int64_t a = anc[j];
        reinterpret_cast<double2*>(rows_out)[t] = reinterpret_cast<const double2*>(rows_in)[a * C + c];
        if (c == 0 && lw_out) lw_out[j] = lw_in[a];
    }
}
// ---- pf_optimal_resize! (resize.jl:149-219) in exact fixed point
// find_inv_w_threshold (resize.jl:203-219) on the DESCENDING order: position d holds kappa = q_(d), A = d weights
// before it and B = S - C[d-1] from it on; the reference's first kappa (ascending) with B / kappa + A <= n is the
// LARGEST such d.  The condition is constant over ties, and d < n is necessary.
__global__ __launch_bounds__(BLOCK) void k_opt_threshold(const uint64_t* __restrict__ cdf_desc, const WSum* ws, int64_t n_new,
                                                         int64_t n_old, Scalars* sc)
{
    const uint64_t S = ws->S;
    const int64_t lim = n_new < n_old ? n_new : n_old;
    long long best = -1;
    for (int64_t d = (int64_t)blockIdx.x * BLOCK + threadIdx.x; d < lim; d += (int64_t)gridDim.x * BLOCK) {
        const uint64_t prev = d > 0 ? cdf_desc[d - 1] : 0;
        const uint64_t kappa = cdf_desc[d] - prev;
        if (kappa > 0 && le_mul(S - prev, (uint64_t)(n_new - d), kappa)) best = d;
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { const long long o = __shfl_xor(best, s, WAVE); best = o > best ? o : best; }
    if (lane_id() == 0 && best >= 0) atomicMax(&sc->opt_d, best);
}
// c = (n - A) / B, or float(n) when no kappa qualifies (resize.jl:215,218), as the pair (a, B): c w_i >= 1 <=> a q_i >= B
__global__ void k_opt_params(const uint64_t* __restrict__ cdf_desc, const WSum* ws, int64_t n_new, Scalars* sc)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const long long d = sc->opt_d;
    sc->opt_a = d < 0 ? (uint64_t)n_new : (uint64_t)(n_new - d);
    sc->opt_B = d <= 0 ? ws->S : ws->S - cdf_desc[d - 1];
}
// parents[1:n_keep] .= findall(keep_idxs) (resize.jl:159,180) from the inclusive scan of the keep flags
__global__ __launch_bounds__(BLOCK) void k_opt_keep_scatter(const uint64_t* __restrict__ keepcdf, int64_t n_old, int32_t* __restrict__ anc)
{
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n_old; i += (int64_t)gridDim.x * BLOCK) {
        const uint64_t c = keepcdf[i], p = i > 0 ? keepcdf[i - 1] : 0;
        if (c != p) anc[p] = (int32_t)i;
    }
}
// log_weights (resize.jl:189-195): kept particles keep theirs, the others get logsumexp - log c; all + log(n / n_old)
__global__ __launch_bounds__(BLOCK) void k_opt_weights(double* __restrict__ lw, int64_t n, const Scalars* sc, const WSum* ws, int K,
                                                       double log_n_ratio)
{
    const int64_t n_keep = (int64_t)sc->Ctot;
    const double rw = lse_from(ws->m, sc->opt_B, K, ws->flags) - log_((double)sc->opt_a);
    for (int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x; j < n; j += (int64_t)gridDim.x * BLOCK)
        lw[j] = (j < n_keep ? lw[j] : rw) + log_n_ratio;
}

// pf_dereplicate! method = :sample (resize.jl:281-293): one categorical draw per block of k replicates, with the
// block's softmax in K_b-bit fixed point (same spec as §3.3 of DESIGN.md, N = k); new weight = logsumexp(block) - log k
__global__ void k_dereplicate_sample(const double* __restrict__ lw, int64_t n_new, int64_t n_old, int k, int interleaved,
                                     uint64_t seed, uint32_t epoch, int Kb, double logk, int32_t* __restrict__ anc,
                                     double* __restrict__ lw_out)
{
    const int64_t stride = interleaved ? n_new : 1;
    for (int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x; j < n_new; j += (int64_t)gridDim.x * BLOCK) {
        const int64_t first = interleaved ? j : j * k;
        double m = -__builtin_huge_val();
        bool nan = false;
        for (int e = 0; e < k; ++e) { const double v = lw[first + e * stride]; if (v != v) nan = true; else m = v > m ? v : m; }
        const bool uniform = !nan && m == -__builtin_huge_val();
        uint64_t S = 0;
        for (int e = 0; e < k; ++e) S += uniform ? 1 : exp_fix(lw[first + e * stride] - m, Kb);
        const Philox b = rng(seed, (uint32_t)j, 0, epoch, TAG_RESAMPLE);
        const uint64_t T = mulhi64(u64(b.w0, b.w1), S);
        uint64_t acc = 0;
        int pick = k - 1;
        for (int e = 0; e < k; ++e) {
            acc += uniform ? 1 : exp_fix(lw[first + e * stride] - m, Kb);
            if (acc > T) { pick = e; break; }
        }
        anc[j] = (int32_t)(first + pick * stride);
        const int f = nan ? FLAG_NAN : (uniform ? FLAG_ALL_NEGINF : 0);
        lw_out[j] = lse_from(m, S, Kb, f) - logk;
    }
}

// ----------------------------------------------------------------------------- trajectory store
// Gen traces are persistent: mean(state, 5 => :moving) (reference README.md:97) reads a PAST choice of every
// surviving particle.  The device keeps, per time step, the step's latent columns (final particle order of that
// step) and the composed ancestor map of the resamples that happened during the step.
__global__ void k_hist_snapshot(const double* __restrict__ rows, int W, int d, int64_t n, double* __restrict__ out)
{
    for (int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x; t < n * d; t += (int64_t)gridDim.x * BLOCK) {
        const int64_t i = t / d;
        out[t] = rows[i * W + (t - i * d)];
    }
}
// B[j] = first resample of the step ? anc[j] : B_old[anc[j]]
__global__ void k_hist_compose(const int32_t* __restrict__ anc, const int32_t* __restrict__ b_old, int64_t n, int32_t* __restrict__ b_new)
{
    for (int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x; j < n; j += (int64_t)gridDim.x * BLOCK)
        b_new[j] = b_old ? b_old[anc[j]] : anc[j];
}
// value of column `col` of step `t` along the ancestry of every current particle: follow B_T, B_{T-1}, ..., B_{t+1}
__global__ void k_hist_column(const int32_t* const* __restrict__ maps, int n_maps, const double* __restrict__ hx, int d, int col,
                              int64_t n, double* __restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        int64_t idx = i;
        for (int s = 0; s < n_maps; ++s) { const int32_t* m = maps[s]; if (m) idx = m[idx]; }
        out[i] = hx[idx * d + col];
    }
}
// sum_i w_i f(v_i) over a plain value array (same weights / reduction as k_wsum)
__global__ __launch_bounds__(BLOCK) void k_wsum_values(const double* __restrict__ lw, const WSum* ws, int K,
                                                       const double* __restrict__ values, int64_t n, int pw,
                                                       const double* center, double match, double* __restrict__ partial)
{
    // pw = 1: sum w v;  2: sum w (v - *center)^2;  3: sum w [v == match]  (proportionmap, statistics.jl:91-101)
    const double m = ws->m;
    const double Sd = (double)ws->S;
    const bool uniform = (ws->flags & FLAG_ALL_NEGINF) != 0;
    const double c = center ? *center : 0.0;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        const uint64_t q = uniform ? 1 : exp_fix(lw[i] - m, K);
        double v = values[i];
        if (pw == 2) { v = v - c; v = v * v; }
        if (pw == 3) v = (v == match) ? 1.0 : 0.0;
        acc += ((double)q / Sd) * v;
    }
    acc = wave_sum_f64(acc);
    __shared__ double s[NWAVES];
    if (lane_id() == 0) s[wave_id()] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int w = 0; w < NWAVES; ++w) t += s[w]; partial[blockIdx.x] = t; }
}

// ----------------------------------------------------------------------------- sub-state views (src/view.jl, resample.jl:205-218)
// after resampling a view: every log-weight = logsumexp(view) - log n (the block keeps its total mass, resample.jl:210)
__global__ void k_fill_from(double* __restrict__ lw, int64_t n, const double* __restrict__ value)
{
    const double v = *value;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) lw[i] = v;
}
__global__ void k_view_fill_weights(double* __restrict__ lw, int64_t n, const WSum* ws, int K, double logN)
{
    const double v = lse_from(ws->m, ws->S, K, ws->flags) - logN;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) lw[i] = v;
}
// with priorities: lw = log_ws + (logsumexp(view weights) - logsumexp(log_ws))   (resample.jl:213-216)
__global__ void k_view_apply_post(const Scalars* sc, int K, const double* __restrict__ lws, double* __restrict__ lw, int64_t n)
{
    const double off = lse_from(sc->raw.m, sc->raw.S, K, sc->raw.flags) - lse_from(sc->post.m, sc->post.S, K, sc->post.flags);
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) lw[i] = lws[i] + off;
}

} // namespace gpf
