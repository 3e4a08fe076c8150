// K1/K2: initialise, propagate (with the fused resample gather / sharded commit), rejuvenation moves  (part of gpf_kernels.hpp; include that header, not this file)
#pragma once

namespace gpf {
// ----------------------------------------------------------------------------- K1/K2: init & step
// per-block (max, flags) of the log-weights a kernel has just written: the first pass of safe_softmax
// (utils.jl:119-128) rides on the kernel that produces the weights instead of re-reading them
__device__ __forceinline__ void track_max(double v, double& m, int& f)
{
    if (v != v) f |= FLAG_NAN;
    else { m = v > m ? v : m; if (v == __builtin_huge_val()) f |= FLAG_POSINF; }
}
// the workgroup's (maximum, flags) into its slot of ms.cur (MaxSlots, gpf_k_common.hpp); workgroup 0 clears ms.clear for the next producer
__device__ __forceinline__ void block_max_store(double m, int f, MaxSlots ms)
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
        unsigned long long* slot = ms.cur + (blockIdx.x % MAX_SLOTS) * SLOT_WORDS;
        atomicMax(slot, max_key(m));                              // (no return value used: fire and forget)
        if (f) atomicOr(slot + 1, (unsigned long long)f);
    }
    if (blockIdx.x == 0 && threadIdx.x < MAX_SLOTS) { ms.clear[threadIdx.x * SLOT_WORDS] = 0; ms.clear[threadIdx.x * SLOT_WORDS + 1] = 0; }
}

// stratified_map! (utils.jl:29-55): K strata, block size B = n div K; particle i < K B belongs to stratum i div B
// (:contiguous) or i mod K (:interleaved); the n - K B remaining particles draw a stratum uniformly (sample(strata, R)),
// here from one more Philox block of the particle (block index NBLK, behind the model's own blocks)
// BLK (gpf_initialize_blocks_strata / gpf_update_blocks_strata): every block of blk_size particles is stratified by itself, as the sub-state
// state[b] would be -- the index inside the block and the block's own particle count take the place of (i, n); the RNG counter stays the particle's
template <class Mo, bool BLK = false>
__device__ __forceinline__ int stratum_of(const ModelArgs& a, uint64_t seed, uint32_t epoch, int64_t gid0, int64_t ig, int64_t ng, uint32_t tag)
{
    int64_t i = ig, n = ng;
    if constexpr (BLK) {
        i = ig % a.blk_size;
        const int64_t b0 = ig - i;
        n = b0 + a.blk_size <= ng ? (int64_t)a.blk_size : ng - b0;
    }
    const int64_t K = a.n_strata, B = n / K;
    if (i < K * B) return (int)(a.interleaved ? i % K : i / B);
    const Philox b = rng(seed, particle_gid(a, gid0, ig), (uint32_t)Mo::NBLK, epoch, tag);
    return (int)mulhi64(u64(b.w0, b.w1), (uint64_t)K);
}

// ---- wide rows (W = 8: 64 bytes, the bearings model with x_{t-1}): a lane that reads its own row issues W / 2 16-byte loads 64 bytes
// apart from its neighbours' -- every wave-instruction touches 32-64 different 128-byte lines and uses a quarter of each.  Here the wave
// moves its 64 rows COOPERATIVELY: W / 2 adjacent lanes fetch one row (64 contiguous bytes per row, whole kilobytes per instruction
// when the rows are consecutive) into the wave's LDS strip, then every lane reads its own row from LDS (pitch W / 2 + 1 sixteen-byte
// words: no bank conflicts); the stores go the other way.  All 64 lanes take part (callers loop wave-uniformly).
template <int W> struct RowStage {
    static constexpr int C = W / 2, PITCH = C + 1, RPI = WAVE / C;        // 16-byte parts per row, LDS pitch, rows per wave-instruction
    static constexpr int WORDS = WAVE * PITCH;                            // double2 words per wave
};
// r[0..W) = row `srow` of rows_in for the calling lane (srow >= 0; any valid row for lanes that have no particle)
template <int W>
__device__ __forceinline__ void wave_rows_load(const double* __restrict__ rows_in, int64_t srow, double2* lds_wave, double (&r)[W])
{
    using RS = RowStage<W>;
    const int lane = lane_id(), part = lane % RS::C, sub = lane / RS::C;
    const int lo = (int)(uint32_t)srow, hi = (int)(uint32_t)((uint64_t)srow >> 32);
    double2 v[RS::C];
#pragma unroll
    for (int c = 0; c < RS::C; ++c) {
        const int rr = c * RS::RPI + sub;                                 // the row this lane helps to fetch
        const int64_t src = (int64_t)(((uint64_t)(uint32_t)__shfl(hi, rr, WAVE) << 32) | (uint32_t)__shfl(lo, rr, WAVE));
        v[c] = reinterpret_cast<const double2*>(rows_in + src * W)[part];
    }
#pragma unroll
    for (int c = 0; c < RS::C; ++c) lds_wave[(c * RS::RPI + sub) * RS::PITCH + part] = v[c];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < RS::C; ++q) { const double2 t = lds_wave[lane * RS::PITCH + q]; r[2 * q] = t.x; r[2 * q + 1] = t.y; }
    __builtin_amdgcn_wave_barrier();
}
// rows_out[i0 + lane] = o for the lanes with i0 + lane < n
template <int W>
__device__ __forceinline__ void wave_rows_store(double* __restrict__ rows_out, int64_t i0, int64_t n, double2* lds_wave, const double (&o)[W])
{
    using RS = RowStage<W>;
    const int lane = lane_id(), part = lane % RS::C, sub = lane / RS::C;
#pragma unroll
    for (int q = 0; q < RS::C; ++q) lds_wave[lane * RS::PITCH + q] = make_double2(o[2 * q], o[2 * q + 1]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < RS::C; ++c) {
        const int rr = c * RS::RPI + sub;
        const double2 t = lds_wave[rr * RS::PITCH + part];
        if (i0 + rr < n) reinterpret_cast<double2*>(rows_out + (i0 + rr) * W)[part] = t;
    }
    __builtin_amdgcn_wave_barrier();
}
#ifndef GPF_STAGE_ROWS
#define GPF_STAGE_ROWS 1
#endif

// pf_initialize (initialize.jl:39-41) / pf_update! (update.jl:15-22): one lane per particle, row in,
// row out, lw += log p(y|x).  Counter-based RNG: no RNG state in memory.
// MODE 0: the model's own sampler; 1: native custom proposal; 2: stratified (the discrete latent constrained per stratum);
// 3 (k_init only): stratified with a native proposal for the other choice
template <int M, int MODE = 0, bool BLK = false>
__global__ __launch_bounds__(BLOCK) void k_init(ModelArgs a, uint64_t seed, uint32_t epoch, int64_t gid0,
                                                int64_t n, int W, double* __restrict__ rows,
                                                double* __restrict__ lw, MaxSlots ms)
{
    using Mo = Model<M>;
    double bm = -__builtin_huge_val(); int bf = 0;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        double x[MAX_DIM];
        double ll;
        const double* const ob = obs_of<BLK>(a, i);
        if constexpr (MODE == 1) ll = Mo::propose(a.P, true, nullptr, ob, seed, particle_gid(a, gid0, i), 0, epoch, TAG_INIT, x);
        else if constexpr (MODE == 2) {
            const double v = a.strata[stratum_of<Mo, BLK>(a, seed, epoch, gid0, i, n, TAG_INIT)];
            const double lp = Mo::sample_stratum(a.P, true, nullptr, ob, v, seed, particle_gid(a, gid0, i), 0, epoch, TAG_INIT, x);
            ll = (lp + Mo::loglik(a.P, x, ob)) + a.logK;                      // initialize.jl:103-104
        } else if constexpr (MODE == 3) {                                        // strata + native proposal, initialize.jl:122-126
            const double v = a.strata[stratum_of<Mo, BLK>(a, seed, epoch, gid0, i, n, TAG_INIT)];
            ll = Mo::propose_stratum(a.P, ob, v, x) + a.logK;
        } else {
            Mo::sample(a.P, true, nullptr, ob, seed, particle_gid(a, gid0, i), 0, epoch, TAG_INIT, x);
            ll = Mo::loglik(a.P, x, ob);
        }
        double* r = rows + i * W;
#pragma unroll
        for (int k = 0; k < Mo::D; ++k) r[k] = x[k];
        for (int k = Mo::D; k < W; ++k) r[k] = 0.0;
        lw[i] = ll;
        track_max(ll, bm, bf);
    }
    block_max_store(bm, bf, ms);
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
    int in_mailbox;            // mf_all / tot_all sit in the shard mailbox (written by peers: system-scope loads)
    // GATHER on a shard whose own slots were resolved in place (k_search_own): anc holds GLOBAL ancestor ids (row anc - anc_off) for the
    // slots this shard serves itself and -1 for the slots whose rows arrive packed (skipped here); sc != nullptr: this launch carries
    // the log-ML update from the gathered summaries
    int masked; int64_t anc_off;
    const int64_t* own_range;  // masked == 2 (stratified): the own hits are the slots [own_range[0], own_range[1]) instead of the slots with anc >= 0
    // masked && ring.base: the slots that are not own hits (outside the own range / anc < 0) are not skipped -- their entries [row | global ancestor id |
    // seal] were stored into this rank's slot-addressed receive window by the peers that serve them (gpf_k_common.hpp): ONE launch commits the whole exchange
    RingIn ring;
    // a propagate enqueued SPECULATIVELY behind the ESS reduction (gpf_step_ess, k_sum_host<GATE>): it forms the verdict from the reduction's
    // accumulators itself (gate_verdict) and, if the ESS is below the threshold -- the filter resamples first --, returns before its first store
    GateIn gate;
};
// Minimum waves per SIMD the compiler plans for (its VGPR budget = 512 / that).  4 was measured on the LG-SSM headline (rows of 2 doubles: 28 - 80 VGPRs,
// nothing near the 128 it allows).  Rows of 8 doubles (bearings with x_{t-1}) reach the cap -- k_step<2, 8, true, false> (config 4's plain propagate) and its
// block-wise form spill 12 / 20 bytes per lane to scratch under it (profiles/r06_kernel_resources.txt): GPF_STEP_WAVES_WIDE is their bound.
#ifndef GPF_STEP_WAVES_WIDE
#define GPF_STEP_WAVES_WIDE 4
#endif
template <int W> constexpr int step_min_waves() { return W >= 8 ? GPF_STEP_WAVES_WIDE : 4; }
template <int M, int W, bool KEEP_PREV, bool GATHER, int MODE = 0, bool PACKED = false, bool BLK = false>
__global__ __launch_bounds__(BLOCK, step_min_waves<W>()) void k_step(ModelArgs a, uint64_t seed, uint32_t epoch, int64_t gid0,
                                                int64_t n, const int32_t* __restrict__ anc,
                                                const double* __restrict__ rows_in,
                                                double* __restrict__ rows_out, double* __restrict__ lw,
                                                MaxSlots ms, PackedCommit pc)
{
    using Mo = Model<M>;
    constexpr int D = Mo::D;
    // a speculative propagate behind the ESS reduction (gpf_step_ess): every wave forms the verdict from the reduction's accumulators and
    // returns before its first store if the filter resamples first (kernel-uniform)
#ifndef GPF_NO_GATE
    if constexpr (!GATHER && !PACKED) {
        if (pc.gate.flag) { if (*pc.gate.flag) return; }
        else if (pc.gate.acc && gate_verdict(pc.gate)) return;
    }
#endif
    double bm = -__builtin_huge_val(); int bf = 0;
    if constexpr (PACKED || GATHER) {
        if (pc.sc && pc.mf_all && blockIdx.x == 0 && threadIdx.x == 0) {
            uint64_t S = 0;
            double mx = -__builtin_huge_val();
            int f = 0;
            for (int g = 0; g < pc.G; ++g) {                 // (system-scope loads: the gathered summaries may sit in the shard mailbox)
                S += (uint64_t)ld_gathered(pc.tot_all + 5 * g, pc.in_mailbox != 0);
                const double v = ld_gathered(pc.mf_all + 2 * g, pc.in_mailbox != 0); mx = v > mx ? v : mx; f |= (int)ld_gathered(pc.mf_all + 2 * g + 1, pc.in_mailbox != 0);
            }
            if (!(f & FLAG_NAN) && mx == -__builtin_huge_val()) f |= FLAG_ALL_NEGINF;
            pc.sc->lml_est = pc.sc->lml_est + (lse_from(mx, S, pc.K, f) - pc.logN);
        }
    }
    constexpr bool STAGE = GPF_STAGE_ROWS && W >= 8 && !PACKED;          // wide rows travel through the wave's LDS strip (wave_rows_load / _store)
    __shared__ double2 s_stage[STAGE ? NWAVES * RowStage<W>::WORDS : 1];
    double2* const lds_wave = s_stage + (STAGE ? wave_id() * RowStage<W>::WORDS : 0);
    for (int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x; STAGE ? e - lane_id() < n : e < n; e += (int64_t)gridDim.x * BLOCK) {
        int64_t i = e;                                  // the slot this lane fills
        const bool live = !STAGE || e < n;              // (STAGE: the whole wave stays in the loop; lanes beyond n help to move rows)
        if (!live) i = 0;
        bool act = true;
        double r[W];
        if constexpr (STAGE) {
            int64_t srow = live ? (GATHER ? (int64_t)anc[i] : i) : 0;
            bool skip = false;
            if (GATHER && pc.masked) {
                skip = live && (pc.masked == 2 ? (i < pc.own_range[0] || i >= pc.own_range[1]) : srow < 0);
                srow = skip ? 0 : srow - (live ? pc.anc_off : 0);
            }
            wave_rows_load<W>(rows_in, srow, lds_wave, r);
            act = live && !skip;                        // (a lane without a particle -- or whose row arrives packed -- computes on row 0 and writes nothing)
            if (GATHER && skip && pc.ring.base) {       // ... or arrives in the receive window: the lane's own wait and load
                pc.anc[i] = (int32_t)ring_load<W>(pc.ring, i, r);
                act = true;
            }
        } else
        if constexpr (PACKED) {
            const double* src = pc.packed + e * (W + 1);
#pragma unroll
            for (int c = 0; c < W; ++c) r[c] = src[c];
            const uint64_t meta = d2u(src[W]);
            i = (int64_t)(meta >> 32);
            pc.anc[i] = (int32_t)(meta & 0xffffffffull);
        } else {
        int64_t srow = GATHER ? (int64_t)anc[i] : i;
        bool windowed = false;
        if (GATHER && pc.masked) {                                                   // (kernel-uniform flag; the other slots' rows arrive packed)
            if (pc.masked == 2 ? (i < pc.own_range[0] || i >= pc.own_range[1]) : srow < 0) {
                if (!pc.ring.base) continue;
                windowed = true;                                                     // ... or sit in the receive window
            }
            srow -= pc.anc_off;
        }
        if (GATHER && windowed) pc.anc[i] = (int32_t)ring_load<W>(pc.ring, i, r);
        else {
        const double2* src = reinterpret_cast<const double2*>(rows_in + srow * W);
#pragma unroll
        for (int c = 0; c < (D + 1) / 2; ++c) { const double2 v = src[c]; r[2 * c] = v.x; r[2 * c + 1] = v.y; }
        }
        }
        double xn[MAX_DIM];
        double ll;
        const double* const ob = obs_of<BLK>(a, i);
        if constexpr (MODE == 1) ll = Mo::propose(a.P, false, r, ob, seed, particle_gid(a, gid0, i), 0, epoch, TAG_UPDATE, xn);
        else if constexpr (MODE == 4) {
            // block-wise update with a proposal PER BLOCK (the per-view updates of test/update.jl:179-189 in one launch): the block's flag
            // selects update.jl:79-96 (native proposal) or update.jl:12-25 (default) for its particles -- the same counters either way
            if (a.blk_prop[(uint32_t)i / (uint32_t)a.blk_size]) ll = Mo::propose(a.P, false, r, ob, seed, particle_gid(a, gid0, i), 0, epoch, TAG_UPDATE, xn);
            else { Mo::sample(a.P, false, r, ob, seed, particle_gid(a, gid0, i), 0, epoch, TAG_UPDATE, xn); ll = Mo::loglik(a.P, xn, ob); }
        }
        else if constexpr (MODE == 2) {
            const double v = a.strata[stratum_of<Mo, BLK>(a, seed, epoch, gid0, i, n, TAG_UPDATE)];
            const double lp = Mo::sample_stratum(a.P, false, r, ob, v, seed, particle_gid(a, gid0, i), 0, epoch, TAG_UPDATE, xn);
            ll = (lp + Mo::loglik(a.P, xn, ob)) + a.logK;                     // update.jl:201-206
        } else {
            Mo::sample(a.P, false, r, ob, seed, particle_gid(a, gid0, i), 0, epoch, TAG_UPDATE, xn);
            ll = Mo::loglik(a.P, xn, ob);
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
        if (STAGE && !(GATHER && pc.masked)) wave_rows_store<W>(rows_out, e - lane_id(), n, lds_wave, o);     // (kernel-uniform condition)
        else if (act) {
            double2* dst = reinterpret_cast<double2*>(rows_out + i * W);
#pragma unroll
            for (int c = 0; c < W / 2; ++c) dst[c] = make_double2(o[2 * c], o[2 * c + 1]);
        }
        if (!act) continue;
        const double nl = (GATHER || PACKED) ? ((GATHER && pc.lw_fill) ? *pc.lw_fill + ll : ll)   // after a resample the incoming
                                             : lw[i] + ll;                                    // log-weights are 0 (or one constant)
        lw[i] = nl;
        track_max(nl, bm, bf);
    }
    block_max_store(bm, bf, ms);
}

// K7/K8: pf_move_accept! with Gen.mh on the current step's latent (rejuvenate.jl:40-53) and
// pf_move_reweight! with move_reweight(trace, selection) (rejuvenate.jl:74-90, :125-132)
// GATHER: a pf_resample! left its ancestor vector pending; the move reads row anc[i] (new_traces .= view(traces, parents),
// resample.jl:60, fused) and the incoming log-weights are 0 (resample.jl:195), exactly like k_step<GATHER>.
// n_accept: [gridDim.x] per-workgroup counts of accepted moves (move-accept only)
static __global__ void k_sum_accepts(const unsigned long long* __restrict__ part, int np, unsigned long long* __restrict__ out)
{
    unsigned long long a = 0;
    for (int i = threadIdx.x; i < np; i += BLOCK) a += part[i];
    a = wave_sum_u64(a);
    __shared__ unsigned long long s_a[NWAVES];
    if (lane_id() == 0) s_a[wave_id()] = a;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned long long t = 0; for (int w = 0; w < NWAVES; ++w) t += s_a[w]; *out = t; }
}
// PROP: the model's native move proposal (Model::move_propose): the new latent comes from the proposal, alpha = weight - fwd_score +
// bwd_score.  With REWEIGHT: move_reweight(trace, proposal, proposal_args) (rejuvenate.jl:134-148), every proposal is taken and alpha is the
// relative weight; without: Gen.mh(trace, proposal, proposal_args) under pf_move_accept! (rejuvenate.jl:40-53) -- accept iff log(rand()) <
// alpha, the weights stay (the uniform: block NBLK behind the proposal's, tag MOVE, as in the selection variant).
// BLK: block-wise (ModelArgs::blk_*): the observation of the particle's block; blocks whose mask bit is clear keep their particles
template <int M, int W, bool REWEIGHT, bool GATHER = false, bool PROP = false, bool BLK = false>
__global__ __launch_bounds__(BLOCK) void k_move(ModelArgs a, uint64_t seed, uint32_t epoch, int64_t gid0,
                                                int64_t n, int has_prev, int n_iters, const int32_t* __restrict__ anc,
                                                const double* __restrict__ rows_in,
                                                double* __restrict__ rows_out, double* __restrict__ lw,
                                                unsigned long long* __restrict__ n_accept, MaxSlots ms)
{
    using Mo = Model<M>;
    constexpr int D = Mo::D, NB = Mo::NBLK;
    unsigned long long acc = 0;
    double bm = -__builtin_huge_val(); int bf = 0;
    // (wide rows through the wave's LDS strip as in k_step: measured SLOWER here -- k_move<bearings> 31.8 -> 34.3 us, the kernel is bound by its
    // two likelihoods and the staging adds LDS latency to every iteration; k_step<2,8> gained 28.7 -> 26.5 us.  GPF_STAGE_MOVE=1 builds it.)
#ifndef GPF_STAGE_MOVE
#define GPF_STAGE_MOVE 0
#endif
    constexpr bool STAGE = GPF_STAGE_MOVE && W >= 8;
    __shared__ double2 s_stage[STAGE ? NWAVES * RowStage<W>::WORDS : 1];
    double2* const lds_wave = s_stage + (STAGE ? wave_id() * RowStage<W>::WORDS : 0);
    for (int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x; STAGE ? e - lane_id() < n : e < n; e += (int64_t)gridDim.x * BLOCK) {
        const bool alive = !STAGE || e < n;             // (STAGE: the whole wave stays in the loop; lanes beyond n help to move rows and write nothing)
        const int64_t i = alive ? e : 0;
        double r[W];
        const int64_t srow = GATHER ? (int64_t)anc[i] : i;
        if constexpr (STAGE) wave_rows_load<W>(rows_in, srow, lds_wave, r);
        else {
        const double2* src = reinterpret_cast<const double2*>(rows_in + srow * W);
#pragma unroll
        for (int c = 0; c < W / 2; ++c) { const double2 v = src[c]; r[2 * c] = v.x; r[2 * c + 1] = v.y; }
        }
        double x[MAX_DIM], xs[MAX_DIM];
#pragma unroll
        for (int k = 0; k < D; ++k) x[k] = r[k];
        const double* xp = r + D;                    // x_{t-1} (valid when has_prev)
        const double* const ob = obs_of<BLK>(a, i);
        const bool live = alive && !(BLK && a.blk_mask && !(a.blk_mask[(uint32_t)i / (uint32_t)a.blk_size] & 1));
        const int iters = live ? n_iters : 0;
        double llx = Mo::loglik(a.P, x, ob);
        double wsum = 0.0;
        const uint32_t gid = particle_gid(a, gid0, i);
        for (int it = 0; it < iters; ++it) {
            if constexpr (PROP) {
                if constexpr (Mo::HAS_MOVE_PROPOSAL && REWEIGHT) {
                    const double rw = Mo::move_propose(a.P, a.q, !has_prev, xp, x, ob, seed, gid, (uint32_t)(it * NB), epoch, TAG_REWEIGHT, xs);
                    wsum = wsum + rw;                                          // rejuvenate.jl:86
#pragma unroll
                    for (int k = 0; k < D; ++k) x[k] = xs[k];
                    ++acc;
                } else if constexpr (Mo::HAS_MOVE_PROPOSAL) {
                    const uint32_t blk0 = (uint32_t)(it * (NB + 1));
                    const double alpha = Mo::move_propose(a.P, a.q, !has_prev, xp, x, ob, seed, gid, blk0, epoch, TAG_MOVE, xs);
                    const Philox b = rng(seed, gid, blk0 + NB, epoch, TAG_MOVE);
                    if (log_(u52(b.w0, b.w1)) < alpha) {                       // Gen.mh: accept iff log(rand()) < weight - fwd_score + bwd_score
#pragma unroll
                        for (int k = 0; k < D; ++k) x[k] = xs[k];
                        ++acc;
                    }
                }
            } else if (REWEIGHT) {
                Mo::sample(a.P, !has_prev, xp, ob, seed, gid, (uint32_t)(it * NB), epoch, TAG_REWEIGHT, xs);
                const double lls = Mo::loglik(a.P, xs, ob);
                wsum = wsum + (lls - llx);
#pragma unroll
                for (int k = 0; k < D; ++k) x[k] = xs[k];
                llx = lls;
                ++acc;
            } else {
                const uint32_t blk0 = (uint32_t)(it * (NB + 1));
                Mo::sample(a.P, !has_prev, xp, ob, seed, gid, blk0, epoch, TAG_MOVE, xs);
                const double lls = Mo::loglik(a.P, xs, ob);
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
        if constexpr (STAGE) wave_rows_store<W>(rows_out, e - lane_id(), n, lds_wave, r);
        else {
        double2* dst = reinterpret_cast<double2*>(rows_out + i * W);
#pragma unroll
        for (int c = 0; c < W / 2; ++c) dst[c] = make_double2(r[2 * c], r[2 * c + 1]);
        }
        if (!alive) continue;
        if (REWEIGHT) {
            double nl;
            if (live) { nl = (GATHER ? 0.0 : lw[i]) + wsum; lw[i] = nl; } else nl = lw[i];       // (a masked block: weights untouched)
            track_max(nl, bm, bf);
        }
        else if (GATHER) lw[i] = 0.0;
    }
    // accepted moves: one plain store per workgroup, summed on demand (k_sum_accepts) when the host asks for the count.  (One
    // same-address atomic per WAVE, as in rounds 1-2, serialises at ~10 ns each: 20-40 us of this kernel at 2 workgroups per CU and
    // the reason why fewer workgroups per CU ran faster.)  Move-reweight kernels move every particle: nothing to count.
    if (!REWEIGHT) {
        const unsigned long long t = wave_sum_u64(acc);
        __shared__ unsigned long long s_acc[NWAVES];
        if (lane_id() == 0) s_acc[wave_id()] = t;
        __syncthreads();
        if (threadIdx.x == 0) { unsigned long long b = 0; for (int w = 0; w < NWAVES; ++w) b += s_acc[w]; n_accept[blockIdx.x] = b; }
    }
    if (REWEIGHT) block_max_store(bm, bf, ms);
}

// K7/K8 + K2 in one launch ("lazy move"): pf_rejuvenate!(state, kern, args, n_iters; method) followed by pf_update! -- the README loop's order
// (README.md:69-76) and BASELINE configs 4 and 5.  The move is not enqueued by gpf_rejuvenate; the plain pf_update! that follows runs
// gather (if a resample is pending) -> move (rejuvenate.jl:40-90 with Gen.mh / move_reweight on the current step's latent, under the OLD
// observation `obs_move` and the move's epoch) -> propagate (update.jl:15-22, new observation, the update's epoch) -> one row store.
// Against k_move + k_step: the moved rows are not written and read back (16 W bytes per particle: 128 B at W = 8) and one launch is
// gone.  Any other consumer of the state runs the stand-alone k_move first (libgpf_core.hip finish_move).  Same arithmetic, same order:
// lw = ((GATHER ? 0 : lw) + sum of relative weights) + log-likelihood.
struct ObsVec { double v[MAX_OBS]; };
// On a sharded filter the resample in front of the move left a pending COMMIT instead of a pending gather (gpf_shard_commit): the same three sources
// as k_step's -- GATHER with pc.masked: the shard's own hits through the ancestor array (global ids; the other slots skipped, or read from the receive
// window when pc.ring is set), PACKED: the received exchange entries [row | slot, ancestor] -- and the launch that carries pc.sc also carries the log-ML
// update.  (Before: materialize() ahead of the move -- k_gather_own 27 us + k_commit_packed 4.8 us at n = 10^6, W = 8 on every resampling step of
// BASELINE configs[3]'s loop, where the unsharded filter pays nothing.)
template <int M, int W, bool REWEIGHT, bool GATHER, bool PACKED = false>
__global__ __launch_bounds__(BLOCK) void k_move_step(ModelArgs a, ObsVec obs_move, uint64_t seed, uint32_t epoch_move, uint32_t epoch, int64_t gid0,
                                                     int64_t n, int has_prev, int n_iters, const int32_t* __restrict__ anc,
                                                     const double* __restrict__ rows_in, double* __restrict__ rows_out, double* __restrict__ lw,
                                                     MaxSlots ms, PackedCommit pc)
{
    using Mo = Model<M>;
    constexpr int D = Mo::D, NB = Mo::NBLK;
    constexpr bool STAGE = GPF_STAGE_ROWS && W >= 8 && !PACKED;           // wide rows through the wave's LDS strip (as k_step)
    __shared__ double2 s_stage[STAGE ? NWAVES * RowStage<W>::WORDS : 1];
    double2* const lds_wave = s_stage + (STAGE ? wave_id() * RowStage<W>::WORDS : 0);
    double bm = -__builtin_huge_val(); int bf = 0;
    if constexpr (PACKED || GATHER) {
        if (pc.sc && pc.mf_all && blockIdx.x == 0 && threadIdx.x == 0) {    // update_lml_est! from the gathered summaries (as k_step)
            uint64_t S = 0;
            double mx = -__builtin_huge_val();
            int f = 0;
            for (int g = 0; g < pc.G; ++g) {
                S += (uint64_t)ld_gathered(pc.tot_all + 5 * g, pc.in_mailbox != 0);
                const double v = ld_gathered(pc.mf_all + 2 * g, pc.in_mailbox != 0); mx = v > mx ? v : mx; f |= (int)ld_gathered(pc.mf_all + 2 * g + 1, pc.in_mailbox != 0);
            }
            if (!(f & FLAG_NAN) && mx == -__builtin_huge_val()) f |= FLAG_ALL_NEGINF;
            pc.sc->lml_est = pc.sc->lml_est + (lse_from(mx, S, pc.K, f) - pc.logN);
        }
    }
    const bool masked = GATHER && pc.masked != 0;                         // kernel-uniform
    for (int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x; STAGE ? e - lane_id() < n : e < n; e += (int64_t)gridDim.x * BLOCK) {
        bool alive = !STAGE || e < n;
        int64_t i = alive ? e : 0;
        double r[W];
        if constexpr (PACKED) {
            const double* src = pc.packed + e * (W + 1);
#pragma unroll
            for (int c = 0; c < W; ++c) r[c] = src[c];
            const uint64_t meta = d2u(src[W]);
            i = (int64_t)(meta >> 32);
            pc.anc[i] = (int32_t)(meta & 0xffffffffull);
        } else {
            int64_t srow = GATHER ? (int64_t)anc[i] : i;
            bool windowed = false;
            if (masked) {
                const bool outside = alive && (pc.masked == 2 ? (i < pc.own_range[0] || i >= pc.own_range[1]) : srow < 0);
                windowed = outside && pc.ring.base != nullptr;
                if (outside && !windowed) alive = false;                     // the slot's row arrives packed: the PACKED launch behind this one moves it
                srow = (outside || !alive) ? 0 : srow - pc.anc_off;
            }
            if constexpr (STAGE) wave_rows_load<W>(rows_in, srow, lds_wave, r);
            else {
                const double2* src = reinterpret_cast<const double2*>(rows_in + srow * W);
#pragma unroll
                for (int c = 0; c < W / 2; ++c) { const double2 v = src[c]; r[2 * c] = v.x; r[2 * c + 1] = v.y; }
            }
            if (GATHER && windowed) pc.anc[i] = (int32_t)ring_load<W>(pc.ring, i, r);
            if (!STAGE && !alive) continue;                                  // (the staged form keeps the whole wave in the loop: its loads and stores are wave-collective)
        }
        // ---- the move (k_move's loop), under the observation and the epoch of the pf_rejuvenate! call
        double x[MAX_DIM], xs[MAX_DIM];
#pragma unroll
        for (int k = 0; k < D; ++k) x[k] = r[k];
        const double* xp = r + D;                       // x_{t-1} (valid when has_prev)
        double llx = Mo::loglik(a.P, x, obs_move.v);
        double wsum = 0.0;
        const uint32_t gid = particle_gid(a, gid0, i);
        const int iters = alive ? n_iters : 0;
        for (int it = 0; it < iters; ++it) {
            if (REWEIGHT) {
                Mo::sample(a.P, !has_prev, xp, obs_move.v, seed, gid, (uint32_t)(it * NB), epoch_move, TAG_REWEIGHT, xs);
                const double lls = Mo::loglik(a.P, xs, obs_move.v);
                wsum = wsum + (lls - llx);                                     // rejuvenate.jl:86
#pragma unroll
                for (int k = 0; k < D; ++k) x[k] = xs[k];
                llx = lls;
            } else {
                const uint32_t blk0 = (uint32_t)(it * (NB + 1));
                Mo::sample(a.P, !has_prev, xp, obs_move.v, seed, gid, blk0, epoch_move, TAG_MOVE, xs);
                const double lls = Mo::loglik(a.P, xs, obs_move.v);
                const Philox b = rng(seed, gid, blk0 + NB, epoch_move, TAG_MOVE);
                const double lu = log_(u52(b.w0, b.w1));
                if (lu < lls - llx) {
#pragma unroll
                    for (int k = 0; k < D; ++k) x[k] = xs[k];
                    llx = lls;
                }
            }
        }
        // ---- the propagate (k_step), from the moved latent, under the new observation and the update's epoch
        double xn[MAX_DIM];
        Mo::sample(a.P, false, x, a.obs, seed, gid, 0, epoch, TAG_UPDATE, xn);
        const double ll = Mo::loglik(a.P, xn, a.obs);
        double o[W];
#pragma unroll
        for (int k = 0; k < W; ++k) o[k] = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) { o[k] = xn[k]; o[D + k] = x[k]; }
        if (STAGE && !masked) wave_rows_store<W>(rows_out, e - lane_id(), n, lds_wave, o);      // (kernel-uniform condition)
        else if (alive) {
            double2* dst = reinterpret_cast<double2*>(rows_out + i * W);
#pragma unroll
            for (int c = 0; c < W / 2; ++c) dst[c] = make_double2(o[2 * c], o[2 * c + 1]);
        }
        if (!alive) continue;
        const double base = (GATHER || PACKED) ? 0.0 : lw[i];                 // resample.jl:195 when a gather / a commit was pending
        const double nl = REWEIGHT ? (base + wsum) + ll : base + ll;
        lw[i] = nl;
        track_max(nl, bm, bf);
    }
    block_max_store(bm, bf, ms);
}

} // namespace gpf
