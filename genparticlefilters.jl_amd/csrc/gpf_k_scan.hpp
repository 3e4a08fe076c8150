// K3/K4: maximum + validity flags, fixed-point weights and their CDF levels in one pass, scalar publication  (part of gpf_kernels.hpp; include that header, not this file)
#pragma once

namespace gpf {
// ----------------------------------------------------------------------------- K3: max + flags
// maximum(vs), any(isnan), all(== -Inf) of safe_softmax (utils.jl:119-128) of weights no kernel of ours produced (host-written
// log-weights, priorities): folded into the maximum slots like the producers do (MaxSlots, gpf_k_common.hpp); the consumers
// (k_scan, k_pack_mflags) read the slots themselves (no finalize launch).
static __global__ __launch_bounds__(BLOCK) void k_max_partial(PrioView pv, int64_t n, MaxSlots ms)
{
    double m = -__builtin_huge_val();
    int f = 0;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) track_max(pv.at(i), m, f);
    block_max_store(m, f, ms);
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
constexpr uint64_t DESC_VALID = 1ull << 63, DESC_MASK = (1ull << 63) - 1;   // (a prefix can be 2^62 itself: N a power of two, all weights at the maximum)
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
    uint32_t* k32s;            // [ntiles*64 >> sample] every (1 << sample)-th key, compact: the LDS table of k_search_multi_s (nullptr: not wanted)
    int sample;
};
constexpr int KEY_SHIFT = 30;  // S <= 2^62: (prefix >> 30) fits 32 bits whatever N is, once the one value 2^62 is saturated
// S = 2^62 exactly when N >= 1024 is a power of two and EVERY weight equals the maximum (a second resample right after a
// resample: all log-weights 0).  For the keys and the 16-bit offsets such a prefix counts as 2^62 - 1: the maps stay monotone, no
// target (T < S) can lie between the two values, and equal keys / offsets are decided by the exact prefixes anyway.
__device__ __forceinline__ uint64_t key_sat(uint64_t prefix) { return prefix < (1ull << 62) ? prefix : (1ull << 62) - 1; }
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
    int64_t* zero128;          // clear the 2 * MAX_SHARDS exchange counters (sharded resample; zero_stride words apart), or nullptr
    int64_t* host_flags;       // pinned host {flags, ticket}: publish the validity flags of the weights, or nullptr
    int64_t ticket;
    int64_t n_slots;           // > 0: also write ws_out->{sB, srem, sinv}, the strata of the total over n_slots slots
    // sharded scans (MODE 3 / 4) with shard mailboxes: wait for the ranks' (max, flags) entries before reading `pmax`; the
    // workgroup that ends up with the shard total S pushes {S, 0, 0, 0, 0} to every peer (MODE 4: k_export_q pushes S with the limbs)
    MboxWait wait;
    MboxPush push;
    int zero_stride;
    // MODE 2 (the ESS getter): the scan publishes {flags, S, limbs of sum q^2} + a ticket to pinned host memory itself -- no publish launch
    // behind it (~5 us per ESS read; BASELINE config 4 and the README loop read the ESS every step).  Fence-free, like every other
    // protocol here (an agent-scope fence at the end of a kernel that has written 10 MB writes the XCD's L2 back: measured, +30 us):
    // every workgroup leaves its limb partials as TAGGED words (tag << 48 | sum, relaxed agent-scope stores), the workgroup that
    // held the last tile -- it owns S -- waits for the words of this launch's tag, folds them and stores the summary.
    // sharded scans (MODE 3 / 4), fused first summary: the launch folds the maximum slots itself, workgroup 0 leaves (max, flags) in mf_out
    // and pushes it to the peers' mailboxes, every workgroup then waits for the peers' pairs -- the separate k_pack_mflags launch (4.5 us)
    // is gone.  mf_me = this shard's index among the gathered pairs (its own pair is the fold, never read back).
    int fuse_mf; int mf_me; double* mf_out; MboxPush mf_push;
    SortedGammaJob sp;         // sp.blocks > 0: the LAST sp.blocks workgroups of the launch compute the tile totals of a sorted multinomial resample
    int64_t* q_host;           // pinned [8]: flags, S, Ql0..3, ticket, check word; nullptr: untagged partials, folded later (k_publish_scalars / k_export_q)
    int64_t q_ticket;
    // MODE 3, a sharded STRATIFIED resample with mailboxes: the workgroup that ends up with the shard total also derives the plan (gpf_k_common.hpp
    // strat_plan_body) -- no k_strat_plan launch behind the scan; it clears the exchange counters itself (zero128 is off then)
    StratPlanJob splan;
};
// Workgroup of the scan kernels: SCAN_BLOCK threads over one 2048-element tile, every wave SCAN_ROWS rows of 128.  256 threads x 4
// rows is the measured optimum: 512 x 2 (twice the waves per CU against the kernel's three dependent round trips) ran 1.3-1.5 us
// SLOWER (13.5 -> 14.9 us with the offset levels, 12.1 -> 13.4 without; profiles/r03_scan_phases.txt) -- the fold of the partial
// maxima and the two block-wide reductions cost more with eight waves than the extra overlap returns.
#ifndef GPF_SCAN_BLOCK
#define GPF_SCAN_BLOCK 256
#endif
constexpr int SCAN_BLOCK = GPF_SCAN_BLOCK;
constexpr int SCAN_NWAVES = SCAN_BLOCK / WAVE;
constexpr int SCAN_ROWS = TILE / (2 * SCAN_BLOCK);
static_assert(SCAN_ROWS * 2 * SCAN_BLOCK == TILE && SCAN_ROWS >= 2 && SCAN_ROWS % 2 == 0, "a wave owns whole 256-element groups");
// MODE 0: plain scan of In; 1: fixed-point weights (folds the max partials); 2: as 1, plus sum q^2 for the ESS;
// 3 / 4: as 1 / 2 with the maximum and flags taken from the np gathered (max, flags) pairs of the shards (mf_all) instead of the slots
template <class In, int MODE>
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan(In in, int64_t n, int64_t ntiles,
                                                const double* __restrict__ mf_all, const unsigned long long* __restrict__ slots,
                                                int np, WSum* __restrict__ ws_out, ScanOut out,
                                                uint64_t* __restrict__ dcur, uint64_t* __restrict__ dnext,
                                                uint64_t* __restrict__ total_out, uint64_t* __restrict__ blockQ,
                                                int32_t* __restrict__ timeout, ScanExtras ex)
{
    // the launch's last ex.sp.blocks workgroups are not part of the scan: they draw the tile totals of a sorted multinomial resample
    // (independent of the weights; hidden behind the scan's latency chain instead of a launch of their own)
    const unsigned G = gridDim.x - (unsigned)ex.sp.blocks;
    if (blockIdx.x >= G) { sorted_gamma_tile(ex.sp, (int64_t)(blockIdx.x - G) * SCAN_BLOCK + threadIdx.x); return; }
    // (ex.n_slots: the thread that ends up with the total also leaves the stratum width of S over n_slots output slots)
    // sharded resamples: the exchange counters of the push pass that follows are cleared here (no memset node)
    if (ex.zero128 && blockIdx.x == 0)
        for (int i = threadIdx.x; i < 2 * MAX_SHARDS * ex.zero_stride; i += SCAN_BLOCK) ex.zero128[i] = 0;
    __shared__ uint64_t s_wave[SCAN_NWAVES];
    __shared__ uint64_t s_red[SCAN_NWAVES];
    __shared__ uint64_t s_Stot;
    uint64_t* const d_agg = dcur;
    uint64_t* const d_pre = dcur + ntiles;
    for (int64_t i = (int64_t)blockIdx.x * SCAN_BLOCK + threadIdx.x; i < 2 * ntiles; i += (int64_t)G * SCAN_BLOCK) dnext[i] = 0;
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
            // the np <= 64 gathered (max, flags) pairs: one lane each, in every wave (system-scope loads: the pairs may sit in this
            // rank's mailbox, written by its peers), folded across the wave
            double m0 = 0.0; int f0 = 0;
            if (ex.fuse_mf) {                                         // kernel-uniform
                fold_slots(slots, m0, f0);
                f0 &= FLAG_NAN | FLAG_POSINF;
                if (blockIdx.x == 0 && wave_id() == 0) {
                    if (lane_id() == 0) { ex.mf_out[0] = m0; ex.mf_out[1] = (double)f0; }
                    const uint64_t words[2] = {d2u(m0), d2u((double)f0)};
                    mbox_push_wave(ex.mf_push, words);
                }
            }
            mbox_wait_block(ex.wait);
            const int g = lane_id();
            const bool mb = ex.wait.tags != nullptr;
            const bool self = ex.fuse_mf && g == ex.mf_me;            // (the own pair: the fold above, not the array another workgroup is writing)
            m = g < np ? (self ? m0 : ld_gathered(mf_all + 2 * g, mb)) : -__builtin_huge_val();
            f = g < np ? (self ? f0 : (int)ld_gathered(mf_all + 2 * g + 1, mb)) : 0;
            m = wave_max_f64(m);
#pragma unroll
            for (int s = 32; s >= 1; s >>= 1) f |= __shfl_xor(f, s, WAVE);
            if (!(f & FLAG_NAN) && m == -__builtin_huge_val()) f |= FLAG_ALL_NEGINF;
        } else fold_slots(slots, m, f);                           // every wave for itself: no LDS, no barrier
        in.m = m; in.flags = f;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            ws_out->m = m; ws_out->flags = f;
            // check = true / :warn (resample.jl:54-55): the host learns safe_softmax's validity flags NOW, from pinned memory,
            // while this kernel and the ancestor search behind it keep running (no stream synchronisation, no idle gap)
            if (ex.host_flags) {
                __hip_atomic_store(ex.host_flags, (int64_t)f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                publish_behind_sys_stores(ex.host_flags + 1, ex.ticket);      // (NOT a system-scope release: that is a write-back of the whole L2)
            }
        }
    }
    uint64_t ql[4] = {0, 0, 0, 0};
    const int lane = lane_id(), wv = wave_id();
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += G) {
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
        for (int w = 0; w < SCAN_NWAVES; ++w) { if (w < wv) wexcl += s_wave[w]; agg += s_wave[w]; }
        if (threadIdx.x == 0) desc_store(d_agg + tile, DESC_VALID | agg);
        // exclusive prefix of this tile: one parallel read of the round's earlier aggregates
        const int64_t first = (tile / G) * G;
        uint64_t acc = 0;
        for (int64_t idx = first + threadIdx.x; idx < tile; idx += SCAN_BLOCK) acc += desc_wait(d_agg + idx, timeout);
        if (first > 0 && threadIdx.x == SCAN_BLOCK - 1) acc += desc_wait(d_pre + first - 1, timeout);
        acc = wave_sum_u64(acc);
        if (lane == 0) s_red[wv] = acc;
        __syncthreads();
        uint64_t excl = 0;
#pragma unroll
        for (int w = 0; w < SCAN_NWAVES; ++w) excl += s_red[w];
        if (threadIdx.x == 0) desc_store(d_pre + tile, DESC_VALID | (excl + agg));
        const uint64_t off = excl + wexcl;
        if (out.cdf) {
#pragma unroll
            for (int k = 0; k < SCAN_ROWS; ++k) {
                const int64_t idx = wbase + k * 2 * WAVE;
                const uint64_t v1 = off + p[2 * k + 1];
                *reinterpret_cast<ulonglong2*>(out.cdf + idx) = make_ulonglong2(off + p[2 * k], v1);
                if ((lane & 7) == 7) out.t16[(idx + 1) >> 4] = v1;                     // element idx+1 = 15 (mod 16)
                if ((lane & 15) == 15) out.k32[(idx + 1) >> 5] = (uint32_t)(key_sat(v1) >> KEY_SHIFT);   // ... = 31 (mod 32)
                if (lane == WAVE - 1 && (k & 1)) out.t256[(idx + 1) >> 8] = v1;         // ... = 255 (mod 256)
                if (out.k32s && lane == WAVE - 1) out.k32s[(idx + 1) >> 7] = (uint32_t)(key_sat(v1) >> KEY_SHIFT);   // (sample == 2) element idx+1 = 127 (mod 128)
                if (out.off16) {                                                        // kernel-uniform
                    // 16-bit offsets inside the key group (16 << logg lanes of this row): klo = key of the previous group
                    const int GL = 16 << out.logg;
                    const uint32_t kv = (uint32_t)(key_sat(v1) >> KEY_SHIFT);
                    const uint32_t khi = (uint32_t)__shfl((int)kv, lane | (GL - 1), WAVE);
                    const uint32_t kprev = (uint32_t)__shfl((int)kv, ((lane & ~(GL - 1)) - 1) & (WAVE - 1), WAVE);
                    const uint32_t klo = lane < GL ? (uint32_t)(key_sat(off + cb[k]) >> KEY_SHIFT) : kprev;
                    const int sh = key_quant_shift(klo, khi);
                    const uint64_t kb = (uint64_t)klo << KEY_SHIFT;
                    const uint32_t o0 = (uint32_t)((key_sat(off + p[2 * k]) - kb) >> sh), o1 = (uint32_t)((key_sat(v1) - kb) >> sh);
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
        if constexpr (MODE == 3) {
            if (tile == ntiles - 1 && wv == SCAN_NWAVES - 1) {       // the wave that ends up with the shard total tells every peer
                const uint64_t Sw = shfl_u64(off + p[2 * SCAN_ROWS - 1], WAVE - 1);
                const uint64_t words[5] = {Sw, 0, 0, 0, 0};
                mbox_push_wave(ex.push, words);
            }
        }
        if (tile == ntiles - 1 && threadIdx.x == SCAN_BLOCK - 1) {
            const uint64_t Stot = off + p[2 * SCAN_ROWS - 1];
            *total_out = Stot;
            if constexpr (WANT_Q) s_Stot = Stot;
            if constexpr (MODE >= 1) {
                if (ex.n_slots > 0) {                   // one thread per launch: a true 64-bit division is fine here
                    const uint64_t Bq = Stot / (uint64_t)ex.n_slots;
                    ws_out->sB = Bq; ws_out->srem = Stot - Bq * (uint64_t)ex.n_slots;
                    ws_out->sinv = (double)ex.n_slots / (double)Stot;
                }
            }
        }
        if constexpr (MODE == 3) {
            if (ex.splan.plan && tile == ntiles - 1) {  // (workgroup-uniform) the stratified plan: this workgroup has just pushed the last total the plan needs
                static_assert(SCAN_BLOCK > MAX_SHARDS, "one thread per shard boundary");
                __shared__ int64_t s_F[MAX_SHARDS + 1];
                const StratPlanJob& jb = ex.splan;
                strat_plan_body(jb, [&](int g) { return shard_bound(jb.n_global, jb.G, g); }, s_F);
            }
        }
        __syncthreads();                                // s_wave / s_red reuse
    }
    if constexpr (WANT_Q) {
        // block partial of the limb sums of sum q^2 (plain stores, folded on demand by k_publish_scalars)
        __shared__ uint64_t s_q[SCAN_NWAVES][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) ql[k] = wave_sum_u64(ql[k]);
        if (lane == 0) { for (int k = 0; k < 4; ++k) s_q[wv][k] = ql[k]; }
        __syncthreads();
        const uint64_t qtag = ex.q_host ? (uint64_t)((ex.q_ticket & 0x7fff) + 1) << 48 : 0;       // (a workgroup scans <= Q_TAG_MAX_TILES tiles: its limb sums stay below 2^48 -- summarize() guards it)
        if (threadIdx.x < 4) {
            uint64_t t = 0;
            for (int w = 0; w < SCAN_NWAVES; ++w) t += s_q[w][threadIdx.x];
            if (ex.q_host) __hip_atomic_store(blockQ + (int64_t)blockIdx.x * 4 + threadIdx.x, qtag | t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else blockQ[(int64_t)blockIdx.x * 4 + threadIdx.x] = t;
        }
        if (ex.q_host && (int64_t)blockIdx.x == (ntiles - 1) % (int64_t)G) {          // the workgroup that held the last tile (workgroup-uniform)
            uint64_t q[4] = {0, 0, 0, 0};
            for (int b = threadIdx.x; b < (int)G; b += SCAN_BLOCK) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    uint64_t v = __hip_atomic_load(blockQ + (int64_t)b * 4 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    unsigned spins = 0;
                    while ((v >> 48) != (qtag >> 48)) {
                        __builtin_amdgcn_s_sleep(1);
                        v = __hip_atomic_load(blockQ + (int64_t)b * 4 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (++spins > SPIN_LIMIT) { *timeout = 1; break; }
                    }
                    q[k] += v & 0xffffffffffffull;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = wave_sum_u64(q[k]);
            __syncthreads();                                                                  // s_q: read above by threads 0..3
            if (lane == 0) { for (int k = 0; k < 4; ++k) s_q[wv][k] = q[k]; }
            __syncthreads();
            if (threadIdx.x == 0) {
                uint64_t t[4] = {0, 0, 0, 0};
                for (int w = 0; w < SCAN_NWAVES; ++w) for (int k = 0; k < 4; ++k) t[k] += s_q[w][k];
                for (int k = 0; k < 4; ++k) ws_out->Ql[k] = t[k];                              // (the device copy: later getters find the limbs folded)
                __hip_atomic_store(ex.q_host + 0, (int64_t)in.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(ex.q_host + 1, (int64_t)s_Stot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                for (int k = 0; k < 4; ++k) __hip_atomic_store(ex.q_host + 2 + k, (int64_t)t[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                // No fence orders the seven words on their way to pinned memory (a system-scope release here would write the XCD's L2
                // back: +30 us).  The payload validates itself instead: word 7 = ticket ^ flags ^ S ^ limbs; the host re-reads until the
                // ticket AND the check word agree with what it sees (a stale word among fresh ones breaks the xor).
                uint64_t chk = (uint64_t)ex.q_ticket ^ (uint64_t)(int64_t)in.flags ^ s_Stot;
                for (int k = 0; k < 4; ++k) chk ^= t[k];
                __hip_atomic_store(ex.q_host + 7, (int64_t)chk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(ex.q_host + 6, ex.q_ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// ----------------------------------------------------------------------------- weight summary WITHOUT a prefix sum (the ESS / log-ML getters)
// effective_sample_size(state) and log_ml_estimate(state) (utils.jl:163-178) need S = sum q and Q = sum q^2 only -- not the CDF.  An
// ESS-triggered filter (README.md:68, BASELINE config 4) reads the ESS every step and resamples on a minority of them: the scan's
// inter-workgroup chain and its 10 MB of CDF stores were paid on every step for a CDF most steps never used.  This is the reduction:
// same fixed-point weights (InFixQ), exact integer sums (order-free, identical to the scan's), no chain.  Workgroups leave
// {S low 31 bits, S high bits, limb sums of Q} as TAGGED words (tag << 48 | sum, relaxed agent-scope stores, as the scan's ESS
// partials: a workgroup folds <= Q_TAG_MAX_TILES tiles); workgroup 0 waits for this launch's tags, folds, fills ws_out and publishes
// {flags, S, limbs, ticket, check word} to pinned host memory (the scan's format: gpf_effective_sample_size reads either).
// SHARD (round 5: the ESS / log-ML getters of a sharded filter, gpf_shard_effective_sample_size, and the verdict of gpf_shard_step_ess): the
// maximum and the flags come from the np gathered (max, flags) pairs of the shards (as k_scan MODE 3 / 4), the sums are taken under that GLOBAL
// maximum; workgroup 0, once it has folded this shard's {S, limbs of sum q^2}, pushes them to every peer's mailbox, waits for the peers' entries,
// folds the GLOBAL summary and publishes THAT to pinned memory ({flags, S, limbs, ticket, check word}: the host polls, no copy, no stream
// synchronisation) -- and leaves the verdict ESS < thr in *go for a propagate enqueued speculatively behind this launch (thr < 0: none).
struct ShardSum {
    const double* mf_all; int np; MboxWait wait_mf;                // the gathered (max, flags) pairs (in the own mailbox)
    MboxPush push_tot; MboxWait wait_tot; const int64_t* tot_all;  // this round's {S, Ql0..3} entries: pushed to the peers / gathered in the own mailbox
    int G, me; double thr; int32_t* go;
    // k_sum_shard only -- the (max, flags) round inside the same launch (slots != nullptr): workgroup 0 folds this shard's maximum slots and sends
    // the pair to every peer before anybody waits for the gathered pairs (no k_pack_mflags launch in front: 4.6 us on one rank)
    const unsigned long long* slots; double* mf_out; MboxPush push_mf;
};
template <bool SHARD>
__global__ __launch_bounds__(SCAN_BLOCK) void k_sum_reduce(InFixQ in, int64_t n, int64_t ntiles, const unsigned long long* __restrict__ slots,
                                                           WSum* __restrict__ ws_out, uint64_t* __restrict__ part, int64_t* __restrict__ q_host,
                                                           int64_t q_ticket, int32_t* __restrict__ timeout, ShardSum ss)
{
    double m; int f;
    if constexpr (SHARD) {
        mbox_wait_block(ss.wait_mf);
        const int g = lane_id();
        const bool mb = ss.wait_mf.tags != nullptr;
        m = g < ss.np ? ld_gathered(ss.mf_all + 2 * g, mb) : -__builtin_huge_val();
        f = g < ss.np ? (int)ld_gathered(ss.mf_all + 2 * g + 1, mb) : 0;
        m = wave_max_f64(m);
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) f |= __shfl_xor(f, s, WAVE);
        if (!(f & FLAG_NAN) && m == -__builtin_huge_val()) f |= FLAG_ALL_NEGINF;
    } else fold_slots(slots, m, f);
    in.m = m; in.flags = f;
    const int lane = lane_id(), wv = wave_id();
    uint64_t acc[6] = {0, 0, 0, 0, 0, 0};             // S, (unused), Ql0..3
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t wbase = tile * TILE + (int64_t)wv * (SCAN_ROWS * 2 * WAVE) + 2 * lane;
#pragma unroll
        for (int k = 0; k < SCAN_ROWS; ++k) {
            uint64_t q0, q1;
            in.load2(wbase + k * 2 * WAVE, n, q0, q1);
            acc[0] += q0 + q1;
            uint64_t lo = q0 * q0, hi = __umul64hi(q0, q0);
            acc[2] += lo & 0xffffffffull; acc[3] += lo >> 32; acc[4] += hi & 0xffffffffull; acc[5] += hi >> 32;
            lo = q1 * q1; hi = __umul64hi(q1, q1);
            acc[2] += lo & 0xffffffffull; acc[3] += lo >> 32; acc[4] += hi & 0xffffffffull; acc[5] += hi >> 32;
        }
    }
    __shared__ uint64_t s_p[SCAN_NWAVES][6];
#pragma unroll
    for (int k = 0; k < 6; ++k) acc[k] = wave_sum_u64(acc[k]);
    if (lane == 0) { for (int k = 0; k < 6; ++k) s_p[wv][k] = acc[k]; }
    __syncthreads();
    const uint64_t tag = (uint64_t)((q_ticket & 0x7fff) + 1) << 48;
    if (threadIdx.x < 6) {
        uint64_t t = 0;
        for (int w = 0; w < SCAN_NWAVES; ++w) t += s_p[w][threadIdx.x];
        if (threadIdx.x == 0) { s_p[0][0] = t & 0x7fffffffull; s_p[0][1] = t >> 31; }      // S in two tagged halves (S < 2^62)
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        uint64_t t = 0;
        if (threadIdx.x < 2) t = s_p[0][threadIdx.x];
        else for (int w = 0; w < SCAN_NWAVES; ++w) t += s_p[w][threadIdx.x];
        __hip_atomic_store(part + (int64_t)blockIdx.x * 6 + threadIdx.x, tag | t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (blockIdx.x != 0) return;
    uint64_t q[6] = {0, 0, 0, 0, 0, 0};
    for (int b = threadIdx.x; b < (int)gridDim.x; b += SCAN_BLOCK) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            uint64_t v = __hip_atomic_load(part + (int64_t)b * 6 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while ((v >> 48) != (tag >> 48)) {
                __builtin_amdgcn_s_sleep(1);
                v = __hip_atomic_load(part + (int64_t)b * 6 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (++spins > SPIN_LIMIT) { *timeout = 1; break; }
            }
            q[k] += v & 0xffffffffffffull;
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) q[k] = wave_sum_u64(q[k]);
    __syncthreads();
    if (lane == 0) { for (int k = 0; k < 6; ++k) s_p[wv][k] = q[k]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int w = 0; w < SCAN_NWAVES; ++w) for (int k = 0; k < 6; ++k) t[k] += s_p[w][k];
        const uint64_t S = t[0] + (t[1] << 31);
        ws_out->m = m; ws_out->flags = f; ws_out->S = S;
        for (int k = 0; k < 4; ++k) ws_out->Ql[k] = t[2 + k];
        if constexpr (SHARD) { s_p[0][0] = S; for (int k = 0; k < 4; ++k) s_p[0][1 + k] = t[2 + k]; }
        else if (q_host) {
            __hip_atomic_store(q_host + 0, (int64_t)f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(q_host + 1, (int64_t)S, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            uint64_t chk = (uint64_t)q_ticket ^ (uint64_t)(int64_t)f ^ S;
            for (int k = 0; k < 4; ++k) { __hip_atomic_store(q_host + 2 + k, (int64_t)t[2 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); chk ^= t[2 + k]; }
            __hip_atomic_store(q_host + 7, (int64_t)chk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(q_host + 6, q_ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if constexpr (SHARD) {
        // ---- this shard's {S, limbs} to every peer; the peers' entries back; the GLOBAL summary to the host (and the verdict to the device)
        __syncthreads();
        if (wv == 0) {
            const uint64_t words[5] = {s_p[0][0], s_p[0][1], s_p[0][2], s_p[0][3], s_p[0][4]};
            mbox_push_wave(ss.push_tot, words);
        }
        mbox_wait_block(ss.wait_tot);
        if (wv == 0) {
            const bool mb = ss.wait_tot.tags != nullptr;
            uint64_t g5[5] = {0, 0, 0, 0, 0};
            if (lane < ss.G) {
#pragma unroll
                for (int k = 0; k < 5; ++k) g5[k] = lane == ss.me ? s_p[0][k] : (uint64_t)ld_gathered(ss.tot_all + 5 * lane + k, mb);   // (the own entry: what was just pushed)
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) g5[k] = wave_sum_u64(g5[k]);
            if (lane == 0) {
                const uint64_t Sg = g5[0];
                const uint64_t lo = g5[1] + (g5[2] << 32);
                const uint64_t hi = (g5[2] >> 32) + g5[3] + (g5[4] << 32) + (lo < g5[1] ? 1u : 0u);
                if (ss.go) *ss.go = ss.thr >= 0.0 && !f && ess_from(Sg, hi, lo) < ss.thr ? 1 : 0;
                __hip_atomic_store(q_host + 0, (int64_t)f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(q_host + 1, (int64_t)Sg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                uint64_t chk = (uint64_t)q_ticket ^ (uint64_t)(int64_t)f ^ Sg;
                for (int k = 0; k < 4; ++k) { __hip_atomic_store(q_host + 2 + k, (int64_t)g5[1 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); chk ^= g5[1 + k]; }
                __hip_atomic_store(q_host + 7, (int64_t)chk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(q_host + 6, q_ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// The same summary with the HOST as the folder (the ESS / log-ML getters block on the result anyway): every workgroup leaves its partial
// sums as eight tagged words in ONE 64-byte line of pinned host memory and is done -- no dependency between workgroups, no collecting
// workgroup (k_sum_reduce: two cross-XCD hops of ~2 us between the partial stores and the publish).  The host waits for the tags of this
// launch and adds <= n_cu lines up.  1024-thread workgroups, 8 weights per lane, every load in flight at once.
//   line b: {S low 31 bits, S high bits | flags << 40, Q limb 0..3, maximum low / high 32 bits}, each (tag << 48) | value
constexpr int SH_BLOCK = 1024, SH_NWAVES = SH_BLOCK / WAVE, SH_ROWS = 2, SH_TILE = SH_BLOCK * 2 * SH_ROWS;      // 4096 weights per workgroup and trip: 245 workgroups at 10^6, every CU busy
// GATE (gpf_step_ess: the README loop's `if effective_sample_size(state) < threshold`, README.md:68, decided where the next launch can see
// it): every workgroup also adds its partial sums into one of GATE_SLOTS accumulator lines in device memory (gate_verdict, gpf_k_common.hpp);
// the launch's first workgroup clears the accumulators of the NEXT gated reduction (two sets alternate).
struct SumGate { uint64_t* acc; uint64_t* acc_next; };
template <bool GATE>
__global__ __launch_bounds__(SH_BLOCK) void k_sum_host(InFixQ in, int64_t n, const unsigned long long* __restrict__ slots, int64_t* __restrict__ h_part, int64_t q_ticket, SumGate gt)
{
    const int lane = lane_id(), wv = wave_id();
    uint64_t acc[5] = {0, 0, 0, 0, 0};                // S, Ql0..3
    // the first trip's weights are requested BEFORE the maximum slots are folded (raw2 needs neither the maximum nor the flags): one
    // memory round trip instead of two in a kernel that is nothing but a chain of them
    double v0[SH_ROWS], v1[SH_ROWS];
    int64_t base = (int64_t)blockIdx.x * SH_TILE;
#pragma unroll
    for (int k = 0; k < SH_ROWS; ++k) in.raw2(base + (int64_t)k * (2 * SH_BLOCK) + 2 * (int64_t)threadIdx.x, n, v0[k], v1[k]);
    double m; int f;
    fold_slots(slots, m, f);
    in.m = m; in.flags = f;
    for (; base < n; base += (int64_t)gridDim.x * SH_TILE) {
#pragma unroll
        for (int k = 0; k < SH_ROWS; ++k) {
            uint64_t q0, q1;
            in.conv2(base + (int64_t)k * (2 * SH_BLOCK) + 2 * (int64_t)threadIdx.x, n, v0[k], v1[k], q0, q1);
            acc[0] += q0 + q1;
            uint64_t lo = q0 * q0, hi = __umul64hi(q0, q0);
            acc[1] += lo & 0xffffffffull; acc[2] += lo >> 32; acc[3] += hi & 0xffffffffull; acc[4] += hi >> 32;
            lo = q1 * q1; hi = __umul64hi(q1, q1);
            acc[1] += lo & 0xffffffffull; acc[2] += lo >> 32; acc[3] += hi & 0xffffffffull; acc[4] += hi >> 32;
        }
        const int64_t nb = base + (int64_t)gridDim.x * SH_TILE;
        if (nb < n) {
#pragma unroll
            for (int k = 0; k < SH_ROWS; ++k) in.raw2(nb + (int64_t)k * (2 * SH_BLOCK) + 2 * (int64_t)threadIdx.x, n, v0[k], v1[k]);
        }
    }
    __shared__ uint64_t s_p[SH_NWAVES][5];
#pragma unroll
    for (int k = 0; k < 5; ++k) acc[k] = wave_sum_u64(acc[k]);
    if (lane == 0) { for (int k = 0; k < 5; ++k) s_p[wv][k] = acc[k]; }
    __syncthreads();
    if (threadIdx.x < 8) {
        const uint64_t tag = (uint64_t)((q_ticket & 0x7fff) + 1) << 48;
        const int t = (int)threadIdx.x;
        uint64_t v = 0;
        if (t < 2) {
            uint64_t S = 0;
            for (int w = 0; w < SH_NWAVES; ++w) S += s_p[w][0];
            v = t == 0 ? (S & 0x7fffffffull) : ((S >> 31) | ((uint64_t)(uint32_t)f << 40));       // (S < 2^62: the high part < 2^31)
        } else if (t < 6) {
            for (int w = 0; w < SH_NWAVES; ++w) v += s_p[w][t - 1];
        } else v = t == 6 ? (d2u(m) & 0xffffffffull) : (d2u(m) >> 32);
        __hip_atomic_store(h_part + (int64_t)blockIdx.x * 8 + t, (int64_t)(tag | v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if constexpr (GATE) {
            uint64_t* const slot = gt.acc + (blockIdx.x & (GATE_SLOTS - 1)) * 8;
            if (t == 0) {
                uint64_t S = 0;
                for (int w = 0; w < SH_NWAVES; ++w) S += s_p[w][0];
                (void)__hip_atomic_fetch_add(slot + 0, S, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (t == 1) { if (f) (void)__hip_atomic_fetch_or(slot + 1, (uint64_t)(uint32_t)f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            else if (t < 6) (void)__hip_atomic_fetch_add(slot + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if constexpr (GATE) { if (blockIdx.x == 0 && threadIdx.x >= WAVE && threadIdx.x < WAVE + GATE_WORDS) gt.acc_next[threadIdx.x - WAVE] = 0; }
}
// The sharded GLOBAL summary (ShardSum, above) in k_sum_host's shape: 1024-thread workgroups, one per CU, the first trip's weights requested
// BEFORE the gathered maxima are waited for; every workgroup adds its five partial sums into one of GATE_SLOTS accumulator lines (k_sum_host<GATE>'s,
// the two sets alternating) and the LAST workgroup to arrive -- not a collecting workgroup that spins on tagged partials, k_sum_reduce<SHARD>: 17.8 us
// per launch at 10^6 against k_sum_host's 9.1 -- folds the lines (one load per lane, three butterfly steps), exchanges the shard's {S, limbs} through the
// mailboxes, publishes the global summary to pinned memory and leaves the verdict ESS < thr on the device.
struct ShardAcc { uint64_t* acc; uint64_t* acc_next; unsigned int* arrive; };
static __global__ __launch_bounds__(SH_BLOCK) void k_sum_shard(InFixQ in, int64_t n, WSum* __restrict__ ws_out, int64_t* __restrict__ q_host, int64_t q_ticket, ShardAcc sa, ShardSum ss)
{
    const int lane = lane_id(), wv = wave_id();
    uint64_t acc[5] = {0, 0, 0, 0, 0};                // S, Ql0..3
    double v0[SH_ROWS], v1[SH_ROWS];
    int64_t base = (int64_t)blockIdx.x * SH_TILE;
#pragma unroll
    for (int k = 0; k < SH_ROWS; ++k) in.raw2(base + (int64_t)k * (2 * SH_BLOCK) + 2 * (int64_t)threadIdx.x, n, v0[k], v1[k]);
    __shared__ double s_m; __shared__ int s_f; __shared__ int s_last;
    __shared__ uint64_t s_p[SH_NWAVES][5];
    if (blockIdx.x == 0 && threadIdx.x >= WAVE && threadIdx.x < WAVE + GATE_WORDS) sa.acc_next[threadIdx.x - WAVE] = 0;    // (the next launch's set)
    if (ss.slots && blockIdx.x == 0 && wv == 0) {              // this shard's (max, flags) to every peer (k_pack_mflags)
        double m0; int f0;
        fold_slots(ss.slots, m0, f0);
        const uint64_t words[2] = {d2u(m0), d2u((double)(f0 & (FLAG_NAN | FLAG_POSINF)))};
        if (lane == 0) { ss.mf_out[0] = m0; ss.mf_out[1] = u2d(words[1]); }
        mbox_push_wave(ss.push_mf, words);
    }
    mbox_wait_block(ss.wait_mf);
    if (wv == 0) {
        const bool mb = ss.wait_mf.tags != nullptr;
        double m = lane < ss.np ? ld_gathered(ss.mf_all + 2 * lane, mb) : -__builtin_huge_val();
        int f = lane < ss.np ? (int)ld_gathered(ss.mf_all + 2 * lane + 1, mb) : 0;
        m = wave_max_f64(m);
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) f |= __shfl_xor(f, s, WAVE);
        if (!(f & FLAG_NAN) && m == -__builtin_huge_val()) f |= FLAG_ALL_NEGINF;
        if (lane == 0) { s_m = m; s_f = f; }
    }
    __syncthreads();
    const double m = s_m; const int f = s_f;
    in.m = m; in.flags = f;
    for (; base < n; base += (int64_t)gridDim.x * SH_TILE) {
#pragma unroll
        for (int k = 0; k < SH_ROWS; ++k) {
            uint64_t q0, q1;
            in.conv2(base + (int64_t)k * (2 * SH_BLOCK) + 2 * (int64_t)threadIdx.x, n, v0[k], v1[k], q0, q1);
            acc[0] += q0 + q1;
            uint64_t lo = q0 * q0, hi = __umul64hi(q0, q0);
            acc[1] += lo & 0xffffffffull; acc[2] += lo >> 32; acc[3] += hi & 0xffffffffull; acc[4] += hi >> 32;
            lo = q1 * q1; hi = __umul64hi(q1, q1);
            acc[1] += lo & 0xffffffffull; acc[2] += lo >> 32; acc[3] += hi & 0xffffffffull; acc[4] += hi >> 32;
        }
        const int64_t nb = base + (int64_t)gridDim.x * SH_TILE;
        if (nb < n) {
#pragma unroll
            for (int k = 0; k < SH_ROWS; ++k) in.raw2(nb + (int64_t)k * (2 * SH_BLOCK) + 2 * (int64_t)threadIdx.x, n, v0[k], v1[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) acc[k] = wave_sum_u64(acc[k]);
    if (lane == 0) { for (int k = 0; k < 5; ++k) s_p[wv][k] = acc[k]; }
    __syncthreads();
    if (threadIdx.x < 5) {
        uint64_t v = 0;
        for (int w = 0; w < SH_NWAVES; ++w) v += s_p[w][threadIdx.x];
        uint64_t* const slot = sa.acc + (blockIdx.x & (GATE_SLOTS - 1)) * 8;
        (void)__hip_atomic_fetch_add(slot + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sys_stores_acknowledged();                     // (the adds have been performed before this workgroup counts as arrived)
    }
    __syncthreads();
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(sa.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    // ---- the last workgroup: this shard's sums, the exchange, the publish
    uint64_t mine[5] = {0, 0, 0, 0, 0};
    if (wv == 0) {
        static_assert(GATE_WORDS == WAVE, "one accumulator word per lane");
        uint64_t v = __hip_atomic_load(sa.acc + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v += shfl_xor_u64(v, 8); v += shfl_xor_u64(v, 16); v += shfl_xor_u64(v, 32);
#pragma unroll
        for (int k = 0; k < 5; ++k) mine[k] = shfl_u64(v, k);
        if (lane == 0) {
            __hip_atomic_store(sa.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ws_out->m = m; ws_out->flags = f; ws_out->S = mine[0];
            for (int k = 0; k < 4; ++k) ws_out->Ql[k] = mine[1 + k];
        }
        mbox_push_wave(ss.push_tot, mine);
    }
    mbox_wait_block(ss.wait_tot);
    if (wv == 0) {
        const bool mb = ss.wait_tot.tags != nullptr;
        uint64_t g5[5] = {0, 0, 0, 0, 0};
        if (lane < ss.G) {
#pragma unroll
            for (int k = 0; k < 5; ++k) g5[k] = lane == ss.me ? mine[k] : (uint64_t)ld_gathered(ss.tot_all + 5 * lane + k, mb);   // (the own entry: what was just pushed)
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) g5[k] = wave_sum_u64(g5[k]);
        if (lane == 0) {
            const uint64_t Sg = g5[0];
            const uint64_t lo = g5[1] + (g5[2] << 32);
            const uint64_t hi = (g5[2] >> 32) + g5[3] + (g5[4] << 32) + (lo < g5[1] ? 1u : 0u);
            if (ss.go) *ss.go = ss.thr >= 0.0 && !f && ess_from(Sg, hi, lo) < ss.thr ? 1 : 0;
            __hip_atomic_store(q_host + 0, (int64_t)f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(q_host + 1, (int64_t)Sg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            uint64_t chk = (uint64_t)q_ticket ^ (uint64_t)(int64_t)f ^ Sg;
            for (int k = 0; k < 4; ++k) { __hip_atomic_store(q_host + 2 + k, (int64_t)g5[1 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); chk ^= g5[1 + k]; }
            __hip_atomic_store(q_host + 7, (int64_t)chk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(q_host + 6, q_ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
// Residual resampling needs TWO prefix sums over the same elements: the copy counts c_i = (N q_i) div S and the
// residual weights r_i = ((N q_i) mod S) >> sh (resample.jl:99,109).  One pass computes both: one read of the weight CDF,
// ONE 64-bit division per element (quotient and remainder), two descriptor channels polled in the same round trip.
// Same tile / descriptor protocol as k_scan (channel A = counts, channel B = residual weights).
struct Scan2Chan { ScanOut out; uint64_t* dcur; uint64_t* dnext; uint64_t* total_out; };
// head_anc != nullptr (the plain residual resample): the deterministic head of the resampler -- particle i into the slots
// [ccdf[i - 1], ccdf[i]) (resample.jl:99-106) -- is written HERE, by the tile that has just learned where its copies start: a lane writes
// the copies of its own cells when there are few (<= HEAD_SMALL each), cells with more are filled by the whole workgroup (coalesced
// runs).  The search kernel then looks up the i.i.d. tail only, and the copy-count CDF and its levels (A.out) are not stored at all.
// A tile with <= HEAD_STAGE copies in all assembles them in LDS first and stores them as one coalesced run.
// tagged limb partials (ScanExtras::q_host): tag << 48 | sum needs sum < 2^48; a limb is < 2^32, so a workgroup may fold at most 2^16
// elements = 32 tiles
constexpr int Q_TAG_MAX_TILES = 32;
constexpr int HEAD_SMALL = 16, HEAD_LIST = 128, HEAD_STAGE = 4096;
// DIRECT (an ESS read came before the resample -- README.md:68-70 -- and left {maximum, flags, S} with the host, k_sum_host): the weights
// are converted here (the same exp_fix as the weight scan's) and the summary arrives as kernel arguments, so the weight scan -- whose only
// product this kernel would read is q_i = cdf[i] - cdf[i - 1] -- is not run at all; workgroup 0 leaves the summary in the device block for
// the search kernel's log-ML update.
// ... on a SHARD (mf_all != nullptr): S is the GLOBAL total the host read from the summary reduction's publish (k_sum_shard), the maximum and the flags are
// folded here from the ranks' gathered (max, flags) pairs of that reduction's round -- m / flags above are ignored.
struct ResidDirect { const double* lw; double m; int32_t flags; int32_t K; uint64_t S; WSum* ws_out;
                     const double* mf_all; int32_t G; int32_t in_mailbox; };
template <bool DIRECT>
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_residual2(const uint64_t* __restrict__ cdf, const WSum* ws, int64_t Nslots,
                                                          int64_t n, int64_t ntiles, Scan2Chan A, Scan2Chan B,
                                                          int32_t* __restrict__ timeout, int32_t* __restrict__ head_anc,
                                                          HeadGiants* __restrict__ giants, uint32_t tag, ResidDirect rd)
{
    __shared__ uint64_t s_wave[2][SCAN_NWAVES];
    __shared__ uint64_t s_red[2][SCAN_NWAVES];
    __shared__ unsigned int s_nheavy;
    __shared__ uint64_t s_hstart[HEAD_LIST];
    __shared__ uint32_t s_hcell[HEAD_LIST], s_hcnt[HEAD_LIST];
    __shared__ int32_t s_stage[HEAD_STAGE];
    for (int64_t i = (int64_t)blockIdx.x * SCAN_BLOCK + threadIdx.x; i < 2 * ntiles; i += (int64_t)gridDim.x * SCAN_BLOCK) { A.dnext[i] = 0; B.dnext[i] = 0; }
    uint64_t S;
    InFixQ in{PrioView{rd.lw, nullptr, 0.0, 0}, nullptr, nullptr, rd.K, rd.m, rd.flags};
    if constexpr (DIRECT) {
        S = rd.S;
        if (rd.mf_all) {                                    // a shard: the global (max, flags) from the gathered pairs (one lane per rank; kernel-uniform branch)
            __shared__ double s_gm; __shared__ int s_gf;
            if (wave_id() == 0) {
                const int l = lane_id();
                double m = l < rd.G ? ld_gathered(rd.mf_all + 2 * l, rd.in_mailbox != 0) : -__builtin_huge_val();
                int f = l < rd.G ? (int)ld_gathered(rd.mf_all + 2 * l + 1, rd.in_mailbox != 0) : 0;
                m = wave_max_f64(m);
#pragma unroll
                for (int q = 32; q >= 1; q >>= 1) f |= __shfl_xor(f, q, WAVE);
                if (!(f & FLAG_NAN) && m == -__builtin_huge_val()) f |= FLAG_ALL_NEGINF;
                if (l == 0) { s_gm = m; s_gf = f; }
            }
            __syncthreads();
            in.m = s_gm; in.flags = s_gf;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) { rd.ws_out->m = in.m; rd.ws_out->flags = in.flags; rd.ws_out->S = rd.S; }
    } else S = ws->S;
    const int sh = residual_shift(S, Nslots);
    const int lane = lane_id(), wv = wave_id();
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t wbase = tile * TILE + (int64_t)wv * (SCAN_ROWS * 2 * WAVE) + 2 * lane;
        uint64_t pa[2 * SCAN_ROWS], pb[2 * SCAN_ROWS];
        uint32_t cnt[2 * SCAN_ROWS];                                                  // copies of the lane's cells (head_anc)
        uint64_t ca = 0, cb = 0;
        if (head_anc && threadIdx.x == 0) s_nheavy = 0;
#pragma unroll
        for (int k = 0; k < SCAN_ROWS; ++k) {
            const int64_t idx = wbase + k * 2 * WAVE;
            uint64_t q0, q1;
            if constexpr (DIRECT) in.load2(idx, n, q0, q1);
            else {
                const ulonglong2 c = *reinterpret_cast<const ulonglong2*>(cdf + idx);      // padded to whole tiles, flat beyond n
                const uint64_t prev = idx > 0 ? cdf[idx - 1] : 0;
                q0 = c.x - prev; q1 = c.y - c.x;
            }
            uint64_t a0 = 0, a1 = 0, b0 = 0, b1 = 0;
            if (S != 0) {
                const uint64_t n0 = (uint64_t)Nslots * q0, n1 = (uint64_t)Nslots * q1;
                a0 = n0 / S; b0 = (n0 - a0 * S) >> sh;
                a1 = n1 / S; b1 = (n1 - a1 * S) >> sh;
            }
            if (idx >= n) { a0 = 0; b0 = 0; }
            if (idx + 1 >= n) { a1 = 0; b1 = 0; }
            cnt[2 * k] = (uint32_t)a0; cnt[2 * k + 1] = (uint32_t)a1;
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
        for (int w = 0; w < SCAN_NWAVES; ++w) {
            if (w < wv) { wexa += s_wave[0][w]; wexb += s_wave[1][w]; }
            agga += s_wave[0][w]; aggb += s_wave[1][w];
        }
        if (threadIdx.x == 0) { desc_store(A.dcur + tile, DESC_VALID | agga); desc_store(B.dcur + tile, DESC_VALID | aggb); }
        const int64_t first = (tile / gridDim.x) * gridDim.x;
        uint64_t acca = 0, accb = 0;
        for (int64_t idx = first + threadIdx.x; idx < tile; idx += SCAN_BLOCK) {
            acca += desc_wait(A.dcur + idx, timeout);
            accb += desc_wait(B.dcur + idx, timeout);
        }
        if (first > 0 && threadIdx.x == SCAN_BLOCK - 1) {
            acca += desc_wait(A.dcur + ntiles + first - 1, timeout);
            accb += desc_wait(B.dcur + ntiles + first - 1, timeout);
        }
        acca = wave_sum_u64(acca); accb = wave_sum_u64(accb);
        if (lane == 0) { s_red[0][wv] = acca; s_red[1][wv] = accb; }
        __syncthreads();
        uint64_t exa = 0, exb = 0;
#pragma unroll
        for (int w = 0; w < SCAN_NWAVES; ++w) { exa += s_red[0][w]; exb += s_red[1][w]; }
        if (threadIdx.x == 0) {
            desc_store(A.dcur + ntiles + tile, DESC_VALID | (exa + agga));
            desc_store(B.dcur + ntiles + tile, DESC_VALID | (exb + aggb));
        }
        const uint64_t offa = exa + wexa, offb = exb + wexb;
        const bool staged = head_anc && agga <= (uint64_t)HEAD_STAGE;                  // (workgroup-uniform)
#pragma unroll
        for (int k = 0; k < SCAN_ROWS; ++k) {
            const int64_t idx = wbase + k * 2 * WAVE;
            const uint64_t va = offa + pa[2 * k + 1], vb = offb + pb[2 * k + 1];
            *reinterpret_cast<ulonglong2*>(B.out.cdf + idx) = make_ulonglong2(offb + pb[2 * k], vb);
            if ((lane & 7) == 7) B.out.t16[(idx + 1) >> 4] = vb;
            if ((lane & 15) == 15) B.out.k32[(idx + 1) >> 5] = (uint32_t)(key_sat(vb) >> KEY_SHIFT);
            if (lane == WAVE - 1 && (k & 1)) B.out.t256[(idx + 1) >> 8] = vb;
            if (!head_anc) {
                *reinterpret_cast<ulonglong2*>(A.out.cdf + idx) = make_ulonglong2(offa + pa[2 * k], va);
                if ((lane & 7) == 7) A.out.t16[(idx + 1) >> 4] = va;
                if ((lane & 15) == 15) A.out.k32[(idx + 1) >> 5] = (uint32_t)(key_sat(va) >> KEY_SHIFT);
                if (lane == WAVE - 1 && (k & 1)) A.out.t256[(idx + 1) >> 8] = va;
            } else {
                // the copies of the lane's two cells start at their exclusive prefixes
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const uint32_t c_ = cnt[2 * k + u];
                    const uint64_t start = offa + pa[2 * k + u] - c_;
                    if (c_ == 0) continue;
                    unsigned slot = HEAD_LIST;
                    if (c_ >= GIANT_COPIES) {
                        // a cell with very many copies: one workgroup would fill them alone (10^6 copies: ~50 us); listed for the search
                        // kernel's whole grid instead.  The list's word carries this resample's tag: claim an entry, or start the list
                        unsigned int w = __hip_atomic_load(&giants->word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), e = GIANT_MAX;
                        while (true) {
                            const unsigned int have = (w >> 8) == tag ? (w & 0xffu) : 0u;
                            if (have >= (unsigned)GIANT_MAX) break;
                            const unsigned int nw = (tag << 8) | (have + 1u);
                            if (__hip_atomic_compare_exchange_strong(&giants->word, &w, nw, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { e = have; break; }
                        }
                        if (e < (unsigned)GIANT_MAX) { giants->start[e] = start; giants->cell[e] = (uint32_t)(idx + u); giants->cnt[e] = c_; continue; }
                    }
                    if (c_ > HEAD_SMALL) slot = atomicAdd(&s_nheavy, 1u);
                    if (slot < HEAD_LIST) { s_hstart[slot] = start; s_hcell[slot] = (uint32_t)(idx + u); s_hcnt[slot] = c_; }
                    else if (staged) for (uint32_t q = 0; q < c_; ++q) s_stage[(start - exa) + q] = (int32_t)(idx + u);
                    else for (uint32_t q = 0; q < c_; ++q) head_anc[start + q] = (int32_t)(idx + u);      // few copies (or the list is full)
                }
            }
        }
        if (head_anc) {
            __syncthreads();
            if (staged) {                                                              // the tile's copies as one coalesced run (the listed cells' ranges
                for (uint32_t q = threadIdx.x; q < (uint32_t)agga; q += SCAN_BLOCK) head_anc[exa + q] = s_stage[q];   // hold leftovers: filled next)
                __syncthreads();
            }
            const unsigned nh = s_nheavy < HEAD_LIST ? s_nheavy : HEAD_LIST;
            for (unsigned hh = 0; hh < nh; ++hh) {
                const uint64_t start = s_hstart[hh]; const uint32_t c_ = s_hcnt[hh]; const int32_t cell = (int32_t)s_hcell[hh];
                for (uint32_t q = threadIdx.x; q < c_; q += SCAN_BLOCK) head_anc[start + q] = cell;
            }
        }
        if (tile == ntiles - 1 && threadIdx.x == SCAN_BLOCK - 1) { *A.total_out = offa + pa[2 * SCAN_ROWS - 1]; *B.total_out = offb + pb[2 * SCAN_ROWS - 1]; }
        __syncthreads();                                // s_wave / s_red reuse
    }
}

// the device scalar block -> its pinned host mirror, ticket last: the host polls the ticket instead of synchronising the
// stream (a hipMemcpyAsync + hipStreamSynchronize pair costs ~13 us of wake-up latency per getter; this costs the launch)
// blockQ != nullptr: the limb partials of sum q^2 that the scan blocks left (only the ESS needs them) are folded into sc->raw.Ql
// on the way -- one launch for "fold + publish" (the ESS-triggered loop of BASELINE config 4 asks for the ESS every step).
// Launched with ONE wave.
static __global__ void k_publish_scalars(Scalars* sc, Scalars* host, long long* host_ticket, long long ticket,
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
    sys_stores_acknowledged();                                   // (every thread's words, before the barrier in front of the ticket: gpf_k_common.hpp)
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(host_ticket, ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

} // namespace gpf
