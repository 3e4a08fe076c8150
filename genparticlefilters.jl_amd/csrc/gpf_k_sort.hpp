// K10: stable descending radix sort (onesweep)  (part of gpf_kernels.hpp; include that header, not this file)
#pragma once

namespace gpf {
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
// (after the histograms: 64 words of tile tickets -- 8 per pass --, then the completion words of k_sort_finish, one per 128-byte line:
//  same-LINE atomics serialise like same-address ones, ~10 ns each)
constexpr int SORT_TICKET_WAYS = 8;
constexpr int SORT_DONE_STRIDE = 32;               // u32 words between completion counters
// (K10d, below: behind the completion words the histogram of the SORT_FINE fine bins and the SORT_BINS + 1 bucket boundaries)
constexpr int SORT_FINE_BITS = 13, SORT_FINE = 1 << SORT_FINE_BITS, SORT_COARSE_BITS = 24;
// (the wide form of K10d for filters above SORT_BINS * 4608 particles: twice the buckets and twice the fine bins -- the mean bucket and the
//  fullest fine bin stay what they are at half the size; the workspace is laid out for the wide form)
constexpr int SORT_BINS_WIDE = 512, SORT_FINE_BITS_WIDE = 14, SORT_FINE_MAX = 1 << SORT_FINE_BITS_WIDE;
// "dead" keys: more than 2^6 = 64 below the maximum (incl. -inf and everything beyond the coarse key's range).  Their fixed-point weight
// is exactly 0 (exp(-64) 2^52 < 1/2): no slot can ever choose them, so their order among themselves cannot be observed
// through pf_resample! -- they only have to stand behind every live key.  K10d sends them all to the last bucket, which k_sort_buckets
// (it orders every bucket IN PLACE) leaves as the partition wrote it, in the partition's (stable) order: -inf-heavy and sharply peaked weight vectors (a bearings filter: half the particles) sort without ranking
// their dead half and without the eight-pass fall-back.  (All weights -inf: every key is dead, the order is the identity -- which is the
// stable sort of equal keys that the uniform fall-back of safe_softmax needs.)
constexpr uint32_t SORT_COARSE_DEAD = 28u << 19;             // sort_coarse of 2^6: binade index 6 + 21, + 1
__host__ __device__ constexpr int sort_fine_dead(int fb) { return (int)(SORT_COARSE_DEAD >> (SORT_COARSE_BITS - fb)); }
__host__ __device__ __forceinline__ size_t sort_ws_fine_offset() { return (size_t)(SORT_PASSES * SORT_BINS + 64 + 18 * SORT_DONE_STRIDE) * sizeof(uint32_t); }
__host__ __device__ __forceinline__ size_t sort_ws_desc_offset() { return sort_ws_fine_offset() + (size_t)(SORT_FINE_MAX + 2 * SORT_BINS_WIDE) * sizeof(uint32_t); }
__host__ inline size_t sort_ws_bytes(int64_t n)
{
    const int64_t nt = (n + SORT_TILE - 1) / SORT_TILE;
    return sort_ws_desc_offset() + (size_t)SORT_PASSES * (nt + (nt + 15) / 16 + (nt + 255) / 256) * SORT_BINS * sizeof(uint64_t);
}
// keys of the log-priorities + the histograms of all eight digits in one pass over the weights
// (FIRST: the lowest digit that will be sorted -- 0: all eight passes)
// clear / clear16: the OTHER of the two sort workspaces (16-byte units), zeroed here for the next sort: histograms, tickets and
// descriptor planes must start at zero, and a hipMemsetAsync in front of every sort costs ~5 us of stream time each
// COARSE: the three digits of the coarse key (sort_coarse, gpf_k_common.hpp) instead, histograms 0..2; `slots` hold the maximum of pv
// (MaxSlots), which workgroup 0 also leaves in *m_out for the passes
constexpr int SORT_M_WORD = 62;                     // the coarse sort's maximum: a double in ticket words 62, 63 of the workspace
#ifndef GPF_KF_BLOCK
#define GPF_KF_BLOCK 1024
#endif
constexpr int KF_BLOCK = GPF_KF_BLOCK;             // the key passes: one workgroup per CU (the flush is global atomics), many waves: every lane has all its loads in flight at once
                                                   // (256-thread workgroups with four serial batches of loads: 16.3 us for k_sort_keys_fine against 10.7)
template <int FIRST, bool COARSE = false>
__global__ __launch_bounds__(KF_BLOCK) void k_sort_keys_hist(PrioView pv, int64_t n, uint64_t* __restrict__ keys, uint32_t* __restrict__ hist,
                                                          uint4* __restrict__ clear, int64_t clear16, const unsigned long long* __restrict__ slots,
                                                          double* __restrict__ m_out)
{
    constexpr int P0 = COARSE ? 0 : FIRST, P1 = COARSE ? 3 : SORT_PASSES;
    __shared__ uint32_t s_h[SORT_PASSES][SORT_BINS];
    double m = 0.0;
    if constexpr (COARSE) { int f; fold_slots(slots, m, f); if (blockIdx.x == 0 && threadIdx.x == 0) *m_out = m; }
    for (int64_t i = (int64_t)blockIdx.x * KF_BLOCK + threadIdx.x; i < clear16; i += (int64_t)gridDim.x * KF_BLOCK) clear[i] = make_uint4(0u, 0u, 0u, 0u);
    for (int i = threadIdx.x; i < SORT_PASSES * SORT_BINS; i += KF_BLOCK) (&s_h[0][0])[i] = 0;
    __syncthreads();
    constexpr int KH_ILP = 4;
    const int64_t stride = (int64_t)gridDim.x * KF_BLOCK;
    for (int64_t i0 = (int64_t)blockIdx.x * KF_BLOCK + threadIdx.x; i0 < n; i0 += KH_ILP * stride) {
        double v[KH_ILP];
#pragma unroll
        for (int q = 0; q < KH_ILP; ++q) v[q] = i0 + q * stride < n ? pv.at(i0 + q * stride) : 0.0;
#pragma unroll
        for (int q = 0; q < KH_ILP; ++q) {
            if (i0 + q * stride >= n) break;
            const uint64_t k = sort_key_desc(v[q]);
            keys[i0 + q * stride] = k;
            const uint64_t dk = COARSE ? (uint64_t)sort_coarse(k, m) : k;
#pragma unroll
            for (int p = P0; p < P1; ++p) atomicAdd(&s_h[p][(dk >> (8 * p)) & 0xff], 1u);
        }
    }
    __syncthreads();
    for (int i = P0 * SORT_BINS + threadIdx.x; i < P1 * SORT_BINS; i += KF_BLOCK) { const uint32_t c = (&s_h[0][0])[i]; if (c) atomicAdd(hist + i, c); }
}

// K10d key pass: the keys + the histogram of the coarse key's top SORT_FINE_BITS bits (the "fine bins": 128 per binade of the
// distance from the maximum) -- what k_sort_pass<2> cuts into SORT_BINS buckets of (nearly) equal counts.
template <int FB>
__global__ __launch_bounds__(KF_BLOCK) void k_sort_keys_fine(PrioView pv, int64_t n, uint64_t* __restrict__ keys, uint32_t* __restrict__ fine,
                                                          uint4* __restrict__ clear, int64_t clear16, const unsigned long long* __restrict__ slots,
                                                          double* __restrict__ m_out)
{
    __shared__ uint32_t s_h[1 << FB];
    double m; int f; fold_slots(slots, m, f);
    if (blockIdx.x == 0 && threadIdx.x == 0) *m_out = m;
    for (int64_t i = (int64_t)blockIdx.x * KF_BLOCK + threadIdx.x; i < clear16; i += (int64_t)gridDim.x * KF_BLOCK) clear[i] = make_uint4(0u, 0u, 0u, 0u);
    for (int i = threadIdx.x; i < (1 << FB); i += KF_BLOCK) s_h[i] = 0;
    __syncthreads();
    constexpr int KH_ILP = 4;
    const int64_t stride = (int64_t)gridDim.x * KF_BLOCK;
    for (int64_t i0 = (int64_t)blockIdx.x * KF_BLOCK + threadIdx.x; i0 < n; i0 += KH_ILP * stride) {
        double v[KH_ILP];
#pragma unroll
        for (int q = 0; q < KH_ILP; ++q) v[q] = i0 + q * stride < n ? pv.at(i0 + q * stride) : 0.0;
#pragma unroll
        for (int q = 0; q < KH_ILP; ++q) {
            if (i0 + q * stride >= n) break;
            const uint64_t k = sort_key_desc(v[q]);
            keys[i0 + q * stride] = k;
            atomicAdd(&s_h[sort_coarse(k, m) >> (SORT_COARSE_BITS - FB)], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (1 << FB); i += KF_BLOCK) { const uint32_t c = s_h[i]; if (c) atomicAdd(fine + i, c); }
}

// one digit pass.  vals_in == nullptr: the payload is the element's index (first pass).
#ifdef GPF_DBG_SORT
__device__ unsigned long long g_dbg_sort[8 * 4096];
#define DBG_SORT(slot) do { if (threadIdx.x == 0 && pass == 1 && blockIdx.x < 4096) g_dbg_sort[8 * blockIdx.x + (slot)] = wall_clock64(); } while (0)
#else
#define DBG_SORT(slot) do {} while (0)
#endif
// MODE 1: the digit is taken from the coarse key (sort_coarse with the maximum *m_ptr) instead of the key itself
// MODE 2 (K10d): ONE most-significant "digit" -- the bucket of the key's fine bin.  The fine bins (the coarse key's top SORT_FINE_BITS
//         bits, histogram `fine` from k_sort_keys_fine) are cut into SORT_BINS buckets of nearly equal counts: bucket(f) = the fine bins
//         whose exclusive count prefix lies in [b n / 256, (b + 1) n / 256) -- monotone in the key, at most n / 256 + (the fullest fine bin)
//         elements each.  Every workgroup derives the same table from the same histogram (integers); the first tile leaves the bucket
//         boundaries in bbase[0 .. SORT_BINS] for k_sort_buckets.
template <int MODE, int NB = SORT_BINS, int FB = SORT_FINE_BITS>
__global__ __launch_bounds__(SORT_BLOCK) void k_sort_pass(const uint64_t* __restrict__ keys_in, const int32_t* __restrict__ vals_in,
                                                     uint64_t* __restrict__ keys_out, int32_t* __restrict__ vals_out, int64_t n,
                                                     int pass, const uint32_t* __restrict__ hist, uint32_t* __restrict__ ticket,
                                                     uint64_t* __restrict__ desc, int32_t* __restrict__ timeout, const double* __restrict__ m_ptr,
                                                     const uint32_t* __restrict__ fine, uint32_t* __restrict__ bbase)
{
    DBG_SORT(0);
    constexpr bool COARSE = MODE == 1, PART = MODE == 2;
    using tab_t = typename std::conditional<(NB > 256), uint16_t, uint8_t>::type;
    constexpr int NFINE = 1 << FB, FINE_DEAD = sort_fine_dead(FB), NB_BITS = NB == 256 ? 8 : 9;
    static_assert(NB == (1 << NB_BITS) && NB <= SORT_BLOCK, "one thread per bucket");
    __shared__ tab_t s_tab[PART ? NFINE : 4];
    __shared__ uint32_t s_bcnt[NB];
    __shared__ uint32_t s_last[SORT_WAVES];
    double cm = 0.0;
    if constexpr (MODE != 0) cm = *m_ptr;
    auto digit_of = [&](uint64_t k) -> uint32_t {
        if constexpr (PART) return s_tab[sort_coarse(k, cm) >> (SORT_COARSE_BITS - FB)];
        else if constexpr (COARSE) return (sort_coarse(k, cm) >> (8 * pass)) & 0xffu;
        else return (uint32_t)(k >> (8 * pass)) & 0xffu;
    };
    __shared__ uint32_t s_cnt[SORT_WAVES][NB];      // per-wave digit counts, then exclusive offsets of the wave inside the tile's bin
    __shared__ uint32_t s_lstart[NB];           // first position of the bin in the tile's sorted order
    __shared__ int64_t s_gbase[NB];             // global position of the bin's first element of this tile, minus s_lstart
    __shared__ uint32_t s_scan[SORT_WAVES];
    __shared__ uint64_t s_keys[SORT_TILE];
    __shared__ int32_t s_vals[SORT_TILE];
    __shared__ uint32_t s_tile;
    const int tid = (int)threadIdx.x, lane = lane_id(), wv = wave_id();
    // tiles are taken by ticket (arrival order, not block index: a tile only ever waits for tiles that are already running).  ONE counter:
    // eight counters (way = blockIdx & 7 taking the tiles way, way + 8, ...), which avoid ~250 serialised same-address atomics in front
    // of the loads, measured 16.4 us per pass against 15.1 (profiles/r03_sort_experiments.txt) -- the interleaved arrival order costs
    // more in the look-back than the atomics cost
    // (tile = workgroup index when the whole launch is resident, without the same-address atomic in front of the loads, was measured too:
    //  16.8-17.5 us per pass against 15.8 -- arrival order is the better look-back order)
    if (tid == 0) s_tile = atomicAdd(ticket + pass * SORT_TICKET_WAYS, 1u);
    for (int i = tid; i < SORT_WAVES * NB; i += SORT_BLOCK) (&s_cnt[0][0])[i] = 0;
    const bool binthr = tid < NB;               // the first four waves double as "thread = bin"
    if constexpr (PART) {
        // fine bins -> buckets: thread t holds bins FPT t .. FPT t + FPT - 1
        constexpr int FPT = NFINE / SORT_BLOCK;
        static_assert(FPT % 4 == 0, "whole uint4 loads");
        uint32_t hq[FPT], hs = 0;
#pragma unroll
        for (int q = 0; q < FPT / 4; ++q) {
            const uint4 h4 = reinterpret_cast<const uint4*>(fine)[tid * (FPT / 4) + q];
            hq[4 * q] = h4.x; hq[4 * q + 1] = h4.y; hq[4 * q + 2] = h4.z; hq[4 * q + 3] = h4.w;
            hs += h4.x + h4.y + h4.z + h4.w;
        }
        uint32_t inc = hs;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) { const uint32_t o = __shfl_up(inc, d, WAVE); if (lane >= d) inc += o; }
        if (lane == WAVE - 1) { s_scan[wv] = inc; s_last[wv] = hq[FPT - 1]; }
        if (binthr) s_bcnt[tid] = (uint32_t)n;             // s_bcnt: where each bucket STARTS (a bucket no fine bin opens starts at the end)
        __syncthreads();
        uint32_t c = inc - hs;
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) if (w < wv) c += s_scan[w];
        // live keys: buckets 0 .. 254 by their exclusive count; dead keys (fine bins from FINE_DEAD on): the last bucket
        const double scale = (double)(NB - 1) / (double)n;
        auto bucket_at = [&](uint32_t cc) { const uint32_t bq = (uint32_t)((double)cc * scale); return bq > NB - 2 ? (uint32_t)(NB - 2) : bq; };   // (monotone in cc, the same in every workgroup: that is all it takes)
        // the bucket of the bin before this thread's first one (-1 in front of bin 0): a bin whose bucket differs from its
        // predecessor's opens every bucket in between at its own exclusive count
        uint32_t hprev = (uint32_t)__shfl_up((int)hq[FPT - 1], 1, WAVE);
        if (lane == 0 && wv > 0) hprev = s_last[wv - 1];
        static_assert(FINE_DEAD % FPT == 0, "a thread's fine bins are all live or all dead");
        const bool dead = FPT * tid >= FINE_DEAD;
        int pb = tid == 0 ? -1 : (FPT * tid - 1 >= FINE_DEAD ? NB - 1 : (int)bucket_at(c - hprev));
#pragma unroll
        for (int q = 0; q < FPT; ++q) {
            const int bk = dead ? NB - 1 : (int)bucket_at(c);
            s_tab[FPT * tid + q] = (tab_t)bk;
            for (int b_ = pb + 1; b_ <= bk; ++b_) s_bcnt[b_] = c;
            pb = bk;
            c += hq[q];
        }
        __syncthreads();
    }
    // exclusive scan of the digit's histogram: where each bin starts in the output
    uint32_t hbase = 0;
    if constexpr (PART) hbase = binthr ? s_bcnt[tid] : 0u;
    else {
        const uint32_t hv = binthr ? hist[pass * NB + tid] : 0u;
        uint32_t hinc = hv;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) { const uint32_t o = __shfl_up(hinc, d, WAVE); if (lane >= d) hinc += o; }
        if (binthr && lane == WAVE - 1) s_scan[wv] = hinc;
        __syncthreads();
        hbase = hinc - hv;
#pragma unroll
        for (int w = 0; w < NB / WAVE; ++w) if (w < wv) hbase += s_scan[w];
    }
    const int64_t tile = s_tile;
    const int64_t t0 = tile * SORT_TILE;
    if constexpr (PART) {
        if (tile == 0 && binthr) { bbase[tid] = hbase; if (tid == 0) bbase[NB] = (uint32_t)n; }
    }
    DBG_SORT(1);
    // ---- load (wave-striped: element = t0 + wave * 1024 + item * 64 + lane), rank inside the wave by digit
    uint64_t key[SORT_ITEMS]; int32_t val[SORT_ITEMS]; uint32_t rank[SORT_ITEMS];
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; ++it) {
        const int64_t i = t0 + wv * (WAVE * SORT_ITEMS) + it * WAVE + lane;
        key[it] = i < n ? keys_in[i] : ~0ull;
        val[it] = i < n ? (vals_in ? vals_in[i] : (int32_t)i) : 0;
    }
    const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    if (key[0] == 1234567ull) DBG_SORT(7);                 // (forces the loads to have landed before the next stamp is meaningful)
    DBG_SORT(2);
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; ++it) {
        const int64_t i = t0 + wv * (WAVE * SORT_ITEMS) + it * WAVE + lane;
        const bool valid = i < n;
        const uint32_t d = digit_of(key[it]);
        uint64_t peers = __ballot(valid);                 // lanes with the same digit (invalid lanes take no part)
#pragma unroll
        for (int b = 0; b < NB_BITS; ++b) { const uint64_t m = __ballot((d >> b) & 1u); peers &= ((d >> b) & 1u) ? m : ~m; }
        const uint32_t prev = s_cnt[wv][d];
        rank[it] = prev + (uint32_t)__popcll(peers & lt_mask);
        __builtin_amdgcn_wave_barrier();
        if (valid && (peers & lt_mask) == 0) s_cnt[wv][d] = prev + (uint32_t)__popcll(peers);     // the group's lowest lane
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    DBG_SORT(3);
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
        if (binthr) __hip_atomic_store(desc + ((size_t)pass * (nt_ + ng_ + nsg_) + (size_t)tile) * NB + tid, SORT_VALID | tcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    uint32_t lstart = linc - tcnt;
#pragma unroll
    for (int w = 0; w < NB / WAVE; ++w) if (w < wv) lstart += s_scan[w];
    if (binthr) s_lstart[tid] = lstart;
    __syncthreads();
    // ---- reorder inside the tile: afterwards every digit's run leaves as contiguous stores
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; ++it) {
        const int64_t i = t0 + wv * (WAVE * SORT_ITEMS) + it * WAVE + lane;
        if (i < n) {
            const uint32_t d = digit_of(key[it]);
            const uint32_t lp = s_lstart[d] + s_cnt[wv][d] + rank[it];
            s_keys[lp] = key[it]; s_vals[lp] = val[it];
        }
    }
    DBG_SORT(4);
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
        uint64_t* const agg = desc + ((size_t)pass * (nt + ng + nsg)) * NB + tid;      // this pass, this bin
        uint64_t* const gt = agg + nt * NB;
        uint64_t* const pre = gt + ng * NB;
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
            for (int e = 0; e < 15; ++e) v[e] = e < cnt ? __hip_atomic_load(const_cast<uint64_t*>(p + (size_t)e * NB), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : SORT_VALID;
#pragma unroll
            for (int e = 0; e < 15; ++e) acc += (v[e] & SORT_VALID) ? (v[e] & SORT_VAL) : wait_word(p + (size_t)e * NB);
            return acc;
        };
        const int64_t k = tile & 15, g = tile >> 4, gk = g & 15, sg = tile >> 8;
        const uint64_t in_group = sum_words(agg + (size_t)(tile - k) * NB, (int)k);
        if (k == 15) __hip_atomic_store(gt + (size_t)g * NB, SORT_VALID | (in_group + tcnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint64_t e_ = in_group + sum_words(gt + (size_t)(g - gk) * NB, (int)gk);
        if (sg > 0) e_ += wait_word(pre + (size_t)(sg - 1) * NB);
        if ((tile & 255) == 255) __hip_atomic_store(pre + (size_t)sg * NB, SORT_VALID | (e_ + tcnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        excl = e_;
    }
    if (binthr) s_gbase[tid] = (int64_t)hbase + (int64_t)excl - (int64_t)lstart;
    __syncthreads();
    DBG_SORT(5);
    const int64_t nvalid = n - t0 < SORT_TILE ? n - t0 : SORT_TILE;
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; ++k) {
        const int lp = k * SORT_BLOCK + tid;
        if (lp < nvalid) {
            const uint64_t kk = s_keys[lp];
            const int64_t g = s_gbase[digit_of(kk)] + lp;
            keys_out[g] = kk; vals_out[g] = s_vals[lp];
        }
    }
    DBG_SORT(6);
}

// ----------------------------------------------------------------------------- K10c: three coarse passes + a finish
// The log-priorities of a filter are continuous and live in a few binades below their maximum.  THREE stable passes over the 24-bit
// coarse key (sort_coarse, gpf_k_common.hpp: the distance from the maximum as {5-bit binade | 19 mantissa bits}) leave almost every
// key where it belongs; what is left are short runs of keys that share their coarse key (~1 per bucket in the populated binades at
// 10^6 particles), in index order.  k_sort_finish orders each such run by the full 64-bit key -- every element counts, inside its run,
// the smaller keys plus the equal ones before it (stable) -- and copies everything else through: one streaming pass instead of five
// latency-chain passes.  A run of more than SORT_RUN_MAX + 1 elements (equal or nearly equal weights; many weights further than 2^9 below the
// maximum, e.g. -inf) raises a flag; the last tile publishes it to pinned host memory and the host re-sorts with all eight passes over
// the key itself.  (Round 3, first form: four passes over the HIGH 32 key bits -- sign, 11 exponent bits, 20 mantissa bits -- + the
// finish: 4 x 15.2 us of passes; the coarse key spends its exponent bits on the binades that are in use.)
constexpr int SORT_RUN_MAX = 48;                    // elements of a run to either side of an element that the finish looks at
#ifndef GPF_FIN_BLOCK
#define GPF_FIN_BLOCK 1024
#endif
// (with the completion counters of the first build on ONE 128-byte line, 512- / 256-thread workgroups took 16.8 / 22.1 us against 14.8:
// atomics to the same LINE serialise at ~10 ns each; on separate lines every shape takes 14.0-14.3 us, profiles/r03_sort_experiments.txt)
constexpr int FIN_BLOCK = GPF_FIN_BLOCK, FIN_TILE = 4 * FIN_BLOCK, FIN_HALO = SORT_RUN_MAX + 1;
static __global__ __launch_bounds__(FIN_BLOCK) void k_sort_finish(const uint64_t* __restrict__ keys_in, const int32_t* __restrict__ vals_in,
                                                           uint64_t* __restrict__ keys_out, int32_t* __restrict__ vals_out, int64_t n,
                                                           uint32_t* __restrict__ done, int64_t* host_flag, int64_t ticket, const double* __restrict__ m_ptr)
{
    __shared__ uint64_t s_k[FIN_TILE + 2 * FIN_HALO];
    __shared__ uint32_t s_c[FIN_TILE + 2 * FIN_HALO];                       // the coarse keys: a run = neighbours with equal coarse keys
    const int tid = (int)threadIdx.x;
    const int64_t t0 = (int64_t)blockIdx.x * FIN_TILE;
    const double cm = *m_ptr;
    for (int p = tid; p < FIN_TILE + 2 * FIN_HALO; p += FIN_BLOCK) {
        const int64_t g = t0 - FIN_HALO + p;
        const uint64_t k = (g >= 0 && g < n) ? keys_in[g] : 0ull;
        s_k[p] = k; s_c[p] = sort_coarse(k, cm);
    }
    int32_t val[FIN_TILE / FIN_BLOCK];                                     // the payloads: in flight while the runs are examined
#pragma unroll
    for (int it = 0; it < FIN_TILE / FIN_BLOCK; ++it) { const int64_t g = t0 + it * FIN_BLOCK + tid; val[it] = g < n ? vals_in[g] : 0; }
    __syncthreads();
    bool too_long = false;
#pragma unroll
    for (int it = 0; it < FIN_TILE / FIN_BLOCK; ++it) {
        const int li = it * FIN_BLOCK + tid;                               // consecutive lanes, consecutive elements: coalesced copy-through
        const int64_t g = t0 + li;
        if (g >= n) continue;
        const int c = li + FIN_HALO;
        const uint64_t key = s_k[c];
        const uint32_t hi = s_c[c];
        // the run of equal coarse keys around this element: lc elements to the left, rc to the right (inside the array)
        int lc = 0, rc = 0;
        while (lc < SORT_RUN_MAX + 1 && g - lc - 1 >= 0 && s_c[c - lc - 1] == hi) ++lc;
        while (rc < SORT_RUN_MAX + 1 && g + rc + 1 < n && s_c[c + rc + 1] == hi) ++rc;
        int64_t pos = g;
        if (lc + rc > 0) {
            // a run of more than SORT_RUN_MAX + 1 elements is left as it stands and flagged (the host re-sorts).  EVERY element of such
            // a run sees that (each looks SORT_RUN_MAX + 1 to either side), so the output is a permutation of the input either way:
            // the weight sums taken over it -- and the log-ML update -- are right even when the order is not
            if (lc + rc > SORT_RUN_MAX) too_long = true;
            else {
                int rank = 0;
                for (int q = -lc; q <= rc; ++q) {
                    const uint64_t k2 = s_k[c + q];
                    rank += (k2 < key || (k2 == key && q < 0)) ? 1 : 0;
                }
                pos = g - lc + rank;
            }
        }
        keys_out[pos] = key; vals_out[pos] = val[it];
    }
    // completion: done[0] = "a run was too long", done[32] = groups finished, done[32 (2 + i)] = workgroups of group i (= blockIdx & 15)
    // finished -- two levels on separate 128-byte lines, so that no line sees more than gridDim / 16 (+ 16) atomics
    if (__syncthreads_or((int)too_long) && tid == 0) { atomicOr(done, 1u); __threadfence(); }
    if (tid == 0) {
        const uint32_t grp = blockIdx.x & 15u, members = (gridDim.x - grp + 15u) / 16u, groups = gridDim.x < 16u ? gridDim.x : 16u;
        if (atomicAdd(done + (2 + grp) * SORT_DONE_STRIDE, 1u) == members - 1 && atomicAdd(done + SORT_DONE_STRIDE, 1u) == groups - 1) {
            const uint32_t v = __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // verdict and ticket in ONE word (ticket << 1 | flag): a relaxed store needs no release -- a system-scope release at the end of a
            // kernel that has just written 12 MB would first write this XCD's L2 back
            __hip_atomic_store(host_flag + 1, (ticket << 1) | (int64_t)(v & 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ----------------------------------------------------------------------------- K10d: key pass + ONE partition pass + a sort per bucket in LDS
// sort_particles=true for filters of up to BK_MAX_N particles.  k_sort_keys_fine (keys + the fine-bin histogram), k_sort_pass<2> (one
// stable onesweep partition into SORT_BINS buckets of nearly equal counts: every bucket is a contiguous range of positions AND of
// keys), then this kernel: one workgroup per bucket orders its <= BK_CAP keys inside LDS and writes them to their final places.
//   sub-key : the distance from the maximum, linear between the bucket's smallest and largest, 14 bits (a few thousand keys over 16 384 values);
//   count   : one LDS atomic per key on the histogram of the sub-keys (its return value = the key's arrival number in its bin), an
//             exclusive scan of the histogram, ord[bin start + arrival number] = the key's position in the bucket;
//   rank    : a key alone in its bin stands at the bin's start; the others count, among their bin's members, the smaller full 64-bit keys
//             + the equal ones that stood before them in the bucket (the partition is stable, so that is index order: the sort is stable
//             although the arrival numbers are not ordered);
//   output  : keys and payloads go to their positions in LDS, then leave as coalesced stores.
// A bin of more than BK_RUN_MAX keys (many equal or nearly equal weights) or a bucket of more than BK_CAP keys (a fine bin fuller than
// the slack) stays in SOME order -- still a permutation of the input: the weight sums taken over it are right -- and raises the
// flag of k_sort_finish (pinned host word; the host then sorts with all eight passes over the full key).
// Three launches against five for the three coarse passes + finish: the two lower coarse passes and the finish's pass over global
// memory happen inside LDS, without ballots or per-wave counters (every key is independent of the others until the scan).
#ifdef GPF_DBG_SORT
__device__ unsigned long long g_dbg_bk[8 * 256];
#define DBG_BK(slot) do { if (threadIdx.x == 0) g_dbg_bk[8 * blockIdx.x + (slot)] = wall_clock64(); } while (0)
#else
#define DBG_BK(slot) do {} while (0)
#endif
constexpr int BK_BLOCK = 1024, BK_WAVES = BK_BLOCK / WAVE, BK_ITEMS = 8, BK_CAP = BK_BLOCK * BK_ITEMS;
constexpr int BK_SUB_BITS = 14, BK_SUB = 1 << BK_SUB_BITS, BK_SUB_PER = BK_SUB / BK_BLOCK;   // (16-bit counters, two to a word: 32 KB)
constexpr int BK_RUN_MAX = 1024;                              // keys of one bin that are still ranked (quadratic work inside the bin)
constexpr int64_t BK_MEAN_MAX = 4608;                          // mean bucket <= 4608 keys: BK_CAP - 4608 left for the fullest fine bin
constexpr int64_t BK_NARROW_N = (int64_t)SORT_BINS * BK_MEAN_MAX, BK_MAX_N = (int64_t)SORT_BINS_WIDE * BK_MEAN_MAX;   // 1 179 648 / 2 359 296 particles
static __global__ __launch_bounds__(BK_BLOCK) void k_sort_buckets(uint64_t* keys, int32_t* vals, int64_t n,
                                                           const uint32_t* __restrict__ bbase, uint32_t* __restrict__ done, int64_t* host_flag,
                                                           int64_t ticket, const double* __restrict__ m_ptr)
{
    // in place: a workgroup has its whole bucket in registers / LDS before it stores anything, and buckets are disjoint ranges
    const uint64_t* const keys_in = keys; const int32_t* const vals_in = vals;
    uint64_t* const keys_out = keys; int32_t* const vals_out = vals;
    __shared__ uint64_t s_key[BK_CAP];                   // by position in the bucket; at the end by final position
    __shared__ int32_t s_val[BK_CAP];                    // by final position
    __shared__ __attribute__((aligned(16))) uint32_t s_binw[BK_SUB / 2 + 4];          // 16-bit counts (<= BK_CAP) two to a word, then exclusive starts (+ the total)
    uint16_t* const s_bin = reinterpret_cast<uint16_t*>(s_binw);
    __shared__ uint16_t s_ord[BK_CAP];                   // bin start + arrival number -> position in the bucket
    __shared__ uint32_t s_scan[BK_WAVES];
    __shared__ double s_mn[BK_WAVES], s_mx[BK_WAVES];
    const int tid = (int)threadIdx.x, lane = lane_id(), wv = wave_id();
    DBG_BK(0);
    const int64_t b0 = bbase[blockIdx.x];
    const int64_t Lg = blockIdx.x == gridDim.x - 1 ? 0 : (int64_t)bbase[blockIdx.x + 1] - b0;     // (the last bucket: the dead keys, already in place)
    const double cm = *m_ptr;
    bool too_long = false;
    if (Lg > BK_CAP) too_long = true;                   // (stays as the partition left it)
    else if (Lg > 0) {
        const int L = (int)Lg;
        uint64_t key[BK_ITEMS]; int32_t val[BK_ITEMS]; uint32_t co[BK_ITEMS];
        double dist[BK_ITEMS];                           // the distance from the maximum (< 64: live keys only in these buckets)
        double dmin = __builtin_huge_val(), dmax = -__builtin_huge_val();
#pragma unroll
        for (int it = 0; it < BK_ITEMS; ++it) {
            const int i = it * BK_BLOCK + tid;
            key[it] = 0ull; val[it] = 0;
            if (i < L) { key[it] = keys_in[b0 + i]; val[it] = vals_in[b0 + i]; }
        }
#pragma unroll
        for (int q = 0; q < BK_SUB_PER / 2; ++q) s_binw[q * BK_BLOCK + tid] = 0u;
#pragma unroll
        for (int it = 0; it < BK_ITEMS; ++it) {
            const int i = it * BK_BLOCK + tid;
            dist[it] = 0.0;
            if (i < L) {
                s_key[i] = key[it];
                dist[it] = cm - sort_key_value(key[it]);
                dmin = dist[it] < dmin ? dist[it] : dmin;
                dmax = dist[it] > dmax ? dist[it] : dmax;
            }
        }
        dmax = wave_max_f64(dmax); dmin = -wave_max_f64(-dmin);
        DBG_BK(1);
        if (lane == 0) { s_mn[wv] = dmin; s_mx[wv] = dmax; }
        __syncthreads();
        DBG_BK(2);
#pragma unroll
        for (int w = 0; w < BK_WAVES; ++w) { dmin = s_mn[w] < dmin ? s_mn[w] : dmin; dmax = s_mx[w] > dmax ? s_mx[w] : dmax; }
        // the sub-key: the distance from the maximum, LINEAR between the bucket's extremes (a bucket is a narrow quantile slice of the
        // weights: their density is nearly flat across it), BK_SUB bins.  Weakly monotone in the key (m - v rounds monotonically, so do the
        // subtraction, the product and the truncation), which is all the ranking below needs.  (First form: the key's own bits, shifted --
        // linear in the weight inside a binade, but a bucket whose weights straddle 0 spans 2^62 key values and collapses into two bins.)
        const double scale = dmax > dmin ? (double)(BK_SUB - 1) / (dmax - dmin) : 0.0;
        uint32_t arr[BK_ITEMS];
#pragma unroll
        for (int it = 0; it < BK_ITEMS; ++it) {
            const int i = it * BK_BLOCK + tid;
            const uint32_t sk = i < L ? (uint32_t)((dist[it] - dmin) * scale) : 0u;
            co[it] = sk > (uint32_t)(BK_SUB - 1) ? (uint32_t)(BK_SUB - 1) : sk;
            const int hsh = 16 * (int)(co[it] & 1u);
            arr[it] = i < L ? (atomicAdd(&s_binw[co[it] >> 1], 1u << hsh) >> hsh) & 0xffffu : 0u;
        }
        __syncthreads();
        // exclusive scan of the BK_SUB counts: thread t owns bins BK_SUB_PER t .. BK_SUB_PER t + BK_SUB_PER - 1 (BK_SUB_PER / 2 packed words)
        static_assert(BK_SUB_PER % 8 == 0, "whole uint4 words");
        uint32_t cnt[BK_SUB_PER], tsum = 0;
#pragma unroll
        for (int q = 0; q < BK_SUB_PER / 8; ++q) {
            const uint4 w4 = reinterpret_cast<const uint4*>(s_binw)[(BK_SUB_PER / 8) * tid + q];
            const uint32_t ww[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) { cnt[8 * q + 2 * u] = ww[u] & 0xffffu; cnt[8 * q + 2 * u + 1] = ww[u] >> 16; tsum += (ww[u] & 0xffffu) + (ww[u] >> 16); }
        }
        uint32_t inc = tsum;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) { const uint32_t o = __shfl_up(inc, d, WAVE); if (lane >= d) inc += o; }
        if (lane == WAVE - 1) s_scan[wv] = inc;
        __syncthreads();
        uint32_t ex = inc - tsum;
#pragma unroll
        for (int w = 0; w < BK_WAVES; ++w) if (w < wv) ex += s_scan[w];
#pragma unroll
        for (int q = 0; q < BK_SUB_PER / 8; ++q) {
            uint32_t ww[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const uint32_t lo = ex; ex += cnt[8 * q + 2 * u]; ww[u] = lo | (ex << 16); ex += cnt[8 * q + 2 * u + 1]; }
            reinterpret_cast<uint4*>(s_binw)[(BK_SUB_PER / 8) * tid + q] = make_uint4(ww[0], ww[1], ww[2], ww[3]);
        }
        if (tid == BK_BLOCK - 1) s_bin[BK_SUB] = (uint16_t)ex;
        __syncthreads();
#pragma unroll
        for (int it = 0; it < BK_ITEMS; ++it) {
            const int i = it * BK_BLOCK + tid;
            if (i < L) s_ord[s_bin[co[it]] + arr[it]] = (uint16_t)i;
        }
        __syncthreads();
        DBG_BK(3);
        uint32_t pos[BK_ITEMS];
#pragma unroll
        for (int it = 0; it < BK_ITEMS; ++it) {
            const int i = it * BK_BLOCK + tid;
            pos[it] = 0u;
            if (i >= L) continue;
            const uint32_t st = s_bin[co[it]], c = s_bin[co[it] + 1] - st;
            uint32_t rank = arr[it];                      // (a bin that is too full keeps its arrival order: distinct positions all the same)
            if (c > (uint32_t)BK_RUN_MAX) too_long = true;
            else if (c > 1) {
                rank = 0;
                for (uint32_t q = 0; q < c; q += 4) {    // (four independent member reads in flight; past the bin's end: the element itself, which counts 0)
                    uint32_t i2[4]; uint64_t k2[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) i2[u] = q + u < c ? (uint32_t)s_ord[st + q + u] : (uint32_t)i;
#pragma unroll
                    for (int u = 0; u < 4; ++u) k2[u] = s_key[i2[u]];
#pragma unroll
                    for (int u = 0; u < 4; ++u) rank += (k2[u] < key[it] || (k2[u] == key[it] && i2[u] < (uint32_t)i)) ? 1u : 0u;
                }
            }
            pos[it] = st + rank;
        }
        DBG_BK(4);
        __syncthreads();                                  // (every read of s_key by bucket position is done)
        DBG_BK(5);
#pragma unroll
        for (int it = 0; it < BK_ITEMS; ++it) {
            const int i = it * BK_BLOCK + tid;
            if (i < L) { s_key[pos[it]] = key[it]; s_val[pos[it]] = val[it]; }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < BK_ITEMS; ++it) {
            const int j = it * BK_BLOCK + tid;
            if (j < L) { keys_out[b0 + j] = s_key[j]; vals_out[b0 + j] = s_val[j]; }
        }
        DBG_BK(6);
#ifdef GPF_DBG_SORT
        if (tid == 0) g_dbg_bk[8 * blockIdx.x + 7] = (unsigned long long)L;
#endif
    }
    // completion and verdict: as k_sort_finish
    if (__syncthreads_or((int)too_long) && tid == 0) { atomicOr(done, 1u); __threadfence(); }
    if (tid == 0) {
        const uint32_t grp = blockIdx.x & 15u, members = (gridDim.x - grp + 15u) / 16u, groups = gridDim.x < 16u ? gridDim.x : 16u;
        if (atomicAdd(done + (2 + grp) * SORT_DONE_STRIDE, 1u) == members - 1 && atomicAdd(done + SORT_DONE_STRIDE, 1u) == groups - 1) {
            const uint32_t v = __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(host_flag + 1, (ticket << 1) | (int64_t)(v & 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

static __global__ void k_extract_column(const double* __restrict__ rows, int W, int col, int64_t n, double* __restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) out[i] = rows[i * W + col];
}
static __global__ void k_parents(const int32_t* __restrict__ anc, int64_t n, int64_t* __restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) out[i] = (int64_t)anc[i] + 1;
}
// get_log_norm_weights / get_norm_weights (utils.jl:100,103-107,148,156)
static __global__ void k_norm_weights(const double* __restrict__ lw, const WSum* ws, int K, int64_t n, int want_log,
                               double* __restrict__ out)
{
    const double m = ws->m;
    const double lse = lse_from(m, ws->S, K, ws->flags);
    const double Sd = (double)ws->S;
    // a NaN (or +Inf) among the weights: softmax is NaN everywhere (utils.jl:103-107 over logsumexp = NaN), not only where the weight is;
    // all -Inf: maximum = -Inf, vs .- maximum = NaN, NaN everywhere as well -- get_norm_weights is the PLAIN softmax; the uniform
    // fallback belongs to safe_softmax inside the resamplers (utils.jl:123-126)
    const bool bad = (ws->flags & (FLAG_NAN | FLAG_POSINF | FLAG_ALL_NEGINF)) != 0;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        if (want_log) out[i] = lw[i] - lse;
        else out[i] = bad ? __builtin_nan("") : (double)exp_fix(lw[i] - m, K) / Sd;
    }
}
static __global__ void k_debug_math(int which, const double* a, const double* b, int64_t n, uint64_t seed, uint32_t epoch,
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
            case 7: out[i] = neglog_u52(d2u(a[i])); break;                     // (the argument's BITS are the slot's 64-bit uniform)
            default: out[i] = 0.0;
        }
    }
}

} // namespace gpf
