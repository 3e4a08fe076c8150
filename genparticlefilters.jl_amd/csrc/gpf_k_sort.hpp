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

// ----------------------------------------------------------------------------- K10b: the same order by a sample sort (two data passes)
// Eight digit passes are eight latency chains (profiles/r02d_sort_phases.txt).  For the sizes a GPU holds per shard in the
// BASELINE configs (up to 2^20 particles) the same permutation comes from TWO passes over the data:
//   k_ssort_splitters  one workgroup: 8192 pseudo-random samples (one per stratum of the index range), sorted in LDS (bitonic),
//                      every 32nd becomes a splitter -> 256 buckets of ~n / 256 elements;
//   k_ssort_partition  every 4096-element tile classifies its keys against the splitters (LDS, 8 probes), reorders the tile by
//                      bucket in LDS and appends each bucket's run to the bucket's fixed-capacity region (one atomic per tile and
//                      bucket; the order inside a region is arbitrary);
//   k_ssort_buckets    one workgroup per bucket sorts its region in LDS (8 keys per lane in registers, then in-place merge-path
//                      rounds) and writes it to its final place (the exclusive sum of the bucket counts).
// Everything compares the COMPOSITE (key, index): a total order, so no pass has to be stable, equal keys split across buckets by
// index (a million equal weights still give 256 equal buckets), and the result is exactly the stable descending sort of K10.
// A bucket that outgrows its region (sampling variance: P ~ 1e-7 per sort at 32 samples per bucket) raises a flag in pinned host
// memory; the host reads it while the bucket sorts run and re-sorts with the eight-pass radix sort.
constexpr int SS_BUCKETS = 256, SS_OVERSAMPLE = 32, SS_SAMPLES = SS_BUCKETS * SS_OVERSAMPLE;   // 8192
constexpr int SS_CAP = 8192;                        // elements per bucket region = elements a workgroup sorts in LDS
constexpr int SS_BLOCK = 1024, SS_TILE = 4096;
constexpr int64_t SS_MIN_N = 1 << 17, SS_MAX_N = (int64_t)SS_BUCKETS * (SS_CAP / 2);   // mean bucket <= half its region
struct SSortArgs {
    uint64_t* skeys; int32_t* sidx;                 // [255] splitters (composite)
    uint64_t* rkeys; int32_t* ridx;                 // [SS_BUCKETS][SS_CAP] bucket regions
    uint32_t* cursor;                               // [SS_BUCKETS] elements appended so far (cleared by k_ssort_splitters)
    uint32_t* done;                                 // tiles finished (cleared by k_ssort_splitters)
    int64_t* host_flag; int64_t ticket;             // pinned {overflow, ticket}
};
__device__ __forceinline__ bool ss_less(uint64_t k1, int32_t i1, uint64_t k2, int32_t i2) { return k1 < k2 || (k1 == k2 && i1 < i2); }

__global__ __launch_bounds__(SS_BLOCK) void k_ssort_splitters(PrioView pv, int64_t n, uint64_t seed, uint32_t epoch, SSortArgs a)
{
    __shared__ uint64_t s_k[SS_SAMPLES];
    __shared__ int32_t s_i[SS_SAMPLES];
    const int tid = (int)threadIdx.x;
    if (tid < SS_BUCKETS) a.cursor[tid] = 0;
    if (tid == 0) *a.done = 0;
    // sample s: one pseudo-random element of stratum [s n / 8192, (s + 1) n / 8192) of the index range (a counter-based hash of
    // (s, epoch): periodic patterns in the weights cannot resonate with the sample positions)
    for (int s = tid; s < SS_SAMPLES; s += SS_BLOCK) {
        const int64_t lo = (int64_t)(((unsigned __int128)(uint64_t)s * (uint64_t)n) >> 13), hi = (int64_t)(((unsigned __int128)(uint64_t)(s + 1) * (uint64_t)n) >> 13);
        uint64_t x = ((uint64_t)s << 32 | epoch) * 0x9E3779B97F4A7C15ull + seed;
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        const int64_t i = lo + (int64_t)(x % (uint64_t)(hi > lo ? hi - lo : 1));
        s_k[s] = sort_key_desc(pv.at(i)); s_i[s] = (int32_t)i;
    }
    __syncthreads();
    // bitonic sort of the 8192 composites (4 compare-exchanges per thread and stage)
    for (int size = 2; size <= SS_SAMPLES; size <<= 1) {
        for (int ls = 31 - __builtin_clz(size >> 1); ls >= 0; --ls) {      // stride = 1 << ls
            const int stride = 1 << ls;
#pragma unroll
            for (int r = 0; r < SS_SAMPLES / 2 / SS_BLOCK; ++r) {
                const int c = tid + r * SS_BLOCK;                          // compare-exchange index
                const int lo_i = ((c >> ls) << (ls + 1)) | (c & (stride - 1)), hi_i = lo_i + stride;
                const bool up = (lo_i & size) == 0;
                const uint64_t k1 = s_k[lo_i], k2 = s_k[hi_i];
                const int32_t i1 = s_i[lo_i], i2 = s_i[hi_i];
                if (ss_less(k2, i2, k1, i1) == up) { s_k[lo_i] = k2; s_k[hi_i] = k1; s_i[lo_i] = i2; s_i[hi_i] = i1; }
            }
            __syncthreads();
        }
    }
    if (tid < SS_BUCKETS - 1) { a.skeys[tid] = s_k[SS_OVERSAMPLE * (tid + 1) - 1]; a.sidx[tid] = s_i[SS_OVERSAMPLE * (tid + 1) - 1]; }
}

__global__ __launch_bounds__(SS_BLOCK) void k_ssort_partition(PrioView pv, int64_t n, SSortArgs a)
{
    __shared__ uint64_t s_sk[SS_BUCKETS];
    __shared__ int32_t s_si[SS_BUCKETS];
    __shared__ uint32_t s_cnt[SS_BUCKETS], s_start[SS_BUCKETS], s_gbase[SS_BUCKETS], s_scan[SS_BUCKETS / WAVE];
    __shared__ uint64_t s_k[SS_TILE];
    __shared__ int32_t s_i[SS_TILE];
    __shared__ uint16_t s_b[SS_TILE];
    const int tid = (int)threadIdx.x, lane = lane_id(), wv = wave_id();
    if (tid < SS_BUCKETS) {
        s_sk[tid] = tid < SS_BUCKETS - 1 ? a.skeys[tid] : ~0ull;         // (a sentinel above every key: never read as a splitter)
        s_si[tid] = tid < SS_BUCKETS - 1 ? a.sidx[tid] : 0x7fffffff;
        s_cnt[tid] = 0;
    }
    __syncthreads();
    const int64_t t0 = (int64_t)blockIdx.x * SS_TILE;
    constexpr int IT = SS_TILE / SS_BLOCK;
    uint64_t key[IT]; int32_t idx[IT]; uint32_t bkt[IT], rnk[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int64_t i = t0 + it * SS_BLOCK + tid;
        idx[it] = (int32_t)i;
        bkt[it] = 0; rnk[it] = 0;
        if (i < n) {
            key[it] = sort_key_desc(pv.at(i));
            uint32_t c = 0;                                              // number of splitters below the element (255 = 2^8 - 1 of them)
#pragma unroll
            for (uint32_t h = 128; h >= 1; h >>= 1) c += ss_less(s_sk[c + h - 1], s_si[c + h - 1], key[it], idx[it]) ? h : 0u;
            bkt[it] = c;
            rnk[it] = atomicAdd(&s_cnt[c], 1u);
        }
    }
    __syncthreads();
    // where each bucket's run starts inside the tile (exclusive scan of the 256 counts), and in its region (one global atomic)
    uint32_t cnt = tid < SS_BUCKETS ? s_cnt[tid] : 0u, inc = cnt;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) { const uint32_t o = __shfl_up(inc, d, WAVE); if (lane >= d) inc += o; }
    if (tid < SS_BUCKETS && lane == WAVE - 1) s_scan[wv] = inc;
    __syncthreads();
    if (tid < SS_BUCKETS) {
        uint32_t st = inc - cnt;
        for (int w = 0; w < wv; ++w) st += s_scan[w];
        s_start[tid] = st;
        s_gbase[tid] = cnt ? atomicAdd(a.cursor + tid, cnt) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int64_t i = t0 + it * SS_BLOCK + tid;
        if (i < n) { const uint32_t p = s_start[bkt[it]] + rnk[it]; s_k[p] = key[it]; s_i[p] = idx[it]; s_b[p] = (uint16_t)bkt[it]; }
    }
    __syncthreads();
    const int64_t nvalid = n - t0 < SS_TILE ? n - t0 : SS_TILE;
    bool over = false;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int p = it * SS_BLOCK + tid;
        if (p < nvalid) {
            const uint32_t b = s_b[p];
            const uint32_t q = s_gbase[b] + ((uint32_t)p - s_start[b]);  // position inside the bucket's region
            if (q < (uint32_t)SS_CAP) { a.rkeys[(size_t)b * SS_CAP + q] = s_k[p]; a.ridx[(size_t)b * SS_CAP + q] = s_i[p]; }
            else over = true;
        }
    }
    // the last tile to finish tells the host whether every bucket stayed inside its region
    if (__syncthreads_or((int)over) && tid == 0) atomicOr(a.done, 0x80000000u);
    __shared__ uint32_t s_last;
    if (tid == 0) s_last = atomicAdd(a.done, 1u);                      // (only the flag travels: the regions are read by the NEXT kernel)
    __syncthreads();
    if (tid == 0 && (s_last & 0x7fffffffu) == gridDim.x - 1) {
        const uint32_t v = __hip_atomic_load(a.done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.host_flag, (int64_t)(v >> 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(a.host_flag + 1, a.ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// LDS index of element i: one pad element per 8 -- a thread's 8 consecutive elements (stride 64 B: a 16-way bank conflict for the
// 64 lanes of a wave) become stride 72 B (2-way), the merge's ~32-byte strides become ~36 bytes (conflict-free)
__device__ __forceinline__ int ss_pad(int i) { return i + (i >> 3); }
constexpr int SS_CAP_PAD = SS_CAP + SS_CAP / 8;
__global__ __launch_bounds__(SS_BLOCK) void k_ssort_buckets(SSortArgs a, uint64_t* __restrict__ keys_out, int32_t* __restrict__ order_out)
{
    __shared__ uint64_t s_k[SS_CAP_PAD];
    __shared__ int32_t s_i[SS_CAP_PAD];
    __shared__ uint32_t s_scan[SS_BUCKETS / WAVE];
    __shared__ uint32_t s_base;
    const int tid = (int)threadIdx.x, lane = lane_id(), wv = wave_id();
    const int b = (int)blockIdx.x;
    // this bucket's place in the output: the exclusive sum of the bucket counts
    {
        const uint32_t c = tid < SS_BUCKETS ? a.cursor[tid] : 0u;
        uint32_t inc = c;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) { const uint32_t o = __shfl_up(inc, d, WAVE); if (lane >= d) inc += o; }
        if (tid < SS_BUCKETS && lane == WAVE - 1) s_scan[wv] = inc;
        __syncthreads();
        if (tid == b) { uint32_t st = inc - c; for (int w = 0; w < wv; ++w) st += s_scan[w]; s_base = st; }
    }
    const uint32_t nb_raw = a.cursor[b];
    const int nb = (int)(nb_raw < (uint32_t)SS_CAP ? nb_raw : (uint32_t)SS_CAP);     // (an overflowing bucket: the host re-sorts everything)
    constexpr int E = SS_CAP / SS_BLOCK;                                 // 8 elements per thread
    int npad = E;                                                        // sorted length needed: smallest E 2^r >= nb
    while (npad < nb) npad <<= 1;
    // ---- load (coalesced), then every thread sorts its 8 consecutive elements in registers
    for (int p = tid; p < npad; p += SS_BLOCK) {
        const bool v = p < nb;
        s_k[ss_pad(p)] = v ? a.rkeys[(size_t)b * SS_CAP + p] : ~0ull;
        s_i[ss_pad(p)] = v ? a.ridx[(size_t)b * SS_CAP + p] : 0x7fffffff;
    }
    __syncthreads();
    uint64_t k[E]; int32_t ix[E];
    const int o0 = tid * E;
    const bool act = o0 < npad;
    if (act) {
#pragma unroll
        for (int e = 0; e < E; ++e) { k[e] = s_k[tid * (E + 1) + e]; ix[e] = s_i[tid * (E + 1) + e]; }     // ss_pad(8 t + e) = 9 t + e
#pragma unroll
        for (int e = 1; e < E; ++e) {                                    // insertion sort (fully unrolled: stays in registers)
#pragma unroll
            for (int f = e; f >= 1; --f) {
                if (ss_less(k[f], ix[f], k[f - 1], ix[f - 1])) { const uint64_t tk = k[f]; k[f] = k[f - 1]; k[f - 1] = tk; const int32_t ti = ix[f]; ix[f] = ix[f - 1]; ix[f - 1] = ti; }
            }
        }
#pragma unroll
        for (int e = 0; e < E; ++e) { s_k[tid * (E + 1) + e] = k[e]; s_i[tid * (E + 1) + e] = ix[e]; }
    }
    __syncthreads();
    // ---- merge rounds, in place: thread t produces outputs [8t, 8t + 8) of its pair of runs (merge path), everybody reads, barrier,
    //      everybody writes.  Runs start at multiples of 8, so ss_pad(base + i) = ss_pad(base) + ss_pad(i).
    for (int L = E; L < npad; L <<= 1) {
        if (act) {
            const int pair = o0 / (2 * L), off = o0 - pair * 2 * L;
            const uint64_t* A = s_k + ss_pad(pair * 2 * L); const int32_t* Ai = s_i + ss_pad(pair * 2 * L);
            const uint64_t* B = s_k + ss_pad(pair * 2 * L + L); const int32_t* Bi = s_i + ss_pad(pair * 2 * L + L);
            // i = number of A elements among the first `off` outputs: the smallest i with A[i] > B[off - i - 1] (composite order, no ties)
            int lo = off > L ? off - L : 0, hi = off < L ? off : L;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const int j = off - mid - 1;                              // 0 <= j < L
                if (ss_less(B[ss_pad(j)], Bi[ss_pad(j)], A[ss_pad(mid)], Ai[ss_pad(mid)])) hi = mid; else lo = mid + 1;
            }
            int i = lo, j = off - lo;
            uint64_t ka = i < L ? A[ss_pad(i)] : ~0ull, kb = j < L ? B[ss_pad(j)] : ~0ull;
            int32_t ia = i < L ? Ai[ss_pad(i)] : 0x7fffffff, ib = j < L ? Bi[ss_pad(j)] : 0x7fffffff;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const bool takeb = j < L && (i >= L || ss_less(kb, ib, ka, ia));
                if (takeb) { k[e] = kb; ix[e] = ib; ++j; kb = j < L ? B[ss_pad(j)] : ~0ull; ib = j < L ? Bi[ss_pad(j)] : 0x7fffffff; }
                else       { k[e] = ka; ix[e] = ia; ++i; ka = i < L ? A[ss_pad(i)] : ~0ull; ia = i < L ? Ai[ss_pad(i)] : 0x7fffffff; }
            }
        }
        __syncthreads();
        if (act) {
#pragma unroll
            for (int e = 0; e < E; ++e) { s_k[tid * (E + 1) + e] = k[e]; s_i[tid * (E + 1) + e] = ix[e]; }
        }
        __syncthreads();
    }
    const uint32_t base = s_base;
    for (int p = tid; p < nb; p += SS_BLOCK) { keys_out[base + p] = s_k[ss_pad(p)]; order_out[base + p] = s_i[ss_pad(p)]; }
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

} // namespace gpf
