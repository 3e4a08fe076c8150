// K5: ancestor searches (two-line core, key-table i.i.d. search, stratified streaming merge)  (part of gpf_kernels.hpp; include that header, not this file)
#pragma once

namespace gpf {
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
    const uint32_t* k32s; int sample;                                                           // ScanOut::k32s / sample (k_search_multi_s)
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
// NS independent slots per lane at once (8 NS line loads in flight per lane; the targets are re-read from the strip instead of kept):
// lds_wave holds NS x 64 entries
template <int NS>
__device__ __forceinline__ void coop_count_le(const uint64_t* const (&line)[NS], const uint64_t (&T)[NS], ulonglong2* lds_wave, int (&cnt)[NS])
{
    const int lane = lane_id(), sub = lane & 7, gbase = lane & ~7;
#pragma unroll
    for (int q = 0; q < NS; ++q) lds_wave[q * WAVE + lane] = make_ulonglong2(reinterpret_cast<uint64_t>(line[q]), T[q]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    ulonglong2 v[8 * NS];
#pragma unroll
    for (int r = 0; r < 8 * NS; ++r) v[r] = load_global_16(lds_wave[(r >> 3) * WAVE + gbase + (r & 7)].x, sub);
#pragma unroll
    for (int q = 0; q < NS; ++q) cnt[q] = 0;
#pragma unroll
    for (int r = 0; r < 8 * NS; ++r) {
        const uint64_t t = lds_wave[(r >> 3) * WAVE + gbase + (r & 7)].y;
        const int c = group8_sum((int)(v[r].x <= t) + (int)(v[r].y <= t));
        cnt[r >> 3] = sub == (r & 7) ? c : cnt[r >> 3];
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
// (struct ShardPlan: gpf_k_common.hpp)
// k_search_strat on a shard packs the exchange entries itself: [row | slot inside its shard << 32 | global ancestor id]
// extra = 1 (a prioritised resample, priority_fn = w -> alpha w, resample.jl:51-52): one more double per entry, log_ws = lw[a] - lp[a]
// (update_weights!, resample.jl:198) -- the receiver does not hold its ancestors' weights
struct PackOut { const double* rows; double* packed; int64_t capacity, gid0; int W; int extra; PrioView pv;
                 int32_t* own_anc; int me;      // own_anc != nullptr: entries for shard `me` itself are not packed -- their GLOBAL ancestor id goes to own_anc[slot inside the shard]
                 // ring.peers != nullptr (the slot-addressed receive windows, gpf_k_common.hpp): the entries for the OTHER shards are not packed either --
                 // [row | global ancestor id | seal] goes straight into the destination rank's window at the slot's local index (needs own_anc)
                 RingOut ring; };
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
    PackOut pack;                                                     // plan != nullptr && pack.packed: no ancestor array, packed rows instead
    int update_lml;                                                   // 0 for sub-state views (resample.jl:185-187); 2: whole-shard
                                                                      // sub-state, the kept mass goes to sc->lw_fill (resample.jl:210)
    int32_t* anc;
    int head_done;                                                    // residual: the deterministic head is written already (k_scan_residual2): tail slots only
    // GPF_RESAMPLE_MULTINOMIAL_SORTED (k_search_strat<true>): sp_g[t] = the gamma total of tile t of SP_TILE slots (k_sorted_gammas; the
    // merge kernel places its own tile: prefix and total of the <= SP_DIRECT_TILES entries in its prologue), or, for more tiles,
    // sp_vlo[t] = where tile t starts among the sorted 64-bit uniforms (k_sorted_tiles), [tiles + 1] entries
    const uint64_t* sp_g; const uint64_t* sp_vlo;
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
    const SearchTop st = search_prologue(a.w, a.c, METHOD == 1 && !a.head_done, a.ntiles, reinterpret_cast<uint64_t*>(smem));
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
    // (residual with the head already written by the scan: the slots [Ctot, n) only)
    const int64_t jfirst = (METHOD == 1 && a.head_done) ? (int64_t)Ctot : 0;
    if (METHOD == 1 && a.head_done) {
        // ... except the copies of the cells the scan listed as too many for one workgroup (HeadGiants): filled here, by the whole grid
        const HeadGiants& gi = a.sc->giants;
        const unsigned int w = gi.word;
        const unsigned int ng = (w >> 8) == (a.epoch & 0xffffffu) ? (w & 0xffu) : 0u;
        for (unsigned int e = 0; e < ng; ++e) {
            const uint64_t start = gi.start[e]; const uint32_t c_ = gi.cnt[e]; const int32_t cell = (int32_t)gi.cell[e];
            for (uint64_t q = (uint64_t)blockIdx.x * SBLOCK + threadIdx.x; q < c_; q += (uint64_t)gridDim.x * SBLOCK) a.anc[start + q] = cell;
        }
    }
    // two slots per lane and iteration (independent dependency chains); the loop is wave-uniform
    for (int64_t base = jfirst + (int64_t)blockIdx.x * 2 * SBLOCK; base < a.n; base += (int64_t)gridDim.x * 2 * SBLOCK) {
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
#ifndef GPF_MULTI_MIN_LOGG
#define GPF_MULTI_MIN_LOGG 0
#endif
__host__ inline int multi_logg(int64_t ntiles)
{
    for (int g = GPF_MULTI_MIN_LOGG; g <= 1; ++g) if (multi_lds_bytes(ntiles, g) <= (size_t)MULTI_LDS_BUDGET) return g;
    return -1;
}
// Beyond that (2.5 M particles) the levels stay those of 32-cell key groups and LDS keeps every (1 << s)-th key (ScanOut::k32s,
// written by the scan): the table search ends at a super-group of 32 << s cells, whose 1 << s keys are one more narrow read
// (16 bytes for s = 2: up to 5 M particles).  Measured against the two-line search (k_search<0>), multinomial, ns per slot:
// 3 M 19.8 / 25.6, 4 M 21.5 / 26.3, 5 M 24.1 / 27.5 (s = 2);  6 M 27.4 / 28.4, 8 M 31.0 / 29.9 (s = 3);  16 M 38.9 / 33.5 (s = 4):
// only s = 2 pays (wider super-groups cost more key reads than the narrow levels save).  0: not in this regime.
constexpr int MULTI_SAMPLE_MAX = 2;
__host__ inline int multi_sample(int64_t ntiles)
{
    if (multi_logg(ntiles) >= 0) return 0;
    for (int s = 2; s <= MULTI_SAMPLE_MAX; ++s) if (multi_lds_bytes(ntiles, s) <= (size_t)MULTI_LDS_BUDGET) return s;
    return 0;
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

// slots per lane and iteration of the key-table searches: 2 (two iterations per workgroup at 10^6 slots) measured 16.0 us against 17.1
// for 4 (one iteration) -- rocprofv3, alternating runs, round 3; 64-cell key groups at this size: 19.2 us
#ifndef GPF_MULTI_NS
#define GPF_MULTI_NS 2
#endif
#ifndef GPF_MULTI_ROW
#define GPF_MULTI_ROW 0
#endif
struct MultiTable { const uint32_t* keys; uint32_t ng, p2; float kscale; };      // the LDS key table of k_search_multi
constexpr uint32_t MULTI_WIN = 512;                // interpolation window of the key search

// the second half of the lookup: pos[u] = the key group that holds T[u] (number of groups that end at or below it); the
// target becomes a 16-bit offset inside the group, then two narrow reads.  key(i) = key of group i (LDS table or global level).
template <int LOGG, int NS>
__device__ __forceinline__ void multi_inside_k(const uint32_t (&g)[NS], const uint32_t (&klo_)[NS], const uint32_t (&khi_)[NS], const CdfLevels& w,
                                               int64_t n_cells, const uint64_t (&T)[NS], uint32_t (&idx)[NS]);
template <int LOGG, int NS, class KeyFn>
__device__ __forceinline__ void multi_inside(KeyFn&& key, uint32_t ng, const CdfLevels& w, int64_t n_cells, const uint64_t (&T)[NS],
                                             const uint32_t (&pos)[NS], uint32_t (&idx)[NS])
{
    uint32_t g[NS], klo[NS], khi[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        g[u] = pos[u] < ng ? pos[u] : ng - 1;
        klo[u] = g[u] ? key(g[u] - 1) : 0u; khi[u] = key(g[u]);
    }
    multi_inside_k<LOGG, NS>(g, klo, khi, w, n_cells, T, idx);
}
// ... with the group and the keys at its two ends given (g = key group of 32 << LOGG cells that holds T)
template <int LOGG, int NS>
__device__ __forceinline__ void multi_inside_k(const uint32_t (&g)[NS], const uint32_t (&klo_)[NS], const uint32_t (&khi_)[NS], const CdfLevels& w,
                                               int64_t n_cells, const uint64_t (&T)[NS], uint32_t (&idx)[NS])
{
    constexpr int G = 32 << LOGG, CS = G / 8;
    uint32_t qq[NS], run[NS];
#if GPF_MULTI_ROW
    // (experiment: the group's G offsets as ONE G*2-byte read -- one line fill and one round trip instead of coarse row + run)
    bool tie[NS]; bool anytie = false;
    {
        uint4 rw[NS][G / 8];
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const uint32_t klo = klo_[u], khi = khi_[u];
            const uint64_t kb = (uint64_t)klo << KEY_SHIFT;
            const uint64_t d = T[u] > kb ? T[u] - kb : 0;
            uint32_t q = (uint32_t)(d >> key_quant_shift(klo, khi));
            q = q < 65535u ? q : 65535u;
            qq[u] = q | (q << 16);
            const uint4* fp = reinterpret_cast<const uint4*>(w.off16 + (size_t)g[u] * (uint32_t)G);
#pragma unroll
            for (int e = 0; e < G / 8; ++e) rw[u][e] = fp[e];
        }
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            uint32_t lt = 0, ne = 0;
#pragma unroll
            for (int e = 0; e < G / 8; ++e) {
                lt += pk_lt(rw[u][e].x, qq[u]) + pk_lt(rw[u][e].y, qq[u]) + pk_lt(rw[u][e].z, qq[u]) + pk_lt(rw[u][e].w, qq[u]);
                ne += pk_ne(rw[u][e].x, qq[u]) + pk_ne(rw[u][e].y, qq[u]) + pk_ne(rw[u][e].z, qq[u]) + pk_ne(rw[u][e].w, qq[u]);
            }
            lt = (lt & 0xffffu) + (lt >> 16); ne = (ne & 0xffffu) + (ne >> 16);
            lt = lt < (uint32_t)G ? lt : (uint32_t)G - 1;
            tie[u] = ne != (uint32_t)G;
            anytie = anytie || tie[u];
            run[u] = lt / (uint32_t)CS;
            idx[u] = g[u] * (uint32_t)G + lt;
        }
    }
#else
    uint4 row[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        const uint32_t klo = klo_[u], khi = khi_[u];
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
#endif
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
// first half: pos[u] = number of table entries (one per G = 32 << GS cells) whose group ends at or below T[u]
template <int GS, int NS>
__device__ __forceinline__ void multi_find(const MultiTable& tb, const CdfLevels& w, const uint64_t (&T)[NS], uint32_t (&pos)[NS])
{
    constexpr int G = 32 << GS;
    constexpr uint32_t WIN = MULTI_WIN;
    uint32_t t[NS];
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
        // equal keys: the exact prefixes decide (rare: one key value in 2^32 S / (2^30 groups) per slot).  The run of groups that
        // share the key can be LONG -- all the mass on one particle leaves tens of thousands of groups with key 0 before it --
        // so it is bisected: its end from the key table (number of keys <= t), then the first group of the run whose exact end
        // prefix exceeds T.  (A linear walk cost 12.7 ms per launch on such weights.)
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            if (!(pos[u] < tb.ng && tb.keys[kpad(pos[u])] == t[u])) continue;
            uint32_t lo = pos[u], hi = tb.ng;                           // keys[lo] == t; find the first index with key > t
            {
                uint32_t a = lo + 1, b = tb.ng;
                while (a < b) { const uint32_t mid = (a + b) >> 1; if (tb.keys[kpad(mid)] <= t[u]) a = mid + 1; else b = mid; }
                hi = a;
            }
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (w.cdf[(int64_t)mid * G + (G - 1)] <= T[u]) lo = mid + 1; else hi = mid;
            }
            pos[u] = lo;
        }
    }
}
template <int LOGG, int NS>
__device__ __forceinline__ void multi_lookup(const MultiTable& tb, const CdfLevels& w, int64_t n_cells, const uint64_t (&T)[NS], uint32_t (&idx)[NS])
{
    uint32_t pos[NS];
    multi_find<LOGG, NS>(tb, w, T, pos);
    multi_inside<LOGG, NS>([&](uint32_t i) { return tb.keys[kpad(i)]; }, tb.ng, w, n_cells, T, pos, idx);
}

// block-collective: the key table into LDS, 16 B per lane from the scan's key level (every (1 << LOGG)-th key).  The loads are
// issued first, `between()` runs while they are in flight (the caller's first targets), then the table is written.
// multi_table_load_s: the shard's total S is itself the result of work that waits on memory (k_search_own: the gathered shard totals, a
// barrier) -- between() returns it, so that work runs behind the table's loads too instead of in front of them
template <int LOGG, class Between>
__device__ __forceinline__ MultiTable multi_table_load_s(const CdfLevels& w, int64_t ntiles, uint32_t* keys, Between&& between);
template <int LOGG, class Between>
__device__ __forceinline__ MultiTable multi_table_load(const CdfLevels& w, int64_t ntiles, uint64_t S, uint32_t* keys, Between&& between)
{
    return multi_table_load_s<LOGG>(w, ntiles, keys, [&]() -> uint64_t { between(); return S; });
}
template <int LOGG, class Between>
__device__ __forceinline__ MultiTable multi_table_load_s(const CdfLevels& w, int64_t ntiles, uint32_t* keys, Between&& between)
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
    const uint64_t S = between();
    tb.kscale = (float)tb.ng / (float)((S >> KEY_SHIFT) + 1);            // groups per key unit: where a key would sit were the CDF linear
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

// ---- the sampled regime (multi_sample): table entry = key of a super-group of 32 << SS cells
template <int SS>
__device__ __forceinline__ MultiTable multi_table_load_sampled(const CdfLevels& w, int64_t ntiles, uint64_t S, uint32_t* keys)
{
    constexpr int KT = (MULTI_LDS_BUDGET / 4 / 4 + SBLOCK - 1) / SBLOCK;
    MultiTable tb;
    tb.keys = keys;
    tb.ng = (uint32_t)multi_groups(ntiles, SS);                         // a multiple of 4 (64 >> SS per tile, SS <= 4)
    const uint32_t nq = tb.ng / 4;
    const uint4* src = reinterpret_cast<const uint4*>(w.k32s);
    uint4 kv[KT];
#pragma unroll
    for (int r = 0; r < KT; ++r) { const uint32_t q = threadIdx.x + (uint32_t)r * SBLOCK; if (q < nq) kv[r] = src[q]; }
    tb.p2 = 1;
    while (2 * tb.p2 <= tb.ng) tb.p2 *= 2;
    tb.kscale = (float)tb.ng / (float)((S >> KEY_SHIFT) + 1);
#pragma unroll
    for (int r = 0; r < KT; ++r) {
        const uint32_t q = threadIdx.x + (uint32_t)r * SBLOCK;
        if (q < nq) { uint32_t* d = keys + kpad(4 * q); d[0] = kv[r].x; d[1] = kv[r].y; d[2] = kv[r].z; d[3] = kv[r].w; }
    }
    __syncthreads();
    return tb;
}
template <int SS, int NS>
__device__ __forceinline__ void multi_lookup_sampled(const MultiTable& tb, const CdfLevels& w, int64_t n_cells, const uint64_t (&T)[NS], uint32_t (&idx)[NS])
{
    constexpr int NK = 1 << SS;                                         // 32-cell key groups per super-group
    uint32_t pos[NS];
    multi_find<SS, NS>(tb, w, T, pos);                                  // the super-group (ties on sampled keys: exact prefixes at super-group ends)
    const uint32_t ng0 = tb.ng << SS;
    uint32_t g[NS], klo[NS], khi[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        const uint32_t sg = pos[u] < tb.ng ? pos[u] : tb.ng - 1;
        const uint32_t t = (uint32_t)(T[u] >> KEY_SHIFT);
        uint32_t kk[NK];                                                // the super-group's keys: NK * 4 contiguous bytes
        const uint4* kp = reinterpret_cast<const uint4*>(w.k32 + ((size_t)sg << SS));
#pragma unroll
        for (int q = 0; q < NK / 4; ++q) { const uint4 v = kp[q]; kk[4 * q] = v.x; kk[4 * q + 1] = v.y; kk[4 * q + 2] = v.z; kk[4 * q + 3] = v.w; }
        uint32_t c = 0;
#pragma unroll
        for (int q = 0; q < NK; ++q) c += (uint32_t)(kk[q] < t);
        c = c < (uint32_t)NK ? c : (uint32_t)NK - 1;                  // (the super-group holds T: its last group ends above it)
        const uint32_t base = sg << SS;
        uint32_t kc = kk[0];                                            // key of group c, from registers
#pragma unroll
        for (int q = 1; q < NK; ++q) kc = (uint32_t)q == c ? kk[q] : kc;
        if (kc == t)                                                    // equal keys: the exact prefixes decide (rare; at most NK - 1 steps)
            while (c < (uint32_t)NK - 1 && w.k32[base + c] == t && w.cdf[(int64_t)(base + c) * 32 + 31] <= T[u]) ++c;
        g[u] = base + c < ng0 ? base + c : ng0 - 1;
        uint32_t lo = sg ? tb.keys[kpad(sg - 1)] : 0u, hi = kk[0];       // keys at the two ends of group base + c, picked from registers
#pragma unroll
        for (int q = 1; q < NK; ++q) { lo = (uint32_t)q == c ? kk[q - 1] : lo; hi = (uint32_t)q == c ? kk[q] : hi; }
        klo[u] = lo; khi[u] = hi;
    }
    multi_inside_k<0, NS>(g, klo, khi, w, n_cells, T, idx);
}
template <int SS>
__global__ __launch_bounds__(SBLOCK, 4) void k_search_multi_s(SearchArgs a)
{
    constexpr int NS = 2;                                               // (the super-group keys cost registers)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (a.update_lml && blockIdx.x == 0 && threadIdx.x == 0) resample_bookkeeping(a);
    const uint64_t S = a.ws->S;
    const MultiTable tb = multi_table_load_sampled<SS>(a.w, a.ntiles, S, reinterpret_cast<uint32_t*>(smem));
    const int64_t stride = (int64_t)gridDim.x * NS * SBLOCK;
    for (int64_t base = (int64_t)blockIdx.x * NS * SBLOCK; base < a.n; base += stride) {
        const int64_t j0 = base + NS * (int64_t)threadIdx.x;
        const uint32_t s0 = (uint32_t)(a.gid0 + j0);
        const Philox b0 = rng(a.seed, s0 >> 1, 0, a.epoch, TAG_RESAMPLE);
        const Philox b1 = (s0 & 1u) ? rng(a.seed, (s0 >> 1) + 1u, 0, a.epoch, TAG_RESAMPLE) : b0;          // kernel-uniform
        uint64_t T[NS] = {mulhi64(resample_pick(b0, s0), S), mulhi64(resample_pick(b1, s0 + 1u), S)};       // resample.jl:59
        uint32_t idx[NS];
        multi_lookup_sampled<SS, NS>(tb, a.w, a.n_cells, T, idx);
        if (j0 < a.n) a.anc[j0] = (int32_t)idx[0];
        if (j0 + 1 < a.n) a.anc[j0 + 1] = (int32_t)idx[1];
    }
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
#ifndef GPF_MONO_WIDE_MULT
#define GPF_MONO_WIDE_MULT 2
#endif
#ifndef GPF_MONO_WIDE2_MULT
#define GPF_MONO_WIDE2_MULT 64
#endif
#ifndef GPF_WIDE_NS
#define GPF_WIDE_NS 4
#endif
constexpr int MONO_WIDE_NS = GPF_WIDE_NS;
constexpr int64_t MONO_WIDE = GPF_MONO_WIDE_MULT * (int64_t)MJB;     // a cell range wider than this is streamed on the per-16 level ...
constexpr int64_t MONO_WIDE2 = GPF_MONO_WIDE2_MULT * (int64_t)MJB;   // ... and beyond this searched per slot

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
    if (guess < 0 && cnt <= 16 * MBLOCK) {
        // no usable guess (a sorted order: the CDF is far from linear) and the level is small: every thread reads 16 entries, the
        // whole level in ONE round trip of coalesced 16-byte loads, and the counts are exact
        ulonglong2 v[8];
        const ulonglong2* src = reinterpret_cast<const ulonglong2*>(arr);
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = (int64_t)(2 * (c * MBLOCK + tid)) < cnt ? src[c * MBLOCK + tid] : make_ulonglong2(~0ull, ~0ull);   // (cnt is even)
        between(Lq[0], Lq[1]);
        int c0 = 0, c1 = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            c0 += (int)(v[c].x <= Lq[0]) + (int)(v[c].y <= Lq[0]);
            c1 += (int)(v[c].x <= Lq[1]) + (int)(v[c].y <= Lq[1]);
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { c0 += __shfl_xor(c0, m, WAVE); c1 += __shfl_xor(c1, m, WAVE); }
        if (lane_id() == 0) { s_cnt[0][0][wave_id()] = c0; s_cnt[0][1][wave_id()] = c1; }
        __syncthreads();
        int64_t k0 = 0, k1 = 0;
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) { k0 += s_cnt[0][0][w]; k1 += s_cnt[0][1][w]; }
        A0 = k0; A1 = k1;
        return;
    }
    if (guess < 0) guess = cnt / 2;
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

// ---- GPF_RESAMPLE_MULTINOMIAL_SORTED: the sorted uniforms (gpf_math.hpp)
// the lane's MSLOTS consecutive spacings from resample slot s0 on (one Philox block per aligned slot pair, like the targets below)
__device__ __forceinline__ void lane_spacings(uint64_t seed, uint32_t epoch, uint32_t s0, uint64_t (&e)[MSLOTS])
{
    constexpr int NPB = MSLOTS / 2;
    const uint32_t sb = s0 >> 1;
    if (!(s0 & 1u)) {                                                 // kernel-uniform
#pragma unroll
        for (int q = 0; q < NPB; ++q) {
            const Philox b = rng(seed, sb + (uint32_t)q, 0, epoch, TAG_RESAMPLE);
            e[2 * q] = spacing_of(u64(b.w0, b.w1)); e[2 * q + 1] = spacing_of(u64(b.w2, b.w3));
        }
    } else {
#pragma unroll
        for (int q = 0; q <= NPB; ++q) {
            const Philox b = rng(seed, sb + (uint32_t)q, 0, epoch, TAG_RESAMPLE);
            if (q > 0) e[2 * q - 1] = spacing_of(u64(b.w0, b.w1));
            if (q < NPB) e[2 * q] = spacing_of(u64(b.w2, b.w3));
        }
    }
}
// (SortedGammaJob / sorted_gamma_tile: gpf_k_common.hpp -- the weight scan runs the same job as extra workgroups of its launch)
static __global__ __launch_bounds__(BLOCK) void k_sorted_gammas(SortedGammaJob job)
{
    sorted_gamma_tile(job, (int64_t)blockIdx.x * BLOCK + threadIdx.x);
}
constexpr int SP_DIRECT_TILES = 1024;              // up to 2.1 M slots the merge kernel sums the tile totals itself (4 loads per lane)
// beyond: vlo[t] = floor((g_0 + ... + g_{t-1}) 2^64 / (sum g + 1)), t = 0 .. ntl, by ONE workgroup (a block-wide scan, a division per tile)
constexpr int STILES_BLOCK = 1024;
static __global__ __launch_bounds__(STILES_BLOCK) void k_sorted_tiles(const uint64_t* __restrict__ g, int64_t ntl, uint64_t* __restrict__ vlo)
{
    __shared__ uint64_t s_w[STILES_BLOCK / WAVE];
    const int tid = (int)threadIdx.x, lane = lane_id(), wv = wave_id();
    const int64_t per = (ntl + STILES_BLOCK - 1) / STILES_BLOCK;       // thread i owns the consecutive tiles [i per, (i + 1) per)
    const int64_t t0 = (int64_t)tid * per, t1 = t0 + per < ntl ? t0 + per : ntl;
    uint64_t mine = 0;
    for (int64_t t = t0; t < t1; ++t) mine += g[t];
    const uint64_t inc = wave_scan_u64(mine);
    if (lane == WAVE - 1) s_w[wv] = inc;
    __syncthreads();
    uint64_t run = inc - mine, tot = 0;
#pragma unroll
    for (int w = 0; w < STILES_BLOCK / WAVE; ++w) { run += w < wv ? s_w[w] : 0; tot += s_w[w]; }
    const Div128 dv = div128_setup(tot + 1);
    for (int64_t t = t0; t < t1; ++t) { vlo[t] = div128(run, dv); run += g[t]; }
    if (tid == 0) vlo[ntl] = div128(tot, dv);
}

#ifdef GPF_DBG_STRAT
__device__ unsigned long long g_dbg_strat[8 * 4096];
#define DBG_STRAT(slot, val) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_dbg_strat[8 * blockIdx.x + (slot)] = (unsigned long long)(val); } while (0)
#else
#define DBG_STRAT(slot, val) do {} while (0)
#endif
// (2-3 workgroups per CU -- 41 KB of LDS each --: a 10^6-slot launch, 489 workgroups, is ONE resident round)
// SORTED: the targets are the sorted uniforms of GPF_RESAMPLE_MULTINOMIAL_SORTED instead of one uniform per stratum: the same merge from
// the cell side, with the number of targets below a prefix found by a binary search of the block's targets in LDS (no closed form)
// Slots per lane: 8 for the sorted uniforms (a workgroup = one tile of SP_TILE slots: the spec), 4 for the strata -- on a CDF in sorted order
// (sort_particles=true) the kernel lasts as long as its widest workgroups, the light tail's, whose cell ranges halve with the slots:
// 15.0 -> 12.9 us at 10^6 (unsorted CDF: 10.2 -> 10.1; profiles/r04_sort_buckets.txt).
#ifndef GPF_MSLOTS_STRAT
#define GPF_MSLOTS_STRAT 4
#endif
constexpr int MJB_STRAT = MBLOCK * GPF_MSLOTS_STRAT;
template <bool SORTED> constexpr int strat_slots_per_block() { return SORTED ? MBLOCK * GPF_MSLOTS : MJB_STRAT; }
template <bool SORTED>
__global__ __launch_bounds__(MBLOCK, 2) void k_search_strat(SearchArgs a)
{
    DBG_STRAT(0, wall_clock64());
    // (these shadow the namespace-scope constants of the same names for the whole kernel)
    constexpr int MSLOTS = SORTED ? GPF_MSLOTS : GPF_MSLOTS_STRAT;
    constexpr int MJB = MBLOCK * MSLOTS;
    constexpr int64_t MONO_WIDE = GPF_MONO_WIDE_MULT * (int64_t)MJB, MONO_WIDE2 = GPF_MONO_WIDE2_MULT * (int64_t)MJB;
    static_assert(MSLOTS % 4 == 0, "ancestors leave the lane as 16-byte stores");
    __shared__ __attribute__((aligned(16))) uint64_t s_T[MJB + 4];   // targets of the block's slots (+inf beyond n, and as padding)
    __shared__ __attribute__((aligned(16))) uint32_t s_mark[MJB];
    __shared__ int s_cnt[2][2][NWAVES];
    __shared__ uint32_t s_wmax[NWAVES];
    __shared__ __attribute__((aligned(16))) ulonglong2 s_coop[NWAVES * MONO_WIDE_NS * WAVE];   // the wide path's line-count strips (coop_count_le)
    __shared__ uint64_t s_sp[NWAVES];                  // SORTED: wave totals of the tile's spacings
    __shared__ uint64_t s_gs[3][NWAVES];               // ... and of the tile totals (before this tile, all, this tile)
    __shared__ uint64_t s_e0, s_eN;
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
    // SORTED on a shard: the spec's tiles are tiles of GLOBAL slots, so the launch's slot 0 is the first slot of the tile that holds the
    // first served slot; the `skip` launch slots below plan->first belong to a lower shard (their spacings still count in the tile's sum)
    const int64_t skip = SORTED && a.plan ? a.plan->first % MJB : 0;
    const int64_t n_out = a.plan ? skip + (a.plan->count < a.n ? a.plan->count : a.n) : a.n;
    if (j0 >= n_out || (a.plan && a.plan->count == 0)) return;        // (the grid of a shard is sized for the send buffer)
    const int64_t sbase = a.plan ? a.plan->first - skip : 0;          // strata / tiles: global slot of the launch's slot 0
    const int64_t pbase = a.plan ? a.plan->first - skip : a.gid0;     // RNG counters
    const int64_t t_off = a.plan ? (int64_t)a.plan->t_off : 0;
    // launch slots that exist at all: a tile's spacings are summed over ALL its slots, whoever serves them
    const int64_t n_all = SORTED && a.plan ? a.n_global - sbase : n_out;
    const int64_t tile = (SORTED && a.plan ? sbase / MJB : 0) + (int64_t)blockIdx.x;   // SORTED: the workgroup's tile of the sorted uniforms
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
    // (under a sorted order the CDF is steep at the front and flat in the tail: no guess, block_count_le_pair reads the whole level)
    const int64_t guess = a.order ? -1 : (int64_t)((double)(j0 > skip ? j0 - skip : 0) * (a.plan ? (double)a.n_cells / (double)(n_out - skip) : (double)a.n_cells * invN)) >> 8;
    block_count_le_pair(a.w.t256, n256, guess, s_cnt, A0, A1, [&](uint64_t& L0, uint64_t& L1) {
        if constexpr (SORTED) {
            // ---- sorted uniforms (gpf_math.hpp): the tile's range [vlo, vlo + W) of the 64-bit uniforms from k_sorted_tiles,
            //      the slots inside it by the tile's own exponential spacings, normalised by their sum
            static_assert(MJB == SP_TILE, "one workgroup = one tile of the sorted uniforms");
            uint64_t gpre = 0, gtot = 0, gown = 0;
            if (!a.sp_vlo) {                                                            // kernel-uniform: <= SP_DIRECT_TILES tile totals
                const int64_t ntl = (sbase + n_all + MJB - 1) / MJB;
                for (int64_t t = tid; t < ntl; t += MBLOCK) {
                    const uint64_t v = a.sp_g[t];
                    gtot += v; gpre += t < tile ? v : 0; gown = t == tile ? v : gown;
                }
            }
            uint64_t e[MSLOTS];
            lane_spacings(a.seed, a.epoch, s0, e);
            uint64_t run = 0;
#pragma unroll
            for (int k = 0; k < MSLOTS; ++k) { run += j0 + (int64_t)t0 + k < n_all ? e[k] : 0; e[k] = run; }   // inclusive inside the lane
            const uint64_t inc = wave_scan_u64(run);
            if (lane == WAVE - 1) s_sp[wv] = inc;
            if (!a.sp_vlo) {
                gpre = wave_sum_u64(gpre); gtot = wave_sum_u64(gtot); gown = wave_sum_u64(gown);
                if (lane == 0) { s_gs[0][wv] = gpre; s_gs[1][wv] = gtot; s_gs[2][wv] = gown; }
            }
            if (tid == 0) {
                s_e0 = e[0];
                // the (N + 1)-th spacing belongs to the last tile
                s_eN = j0 + MJB >= n_all ? spacing_of(resample_u64(a.seed, (uint32_t)(pbase + n_all), a.epoch)) : 0;
            }
            __syncthreads();
            uint64_t st = 1 + s_eN, wex = 0;                                           // s_t = the tile's sum + 1 (+ e_N)
#pragma unroll
            for (int w = 0; w < NWAVES; ++w) { st += s_sp[w]; wex += w < wv ? s_sp[w] : 0; }
            uint64_t vlo, Wt;
            if (a.sp_vlo) { vlo = a.sp_vlo[tile]; Wt = a.sp_vlo[tile + 1] - vlo; }
            else {
                gpre = 0; gtot = 0; gown = 0;
#pragma unroll
                for (int w = 0; w < NWAVES; ++w) { gpre += s_gs[0][w]; gtot += s_gs[1][w]; gown += s_gs[2][w]; }
                const Div128 dg = div128_setup(gtot + 1);                               // (as k_sorted_tiles)
                vlo = div128(gpre, dg); Wt = div128(gpre + gown, dg) - vlo;
            }
            const uint64_t S = a.ws->S;
            const uint64_t Tlo = mulhi64(vlo, S), Tw = mulhi64(vlo + Wt, S) - Tlo;     // the tile's targets lie in [Tlo, Tlo + Tw]
            const double inv_s = 1.0 / (double)st, dTw = (double)Tw;
            const uint64_t off = wex + (inc - run);
            uint64_t T[MSLOTS];
#pragma unroll
            for (int k = 0; k < MSLOTS; ++k) {
                const int64_t j = j0 + (int64_t)t0 + k;
                // (a shard: slots above its served range -- a higher shard's, targets beyond its CDF -- read as padding; the `skip` slots below
                //  it -- targets below its CDF -- as 0: the targets stay ascending, neither kind is written out)
                const uint64_t Tg = j < n_out ? sorted_target(off + e[k], inv_s, Tlo, Tw, dTw) : ~0ull;     // resample.jl:59 on the sorted uniform
                T[k] = j < n_out ? (j >= skip ? Tg - (uint64_t)t_off : 0) : ~0ull;
            }
            Lj0 = sorted_target(s_e0, inv_s, Tlo, Tw, dTw);                             // the block's first target ...
            Lj1 = sorted_target(st - 1 - s_eN, inv_s, Tlo, Tw, dTw) + 1;                // ... and one past its last (real) one
            Lj0 = Lj0 > (uint64_t)t_off ? Lj0 - (uint64_t)t_off : 0;                    // (local; the tile may start below the shard's CDF
            Lj1 -= (uint64_t)t_off;                                                     //  and end above it: a block with a served slot has Lj1 > t_off)
            Lj0s = (int64_t)Lj0;
            L0 = Lj0; L1 = Lj1 - 1;
#pragma unroll
            for (int k = 0; k < MSLOTS; k += 2) *reinterpret_cast<ulonglong2*>(s_T + MSLOTS * tid + k) = make_ulonglong2(T[k], T[k + 1]);
            if (tid < 4) s_T[MJB + tid] = ~0ull;
#pragma unroll
            for (int k = 0; k < MSLOTS; k += 4) *reinterpret_cast<uint4*>(s_mark + MSLOTS * tid + k) = make_uint4(0u, 0u, 0u, 0u);
            return;
        }
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
    // Three regimes by the width of the block's cell range (all block-uniform):  <= MONO_WIDE cells: stream the cells;  <= MONO_WIDE2
    // (the light tail of a sorted order: many cells per slot): stream the per-16 level -- a sixteenth of the entries -- for every
    // slot's 16-cell group, then ONE line count per slot inside it;  beyond: per-slot search from the per-256 level down.
    const bool two_level = i_end - i_start > MONO_WIDE && i_end - i_start <= MONO_WIDE2;
    const int64_t n16 = a.ntiles * (TILE / 16);
    if (tid == 0) s_mark[0] = (uint32_t)(two_level ? i_start / 16 : i_start);
    __syncthreads();
    DBG_STRAT(1, wall_clock64()); DBG_STRAT(4, i_end - i_start);
    uint32_t res[MSLOTS];
    if (i_end - i_start <= MONO_WIDE2) {
        // ---- stream the cells (or the per-16 entries); entry i resolves e = #{slots of the block with a target < value[i]} slots
        const double inv_step = a.ws->sinv;
        const uint64_t* cbase = two_level ? a.w.t16 + i_start / 16 : a.w.cdf + i_start;
        const uint32_t ncell = (uint32_t)(two_level ? (i_end - i_start) / 16 : i_end - i_start), ibase = (uint32_t)(two_level ? i_start / 16 : i_start) + 1u;
        constexpr int CPF = 6;                                            // 16-byte loads in flight per lane: 3072 cells per sweep
        for (uint32_t i0 = 0; i0 < ncell; i0 += 2u * MBLOCK * CPF) {
            ulonglong2 cc[CPF];
#pragma unroll
            for (int r = 0; r < CPF; ++r) {
                const uint32_t i = i0 + 2u * MBLOCK * r + 2u * (uint32_t)tid;
                cc[r] = i < ncell ? *reinterpret_cast<const ulonglong2*>(cbase + i) : make_ulonglong2(~0ull, ~0ull);
            }
            if constexpr (SORTED) {
                // e = number of the block's targets below the prefix, for the lane's 2 CPF prefixes at once: branch-free binary searches
                // of the (ascending) targets in LDS, interleaved -- 12 independent chains of 12 dependent reads
                uint64_t cv[2 * CPF]; uint32_t ee[2 * CPF];
#pragma unroll
                for (int r = 0; r < CPF; ++r) {
                    // a prefix at or above Lj1 resolves nothing (every slot is resolved by then); padding reads as ~0
                    cv[2 * r] = cc[r].x < Lj1 ? cc[r].x : 0; cv[2 * r + 1] = cc[r].y < Lj1 ? cc[r].y : 0;
                    ee[2 * r] = 0; ee[2 * r + 1] = 0;
                }
#pragma unroll
                for (uint32_t hh = MJB / 2; hh >= 1; hh >>= 1) {
#pragma unroll
                    for (int q = 0; q < 2 * CPF; ++q) ee[q] += s_T[ee[q] + hh - 1] < cv[q] ? hh : 0u;
                }
#pragma unroll
                for (int q = 0; q < 2 * CPF; ++q) ee[q] += s_T[ee[q]] < cv[q] ? 1u : 0u;      // (cv <= the last target: ee <= MJB - 1)
#pragma unroll
                for (int r = 0; r < CPF; ++r) {
                    const uint32_t i = i0 + 2u * MBLOCK * r + 2u * (uint32_t)tid;
                    if (i0 + 2u * MBLOCK * r >= ncell) break;             // block-uniform
                    // prefixes at or below the first target resolve nothing: only the LAST of them (they ascend) bounds slot 0
                    const uint64_t below = __ballot(cc[r].y <= Lj0);
                    if (cc[r].y <= Lj0) {
                        if (lane == (int)__popcll(below) - 1) atomicMax(&s_mark[0], ibase + i + 1u);
                        continue;
                    }
                    if (cc[r].x < Lj1) atomicMax(&s_mark[ee[2 * r]], ibase + i);
                    if (cc[r].y < Lj1) atomicMax(&s_mark[ee[2 * r + 1]], ibase + i + 1u);
                }
            } else
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
        if (two_level) {
            // res = number of per-16 entries <= target = the slot's 16-cell group; the cells <= target inside it: one 128-byte line
            // per slot, 8 lanes to a line, MONO_WIDE_NS slots of the lane in flight (coop_count_le)
            constexpr int WNS = MONO_WIDE_NS;
            static_assert(MSLOTS % WNS == 0, "whole rounds");
            ulonglong2* const strip = s_coop + wv * (WNS * WAVE);
#pragma unroll
            for (int k = 0; k < MSLOTS; k += WNS) {
                uint64_t Tq[WNS]; const uint64_t* ln[WNS]; int cq[WNS];
#pragma unroll
                for (int q = 0; q < WNS; ++q) {
                    Tq[q] = s_T[MSLOTS * tid + k + q];
                    res[k + q] = (int64_t)res[k + q] < n16 ? res[k + q] : (uint32_t)(n16 - 1);
                    ln[q] = a.w.cdf + (int64_t)res[k + q] * 16;
                }
                coop_count_le<WNS>(ln, Tq, strip, cq);
#pragma unroll
                for (int q = 0; q < WNS; ++q) res[k + q] = res[k + q] * 16u + (uint32_t)cq[q];
            }
        }
    } else {
        // ---- a few slots over a very large part of the CDF: per-slot search of the range.  The range's per-256 entries are
        //      staged in LDS (over s_mark) and searched there; the two line counts below them (per-16 level, cells) are dependent
        //      global reads, MONO_WIDE_NS slots of the lane in flight at a time.  Targets are read from and results written to the
        //      lane's own s_T entries (rolled loop: the streaming path's registers).
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
        constexpr int WNS = MONO_WIDE_NS;                                  // slots of the lane in flight at a time
        static_assert(MSLOTS % WNS == 0, "whole rounds");
        ulonglong2* const strip = s_coop + wv * (WNS * WAVE);
#pragma unroll 1
        for (int k = 0; k < MSLOTS; k += WNS) {
            uint64_t Tq[WNS]; int64_t gq[WNS]; const uint64_t* ln[WNS]; int cq[WNS];
#pragma unroll
            for (int q = 0; q < WNS; ++q) { Tq[q] = s_T[MSLOTS * tid + k + q]; gq[q] = group_of(Tq[q]); ln[q] = a.w.t16 + gq[q] * 16; }
            coop_count_le<WNS>(ln, Tq, strip, cq);
#pragma unroll
            for (int q = 0; q < WNS; ++q) {
                int64_t sq = gq[q] * 16 + cq[q];
                gq[q] = sq < n16 ? sq : n16 - 1;
                ln[q] = a.w.cdf + gq[q] * 16;
            }
            coop_count_le<WNS>(ln, Tq, strip, cq);
#pragma unroll
            for (int q = 0; q < WNS; ++q) s_T[MSLOTS * tid + k + q] = (uint64_t)(gq[q] * 16 + cq[q]);
        }
#pragma unroll
        for (int k = 0; k < MSLOTS; ++k) res[k] = (uint32_t)s_T[MSLOTS * tid + k];
    }
    DBG_STRAT(2, wall_clock64());
    const int64_t jb = j0 + MSLOTS * tid;
    const uint32_t last = (uint32_t)(a.n_cells - 1);
    if (a.plan && (a.pack.packed || a.pack.ring.peers)) {
        // ---- a shard: the served slots leave as exchange entries in slot order (which is grouped by destination shard).  The
        //      ancestors go through LDS so that neighbouring LANES take neighbouring entries: the packed stores are contiguous
        //      and the row reads (ascending ancestors) coalesce
        if (a.pack.own_anc) {
            // own-direct: a workgroup whose slots ALL belong to this shard itself (on one rank: every workgroup) writes the ancestors in
            // place from its registers, like the unsharded kernel -- no LDS round trip, no packed entries
            const int64_t b0 = a.plan->bounds[a.pack.me], b1 = a.plan->bounds[a.pack.me + 1];
            const int64_t g0 = sbase + (j0 > skip ? j0 : skip), g1 = sbase + (j0 + MJB < n_out ? j0 + MJB : n_out);
            if (g0 >= b0 && g1 <= b1 && n_out - skip <= a.pack.capacity) {     // block-uniform
                int32_t* dst = a.pack.own_anc + (sbase + jb - b0);
                int32_t out[MSLOTS];
#pragma unroll
                for (int k = 0; k < MSLOTS; ++k) out[k] = (int32_t)(a.pack.gid0 + (int64_t)(res[k] < last ? res[k] : last));
                if (jb >= skip && jb + MSLOTS <= n_out && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
#pragma unroll
                    for (int k = 0; k < MSLOTS; k += 4) *reinterpret_cast<int4*>(dst + k) = make_int4(out[k], out[k + 1], out[k + 2], out[k + 3]);
                } else {
#pragma unroll
                    for (int k = 0; k < MSLOTS; ++k) if (jb + k >= skip && jb + k < n_out) dst[k] = out[k];
                }
                return;
            }
        }
        int64_t* const s_bnd = reinterpret_cast<int64_t*>(s_T);        // the targets are spent
        __syncthreads();
#pragma unroll
        for (int k = 0; k < MSLOTS; ++k) s_mark[MSLOTS * tid + k] = res[k] < last ? res[k] : last;
        const int G = a.plan->n_shards;
        for (int g = tid; g <= G; g += MBLOCK) s_bnd[g] = a.plan->bounds[g];
        __syncthreads();
        const int64_t lim = n_out - skip < a.pack.capacity ? n_out : skip + a.pack.capacity;
        int lo = 0;
        {
            const int64_t jg = sbase + j0 + tid;                         // the shard that holds the lane's first entry
            int hi = G - 1;
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_bnd[mid] <= jg) lo = mid; else hi = mid - 1; }
        }
        const int W = a.pack.W;
#pragma unroll
        for (int k = 0; k < MSLOTS; ++k) {
            const int64_t e = j0 + k * MBLOCK + tid;
            if (e >= lim) break;
            if (e < skip) continue;
            const int64_t jg = sbase + e;
            while (lo < G - 1 && s_bnd[lo + 1] <= jg) ++lo;
            const int64_t i = s_mark[k * MBLOCK + tid];
            if (a.pack.own_anc && lo == a.pack.me) {                  // this shard's own slot: no packed entry, the ancestor in place
                a.pack.own_anc[jg - s_bnd[lo]] = (int32_t)(a.pack.gid0 + i);
                continue;
            }
            const double2* src = reinterpret_cast<const double2*>(a.pack.rows + i * W);
            if (a.pack.ring.peers) {                                  // straight into the window of the rank that holds the slot (peer stores)
                double row[8];
                for (int c = 0; c < W / 2; ++c) { const double2 v = src[c]; row[2 * c] = v.x; row[2 * c + 1] = v.y; }
                ring_store(a.pack.ring.peers[lo] + a.pack.ring.off + (jg - s_bnd[lo]) * (W + 2), row, W, (uint64_t)(a.pack.gid0 + i), a.pack.ring.seq);
                continue;
            }
            double* dst = a.pack.packed + (e - skip) * (W + 1 + a.pack.extra);
            if (a.pack.extra) dst[W + 1] = a.pack.pv.lw[i] - a.pack.pv.at(i);
            for (int c = 0; c < W / 2; ++c) { const double2 v = src[c]; dst[2 * c] = v.x; dst[2 * c + 1] = v.y; }
            dst[W] = u2d(((uint64_t)(jg - s_bnd[lo]) << 32) | (uint64_t)(a.pack.gid0 + i));
        }
        return;
    }
    // ---- parents[j] = order[i_old]   (resample.jl:168)
    int32_t out[MSLOTS];
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
    DBG_STRAT(3, wall_clock64());
}

} // namespace gpf
