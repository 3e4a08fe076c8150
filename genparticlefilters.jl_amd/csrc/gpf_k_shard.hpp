// shard-level kernels: summaries, the push exchange, the stratified plan, packed commit  (part of gpf_kernels.hpp; include that header, not this file)
#pragma once

namespace gpf {
// ----------------------------------------------------------------------------- shard-level kernels (multi-GPU)
// Sharding: GPU g owns the contiguous global particle range [gid0, gid0+n).  The weight
// CDF is global = local inclusive scan + the sum of the lower shards' totals; output slot j (global id)
// draws a target in GLOBAL fixed-point coordinates, the shard that owns that CDF cell looks the ancestor
// up and returns the row.  Integer arithmetic makes the ancestors independent of the number of shards.
constexpr int64_t SPACE_COUNTS = (int64_t)1 << 62;   // residual: target lives in the copy-count CDF
constexpr uint64_t STAGE_T_MASK = (1ull << 62) - 1;  // a staged target (T_local < S_local <= 2^62) below its space bit

static __global__ void k_pack_mflags(const unsigned long long* __restrict__ slots, double* out2, MboxPush push)
{
    double m; int f;
    fold_slots(slots, m, f);
    if (threadIdx.x == 0) { out2[0] = m; out2[1] = (double)(f & (FLAG_NAN | FLAG_POSINF)); }
    if (wave_id() == 0) {                                   // every lane holds (m, f): straight into the peers' mailboxes
        const uint64_t words[2] = {d2u(m), d2u((double)(f & (FLAG_NAN | FLAG_POSINF)))};
        mbox_push_wave(push, words);
    }
}
// host getters: wait for two mailbox rounds and copy their gathered arrays (na / nb words) into ordinary device memory
// (system-scope loads; the host then copies from there)
static __global__ void k_mbox_collect(MboxWait a, const uint64_t* srca, uint64_t* dsta, int na, MboxWait b, const uint64_t* srcb, uint64_t* dstb, int nb)
{
    mbox_wait_block(a); mbox_wait_block(b);
    for (int i = threadIdx.x; i < na; i += blockDim.x) dsta[i] = ld_sys(srca + i);
    for (int i = threadIdx.x; i < nb; i += blockDim.x) dstb[i] = ld_sys(srcb + i);
}
// gpf_comm_calibrate: `reps` DEPENDENT mailbox rounds inside one launch -- every rank stores its entry into every peer's mailbox, then waits for all
// G entries of the round; nobody can begin round r + 1 before it holds everybody's entry of round r.  Launch time / reps = what one summary round
// costs between real GPUs (the store's way over xGMI + the poll that sees it).  One wave.
static __global__ void k_mbox_rounds(uint64_t* const* peers, const uint64_t* own, int G, int me, uint64_t seq0, int reps, int32_t* timeout)
{
    for (int r = 0; r < reps; ++r) {
        const uint64_t seq = seq0 + 1 + (uint64_t)r;
        const int slot = (int)(seq & (MB_SLOTS - 1));
        const MboxPush p{peers, mb_payload_off(MB_CAL, slot), mb_tag_off(MB_CAL, slot), seq, G, me, 2};
        const uint64_t words[2] = {seq, (uint64_t)me};
        mbox_push_wave(p, words);
        const MboxWait w{own + mb_tag_off(MB_CAL, slot), seq, G, 2, timeout};
        mbox_wait_block(w);
    }
}
// gpf_comm_create's self-test of the receive windows, round d (1 <= d < G): this rank stores one entry into slot 0 of rank (me + d) % G's window and waits
// for the entry rank (me - d) % G stores into its own -- the very stores, loads and seals of a slab exchange (ring_store / ring_load), across the very
// mappings.  One lane; W at run time.  A stalled or invisible peer store ends in the bounded wait's flag 3, which the host turns into "no windows".
static __global__ void k_ring_selftest(uint64_t* const* peers, const uint64_t* own, int G, int me, int d, int64_t parity_words, uint64_t seq, int W, int32_t* timeout,
                                       int32_t* bad)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int64_t off = (int64_t)(seq & (RING_PARITIES - 1)) * parity_words;
    double row[8];
    for (int c = 0; c < 8; ++c) row[c] = (double)(me * 16 + c);
    ring_store(peers[(me + d) % G] + off, row, W, (uint64_t)me, seq);
    const int src = (me - d + G) % G;
    const uint64_t* e = own + off;
    bool ok = false;
    for (unsigned spins = 0; spins <= RING_SPIN_LIMIT; ++spins) {
        uint64_t x = seq, last = 0;
        for (int k = 0; k < W + 2; ++k) { last = ld_sys(e + k); if (k < W + 1) x = (x ^ last) * 0x9E3779B97F4A7C15ull; }
        if (x == last) { ok = true; break; }
        __builtin_amdgcn_s_sleep(8);
    }
    if (!ok) { __hip_atomic_store(timeout, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return; }
    bool same = ld_sys(e + W) == (uint64_t)src;
    for (int c = 0; c < W; ++c) same = same && u2d(ld_sys(e + c)) == (double)(src * 16 + c);
    if (!same) *bad = 1;
}
// {Ql0..3} -> out5[1..4]: limb sums of sum q^2 folded over the scan blocks (exact integers); out5[0] = S_local is written by the scan
static __global__ void k_export_q(const uint64_t* __restrict__ blockQ, int nblk, int64_t* out5, MboxPush push)
{
    __shared__ uint64_t s_q[NWAVES][4];
    __shared__ uint64_t s_tot[4];
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
        s_tot[threadIdx.x] = t;
    }
    if (push.peers) {                                       // {S, Ql0..3} to every peer (S was written by the scan before this kernel)
        __syncthreads();
        if (wave_id() == 0) {
            const uint64_t words[5] = {(uint64_t)out5[0], s_tot[0], s_tot[1], s_tot[2], s_tot[3]};
            mbox_push_wave(push, words);
        }
    }
}
// global S (and residual shift) into the device scalar block from the gathered shard totals
// tot_all = the gathered {S_local, Ql0..3} of all G shards -> the global S
static __global__ void k_set_global(const int64_t* __restrict__ tot_all, int G, WSum* ws, MboxWait wait)
{
    mbox_wait_block(wait);
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        uint64_t S = 0;
        for (int g = 0; g < G; ++g) S += (uint64_t)ld_gathered(tot_all + 5 * g, wait.tags != nullptr);
        ws->S = S;
    }
}
// (one wave)  zero: the exchange counters of the resample, cleared here when no weight scan runs in front (the direct residual scans of a shard)
static __global__ void k_export_residual(const Scalars* sc, int64_t* out2, MboxPush push, int64_t* zero, int zero_words)
{
    if (zero) for (int i = threadIdx.x; i < zero_words; i += blockDim.x) zero[i] = 0;
    const uint64_t words[2] = {sc->Ctot, sc->Rs};
    if (threadIdx.x == 0 && blockIdx.x == 0) { out2[0] = (int64_t)words[0]; out2[1] = (int64_t)words[1]; }
    mbox_push_wave(push, words);
}

// ---- sharded resampling, the PUSH exchange.  RNG counters are keyed by the GLOBAL slot id, so every
// shard can evaluate the target of EVERY output slot itself (Philox is pure ALU work): the owner of a target finds
// out on its own which slots draw from it, looks the ancestors up and pushes [row | slot | ancestor id] to the shard
// that holds the slot.  No request message exists.  Pass 1 walks all slots in chunks that never straddle a shard
// boundary, compacts each chunk's hits in LDS and appends them to the staging list of the slot's shard (one global
// atomic per chunk; the order of chunks inside a list is arbitrary, every entry names its slot); it also counts what
// this shard will receive from whom.  Pass 2 looks the staged hits up (same core as k_search) and packs the rows.
constexpr int PUSH_CHUNK = 2048;                  // output slots per chunk
// the 2G exchange counters sit on separate 128-byte lines: every chunk does one returning atomic add on its destination's counter, and
// atomics to the same cache LINE serialise (~7-10 ns each: with G = 8 and all 16 counters on one line, ~4000 of them per launch)
// (COUNT_STRIDE: gpf_k_common.hpp)
// The RECEIVE counters (who serves this shard's slots) take one atomic add per workgroup and owner at the END of the kernels that count them
// (k_search_own, k_search_own_res, k_push_scan): 256 adds on one address, ~5 ns each in a row = 1.3 us of tail per launch.  On one node (G <= 8) they
// are striped: stripe 0 is the counter itself (index G + q), stripes 1 .. 7 use the lines of counters no shard owns (from index 2 G on; the scan
// clears all 2 MAX_SHARDS lines every round), a workgroup adds to stripe blockIdx & 7, the reader sums.  Kernels that SET the counters (the plan
// kernels, the pull plan) write stripe 0 and find the others zero.
constexpr int RECV_STRIPES = 8;
__host__ __device__ inline int recv_stripes(int G) { return G <= 8 ? RECV_STRIPES : 1; }
__host__ __device__ inline int recv_counter_index(int G, int q, int stripe) { return (G <= 8 && stripe) ? 2 * G + (stripe - 1) * G + q : G + q; }
static_assert(2 * 8 + (RECV_STRIPES - 1) * 8 <= 2 * MAX_SHARDS, "the stripes fit behind the counters of a node");
__device__ __forceinline__ int64_t recv_count(const int64_t* counts, int G, int q)
{
    int64_t v = 0;
    for (int s = 0; s < recv_stripes(G); ++s) v += counts[recv_counter_index(G, q, s) * COUNT_STRIDE];
    return v;
}
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
    MboxWait wait_tot, wait_cr;                   // shard mailboxes: tot_all / cr_all are filled by the peers' kernels -- wait before reading
    int extra; PrioView pv;                       // prioritised resample: packed entries carry one more double, lw[a] - lp[a] (PackOut::extra)
    int skip_own;                                 // the shard's OWN slots are resolved by k_search_own (ancestors in place, no packed entry): pass 1 skips their chunks
    int64_t* traffic;                             // the plan kernels (window exchange: the host never learns the counts): {entries sent to, received from OTHER shards} so far, or nullptr
    RingOut ring;                                 // k_push / k_push_multi: ring.peers != nullptr -- the entries go straight into the destination ranks' receive windows (GPF_SHARD_EXCHANGE_P2P_ALL)
};
// (k_push / k_push_multi, window exchange: what this shard serves to and is served by the OTHER shards, for gpf_comm_traffic)
__device__ __forceinline__ void push_traffic(const PushArgs& a)
{
    if (!a.traffic) return;
    unsigned long long sent = 0, recv = 0;
    for (int g = 0; g < a.G; ++g) if (g != a.me) { sent += (unsigned long long)a.counts[g * COUNT_STRIDE]; recv += (unsigned long long)recv_count(a.counts, a.G, g); }
    if (sent) atomicAdd(reinterpret_cast<unsigned long long*>(a.traffic), sent);
    if (recv) atomicAdd(reinterpret_cast<unsigned long long*>(a.traffic + 1), recv);
}
struct PushTables {                               // LDS copy of the per-shard tables
    int64_t w_incl[MAX_SHARDS], c_incl[MAX_SHARDS], bounds[MAX_SHARDS + 1], chunk0[MAX_SHARDS + 1];
};
__device__ __forceinline__ void push_tables(const PushArgs& a, PushTables& t)
{
    // inclusive shard totals of the sampled space (weights, or residual weights) and of the residual copy counts: the
    // first wave, one shard per lane (G <= MAX_SHARDS = 64)
    static_assert(MAX_SHARDS <= WAVE, "one lane per shard");
    mbox_wait_block(a.wait_tot);
    mbox_wait_block(a.wait_cr);
    if (threadIdx.x < WAVE) {
        const int g = (int)threadIdx.x;
        const bool mb = a.wait_tot.tags != nullptr;
        uint64_t w = g < a.G ? (uint64_t)(a.cr_all ? ld_gathered(a.cr_all + 2 * g + 1, mb) : ld_gathered(a.tot_all + 5 * g, mb)) : 0;
        uint64_t c = g < a.G && a.cr_all ? (uint64_t)ld_gathered(a.cr_all + 2 * g, mb) : 0;
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
// ... with the inclusive totals of shards 0 .. 6 in REGISTERS (they are the same for every lane: readfirstlane makes them scalars), one node
// (G <= 8): seven compares against scalars and a select chain for the owner's lower end -- no LDS read, no dependent read indexed by the owner.
// (push_owner per slot: 7 LDS broadcasts + 1 dependent LDS read, SQ_WAIT_INST_LDS 505 K cycles per launch of k_search_own against 70 K in
//  k_search_multi, + 90 VALU instructions per slot.)  Totals of the shards from G - 1 on read as ~0: never <= T.
struct OwnerBounds { uint64_t b[7]; };
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v)
{
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}
__device__ __forceinline__ OwnerBounds owner_bounds(const PushTables& t, int G, int space)
{
    OwnerBounds ob;
    const int64_t* incl = space ? t.c_incl : t.w_incl;
#pragma unroll
    for (int g = 0; g < 7; ++g) ob.b[g] = uniform_u64(g < G - 1 ? (uint64_t)incl[g] : ~0ull);
    return ob;
}
__device__ __forceinline__ int push_owner(const OwnerBounds& ob, uint64_t T, uint64_t& T_local)
{
    int h = 0; uint64_t lo = 0;
#pragma unroll
    for (int g = 0; g < 7; ++g) { const bool le = ob.b[g] <= T; h += le ? 1 : 0; lo = le ? ob.b[g] : lo; }
    T_local = T - lo;
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
        if (a.skip_own && g == a.me) continue;                                        // block-uniform: k_search_own takes this shard's slots
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
                if (threadIdx.x == 0) s_base = atomicAdd(reinterpret_cast<unsigned long long*>(a.counts + g * COUNT_STRIDE), (unsigned long long)total);
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
            s_base = atomicAdd(reinterpret_cast<unsigned long long*>(a.counts + g * COUNT_STRIDE), (unsigned long long)total);
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
        atomicAdd(reinterpret_cast<unsigned long long*>(a.counts + recv_counter_index(a.G, (int)threadIdx.x, (int)(blockIdx.x & (RECV_STRIPES - 1))) * COUNT_STRIDE),
                  (unsigned long long)s_recv[threadIdx.x]);
}
// ---- the shard's OWN slots (multinomial).  A slot of this shard whose target falls into this shard's part of the CDF needs no exchange
// entry at all: its ancestor is a local particle, exactly the unsharded case.  This kernel is k_search_multi over the shard's own slots
// with the GLOBAL target arithmetic: own hits get anc[j] = global id of the ancestor (the next propagate gathers the row through it,
// k_step<GATHER> with PackedCommit::masked), slots whose target another shard owns get anc[j] = -1 (their rows arrive packed) and are
// counted per owner (the receive counts of the exchange; counts[G + me] = own hits).  Pass 1 (k_push_scan) then walks the OTHER shards'
// slots only, and nothing is staged, packed or copied for what never leaves the GPU: on one rank the sharded resample is the unsharded one.
template <int LOGG>
__global__ __launch_bounds__(SBLOCK, 4) void k_search_own(PushArgs a, CdfLevels lw_, int64_t n, int64_t ntiles, int64_t gid0, int32_t* __restrict__ anc)
{
    constexpr int NS = GPF_MULTI_NS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ PushTables t;
    __shared__ unsigned int s_recv[MAX_SHARDS];
    if (threadIdx.x < MAX_SHARDS) s_recv[threadIdx.x] = 0;
    uint64_t Sw = 0;
    // the lane's NS consecutive slots: one Philox block per aligned slot pair, the first trip's while the key table is on its way, the next trip's
    // behind this trip's stores (as k_search_multi).  What the 4 us between this kernel and k_search_multi were (one rank, 19.4 against 15.2 us;
    // profiles/r05d_sharded_one_rank.txt): the barrier of push_tables in front of the Philox blocks 1.4, the owner logic 1.1, 256 same-address
    // atomics on the receive counter at the end of the workgroups 1.3 -- not the doubled Philox work, not the order of the prologue
    const int64_t stride = (int64_t)gridDim.x * NS * SBLOCK;
    int64_t base = (int64_t)blockIdx.x * NS * SBLOCK;
    uint64_t Tgl[NS], U[NS];
    auto draw = [&](int64_t b0) { resample_u64_run<NS>(a.seed, (uint32_t)(gid0 + b0 + NS * (int64_t)threadIdx.x), a.epoch, U); };
    auto scale = [&]() {
#pragma unroll
        for (int u = 0; u < NS; ++u) Tgl[u] = mulhi64(U[u], Sw);                     // resample.jl:59, global coordinates
    };
    // (the shard totals -- a mailbox wait, a load, a barrier -- and the first targets behind the key table's loads, not in front of them)
    const MultiTable tb = multi_table_load_s<LOGG>(lw_, ntiles, reinterpret_cast<uint32_t*>(smem), [&]() -> uint64_t {
        draw(base);                                                                  // (Philox needs no total: in front of the barrier of push_tables, not behind it)
        push_tables(a, t);
        Sw = (uint64_t)t.w_incl[a.G - 1];
        scale();
        return (uint64_t)t.w_incl[a.me] - (a.me ? (uint64_t)t.w_incl[a.me - 1] : 0);
    });
    const int lane = lane_id();
    unsigned recv_cnt = 0;                                                           // lane q: this wave's slots served by shard q
    const bool node = a.G <= 8;                                                      // kernel-uniform
    const OwnerBounds ob = owner_bounds(t, node ? a.G : 1, 0);
    for (; base < n; base += stride) {
        const int64_t j0 = base + NS * (int64_t)threadIdx.x;
        uint64_t T[NS]; int own[NS];
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const uint64_t Tg = Tgl[u];
            uint64_t Tl;
            if (a.G == 1) { own[u] = j0 + u < n ? 0 : -1; T[u] = Tg; }               // kernel-uniform: one shard owns every target (the own count: n)
            else {
                own[u] = node ? push_owner(ob, Tg, Tl) : push_owner(t, a.G, 0, Tg, Tl);
                if (j0 + u >= n) own[u] = -1;
                T[u] = own[u] == a.me ? Tl : 0;                                      // (another shard's target: the lane rides along with a dummy)
                for (int q = 0; q < a.G; ++q) {
                    const unsigned c = (unsigned)__popcll(__ballot(own[u] == q));
                    if (lane == q) recv_cnt += c;
                }
            }
        }
        uint32_t idx[NS];
        multi_lookup<LOGG, NS>(tb, lw_, n, T, idx);
#pragma unroll
        for (int u = 0; u < NS; ++u) idx[u] = own[u] == a.me ? (uint32_t)(int32_t)(gid0 + (int64_t)idx[u]) : 0xffffffffu;      // (-1: another shard serves the slot)
        int32_t* dst = anc + j0;
        if (NS == 2 && j0 + NS <= n && (reinterpret_cast<uintptr_t>(dst) & 7) == 0) *reinterpret_cast<int2*>(dst) = make_int2((int32_t)idx[0], (int32_t)idx[1]);
        else {
#pragma unroll
            for (int u = 0; u < NS; ++u) if (j0 + u < n) dst[u] = (int32_t)idx[u];
        }
        if (base + stride < n) { draw(base + stride); scale(); }
    }
    if (recv_cnt) atomicAdd(&s_recv[lane], recv_cnt);
    __syncthreads();
    if (a.G == 1) { if (blockIdx.x == 0 && threadIdx.x == 0) a.counts[(a.G + a.me) * COUNT_STRIDE] = n; }    // (every slot is an own hit)
    else if (threadIdx.x < (unsigned)a.G && s_recv[threadIdx.x])
        atomicAdd(reinterpret_cast<unsigned long long*>(a.counts + recv_counter_index(a.G, (int)threadIdx.x, (int)(blockIdx.x & (RECV_STRIPES - 1))) * COUNT_STRIDE),
                  (unsigned long long)s_recv[threadIdx.x]);
}
// ... and for RESIDUAL resampling (resample.jl:96-115): an own slot below the global copy total is the jg-th deterministic copy (target jg in
// the copy-count space, owner by the shards' inclusive copy totals), the others draw from the residual weights; own hits are looked up
// in this shard's copy-count / residual-weight CDF (k_search's two-line core, both top tables in LDS), the rest get -1 and a count per owner.
static __global__ __launch_bounds__(SBLOCK, SEARCH_WAVES_PER_SIMD) void k_search_own_res(PushArgs a, CdfLevels lw_, CdfLevels lc_, int64_t n, int64_t ntiles,
                                                                                   int64_t gid0, int32_t* __restrict__ anc)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const SearchTop st = search_prologue(lw_, lc_, true, ntiles, reinterpret_cast<uint64_t*>(smem));
    __shared__ PushTables t;
    __shared__ unsigned int s_recv[MAX_SHARDS];
    __shared__ ulonglong2 s_coop[2 * SBLOCK];
    ulonglong2* const lds_wave = s_coop + wave_id() * (2 * WAVE);
    if (threadIdx.x < MAX_SHARDS) s_recv[threadIdx.x] = 0;
    // (the first trip's uniforms need no total: drawn in front of the mailbox wait + barrier of push_tables, the next trip's behind this trip's stores --
    //  as k_search_own, where the barrier in front of the Philox blocks cost 1.4 us)
    uint64_t U[2];
    const int64_t stride = (int64_t)gridDim.x * 2 * SBLOCK;
    int64_t base = (int64_t)blockIdx.x * 2 * SBLOCK;
    resample_u64_run<2>(a.seed, (uint32_t)(gid0 + base + 2 * (int64_t)threadIdx.x), a.epoch, U);      // (one Philox block for the lane's slot pair)
    push_tables(a, t);
    const PushScal sc = push_scalars<1>(a, t);
    const int lane = lane_id();
    unsigned recv_cnt = 0;
    for (; base < n; base += stride) {
        uint64_t T[2]; int own[2]; const uint64_t* top[2]; const CdfLevels* L[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t j = base + 2 * (int64_t)threadIdx.x + u;
            const uint64_t jg = (uint64_t)(gid0 + j);
            uint64_t Tg = 0, Tl = 0; int space = 0;
            push_target<1>(a, sc, jg, U[u], Tg, space);
            own[u] = j < n ? push_owner(t, a.G, space, Tg, Tl) : -1;
            const bool mine = own[u] == a.me;
            T[u] = mine ? Tl : 0;                                                    // (another shard's target: the lane rides along with a dummy)
            top[u] = (mine && space) ? st.topc : st.topw; L[u] = (mine && space) ? &lc_ : &lw_;
            for (int q = 0; q < a.G; ++q) {
                const unsigned c = (unsigned)__popcll(__ballot(own[u] == q));
                if (lane == q) recv_cnt += c;
            }
        }
        int64_t idx[2];
        search_pair(st, L, top, T, true, lds_wave, n, ntiles, idx);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t j = base + 2 * (int64_t)threadIdx.x + u;
            if (j < n) anc[j] = own[u] == a.me ? (int32_t)(gid0 + idx[u]) : -1;
        }
        if (base + stride < n) resample_u64_run<2>(a.seed, (uint32_t)(gid0 + base + stride + 2 * (int64_t)threadIdx.x), a.epoch, U);
    }
    if (recv_cnt) atomicAdd(&s_recv[lane], recv_cnt);
    __syncthreads();
    if (threadIdx.x < (unsigned)a.G && s_recv[threadIdx.x])
        atomicAdd(reinterpret_cast<unsigned long long*>(a.counts + recv_counter_index(a.G, (int)threadIdx.x, (int)(blockIdx.x & (RECV_STRIPES - 1))) * COUNT_STRIDE),
                  (unsigned long long)s_recv[threadIdx.x]);
}
// materialize() of a commit with own hits: rows_out[j] = rows_in[anc[j] - gid0], lw[j] = 0 for the slots with anc[j] >= 0
// (own_range != nullptr: the own hits are the slots [own_range[0], own_range[1]) -- stratified resampling; else the slots with anc >= 0)
template <int W>
__global__ __launch_bounds__(BLOCK) void k_gather_own(const int32_t* __restrict__ anc, int64_t gid0, const double* __restrict__ rows_in,
                                                      double* __restrict__ rows_out, double* __restrict__ lw, int64_t n, const int64_t* __restrict__ own_range)
{
    constexpr int C = W / 2;
    const int64_t lo = own_range ? own_range[0] : 0, hi = own_range ? own_range[1] : n;
    for (int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x; t < n * C; t += (int64_t)gridDim.x * BLOCK) {
        const int64_t j = t / C;
        const int c = (int)(t - j * C);
        if (j < lo || j >= hi) continue;
        const int32_t a = anc[j];
        if (!own_range && a < 0) continue;
        reinterpret_cast<double2*>(rows_out)[t] = reinterpret_cast<const double2*>(rows_in)[((int64_t)a - gid0) * C + c];
        if (c == 0) lw[j] = 0.0;                                                     // update_weights!, resample.jl:195
    }
}

// ---- the PULL plan of the i.i.d. resamplers (gpf_comm_set_plan / GPF_SHARD_PLAN=pull).  Every shard evaluates only
// its OWN slots -- n targets instead of the push plan's n_global -- and asks for them: request {T_local | space << 62, slot inside the
// requester} (the entry format of the push stage) goes to the shard whose range of the sampled space holds the target, grouped by
// owner.  The owners receive the requests straight into their staging lists and run pass 2 (k_push / k_push_multi) unchanged.  The
// price is a second exchange (requests out, rows back) and a host wait before it; which plan wins depends on G and on the links.
// req: [G][req_stride] request lists, req_counts: G counters COUNT_STRIDE words apart (cleared by the caller)
template <int METHOD>
__global__ __launch_bounds__(PUSH_SCAN_BLOCK) void k_pull_scan(PushArgs a, ulonglong2* __restrict__ req, int64_t req_stride, int64_t* __restrict__ req_counts)
{
    constexpr int R = PUSH_CHUNK / PUSH_SCAN_BLOCK, NW = PUSH_SCAN_BLOCK / WAVE;
    __shared__ PushTables t;
    __shared__ unsigned int s_wcnt[NW][MAX_SHARDS];                                   // per wave and owner: count, then exclusive prefix over waves
    __shared__ unsigned long long s_base[MAX_SHARDS];
    push_tables(a, t);
    const PushScal sc = push_scalars<METHOD>(a, t);
    const int lane = lane_id(), wv = (int)threadIdx.x / WAVE;
    const uint64_t lt = lane ? (~0ull >> (WAVE - lane)) : 0ull;                       // lanes below this one
    const int64_t lo = t.bounds[a.me], hi = t.bounds[a.me + 1];
    const int64_t nch = (hi - lo + PUSH_CHUNK - 1) / PUSH_CHUNK;
    for (int64_t c = blockIdx.x; c < nch; c += gridDim.x) {
        const int64_t j0 = lo + c * PUSH_CHUNK, j1 = j0 + PUSH_CHUNK < hi ? j0 + PUSH_CHUNK : hi;
        uint64_t U[R];                                                                // (the slot uniforms: as in k_push_scan)
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
        uint64_t Tl[R]; int own[R]; unsigned rank[R];
        unsigned wcnt = 0;                                                            // lane q: this wave's requests to shard q so far
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t j = j0 + (int64_t)threadIdx.x * R + r;
            own[r] = -1; Tl[r] = 0; rank[r] = 0;
            if (j < j1) {
                uint64_t T = 0; int space = 0;
                push_target<METHOD>(a, sc, (uint64_t)j, U[r], T, space);
                own[r] = push_owner(t, a.G, space, T, Tl[r]);
                Tl[r] |= (uint64_t)space << 62;
            }
            for (int q = 0; q < a.G; ++q) {
                const uint64_t m = __ballot(own[r] == q);
                const unsigned prev = (unsigned)__shfl((int)wcnt, q, WAVE);
                if (own[r] == q) rank[r] = prev + (unsigned)__popcll(m & lt);
                if (lane == q) wcnt += (unsigned)__popcll(m);
            }
        }
        if (lane < a.G) s_wcnt[wv][lane] = wcnt;
        __syncthreads();
        if ((int)threadIdx.x < a.G) {                                                 // one returning atomic per chunk and owner
            unsigned run = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) { const unsigned v = s_wcnt[w][threadIdx.x]; s_wcnt[w][threadIdx.x] = run; run += v; }
            s_base[threadIdx.x] = run ? atomicAdd(reinterpret_cast<unsigned long long*>(req_counts + threadIdx.x * COUNT_STRIDE), (unsigned long long)run) : 0ull;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (own[r] >= 0)
                req[(int64_t)own[r] * req_stride + (int64_t)s_base[own[r]] + s_wcnt[wv][own[r]] + rank[r]] =
                    make_ulonglong2(Tl[r], (uint64_t)(j0 + (int64_t)threadIdx.x * R + r - lo));
        __syncthreads();                                                              // s_wcnt / s_base
    }
}
// the request counters, densely: out[q] = requests of this shard to shard q
static __global__ void k_pull_counts(const int64_t* __restrict__ req_counts, int G, int64_t* __restrict__ out)
{
    if ((int)threadIdx.x < G) out[threadIdx.x] = req_counts[threadIdx.x * COUNT_STRIDE];
}
// after the gathered request matrix M[requester][owner] is known: the exchange counters pass 2 and the row exchange work from --
// entries this shard serves to g = what g asked of it, entries it receives from g = what it asked of g
static __global__ void k_pull_set_counts(const int64_t* __restrict__ M, int G, int me, int64_t* __restrict__ counts)
{
    const int g = (int)threadIdx.x;
    if (g < G) { counts[g * COUNT_STRIDE] = M[g * G + me]; counts[(G + g) * COUNT_STRIDE] = M[me * G + g]; }
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
        for (int g = 0; g < a.G; ++g) { s_off[g] = o; o += a.counts[g * COUNT_STRIDE]; s_bnd[g] = a.bounds[g]; }
        s_off[a.G] = o;
        // the host needs the counts for the all-to-all split sizes: publish them to pinned host memory NOW, so the host
        // reads them while this kernel is still looking ancestors up (system-scope stores, ticket last)
        if (blockIdx.x == 0 && a.host_counts) {
            for (int g = 0; g < 2 * a.G; ++g)
                __hip_atomic_store(a.host_counts + g, g < a.G ? a.counts[g * COUNT_STRIDE] : recv_count(a.counts, a.G, g - a.G), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            publish_behind_sys_stores(a.host_counts + 2 * MAX_SHARDS, a.ticket);
        }
        if (blockIdx.x == 0) push_traffic(a);
    }
    __syncthreads();
    const int64_t total = s_off[a.G] < capacity ? s_off[a.G] : capacity;
    for (int64_t base = (int64_t)blockIdx.x * 2 * SBLOCK; base < total; base += (int64_t)gridDim.x * 2 * SBLOCK) {
        int64_t e[2]; bool act[2]; uint64_t T[2], slot[2]; const uint64_t* top[2]; const CdfLevels* L[2]; int gdst[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            e[u] = base + u * SBLOCK + threadIdx.x;
            act[u] = e[u] < total;
            const int64_t ee = act[u] ? e[u] : total - 1;
            int g = 0;
            while (g < a.G - 1 && ee >= s_off[g + 1]) ++g;
            gdst[u] = g;
            const ulonglong2 q = a.stage[s_bnd[g] + (ee - s_off[g])];
            const bool incounts = (q.x >> 62) != 0;
            T[u] = q.x & STAGE_T_MASK;
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
            if (a.ring.peers) {                                   // the window exchange: straight into the slot of the rank that holds it
                ring_store(a.ring.peers[gdst[u]] + a.ring.off + (int64_t)slot[u] * (W + 2), src, W, (uint64_t)(gid0 + idx[u]), a.ring.seq);
                continue;
            }
            double* dst = packed_out + e[u] * (W + 1 + a.extra);
            if (a.extra) dst[W + 1] = a.pv.lw[idx[u]] - a.pv.at(idx[u]);
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
// counts follow by intersecting slot ranges, and the served slots' ancestors AND their packed rows come from k_search_strat
// (streaming merge over the shard's own CDF, SearchArgs::pack) -- no pass over the global slots, no staging list.
static __global__ __launch_bounds__(128) void k_strat_plan(PushArgs a, ShardPlan* plan)
{
    __shared__ int64_t F[MAX_SHARDS + 1];
    const StratPlanJob jb{plan, a.seed, a.epoch, a.G, a.me, a.n_global, a.tot_all, a.wait_tot, a.counts, a.host_counts, a.ticket, a.traffic, 0};
    strat_plan_body(jb, [&](int g) { return a.bounds[g]; }, F);
}
// ---- sharded SORTED MULTINOMIAL (GPF_RESAMPLE_MULTINOMIAL_SORTED): the targets of the N slots are non-decreasing in
// the slot index, like the strata, so the slots shard h serves are again ONE range [F[h], F[h+1]), F[h] = the first slot whose target is
// >= lo_h.  The tile totals depend on (seed, epoch, N) alone: every shard draws all of them (k_sorted_gammas over the GLOBAL tiles, as extra
// workgroups of its weight scan), so every shard can place every tile -- mulhi(vlo[t], S) ascends in t -- and, with the spacings of ONE
// tile, every slot of it.  Workgroup h of this kernel finds F[h]: the first tile whose END reaches lo_h (the tiles below it lie entirely
// below lo_h), then that tile's 2048 targets exactly as k_search_strat<true> forms them, and the number of them below lo_h.  The workgroup
// that arrives last turns F into the plan and the exchange counts (as k_strat_plan).  One workgroup per INTERIOR boundary (F[0] = 0 and
// F[G] = N need no search; one shard: one workgroup, no search, no arrival).  Up to SP_DIRECT_TILES tiles every workgroup scans the tile
// totals itself (the merge kernel does the same for its own tile); beyond, k_sorted_tiles has run and vlo[] is searched in place.
struct SortedPlanJob {
    const uint64_t* g;            // [ntl] gamma totals of the GLOBAL tiles
    const uint64_t* vlo;          // have_vlo: [ntl + 1] where every tile starts among the sorted 64-bit uniforms (k_sorted_tiles)
    int64_t ntl; int have_vlo;
    int64_t* F;                   // [MAX_SHARDS + 1] first slot served by every shard (scratch between the workgroups)
    unsigned int* arrive;         // arrival counter, zero between launches
};
// number of t in [0, cnt) with pred(t), pred monotone (true ... true false ... false); block-cooperative: one 256-entry window around `guess`
// (one round trip when it brackets the answer), else 256-ary rounds over the whole range
template <class P>
__device__ __forceinline__ int64_t block_partition_point(int64_t cnt, int64_t guess, int (*s_cnt)[NWAVES], P&& pred)
{
    const int tid = (int)threadIdx.x;
    int par = 0;
    auto count = [&](bool p) {
        const int c = (int)__popcll(__ballot(p));
        if (lane_id() == 0) s_cnt[par][wave_id()] = c;
        __syncthreads();
        int64_t k = 0;
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) k += s_cnt[par][w];
        par ^= 1;
        return k;
    };
    {
        int64_t w_lo = guess - MBLOCK / 2;
        w_lo = w_lo + MBLOCK > cnt ? cnt - MBLOCK : w_lo;
        w_lo = w_lo < 0 ? 0 : w_lo;
        const int64_t w_hi = w_lo + MBLOCK < cnt ? w_lo + MBLOCK : cnt;
        const int64_t k = count(w_lo + tid < w_hi ? pred(w_lo + tid) : false);
        if ((k > 0 || w_lo == 0) && (k < w_hi - w_lo || w_hi == cnt)) return w_lo + k;                       // block-uniform
    }
    int64_t lo = 0, hi = cnt;                      // invariant: lo <= answer <= hi
    while (hi > lo) {                              // block-uniform
        const int64_t len = hi - lo, step = len <= MBLOCK ? 1 : (len + MBLOCK - 1) / MBLOCK;
        const int64_t p = lo + (int64_t)(tid + 1) * step - 1;
        const int64_t k = count(p < hi ? pred(p) : false);
        const int64_t nlo = lo + k * step, cap = step == 1 ? nlo : nlo + step - 1;                          // the first probe that failed bounds the answer
        hi = cap < hi ? cap : hi;
        lo = nlo;
    }
    return lo;
}
static __global__ __launch_bounds__(MBLOCK) void k_sorted_plan(PushArgs a, ShardPlan* plan, SortedPlanJob jb)
{
    static_assert(MJB == SP_TILE && MAX_SHARDS + 1 <= MBLOCK, "one workgroup = one tile; one thread per shard boundary in the tail");
    __shared__ uint64_t s_v[SP_DIRECT_TILES + 1];
    __shared__ uint64_t s_w[NWAVES], s_sp[NWAVES];
    __shared__ int s_cnt[2][NWAVES];
    __shared__ int64_t s_F[MAX_SHARDS + 1];
    __shared__ uint64_t s_eN;
    __shared__ int s_last;
    const int h = a.G > 1 ? (int)blockIdx.x + 1 : 0, tid = (int)threadIdx.x, lane = lane_id(), wv = wave_id();   // the workgroup's boundary (interior; one shard: none)
    const uint64_t N = (uint64_t)a.n_global;
    uint64_t S = 0, lo = 0, lo_me = 0;
    mbox_wait_block(a.wait_tot);
    for (int g = 0; g < a.G; ++g) {
        const uint64_t v = (uint64_t)ld_gathered(a.tot_all + 5 * g, a.wait_tot.tags != nullptr);
        if (g < h) lo += v;
        if (g < a.me) lo_me += v;
        S += v;
    }
    const int64_t ntl = jb.ntl;
    if (!jb.have_vlo && a.G > 1) {
        // vlo[t] = floor((g_0 + ... + g_{t-1}) 2^64 / (sum g + 1)) for every tile (k_sorted_tiles' arithmetic): thread i owns consecutive tiles
        const int64_t per = (ntl + MBLOCK - 1) / MBLOCK;
        const int64_t t0 = (int64_t)tid * per, t1 = t0 + per < ntl ? t0 + per : ntl;
        uint64_t mine = 0;
        for (int64_t t = t0; t < t1; ++t) mine += jb.g[t];
        const uint64_t inc = wave_scan_u64(mine);
        if (lane == WAVE - 1) s_w[wv] = inc;
        __syncthreads();
        uint64_t run = inc - mine, tot = 0;
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) { run += w < wv ? s_w[w] : 0; tot += s_w[w]; }
        const Div128 dv = div128_setup(tot + 1);
        for (int64_t t = t0; t < t1; ++t) {
            s_v[t] = div128(run, dv);
            run += jb.g[t];
        }
        if (tid == 0) s_v[ntl] = div128(tot, dv);
        __syncthreads();
    }
    const uint64_t* const V = jb.have_vlo ? jb.vlo : s_v;
    int64_t f;
    if (h == 0) f = 0;
    else if (h >= a.G || lo >= S) f = (int64_t)N;                     // (block-uniform branches)
    else {
        // tiles whose targets all lie below lo: the tile's targets are <= mulhi(vlo[t + 1], S); the sorted uniforms spread evenly over the
        // tiles whatever the weights, so lo / S names the tile up to a few
        const int64_t guess = (int64_t)((double)lo / (double)S * (double)ntl);
        const int64_t ts = block_partition_point(ntl, guess, s_cnt, [&](int64_t t) { return mulhi64(V[t + 1], S) < lo; });
        if (ts >= ntl) f = (int64_t)N;
        else {
            // the targets of tile ts, as k_search_strat<true> forms them (global coordinates), and how many of them lie below lo
            const int64_t base = ts * MJB, n_all = (int64_t)N - base;
            uint64_t e[MSLOTS];
            lane_spacings(a.seed, a.epoch, (uint32_t)(base + MSLOTS * tid), e);
            uint64_t run = 0;
#pragma unroll
            for (int k = 0; k < MSLOTS; ++k) { run += MSLOTS * tid + k < n_all ? e[k] : 0; e[k] = run; }
            const uint64_t inc = wave_scan_u64(run);
            if (lane == WAVE - 1) s_sp[wv] = inc;
            if (tid == 0) s_eN = MJB >= n_all ? spacing_of(resample_u64(a.seed, (uint32_t)N, a.epoch)) : 0;   // the (N + 1)-th spacing belongs to the last tile
            __syncthreads();
            uint64_t st = 1 + s_eN, wex = 0;
#pragma unroll
            for (int w = 0; w < NWAVES; ++w) { st += s_sp[w]; wex += w < wv ? s_sp[w] : 0; }
            const uint64_t vlo = V[ts], Wt = V[ts + 1] - vlo;
            const uint64_t Tlo = mulhi64(vlo, S), Tw = mulhi64(vlo + Wt, S) - Tlo;
            const double inv_s = 1.0 / (double)st, dTw = (double)Tw;
            const uint64_t off = wex + (inc - run);
            int c = 0;
#pragma unroll
            for (int k = 0; k < MSLOTS; ++k)
                c += MSLOTS * tid + k < n_all && sorted_target(off + e[k], inv_s, Tlo, Tw, dTw) < lo ? 1 : 0;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m, WAVE);
            if (lane == 0) s_cnt[0][wv] = c;                          // (the partition point's last round used the other parity or is behind a barrier)
            __syncthreads();
            int64_t cb = 0;
#pragma unroll
            for (int w = 0; w < NWAVES; ++w) cb += s_cnt[0][w];
            f = base + cb;
        }
    }
    // ---- the last workgroup to arrive turns the boundaries into the plan
    if (gridDim.x > 1) {
        if (tid == 0) {
            __hip_atomic_store(jb.F + h, f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sys_stores_acknowledged();                   // (F[h] is an agent-scope atomic store, read back with agent-scope atomic loads: no L2 write-back needed)
            const unsigned int old = __hip_atomic_fetch_add(jb.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = old == gridDim.x - 1 ? 1 : 0;
        }
        __syncthreads();
        if (!s_last) return;
    }
    if (tid <= a.G) s_F[tid] = tid == 0 ? 0 : (tid == a.G ? (int64_t)N : (gridDim.x > 1 ? __hip_atomic_load(jb.F + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : f));
    __syncthreads();
    const int q = tid;
    if (q < a.G) {
        // sent to shard q: the served slots that lie in q's slot range; received from shard q: q's served slots in this shard's range
        const int64_t s0 = s_F[a.me] > a.bounds[q] ? s_F[a.me] : a.bounds[q], s1 = s_F[a.me + 1] < a.bounds[q + 1] ? s_F[a.me + 1] : a.bounds[q + 1];
        const int64_t r0 = s_F[q] > a.bounds[a.me] ? s_F[q] : a.bounds[a.me], r1 = s_F[q + 1] < a.bounds[a.me + 1] ? s_F[q + 1] : a.bounds[a.me + 1];
        const int64_t ns = s1 > s0 ? s1 - s0 : 0, nr = r1 > r0 ? r1 - r0 : 0;
        a.counts[q * COUNT_STRIDE] = ns; a.counts[(a.G + q) * COUNT_STRIDE] = nr;
        if (a.traffic && q != a.me) {
            if (ns) atomicAdd(reinterpret_cast<unsigned long long*>(a.traffic), (unsigned long long)ns);
            if (nr) atomicAdd(reinterpret_cast<unsigned long long*>(a.traffic + 1), (unsigned long long)nr);
        }
        if (q == a.me) { plan->own_range[0] = nr > 0 ? r0 - a.bounds[a.me] : 0; plan->own_range[1] = nr > 0 ? r1 - a.bounds[a.me] : 0; }
        if (a.host_counts) {
            __hip_atomic_store(a.host_counts + q, ns, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(a.host_counts + a.G + q, nr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            sys_stores_acknowledged();                       // (before the barrier in front of the ticket)
        }
    }
    if (q == 0) {
        plan->ws.S = S; plan->ws.sB = S / N; plan->ws.srem = S % N; plan->ws.sinv = (double)N / (double)S;
        plan->first = s_F[a.me]; plan->count = s_F[a.me + 1] - s_F[a.me]; plan->t_off = lo_me;
        plan->n_shards = a.G;
        if (gridDim.x > 1) __hip_atomic_store(jb.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (q <= a.G) plan->bounds[q] = a.bounds[q];
    __syncthreads();
    if (q == 0 && a.host_counts) publish_behind_sys_stores(a.host_counts + 2 * MAX_SHARDS, a.ticket);   // (the counts: acknowledged before the barrier)
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
        for (int g = 0; g < a.G; ++g) { s_off[g] = o; o += a.counts[g * COUNT_STRIDE]; s_bnd[g] = a.bounds[g]; }
        s_off[a.G] = o;
        if (blockIdx.x == 0 && a.host_counts) {   // (as in k_push: the host reads the counts while the look-ups run)
            for (int g = 0; g < 2 * a.G; ++g)
                __hip_atomic_store(a.host_counts + g, g < a.G ? a.counts[g * COUNT_STRIDE] : recv_count(a.counts, a.G, g - a.G), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            publish_behind_sys_stores(a.host_counts + 2 * MAX_SHARDS, a.ticket);
        }
        if (blockIdx.x == 0) push_traffic(a);
    }
    const MultiTable tb = multi_table_load<LOGG>(lw_, ntiles, (uint64_t)ld_gathered(a.tot_all + 5 * a.me, a.wait_tot.tags != nullptr), reinterpret_cast<uint32_t*>(smem), [] {});
    const int64_t total = s_off[a.G] < capacity ? s_off[a.G] : capacity;
    for (int64_t base = (int64_t)blockIdx.x * NE * SBLOCK; base < total; base += (int64_t)gridDim.x * NE * SBLOCK) {
        int64_t e[NE]; bool act[NE]; uint64_t T[NE]; uint32_t slot[NE]; int gdst[NE];
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            e[u] = base + u * SBLOCK + threadIdx.x;
            act[u] = e[u] < total;
            const int64_t ee = act[u] ? e[u] : total - 1;
            int g = 0;
            while (g < a.G - 1 && ee >= s_off[g + 1]) ++g;
            gdst[u] = g;
            const ulonglong2 q = a.stage[s_bnd[g] + (ee - s_off[g])];
            T[u] = q.x & STAGE_T_MASK;
            slot[u] = (uint32_t)q.y;
        }
        uint32_t idx[NE];
        multi_lookup<LOGG, NE>(tb, lw_, n, T, idx);
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            if (!act[u]) continue;
            const double2* src = reinterpret_cast<const double2*>(rows + (int64_t)idx[u] * W);
            if (a.ring.peers) {                                   // the window exchange (as k_push)
                ring_store(a.ring.peers[gdst[u]] + a.ring.off + (int64_t)slot[u] * (W + 2), rows + (int64_t)idx[u] * W, W, (uint64_t)(gid0 + idx[u]), a.ring.seq);
                continue;
            }
            double* dst = packed_out + e[u] * (W + 1 + a.extra);
            if (a.extra) dst[W + 1] = a.pv.lw[idx[u]] - a.pv.at(idx[u]);
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
                                                         double logN, Scalars* sc, int in_mailbox)
{
    // update_lml_est! (resample.jl:178-182) from the gathered global summary: lml += (m + log(S 2^-K)) - log N
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        uint64_t S = 0;
        double mx = -__builtin_huge_val();
        int f = 0;
        for (int g = 0; g < G; ++g) {
            S += (uint64_t)ld_gathered(tot_all + 5 * g, in_mailbox != 0);
            const double v = ld_gathered(mf_all + 2 * g, in_mailbox != 0); mx = v > mx ? v : mx; f |= (int)ld_gathered(mf_all + 2 * g + 1, in_mailbox != 0);
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

// ... the same out of the slot-addressed receive window (gpf_k_common.hpp): the slots outside the shard's own range [own_range[0], own_range[1])
// were written by the peers that serve them -- entry j = [row | global ancestor id | seal], waited for per entry
template <int W>
__global__ __launch_bounds__(BLOCK) void k_commit_ring(RingIn ring, int64_t n, const int64_t* __restrict__ own_range, double* __restrict__ rows_new,
                                                       int32_t* __restrict__ anc, double* __restrict__ lw,
                                                       const double* __restrict__ mf_all, const int64_t* __restrict__ tot_all, int G, int K,
                                                       double logN, Scalars* sc, int in_mailbox)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {             // update_lml_est! (resample.jl:178-182), as k_commit_packed
        uint64_t S = 0;
        double mx = -__builtin_huge_val();
        int f = 0;
        for (int g = 0; g < G; ++g) {
            S += (uint64_t)ld_gathered(tot_all + 5 * g, in_mailbox != 0);
            const double v = ld_gathered(mf_all + 2 * g, in_mailbox != 0); mx = v > mx ? v : mx; f |= (int)ld_gathered(mf_all + 2 * g + 1, in_mailbox != 0);
        }
        if (!(f & FLAG_NAN) && mx == -__builtin_huge_val()) f |= FLAG_ALL_NEGINF;
        sc->lml_est = sc->lml_est + (lse_from(mx, S, K, f) - logN);
    }
    // own_range: the shard's own hits are one slot range (ascending targets); nullptr: they are the slots with anc >= 0 (k_search_own: i.i.d. targets),
    // every slot with anc < 0 sits in the window
    const int64_t olo = own_range ? own_range[0] : 0, cnt = own_range ? own_range[1] - own_range[0] : 0;
    for (int64_t k = (int64_t)blockIdx.x * BLOCK + threadIdx.x; k < n - cnt; k += (int64_t)gridDim.x * BLOCK) {
        const int64_t j = k < olo ? k : k + cnt;
        if (!own_range && anc[j] >= 0) continue;
        double r[W];
        const uint64_t ga = ring_load<W>(ring, j, r);
        double* dst = rows_new + j * W;
#pragma unroll
        for (int c = 0; c < W; ++c) dst[c] = r[c];
        anc[j] = (int32_t)ga;
        lw[j] = 0.0;                                   // update_weights!, resample.jl:195
    }
}

// a prioritised sharded resample: entries are [row | meta | log_ws]; the log-ML estimate moves by the RAW weights' summary
// (update_lml_est!, resample.jl:57,178-182), the new log-weights wait for the global logsumexp of log_ws (k_shard_apply_post)
template <int W>
__global__ __launch_bounds__(BLOCK) void k_commit_packed_ws(const double* __restrict__ packed, int64_t m, double* __restrict__ rows_new,
                                                            int32_t* __restrict__ anc, double* __restrict__ lws,
                                                            const double* __restrict__ mf_raw, const int64_t* __restrict__ tot_raw, int G, int K,
                                                            double logN, Scalars* sc, int in_mailbox)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        uint64_t S = 0;
        double mx = -__builtin_huge_val();
        int f = 0;
        for (int g = 0; g < G; ++g) {
            S += (uint64_t)ld_gathered(tot_raw + 5 * g, in_mailbox != 0);
            const double v = ld_gathered(mf_raw + 2 * g, in_mailbox != 0); mx = v > mx ? v : mx; f |= (int)ld_gathered(mf_raw + 2 * g + 1, in_mailbox != 0);
        }
        if (!(f & FLAG_NAN) && mx == -__builtin_huge_val()) f |= FLAG_ALL_NEGINF;
        sc->lml_est = sc->lml_est + (lse_from(mx, S, K, f) - logN);
    }
    for (int64_t k = (int64_t)blockIdx.x * BLOCK + threadIdx.x; k < m; k += (int64_t)gridDim.x * BLOCK) {
        const double* src = packed + k * (W + 2);
        const uint64_t meta = d2u(src[W]);
        const int64_t j = (int64_t)(meta >> 32);
        double* dst = rows_new + j * W;
#pragma unroll
        for (int c = 0; c < W; ++c) dst[c] = src[c];
        anc[j] = (int32_t)(meta & 0xffffffffull);
        lws[j] = src[W + 1];                               // log_ws = lw[parents] - lp[parents], resample.jl:198
    }
}
// lw = log_ws + (log N - logsumexp(log_ws)) with the logsumexp over ALL shards (resample.jl:200), from the gathered post summaries
static __global__ __launch_bounds__(BLOCK) void k_shard_apply_post(const double* __restrict__ mf_all, const int64_t* __restrict__ tot_all, int G, int K,
                                                            double logN, const double* __restrict__ lws, double* __restrict__ lw, int64_t n,
                                                            MboxWait wait)
{
    mbox_wait_block(wait);
    const bool mb = wait.tags != nullptr;
    uint64_t S = 0;
    double mx = -__builtin_huge_val();
    int f = 0;
    for (int g = 0; g < G; ++g) {
        S += (uint64_t)ld_gathered(tot_all + 5 * g, mb);
        const double v = ld_gathered(mf_all + 2 * g, mb); mx = v > mx ? v : mx; f |= (int)ld_gathered(mf_all + 2 * g + 1, mb);
    }
    if (!(f & FLAG_NAN) && mx == -__builtin_huge_val()) f |= FLAG_ALL_NEGINF;
    const double off = logN - lse_from(mx, S, K, f);
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK)
        lw[i] = lws[i] + off;
}

// ---- sharded STRATIFIED resampling with the reference's default sort_particles = true (src/resample.jl:145,156-157): the REPLICATED plan.  A global
// descending sort has no shard-local form -- position p of the sorted order can hold any shard's particle -- so every rank gathers ALL log-weights (8 bytes
// per global particle, one all-gather) and runs the unsharded sort + scan + search on them itself (a planner filter of n_global particles without rows: the
// very kernels of the single-GPU path on the very same numbers, hence the same ancestors for any number of shards).  Every rank then holds the ancestor of
// EVERY global slot: who serves which slot needs no message, the exchange counts no exchange.  The rows travel as the i.i.d. resamplers' do (packed
// entries [row | slot | ancestor id], own hits in place through the ancestor array).
struct AncPlan {
    const int32_t* anc_g;                         // [n_global] ancestor (global id) of every global slot, on every rank
    int64_t n_global; int G, me;
    int64_t base, extra;                          // the contiguous-range rule: shard g holds base + (g < extra) particles
    int64_t lo, hi;                               // this shard's particles / slots [lo, hi)
};
__device__ __forceinline__ int anc_owner(const AncPlan& p, int64_t x)
{
    const int64_t wide = p.extra * (p.base + 1);
    return x < wide ? (int)(x / (p.base + 1)) : (int)(p.extra + (x - wide) / p.base);
}
__device__ __forceinline__ int64_t anc_bound(const AncPlan& p, int g) { return (int64_t)g * p.base + (g < p.extra ? g : p.extra); }
constexpr int ANC_BLOCK = 256, ANC_PER = 8, ANC_CHUNK = ANC_BLOCK * ANC_PER;
// counts[g] = entries this shard sends to shard g, counts[G + q] = entries (own hits included) shard q serves to this shard's slots; own != 0: the
// shard's own hits go into anc_local as global ids (-1 = arrives packed) and are not counted as sent
static __global__ __launch_bounds__(ANC_BLOCK) void k_anc_count(AncPlan p, int own, int32_t* __restrict__ anc_local, int64_t* __restrict__ counts)
{
    __shared__ unsigned int s_send[MAX_SHARDS], s_recv[MAX_SHARDS];
    for (int g = threadIdx.x; g < MAX_SHARDS; g += ANC_BLOCK) { s_send[g] = 0; s_recv[g] = 0; }
    __syncthreads();
    for (int64_t j = (int64_t)blockIdx.x * ANC_BLOCK + threadIdx.x; j < p.n_global; j += (int64_t)gridDim.x * ANC_BLOCK) {
        const int64_t a = p.anc_g[j];
        const bool mine = a >= p.lo && a < p.hi, to_me = j >= p.lo && j < p.hi;
        if (to_me) {
            atomicAdd(&s_recv[mine ? p.me : anc_owner(p, a)], 1u);
            if (own) anc_local[j - p.lo] = mine ? (int32_t)a : -1;
        }
        if (mine && !(own && to_me)) atomicAdd(&s_send[anc_owner(p, j)], 1u);
    }
    __syncthreads();
    for (int g = threadIdx.x; g < p.G; g += ANC_BLOCK) {
        if (s_send[g]) atomicAdd(reinterpret_cast<unsigned long long*>(counts + (int64_t)g * COUNT_STRIDE), (unsigned long long)s_send[g]);
        if (s_recv[g]) atomicAdd(reinterpret_cast<unsigned long long*>(counts + (int64_t)(p.G + g) * COUNT_STRIDE), (unsigned long long)s_recv[g]);
    }
}
// the rows this shard serves, packed by destination ([row | slot inside the destination << 32 | ancestor id]; any order inside a destination's group:
// every entry names its slot); cursors[G] start at zero; stops at the capacity (the host repeats the call with a larger buffer if the counts say so)
template <int W>
__global__ __launch_bounds__(ANC_BLOCK) void k_anc_pack(AncPlan p, int own, const double* __restrict__ rows, const int64_t* __restrict__ counts,
                                                        unsigned long long* __restrict__ cursors, int64_t capacity, double* __restrict__ packed_out)
{
    __shared__ unsigned int s_cnt[MAX_SHARDS];
    __shared__ int64_t s_base[MAX_SHARDS];
    __shared__ int64_t s_off[MAX_SHARDS + 1];
    if (threadIdx.x == 0) {
        int64_t o = 0;
        for (int g = 0; g < p.G; ++g) { s_off[g] = o; o += counts[(int64_t)g * COUNT_STRIDE]; }
        s_off[p.G] = o;
    }
    const int64_t nch = (p.n_global + ANC_CHUNK - 1) / ANC_CHUNK;
    for (int64_t c = blockIdx.x; c < nch; c += gridDim.x) {
        __syncthreads();
        for (int g = threadIdx.x; g < p.G; g += ANC_BLOCK) s_cnt[g] = 0;
        __syncthreads();
        int dst[ANC_PER]; unsigned int rank[ANC_PER]; int64_t anc[ANC_PER];
#pragma unroll
        for (int u = 0; u < ANC_PER; ++u) {
            const int64_t j = c * ANC_CHUNK + (int64_t)u * ANC_BLOCK + threadIdx.x;
            dst[u] = -1;
            if (j >= p.n_global) continue;
            const int64_t a = p.anc_g[j];
            if (a < p.lo || a >= p.hi) continue;
            if (own && j >= p.lo && j < p.hi) continue;
            anc[u] = a;
            dst[u] = anc_owner(p, j);
            rank[u] = atomicAdd(&s_cnt[dst[u]], 1u);
        }
        __syncthreads();
        for (int g = threadIdx.x; g < p.G; g += ANC_BLOCK)
            s_base[g] = s_cnt[g] ? (int64_t)atomicAdd(cursors + g, (unsigned long long)s_cnt[g]) : 0;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < ANC_PER; ++u) {
            if (dst[u] < 0) continue;
            const int64_t j = c * ANC_CHUNK + (int64_t)u * ANC_BLOCK + threadIdx.x;
            const int64_t e = s_off[dst[u]] + s_base[dst[u]] + rank[u];
            if (e >= capacity) continue;
            const double* src = rows + (anc[u] - p.lo) * W;
            double* out = packed_out + e * (W + 1);
#pragma unroll
            for (int k = 0; k < W; ++k) out[k] = src[k];
            out[W] = u2d(((uint64_t)(j - anc_bound(p, dst[u])) << 32) | (uint64_t)anc[u]);
        }
    }
}
// the gathered log-weights of shards of unequal size (n_global % G != 0: every rank contributes `per` = base + 1 words, the last of some unused) -> dense
static __global__ __launch_bounds__(BLOCK) void k_anc_compact(AncPlan p, const double* __restrict__ gathered, int64_t per, double* __restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < p.n_global; i += (int64_t)gridDim.x * BLOCK) {
        const int g = anc_owner(p, i);
        out[i] = gathered[(int64_t)g * per + (i - anc_bound(p, g))];
    }
}

} // namespace gpf
