// constants, device scalars, wave helpers  (part of gpf_kernels.hpp; include that header, not this file)
#pragma once

namespace gpf {

constexpr int BLOCK = 256;
constexpr int WAVE = 64;
constexpr int NWAVES = BLOCK / WAVE;
constexpr int SCAN_ITEMS = 8;
constexpr int TILE = BLOCK * SCAN_ITEMS;          // 2048 weights per scan tile
constexpr int MAX_PARTIALS = 2048;                // most workgroups of a kernel that leaves the maximum of the log-weights behind
// maximum(vs) / any(isnan) of safe_softmax (utils.jl:119-128), left behind by the kernel that WRITES the log-weights: every workgroup
// folds its block maximum into one of MAX_SLOTS slots with ONE fire-and-forget integer atomic max on an order-preserving key (and an
// atomic or of its flags, only when it has any); the consumer reads the 32 slots with one load per lane and a wave reduction -- no
// LDS, no barrier.  (Before: one partial per workgroup and a fold of <= 1024 of them in every workgroup of the scan, ~3.5 us of its
// 13.5, profiles/r03_scan_phases.txt.)  A slot is a 128-byte line of its own: atomics to one line serialise (~8 ns each), ~32 per
// line and launch stay invisible.  Two slot arrays alternate: a producer folds into one and clears the other for the next producer.
constexpr int MAX_SLOTS = 32;
constexpr int SLOT_WORDS = 16;                    // 8-byte words per slot: {key, flags, pad}
struct MaxSlots { unsigned long long* cur; unsigned long long* clear; };
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
// residual resampling, deterministic head: cells with very many copies are not filled by the one workgroup that meets them in
// k_scan_residual2 but listed here and filled by the whole grid of the search kernel that follows.  No reset between resamples: the word
// carries the resample's tag (epoch), a list with another tag is empty.
constexpr int GIANT_MAX = 32;
constexpr uint32_t GIANT_COPIES = 16384;
struct HeadGiants {
    unsigned int word;                    // tag << 8 | entries
    unsigned int pad;
    uint64_t start[GIANT_MAX];
    uint32_t cell[GIANT_MAX], cnt[GIANT_MAX];
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
    int32_t  gate_go;          // gpf_step_ess: 1 = the ESS fell below the threshold (k_sum_host<GATE>); the speculative propagate behind it returns at once
    long long opt_d;           // optimal resize: threshold position in the descending order (-1: none)
    uint64_t opt_a, opt_B;     // optimal resize: inverse weight threshold c = a S / B as the exact pair (a, B)
    HeadGiants giants;         // residual: cells with >= GIANT_COPIES copies of the current resample
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
// order-preserving key of a non-NaN double: larger value <=> larger key; key 0 is below every value ("no workgroup wrote")
__device__ __forceinline__ unsigned long long max_key(double v)
{
    const uint64_t b = d2u(v);
    return (b >> 63) ? ~b : (b | (1ull << 63));
}
__device__ __forceinline__ double max_unkey(unsigned long long k)
{
    if (k == 0) return -__builtin_huge_val();
    return u2d((k >> 63) ? (k & ~(1ull << 63)) : ~k);
}
// every lane of the calling wave ends with (maximum, flags) of the slots; any number of waves may call it, no LDS, no barrier
__device__ __forceinline__ void fold_slots(const unsigned long long* __restrict__ slots, double& m_out, int& f_out)
{
    static_assert(MAX_SLOTS <= WAVE, "one lane per slot");
    const int l = (int)(threadIdx.x % WAVE);
    const unsigned long long k = l < MAX_SLOTS ? slots[l * SLOT_WORDS] : 0ull;
    int f = l < MAX_SLOTS ? (int)slots[l * SLOT_WORDS + 1] : 0;
    const double m = wave_max_f64(max_unkey(k));
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) f |= __shfl_xor(f, s, WAVE);
    if (!(f & FLAG_NAN) && m == -__builtin_huge_val()) f |= FLAG_ALL_NEGINF;
    m_out = m; f_out = f;
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

// ----------------------------------------------------------------------------- shard mailboxes (multi-GPU summaries)
// The summaries a sharded resample exchanges are 16-40 bytes per rank: (max, flags), {S, sum q^2 limbs}, residual {Ctot, Rs}
// (SURVEY.md §2.3 C1-C4).  As RCCL all-gathers they cost ~24 us of launch + protocol each, more than the kernels between them.
// Here the PRODUCING kernel stores them straight into every peer's mailbox (device memory mapped through hipIpc; xGMI peer
// writes) and the CONSUMING kernel waits for the G tagged entries it needs: no collective, no host involvement, no extra launch.
//   mailbox = MB_KINDS x MB_SLOTS x { payload [G][words(kind)] u64, dense like the gathered arrays ; tags [MAX_SHARDS] u64 }
//   round `seq` (1, 2, ... per kind, the same on every rank: SPMD call order) uses slot seq & (MB_SLOTS - 1); entry [src] is
//   written by rank src alone: payload words (system-scope stores), then its tag = seq (system-scope RELEASE).  A reader spins
//   on tag == seq (system-scope ACQUIRE), then reads the payload with system-scope loads (they bypass the non-coherent L2, so
//   the memory type of the mailbox does not matter).  Tags only grow: slots are never cleared; a peer can run at most one
//   round ahead (it needs this rank's entry of round r to finish round r), so four slots never alias.
constexpr int MB_KINDS = 4, MB_SLOTS = 4;
enum : int { MB_MF = 0, MB_TOT = 1, MB_CR = 2, MB_CAL = 3 /* gpf_comm_calibrate's rounds */ };
__host__ __device__ constexpr int mb_words(int kind) { return kind == MB_TOT ? 5 : 2; }
// offsets in u64 words inside a mailbox
__host__ __device__ constexpr int64_t mb_payload_off(int kind, int slot)
{
    int64_t o = 0;
    for (int k = 0; k < kind; ++k) o += (int64_t)MB_SLOTS * (MAX_SHARDS * mb_words(k) + MAX_SHARDS);
    return o + (int64_t)slot * (MAX_SHARDS * mb_words(kind) + MAX_SHARDS);
}
__host__ __device__ constexpr int64_t mb_tag_off(int kind, int slot) { return mb_payload_off(kind, slot) + (int64_t)MAX_SHARDS * mb_words(kind); }
constexpr int64_t MB_TOTAL_WORDS = mb_payload_off(MB_KINDS, 0);

__device__ __forceinline__ uint64_t ld_sys(const uint64_t* p) { return __hip_atomic_load(const_cast<uint64_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ int64_t ld_sys(const int64_t* p) { return (int64_t)ld_sys(reinterpret_cast<const uint64_t*>(p)); }
__device__ __forceinline__ double ld_sys(const double* p) { return u2d(ld_sys(reinterpret_cast<const uint64_t*>(p))); }

// gathered summaries that MAY sit in the mailbox: system-scope loads there, ordinary (cached) loads when they came by a collective
template <class T> __device__ __forceinline__ T ld_gathered(const T* p, bool in_mailbox) { return in_mailbox ? ld_sys(p) : *p; }

// A ticket / seal published BEHIND payload words that the same thread stored with system-scope atomic stores.  Those stores are write-through
// (sc0 sc1): all that "behind" needs is their acknowledgement, s_waitcnt vmcnt(0).  A system-scope RELEASE store puts `buffer_wbl2 sc0 sc1` in
// front instead -- a write-back of the whole L2, i.e. of the tens of MB of rows the propagate has just written: 3 - 5 us in kernels that are
// otherwise one wave (k_pack_mflags 5.5 us, k_set_global / k_export_residual 4.8, +3 us on the weight scan behind a mailbox wait).
// ISA DEPENDENCY (checked at compile time below): on the gfx9 family (gfx90a / gfx942 / gfx950) stores and non-returning atomics are counted by vmcnt;
// gfx10+ counts them in vscnt, where this wait would order nothing.  What stands behind this wait must be (a) atomic stores / non-returning atomics
// of the SAME thread -- write-through at their scope, so "acknowledged" means "performed at that scope" --, never plain stores (those may still sit
// in the L2 of this XCD), and (b) read back with atomic loads of at least the same scope.  The mailbox entries and window entries do not even rely
// on this: their seals validate the words whatever order they arrive in; the accumulator lines + arrival counters of k_sum_shard / k_sorted_plan
// and the pinned tickets do (agent- / system-scope atomics on both sides).
#if defined(__HIP_DEVICE_COMPILE__) && !(defined(__gfx90a__) || defined(__gfx942__) || defined(__gfx950__))
#error "sys_stores_acknowledged() relies on vmcnt counting stores (gfx90a / gfx942 / gfx950); on another ISA use __builtin_amdgcn_fence(__ATOMIC_RELEASE, \"agent\")"
#endif
__device__ __forceinline__ void sys_stores_acknowledged() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <class T> __device__ __forceinline__ void publish_behind_sys_stores(T* p, T v)
{
    sys_stores_acknowledged();
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// the seal of a mailbox entry: the round's sequence number folded with the entry's words (odd multiplier: every step a bijection; never 0 for
// seq != 0 over the zeroed mailbox).  The consumer recomputes it from the words IT reads: an entry whose words and seal did not arrive together
// (in whatever order the fabric delivered them) does not match and is polled again -- the protocol needs no ordering between the stores.
__device__ __forceinline__ uint64_t mbox_seal(uint64_t seq, const uint64_t* words, int n)
{
    uint64_t x = seq;
    for (int k = 0; k < n; ++k) x = (x ^ words[k]) * 0x9E3779B97F4A7C15ull;
    return x;
}
struct MboxPush {              // where a producer's summary goes (peers == nullptr: nowhere, the caller gathers it with a collective)
    uint64_t* const* peers;    // device array [G]: base of every rank's mailbox as mapped HERE (peers[me] = the own one)
    int64_t payload_off, tag_off;   // of (kind, slot), before the [me] index
    uint64_t tag;              // = seq
    int G, me, nwords;
};
// wave-collective (all 64 lanes call it with the same words): lane l < G stores this rank's entry into rank l's mailbox
__device__ __forceinline__ void mbox_push_wave(const MboxPush& p, const uint64_t* words)
{
    const int l = lane_id();
    if (p.peers && l < p.G) {
        uint64_t* base = p.peers[l];
        uint64_t* dst = base + p.payload_off + (int64_t)p.me * p.nwords;
        for (int k = 0; k < p.nwords; ++k) __hip_atomic_store(dst + k, words[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // (the seal validates the words whatever the order of arrival; sent behind their acknowledgement all the same, so that the consumer's FIRST
        //  look at a sealed entry finds its words -- a mismatch costs it a sleep and one more round trip: measured +2 - 3 us per step without the wait)
        publish_behind_sys_stores(base + p.tag_off + p.me, mbox_seal(p.tag, words, p.nwords));
    }
}
struct MboxWait {              // what a consumer waits for (tags == nullptr: nothing -- the data came by a collective or is local)
    const uint64_t* tags;      // own mailbox + tag_off(kind, slot): the seals; the entries ([MAX_SHARDS][nwords]) stand right in front of them
    uint64_t want;             // = seq
    int n, nwords;             // ranks; words per entry
    int32_t* timeout;          // pinned host flag: set to 2 when a peer's entry did not arrive in time
};
constexpr unsigned MB_SPIN_LIMIT = 1u << 23;       // x ~1-2 us per probe: a peer may be ~10 s late (host preempted) before the wait gives up and flags the run
// block-collective: returns when the entries of all a.n ranks have arrived (threads < n poll one tag each)
__device__ __forceinline__ void mbox_wait_block(const MboxWait& w)
{
    if (!w.tags) return;
    if ((int)threadIdx.x < w.n) {
        const uint64_t* entry = w.tags - (int64_t)MAX_SHARDS * w.nwords + (int64_t)threadIdx.x * w.nwords;
        unsigned spins = 0;
        for (;;) {
            // seal and words in one round trip (independent system-scope loads: they bypass the caches, as the consumers' ld_gathered will)
            const uint64_t seal = ld_sys(w.tags + threadIdx.x);
            uint64_t x = w.want;
            for (int k = 0; k < w.nwords; ++k) x = (x ^ ld_sys(entry + k)) * 0x9E3779B97F4A7C15ull;
            if (seal == x) break;
            __builtin_amdgcn_s_sleep(8);
            if (++spins > MB_SPIN_LIMIT) { __hip_atomic_store(w.timeout, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
        }
    }
    __syncthreads();
}

// ----------------------------------------------------------------------------- slot-addressed receive window (multi-GPU slab exchange)
// The resamplers with ascending targets (stratified, sorted multinomial) exchange boundary slabs: a few thousand rows per shard boundary
// (DESIGN.md 6.7).  As a grouped ncclSend / ncclRecv that costs a host wait for the split sizes and the group's latency -- all of the
// predicted weak-scaling loss of BASELINE configs[2].  Instead every rank exports a WINDOW of one entry per local slot (device memory mapped
// by its peers through hipIpc, like the mailboxes); the merge kernel of the shard that SERVES slot j of rank g (k_search_strat's pack loop)
// stores the entry [row (W words) | global ancestor id | seal] straight into window_g[j] (xGMI peer stores), and rank g's next propagate
// (k_step<GATHER>, PackedCommit::ring) reads window[j] for the slots outside its own range.  Every slot is served by exactly one shard and
// the window is addressed by the DESTINATION slot: no counts, no offsets, no capacity to overflow, nothing for the host to wait for.
// seal = the exchange's sequence number folded with the entry's words (mbox_seal): the consumer recomputes it from the words it reads and
// polls until it matches, so the protocol needs no ordering between the stores, and an entry of an earlier exchange (another seq) never
// validates.  Two parities alternate (a peer cannot start exchange r + 1 before this rank has sent it the summaries of r + 1, which this
// rank's stream only does behind the propagate that consumed r: one parity would do; the second is slack).
constexpr int RING_PARITIES = 2;
struct RingOut { uint64_t* const* peers; int64_t off; uint64_t seq; };     // producer: peers[g] = rank g's window as mapped HERE (nullptr: no window), off = the parity's word offset
struct RingIn  { const uint64_t* base; uint64_t seq; int32_t* timeout; };  // consumer: own window + the parity's offset (nullptr: no window)
constexpr unsigned RING_SPIN_LIMIT = 1u << 22;     // x ~2 us per probe: as the mailbox wait, a peer may be ~10 s late before the wait gives up and flags the run
// one entry, W row words: every lane for itself (system-scope write-through stores; no ordering needed)
__device__ __forceinline__ void ring_store(uint64_t* dst, const double* row, int W, uint64_t anc, uint64_t seq)
{
    uint64_t x = seq;
    for (int c = 0; c < W; ++c) {
        const uint64_t w = d2u(row[c]);
        x = (x ^ w) * 0x9E3779B97F4A7C15ull;
        __hip_atomic_store(dst + c, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    x = (x ^ anc) * 0x9E3779B97F4A7C15ull;
    __hip_atomic_store(dst + W, anc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(dst + W + 1, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// the entry of local slot `slot`: polls until the words it reads carry the seal of exchange in.seq (system-scope loads: the peers wrote them)
template <int W>
__device__ __forceinline__ uint64_t ring_load(const RingIn& in, int64_t slot, double (&r)[W])
{
    const uint64_t* e = in.base + slot * (W + 2);
    uint64_t w[W + 2];
    for (unsigned spins = 0;; ++spins) {
#pragma unroll
        for (int k = 0; k < W + 2; ++k) w[k] = ld_sys(e + k);
        uint64_t x = in.seq;
#pragma unroll
        for (int k = 0; k < W + 1; ++k) x = (x ^ w[k]) * 0x9E3779B97F4A7C15ull;
        if (x == w[W + 1]) break;
        __builtin_amdgcn_s_sleep(8);
        // (a lane that gave up has flagged the run: the others follow at their next look instead of waiting the limit out one by one)
        if (spins > RING_SPIN_LIMIT || ((spins & 1023u) == 1023u && __hip_atomic_load(in.timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0)) {
            __hip_atomic_store(in.timeout, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
    }
#pragma unroll
    for (int c = 0; c < W; ++c) r[c] = u2d(w[c]);
    return w[W];
}

// ----------------------------------------------------------------------------- the plan of a sharded STRATIFIED resample
// Stratum j is [L(j), L(j+1)) with L ascending in j, and shard h owns the targets in [lo_h, lo_(h+1)) (lo = exclusive shard totals): the slots shard h
// serves are the contiguous range [F[h], F[h+1]), F[h] = first slot whose target is >= lo_h -- the stratum that contains lo_h, or the one after it,
// decided by that one slot's target.  Every shard derives all of F from the gathered totals (the same integers everywhere), the exchange counts follow
// by intersecting slot ranges.  The body runs in a launch of its own (k_strat_plan: totals gathered by a collective, or per-phase hosts) or -- round 6 --
// in the workgroup of the weight scan that ends up with the shard total (k_scan MODE 3, ScanExtras::splan: the plan needs nothing but the G totals,
// and that workgroup has just pushed the last of them; one launch and its gap less on every stratified resample across shards).
#ifndef GPF_COUNT_STRIDE
#define GPF_COUNT_STRIDE 16
#endif
constexpr int COUNT_STRIDE = GPF_COUNT_STRIDE;     // int64 words between the exchange counters (each on a 128-byte line of its own)
struct ShardPlan {
    WSum ws;                                                          // the GLOBAL weight sum and its strata constants
    int64_t first, count;                                             // this shard serves the global slots [first, first + count)
    uint64_t t_off;                                                   // where this shard's CDF starts in the global one
    int32_t n_shards, pad;
    int64_t bounds[MAX_SHARDS + 1];                                   // first global slot of every shard (the packed entries name slots inside their shard)
    int64_t own_range[2];                                             // the slots of THIS shard (local indices) that it serves itself: [lo, hi)
};
struct StratPlanJob {                              // plan == nullptr: no job
    ShardPlan* plan;
    uint64_t seed; uint32_t epoch; int G, me; int64_t n_global;
    const int64_t* tot_all; MboxWait wait_tot;     // [G][5] gathered {S_local, ...} (own mailbox: wait for the round first)
    int64_t* counts; int64_t* host_counts; int64_t ticket; int64_t* traffic;   // as PushArgs
    int zero_words;                                // > 0 (inside the weight scan): clear this many words of `counts` first -- the scan's own clearing is off then
};
// contiguous shard ranges: the first n_global % G shards hold one particle more (the rule every rank applies to its own gpf_config)
__host__ __device__ inline int64_t shard_bound(int64_t n_global, int G, int g)
{
    const int64_t b = n_global / G, x = n_global % G;
    return (int64_t)g * b + (g < x ? g : x);
}
// block-collective (>= MAX_SHARDS + 1 threads; F: LDS scratch [MAX_SHARDS + 1]); bnd(g) = first global slot of shard g
template <class Bnd>
__device__ __forceinline__ void strat_plan_body(const StratPlanJob& a, Bnd bnd, int64_t* F)
{
    const int h = (int)threadIdx.x;
    ShardPlan* const plan = a.plan;
    const uint64_t N = (uint64_t)a.n_global;
    if (a.zero_words > 0) {
        for (int i = h; i < a.zero_words; i += (int)blockDim.x) a.counts[i] = 0;
        __syncthreads();                                   // (the same workgroup writes its counts below)
    }
    uint64_t S = 0, lo = 0, lo_me = 0;
    mbox_wait_block(a.wait_tot);
    for (int g = 0; g < a.G; ++g) {
        const uint64_t v = (uint64_t)ld_gathered(a.tot_all + 5 * g, a.wait_tot.tags != nullptr);
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
        const int64_t bh0 = bnd(h), bh1 = bnd(h + 1), bm0 = bnd(a.me), bm1 = bnd(a.me + 1);
        const int64_t s0 = F[a.me] > bh0 ? F[a.me] : bh0, s1 = F[a.me + 1] < bh1 ? F[a.me + 1] : bh1;
        const int64_t r0 = F[h] > bm0 ? F[h] : bm0, r1 = F[h + 1] < bm1 ? F[h + 1] : bm1;
        const int64_t ns = s1 > s0 ? s1 - s0 : 0, nr = r1 > r0 ? r1 - r0 : 0;
        a.counts[h * COUNT_STRIDE] = ns; a.counts[(a.G + h) * COUNT_STRIDE] = nr;
        if (a.traffic && h != a.me) {
            if (ns) atomicAdd(reinterpret_cast<unsigned long long*>(a.traffic), (unsigned long long)ns);
            if (nr) atomicAdd(reinterpret_cast<unsigned long long*>(a.traffic + 1), (unsigned long long)nr);
        }
        if (h == a.me) { plan->own_range[0] = nr > 0 ? r0 - bm0 : 0; plan->own_range[1] = nr > 0 ? r1 - bm0 : 0; }
        if (a.host_counts) {
            __hip_atomic_store(a.host_counts + h, ns, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(a.host_counts + a.G + h, nr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            sys_stores_acknowledged();                       // (before the barrier in front of the ticket)
        }
    }
    if (h == 0) {
        plan->ws.S = S; plan->ws.sB = B; plan->ws.srem = rem; plan->ws.sinv = (double)N / (double)S;
        plan->first = F[a.me]; plan->count = F[a.me + 1] - F[a.me]; plan->t_off = lo_me;
        plan->n_shards = a.G;
    }
    if (h <= a.G) plan->bounds[h] = bnd(h);
    __syncthreads();
    if (h == 0 && a.host_counts) publish_behind_sys_stores(a.host_counts + 2 * MAX_SHARDS, a.ticket);   // (the counts: acknowledged before the barrier)
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
// Coarse 24-bit sort key (K10c): the distance d = m - v >= 0 of a log-priority from the maximum m, as {5-bit binade | 19 mantissa
// bits}: binades 2^-21 .. 2^9 (below: 0, above -- weights that underflow anyway, -inf, NaN --: all ones).  Weakly monotone in the full
// key (v1 > v2 => d1 <= d2: the subtraction rounds monotonically), so three stable 8-bit passes over it leave the keys sorted up to
// runs of equal coarse keys, which k_sort_finish orders by the full key.  Against the high 32 bits of the key itself (sign, 11 exponent
// bits, 20 mantissa bits = four passes) the exponent field shrinks to the binades log-weights actually occupy below their maximum.
constexpr int COARSE_E0 = 1023 - 21;
__device__ __forceinline__ uint32_t sort_coarse(uint64_t key, double m)
{
    const uint64_t b = d2u(m - sort_key_value(key)) & 0x7fffffffffffffffull;
    const int e = (int)(b >> 52) - COARSE_E0;
    if (e < 0) return 0u;
    if (e > 29) return 0xFFFFFFu;
    return ((uint32_t)(e + 1) << 19) | (uint32_t)((b >> 33) & 0x7FFFFu);
}


// GPF_RESAMPLE_MULTINOMIAL_SORTED (gpf_math.hpp): g[t] = the gamma total of tile t of SP_TILE slots (shape = the tile's
// slots, + 1 in the last tile: the (N + 1)-th spacing), one lane per tile.  Depends on (seed, epoch, n) alone -- not on the weights: it
// runs as extra workgroups of the weight scan's launch when there is one (k_scan, ScanExtras::sp), else on its own (k_sorted_gammas).
struct SortedGammaJob { uint64_t seed; uint64_t* g; int64_t gid0, n, ntl; uint32_t epoch; int Eg; int blocks; };
__device__ __forceinline__ void sorted_gamma_tile(const SortedGammaJob& j, int64_t t)
{
    if (t >= j.ntl) return;
    const int64_t first = t * SP_TILE, cnt = first + SP_TILE <= j.n ? SP_TILE : j.n - first;
    j.g[t] = gamma_tile(j.seed, (uint32_t)(j.gid0 + first), j.epoch, cnt + (t == j.ntl - 1 ? 1 : 0), j.Eg);
}

// gpf_step_ess: k_sum_host<GATE> adds every workgroup's partial sums into one of GATE_SLOTS accumulator lines (fire-and-forget device
// atomics: slot = workgroup & 7, words {S, flags, Q limb 0..3, -, -}); the propagate enqueued speculatively behind it reads the 64 words
// with ONE load per lane (lane = slot * 8 + word), adds the slots up with three butterfly steps and forms the verdict ESS = S^2 / Q < thr
// with the host's own arithmetic (normalise_Q + ess_from: the same IEEE operations, the same result; invalid weights: the ESS is NaN, NaN <
// thr is false).  Every wave for itself, all lanes converged, no LDS, no barrier.  Measured alternatives (profiles/r05_step_ess.txt): a fold
// by k_sum_host's last-arriving workgroup (+6 us on the reduction), a fold of per-workgroup lines by every workgroup of the propagate
// through LDS (+3.5 us on the propagate), by every wave (12 loads per lane: +10 us).
constexpr int GATE_SLOTS = 8, GATE_WORDS = GATE_SLOTS * 8;
struct GateIn { const uint64_t* acc; double thr; int32_t* go_dev; int64_t* h_gate; int64_t ticket;
                const int32_t* flag; };     // flag != nullptr (sharded filters: the verdict needs the GLOBAL sums, k_sum_reduce<SHARD> formed it): just read it
__device__ __forceinline__ bool gate_verdict(const GateIn& g)
{
    static_assert(GATE_WORDS == WAVE, "one accumulator word per lane");
    uint64_t v = g.acc[lane_id()];
    v += shfl_xor_u64(v, 8); v += shfl_xor_u64(v, 16); v += shfl_xor_u64(v, 32);      // (flags: a sum of ORs -- zero iff no flag anywhere)
    const uint64_t S = shfl_u64(v, 0), fl = shfl_u64(v, 1), q0 = shfl_u64(v, 2), q1 = shfl_u64(v, 3), q2 = shfl_u64(v, 4), q3 = shfl_u64(v, 5);
    const uint64_t lo = q0 + (q1 << 32);                                        // Q = sum of the limb sums << 32 k (Q <= S^2 < 2^124)
    const uint64_t hi = (q1 >> 32) + q2 + (q3 << 32) + (lo < q0 ? 1u : 0u);
    const bool go = fl == 0 && ess_from(S, hi, lo) < g.thr;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *g.go_dev = go ? 1 : 0;
        __hip_atomic_store(g.h_gate, (int64_t)((g.ticket << 1) | (go ? 1 : 0)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return go;
}

} // namespace gpf
