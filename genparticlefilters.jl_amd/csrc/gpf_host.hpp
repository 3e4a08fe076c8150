// gpf_host.hpp -- host-side internals shared by the translation units of libgpf_hip.so (libgpf_core / _resample / _aux / _shard .hip):
// the filter handle, the launch / error macros and the declarations of the helpers one unit defines and another calls.  Nothing here is
// part of the C ABI (include/gpf.h); everything lives in namespace gpfh and is built with hidden visibility.
#pragma once
#include "../../include/gpf.h"
#include "gpf_kernels.hpp"

#include <hip/hip_ext.h>
#include <rccl/rccl.h>        // types and enums only: librccl is loaded with dlopen (gpf_comm_create), nothing links against it
#include <dlfcn.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

using namespace gpf;

namespace gpfh {

// polite busy-wait on a pinned-memory ticket (x86 PAUSE; a plain compiler barrier elsewhere)
static inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    __asm__ __volatile__("" ::: "memory");
#endif
}
inline thread_local std::string g_err;   // errors before a handle exists

// Kernel timing (gpf_kernel_timing): inside timed() the launch carries a start/stop event pair that the runtime
// stamps at the kernel's own begin and end (hipExtLaunchKernel), so the elapsed time is the dispatch's duration, the
// same quantity rocprofv3 --kernel-trace reports -- not launch gap + kernel as with events recorded around the launch.
inline thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;
#define GPF_LAUNCH(kernel, grid, block, lds, stream, ...) \
    hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, g_ev_start, g_ev_stop, 0, __VA_ARGS__)

struct Timer {
    bool on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
};
// gpf_phase_timing: events on the handle's stream at the phase boundaries of a sharded resample and of the propagate that commits it.  A mark names the
// phase that ENDS there (-1: a sequence begins); the time of a phase = from the mark in front of it, idle gaps included (GPU timeline, not kernel time).
struct PhaseTimer {
    bool on = false;
    std::vector<std::pair<int, hipEvent_t>> marks;
    double host_wait_us = 0.0;          // wall clock the host spent blocked on the exchange's split sizes
    int64_t resamples = 0;
};

} // namespace gpfh

struct __attribute__((visibility("hidden"))) gpf_filter {
    gpf_config cfg{};
    ModelArgs args{};
    int d = 0, W = 0, K = 0;
    double logN = 0.0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int n_cu = 256;
    int64_t n = 0, ntiles = 0;
    double* rows[2] = {nullptr, nullptr};
    int cur = 0;
    double *lw = nullptr, *lws = nullptr, *lp = nullptr, *dtmp = nullptr;
    uint64_t* cdf[3] = {nullptr, nullptr, nullptr};     // padded to whole tiles
    uint64_t* t16[3] = {nullptr, nullptr, nullptr};     // coarser levels written by the scan (gpf_kernels.hpp ScanOut)
    uint64_t* t256[3] = {nullptr, nullptr, nullptr};
    uint64_t* desc[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};   // [channel][ping-pong]: agg | prefix
    int dcur[3] = {0, 0, 0};
    const uint64_t* table[3] = {nullptr, nullptr, nullptr};   // per-tile inclusive prefixes of the last scan per channel
    int32_t *anc = nullptr, *order = nullptr, *idx_in = nullptr;
    uint64_t *keys = nullptr, *keys_out = nullptr;
    void* sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    int sort_ws_cur = 0;
    int64_t* h_sort_flag = nullptr; int64_t sort_ticket = 0;   // pinned {a run too long for k_sort_finish, ticket}
    unsigned long long* mslots[2] = {nullptr, nullptr};   // MaxSlots (gpf_k_common.hpp): maximum + flags of the log-weights, two alternating arrays
    int mcur = 0;                                         // mslots[mcur]: written by the latest producer
    uint64_t* blockQ = nullptr;
    double *partial = nullptr, *dscal = nullptr;
    unsigned long long* acc_part = nullptr;   // [MAX_PARTIALS] accepted moves per workgroup of the last move-accept kernel
    double* tree_buf = nullptr; int64_t tree_cap = 0;   // partials of the weighted tree sums (statistics)
    Scalars* sc = nullptr;
    Scalars* h_sc = nullptr;       // pinned mirror
    long long* h_sc_ticket = nullptr; long long sc_ticket = 0;   // k_publish_scalars -> host polling (fetch_scalars)
    uint32_t epoch = 0;
    bool initialized = false, has_prev = false, raw_valid = false, residual_scanned = false;
    bool max_valid = false;        // mslots[mcur] describes the current log-weights (written by the kernel that produced them)
    bool raw_sum_valid = false;    // sc->raw holds {m, flags, S, Ql} of the current log-weights WITHOUT a CDF (k_sum_reduce: the ESS / log-ML getters)
    WSum sum_cache{};              // ... and the host's copy of it
    uint64_t* sum_part = nullptr;  // [6][workgroups] tagged partials of k_sum_reduce
    int64_t* h_spart = nullptr;    // pinned: [n_cu][8] tagged partials of k_sum_host, folded by the host
    bool sum_on_host = false;      // sum_cache came from k_sum_host: it holds m too, and sc->raw on the device was NOT updated
    uint64_t* gate_part = nullptr; int gate_cur = 0; int64_t* h_gate = nullptr;   // gpf_step_ess: two sets of k_sum_host<GATE>'s accumulator lines (gate_verdict), pinned {ticket << 1 | verdict} of the launch that read them
    bool raw_has_q = false;        // the raw summary's scan also accumulated sum q^2 (blockQ)
    bool raw_q_folded = false;     // sc->raw.Ql folded from blockQ
    bool pending_gather = false;   // a resample left (rows[cur], anc) un-gathered; log-weights are 0
    bool pending_fill = false;     // ... or, after gpf_resample_local, the constant sc->lw_fill
    // "lazy search" (gpf_k_fused.hpp): pf_resample!(:multinomial) enqueued only the weight scan; the ancestors (h->anc) and the log-ML
    // update are still to come -- from k_step_search when a plain pf_update! follows, else from finish_search().  Implies pending_gather.
    bool pending_search = false;
    SearchArgs pend_sa{};
    // "lazy move" (gpf_k_step.hpp k_move_step): pf_rejuvenate! enqueued nothing; the move runs inside the plain pf_update! that follows
    // (gather -> move -> propagate -> one row store), or on its own (finish_move) as soon as anything else looks at the state.  The move's
    // epoch is consumed at the call (pm_epoch = the epoch the stand-alone k_move would have used).
    bool pending_move = false; int pm_method = 0, pm_iters = 0; uint32_t pm_epoch = 0; ModelArgs pm_args{};
    bool lazy_move = true;         // GPF_LAZY_MOVE=0 in the environment: every pf_rejuvenate! launches its kernel at once
    bool lazy_search = false;      // gpf_set_lazy_search (default: GPF_LAZY_SEARCH=1 in the environment, else off)
    // the 16-bit offset levels of the weight channel (k_search_multi / k_push_multi) cost the scan ~1.4 us: only written when a
    // multinomial search will read them
    bool want_offsets = true;      // what the next scan of channel 0 writes
    bool ch0_offsets = false;      // what the last scan of channel 0 wrote
    bool offsets_hint = true;      // was the last resample multinomial?  (scans that run ahead of a resample: the ESS getter)
    bool pending_packed = false;   // sharded: the resampled population is still the received exchange buffer (gpf_shard_commit)
    const double* pend_packed = nullptr; const double* pend_mf = nullptr; const int64_t* pend_tot = nullptr; int pend_G = 0;
    bool pend_mailbox = false;     // pend_mf / pend_tot sit in the shard mailbox
    // own-direct commit (k_search_own): the packed buffer holds pend_m < n entries (the slots other shards serve), the shard's own hits
    // sit in h->anc as global ancestor ids (-1 elsewhere) and are gathered through it
    bool pend_own = false; int64_t pend_m = 0; bool pend_own_range = false;   // (own hits named by ShardPlan::own_range: stratified)
    double* fuse_mf_out = nullptr; // set by shard_summary around gpf_shard_weight_scan: the scan produces / pushes the (max, flags) summary itself
    bool own_direct = false;       // set by the library engine around its phase calls: gpf_shard_push_count resolves the own slots in place
    bool own_direct_range = false; // ... stratified: the own hits are one slot range (ShardPlan::own_range), written by k_search_strat's pack loop
    // trajectory store (gpf_history_enable): per recorded step the d latent columns in the step's final particle
    // order, and the composed ancestor map of the resamples that happened during that step (nullptr = identity)
    bool hist_on = false;
    int hist_cap = 0;
    std::vector<double*> hist_x;         // [step] n*d doubles (nullptr until snapshotted)
    std::vector<int32_t*> hist_map;      // [step] n int32 or nullptr
    int hist_step = -1;                  // index of the current step (0 = after gpf_initialize)
    const int32_t** hist_dev_maps = nullptr;
    // sub-state view (src/view.jl:16-48): this handle aliases particles [view_start, view_start + n) of `parent`
    gpf_filter* parent = nullptr;
    std::vector<gpf_filter*> views;      // the live view handles of THIS filter: gpf_destroy orphans them (a view used after its filter is gone fails loudly)
    bool orphaned = false;               // view: its filter was destroyed -- parent dangles and is never followed, the stream is gone with it
    int64_t view_start = 0;
    int64_t view_step = 1;               // > 1: strided view (state[start:step:stop]); works on the compact copies below
    double* vrows[2] = {nullptr, nullptr}; double* vlw = nullptr; int32_t* vanc = nullptr;
    int32_t* vidx = nullptr; int32_t* vgid = nullptr;   // view over an index vector (view_step == 0): the particles' indices in the parent, and idx - idx[0] (ModelArgs::gid_map)
    uint64_t generation = 0;             // bumped when the per-particle buffers are reallocated (views check it)
    uint64_t parent_generation = 0;
    uint64_t mutations = 0;              // bumped by every change of the rows / log-weights of this filter (through any handle)
    uint64_t seen_mutations = 0;         // view: the parent's counter when this view's cached summaries were valid
    // multi-GPU: the library's own RCCL communicator and the device scratch of gpf_shard_resample
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    double *sh_mf = nullptr, *sh_mf_all = nullptr; int64_t *sh_tot = nullptr, *sh_tot_all = nullptr, *sh_cr = nullptr, *sh_cr_all = nullptr;
    double *sh_send = nullptr, *sh_recv = nullptr; int64_t sh_send_cap = 0, sh_recv_cap = 0;
    // shard mailboxes (gpf_k_common.hpp): the three small summaries of a sharded resample travel as peer stores from the
    // producing kernel into every rank's mailbox instead of RCCL all-gathers
    uint64_t* mbox = nullptr;            // this rank's mailbox (device memory, exported through hipIpc)
    uint64_t** mb_peers = nullptr;       // device array [world]: every rank's mailbox as mapped in this process
    std::vector<void*> mb_opened;        // peers' mailboxes opened with hipIpcOpenMemHandle (closed by gpf_comm_destroy)
    bool mb_active = false;
    bool mb_fuse_default = false;        // every rank of the communicator has a device of its own: the (max, flags) round rides in its consumer's launch (mailbox_fuse_mf)
    // the GLOBAL weight summary of the latest one-launch reduction (k_sum_shard: the sharded ESS getter / gpf_shard_step_ess), as the host read it, and what
    // it describes: a gpf_shard_resample that finds it still valid (same weights, no newer mailbox round) does not repeat the (max, flags) round, and a
    // residual one runs no weight scan at all (shard_resample_impl)
    bool gsum_ok = false; uint64_t gsum_mut = 0, gsum_mf_seq = 0, gsum_tot_seq = 0; WSum gsum{};
    bool comm_poisoned = false;          // a sharded call failed on THIS rank after its mailbox rounds / collectives had begun: the peers are out of step with it
    bool mb_engine = false;              // set by the library engine around its phase calls: they push / wait through the mailbox
    // the slot-addressed receive window (gpf_k_common.hpp RingOut / RingIn): one entry of W + 2 words per local slot and parity, mapped by every peer
    uint64_t* ring = nullptr;            // this rank's window (device memory, exported through hipIpc)
    uint64_t** ring_peers = nullptr;     // device array [world]: every rank's window as mapped in this process
    std::vector<void*> ring_opened;
    bool ring_active = false;
    int64_t ring_parity_words = 0;       // words of one parity: (slots of the largest shard) x (W + 2)
    uint64_t ring_seq = 0;               // window exchanges so far: the same on every rank (SPMD call order)
    int exchange_mode = 0;               // gpf_comm_set_exchange: GPF_SHARD_EXCHANGE_RCCL | _P2P (the resamplers with ascending targets)
    bool ring_now = false;               // set by the library engine around its phase calls: this resample exchanges through the windows
    bool splan_ride = false;             // ... this resample is stratified: its plan rides in the weight scan's launch (k_scan MODE 3, ScanExtras::splan)
    bool splan_done = false;             // ... and did: gpf_shard_push_count launches no k_strat_plan
    bool pend_ring = false; uint64_t pend_ring_seq = 0;   // the pending commit's entries sit in the window (exchange pend_ring_seq)
    int64_t* tr_dev = nullptr;           // device {entries sent, received} of the window exchanges (the host never learns their counts: gpf_comm_traffic reads these)
    // what the shard phases summarise / pack on behalf of the engine (defaults: the raw log-weights, no extra field)
    PrioView sum_pv{nullptr, nullptr, 0.0, 0}; bool sum_pv_set = false; WSum* sum_slot = nullptr; bool sum_no_cdf = false;
    int push_extra = 0; PrioView push_pv{nullptr, nullptr, 0.0, 0};
    uint64_t sh_round = 0;               // summary rounds so far: the local / gathered arrays are rings of SH_RING rounds
    const double* cur_mf_all = nullptr; const int64_t* cur_tot_all = nullptr; const int64_t* cur_cr_all = nullptr;   // the gathered summaries of the current round
    uint64_t mb_seq[MB_KINDS] = {0, 0, 0, 0};   // rounds so far per kind: the same on every rank (SPMD call order)
    uint64_t mb_cur[MB_KINDS] = {0, 0, 0, 0};   // the round whose entries the current gathered pointers name
    int32_t* h_timeout = nullptr;        // pinned: set by a scan whose bounded inter-workgroup wait gave up (checked on the host)
    int scan_blocks_per_cu = 2;          // resident scan workgroups per CU the launch may rely on (occupancy query)
    int wscan_blocks_per_cu = 2;         // ... of the weight scans k_scan<InFixQ, *> alone (fewer registers than the residual scan)
    int64_t* shard_counts = nullptr;     // [2 * MAX_SHARDS] exchange counters of the current resample (device) + pinned mirror
    int64_t* h_shard_counts = nullptr;
    // block-wise resampling (gpf_resample_blocks): {flags, count} words, the per-block mask, per-block statistics
    int64_t* h_qpub = nullptr; int64_t q_ticket = 0;   // the ESS getter's scan publishes {flags, S, limbs of sum q^2} itself (ScanExtras::q_host)
    bool q_published = false;                          // ... and the scan of THIS call did
    int32_t* blk_words = nullptr; int32_t* blk_mask = nullptr; double* blk_stats = nullptr; int64_t blk_cap = 0, blk_last = 0;
    // blocks of more than BLK_MAX particles: gpf_resample_blocks / gpf_block_stats run the loop over sub-states themselves, through view
    // handles kept on the filter (one per block; rebuilt when the block size or the particle buffers change)
    std::vector<gpf_filter*> blk_views; int64_t blk_views_size = 0; uint64_t blk_views_gen = 0;
    double* blk_obs = nullptr; int64_t blk_obs_cap = 0;                                // per-block observations [n_blocks][MAX_OBS] on the device
    static constexpr int BLK_STAGE = 4;                                                // pinned staging buffers, used in turn (no stream sync per step)
    double* h_blk_obs[BLK_STAGE] = {nullptr, nullptr, nullptr, nullptr};
    int64_t blk_stage_next = 0;                                                        // staging copies issued so far (ticket of the next one - 1)
    int64_t* h_blk_done = nullptr; unsigned int* blk_stage_counter = nullptr;          // pinned: ticket of the last finished staging copy; device: its workgroup counter
    int64_t blk_obs_size = 0;                                                          // > 0: the latest observations are per block, blocks of this size
    // the pull plan (gpf_comm_set_plan): request lists [G][n], their counters, the dense / gathered request matrix and its pinned mirror
    int shard_plan_kind = 0;
    ulonglong2* pull_req = nullptr; int64_t pull_req_cap = 0;
    int64_t* pull_counts = nullptr; int64_t* pull_pc = nullptr; int64_t* pull_pc_all = nullptr; int64_t* h_pull_pc_all = nullptr;
    int64_t* h_flags = nullptr;          // pinned {validity flags, ticket} published by the weight scan of a checked resample
    int64_t flag_ticket = 0;
    int64_t push_ticket = 0;             // bumped by every gpf_shard_push launch; k_push publishes it with the counts
    bool counts_published = false;
    // GPF_RESAMPLE_MULTINOMIAL_SORTED: gamma totals of the tiles of SP_TILE slots (k_sorted_gammas) and, for many tiles, their starting points (k_sorted_tiles)
    uint64_t* sp_g = nullptr; uint64_t* sp_vlo = nullptr; int64_t sp_cap = 0;
    SortedGammaJob sp_job{}; bool sp_job_set = false;    // tile totals wanted: the next weight scan of this call carries them (scan_launch), else k_sorted_gammas
    ulonglong2* push_stage = nullptr;    // push exchange: staged hits, one 16-byte entry per global output slot at most
    // sharded stratified resampling with sort_particles = true (gpf_shard_resample_sorted; gpf_k_shard.hpp AncPlan): the planner filter of n_global particles
    // every rank sorts / scans / searches the gathered log-weights with, the all-gather staging of unequal shards, the pack cursors
    gpf_filter* planner = nullptr;
    double* sorted_src = nullptr; double* sorted_gath = nullptr; int64_t sorted_per = 0;
    unsigned long long* anc_cursors = nullptr;
    ShardPlan* shard_plan = nullptr;     // sharded stratified / sorted multinomial resampling: the slot range this shard serves (k_strat_plan, k_sorted_plan)
    int64_t* splan_F = nullptr; unsigned int* splan_arrive = nullptr;   // k_sorted_plan: boundary scratch [MAX_SHARDS + 1], arrival counter (zero between launches)
    int64_t push_cap = 0;
    bool push_counted = false;
    // exchange volume of the sharded resamples so far (gpf_comm_traffic): calls, entries sent to / received from OTHER ranks, bytes of one entry of the latest call
    int64_t tr_calls = 0, tr_sent = 0, tr_recv = 0, tr_entry_bytes = 0;
    int last_flags = 0;                  // safe_softmax flags (FLAG_*) of the latest resample that read them on the host (resample_impl)
    hipEvent_t chain_ev = nullptr;       // ChainGate: recorded behind this filter's chained kernels while other filters live on the device
    bool chain_counted = false;
    gpfh::Timer timers[GPF_K_COUNT];
    gpfh::PhaseTimer phases;
    std::string err;
};

namespace gpfh {

#define HIP_TRY(h, expr)                                                                        \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                       \
            return GPF_ERR_HIP;                                                                 \
        }                                                                                       \
    } while (0)

inline gpf_status fail(gpf_handle h, gpf_status s, const std::string& msg)
{
    if (h) h->err = msg; else g_err = msg;
    return s;
}

constexpr int row_width(int D, bool keep) { return ((keep ? 2 * D : D) + 1) & ~1; }
// one buffer per scan channel holds the per-256 level (8 u64 per tile) followed by the 4-byte key level (64 u32 per tile)
// one buffer per scan channel: the per-256 level (8 u64 per tile), the 4-byte key level (64 u32 per tile), the 16-bit
// in-group offsets (2048 u16 per tile) and their coarse rows (<= 512 u16 per tile) -- gpf_kernels.hpp ScanOut
// ... and, beyond 2.5 M particles, the compact copy of every 4th / 8th / 16th key (<= 16 u32 per tile)
inline size_t t256_bytes(int64_t ntiles) { return (size_t)ntiles * ((TILE / 256) * sizeof(uint64_t) + (TILE / 32) * sizeof(uint32_t) + TILE * sizeof(uint16_t) + (TILE / 4) * sizeof(uint16_t) + 16 * sizeof(uint32_t)); }
inline uint32_t* k32_of(uint64_t* t256, int64_t ntiles) { return reinterpret_cast<uint32_t*>(t256 + ntiles * (TILE / 256)); }
inline uint16_t* off16_of(uint64_t* t256, int64_t ntiles) { return reinterpret_cast<uint16_t*>(k32_of(t256, ntiles) + ntiles * (TILE / 32)); }
inline uint16_t* coarse_of(uint64_t* t256, int64_t ntiles) { return off16_of(t256, ntiles) + ntiles * TILE; }
inline uint32_t* k32s_of(uint64_t* t256, int64_t ntiles) { return reinterpret_cast<uint32_t*>(coarse_of(t256, ntiles) + ntiles * (TILE / 4)); }
// channel 0 (the weights) carries the offset levels when k_search_multi can take the filter (multi_logg >= 0)
inline ScanOut scan_out(uint64_t* cdf, uint64_t* t16, uint64_t* t256, int64_t ntiles, bool with_offsets)
{
    int logg = with_offsets ? multi_logg(ntiles) : -1;
    const int sample = with_offsets && logg < 0 ? multi_sample(ntiles) : 0;       // beyond 2.5 M particles: 32-cell levels + sampled keys
    if (sample > 0) logg = 0;
    return ScanOut{cdf, t16, t256, k32_of(t256, ntiles), logg >= 0 ? off16_of(t256, ntiles) : nullptr, logg >= 0 ? coarse_of(t256, ntiles) : nullptr, logg,
                   sample > 0 ? k32s_of(t256, ntiles) : nullptr, sample};
}

inline int grid_for(const gpf_filter* h, int64_t work_items, int blocks_per_cu)
{
    const int64_t need = (work_items + BLOCK - 1) / BLOCK;
    const int64_t cap = (int64_t)h->n_cu * blocks_per_cu;
    return (int)std::max<int64_t>(1, std::min(need, cap));
}

// the slots the next producer of log-weights folds its maximum into (and the array it clears for the producer after it)
inline MaxSlots next_slots(gpf_filter* h) { h->mcur ^= 1; return MaxSlots{h->mslots[h->mcur], h->mslots[1 - h->mcur]}; }

template <class F>
gpf_status timed(gpf_filter* h, int id, F&& launch)
{
    Timer& t = h->timers[id];
    if (!t.on) { launch(); return GPF_OK; }
    hipEvent_t a, b;
    HIP_TRY(h, hipEventCreate(&a));
    HIP_TRY(h, hipEventCreate(&b));
    g_ev_start = a; g_ev_stop = b;
    launch();                       // exactly one GPF_LAUNCH
    g_ev_start = g_ev_stop = nullptr;
    t.ev.emplace_back(a, b);
    return GPF_OK;
}

// ------------------------------------------------------------------ chained kernels of SEVERAL filters on one device
// The scans (k_scan, k_scan_residual2) and the sort's partition passes (k_sort_pass) chain their workgroups: a workgroup waits for the prefix of the tiles
// below its own.  With ONE such kernel on the device that is safe -- its grid is sized to be resident at once (resample_device_setup), and a workgroup
// only ever waits for workgroups dispatched before it.  TWO of them in flight on different streams (independent filters stepped from one process:
// tools/replicas.py, R = 4 filters of 2 x 10^6 particles) can each hold the slots the other's not-yet-dispatched workgroups need: every resident
// workgroup waits, nothing retires -- the bounded spins give up after seconds and the run fails loudly ("bounded inter-workgroup wait timed out").
// So: while a process holds MORE THAN ONE filter on a device, its chained kernels run one after the other -- each such launch waits for the event
// behind the previous one (of whichever stream) and records its own.  Everything else (propagate, search, gather, reductions) still overlaps across
// the streams, and a process with one filter per device -- the headline, every sharded rank -- never touches the gate.
struct ChainGate {
    std::mutex mu;
    int live = 0;                        // filters (not views) alive on the device
    hipEvent_t last = nullptr;           // behind the latest chained launch ...
    hipStream_t last_stream = nullptr;   // ... which went to this stream
};
inline ChainGate g_chain[16];
struct ChainScope {                      // around the launch (or the launches, same stream) of chained kernels
    gpf_filter* h; ChainGate* g; bool on;
    explicit ChainScope(gpf_filter* f);
    ~ChainScope();
};

inline void phase_mark(gpf_filter* h, int id)
{
    if (!h->phases.on) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, h->stream);
    h->phases.marks.emplace_back(id, e);
}

inline ChainScope::ChainScope(gpf_filter* f) : h(f), g(nullptr), on(false)
{
    const int dev = h->cfg.device;
    if (dev < 0 || dev >= 16) return;
    g = &g_chain[dev];
    g->mu.lock();
    on = g->live > 1;
    if (on && g->last && g->last_stream != h->stream) (void)hipStreamWaitEvent(h->stream, g->last, 0);
}
inline ChainScope::~ChainScope()
{
    if (!g) return;
    if (on) {
        gpf_filter* owner = h->parent ? h->parent : h;            // (a view launches on its filter's stream; the event lives with the filter)
        if (!owner->chain_ev && hipEventCreateWithFlags(&owner->chain_ev, hipEventDisableTiming) != hipSuccess) owner->chain_ev = nullptr;
        if (owner->chain_ev && hipEventRecord(owner->chain_ev, h->stream) == hipSuccess) { g->last = owner->chain_ev; g->last_stream = h->stream; }
    }
    g->mu.unlock();
}

// ------------------------------------------------------------------ shard mailboxes (host side)
// begin a new round of `kind` (the producing kernel of this call pushes it); no mailbox / not the library engine: push nowhere
inline MboxPush mb_begin(gpf_filter* h, int kind)
{
    MboxPush p{};
    if (!(h->mb_active && h->mb_engine)) return p;
    const uint64_t seq = ++h->mb_seq[kind];
    h->mb_cur[kind] = seq;
    const int slot = (int)(seq & (MB_SLOTS - 1));
    p.peers = h->mb_peers; p.payload_off = mb_payload_off(kind, slot); p.tag_off = mb_tag_off(kind, slot);
    p.tag = seq; p.G = h->comm_world; p.me = h->comm_rank; p.nwords = mb_words(kind);
    return p;
}
// what a consumer of the current round of `kind` waits for
inline MboxWait mb_wait(const gpf_filter* h, int kind)
{
    MboxWait w{};
    if (!(h->mb_active && h->mb_engine)) return w;
    const uint64_t seq = h->mb_cur[kind];
    w.tags = h->mbox + mb_tag_off(kind, (int)(seq & (MB_SLOTS - 1))); w.want = seq; w.n = h->comm_world; w.nwords = mb_words(kind); w.timeout = h->h_timeout;
    return w;
}
// the gathered array of the current round of `kind` inside the own mailbox ([G][words], dense like the all-gather's output)
inline const void* mb_gathered(const gpf_filter* h, int kind)
{
    return h->mbox + mb_payload_off(kind, (int)(h->mb_cur[kind] & (MB_SLOTS - 1)));
}

// ------------------------------------------------------------------ per-N device buffers
struct Bufs {                        // everything whose size depends on the particle count (detached copy, for resizing)
    int64_t n = 0, ntiles = 0;
    double* rows[2] = {nullptr, nullptr};
    int cur = 0;
    double *lw = nullptr, *lws = nullptr, *lp = nullptr, *dtmp = nullptr;
    uint64_t *cdf[3] = {}, *t16[3] = {}, *t256[3] = {}, *desc[3][2] = {};
    int32_t *anc = nullptr, *order = nullptr, *idx_in = nullptr;
    uint64_t *keys = nullptr, *keys_out = nullptr;
    void* sort_tmp = nullptr;
};

#ifndef STEP_BLOCKS_PER_CU
#define STEP_BLOCKS_PER_CU 4
#endif
inline int step_grid(const gpf_filter* h) { return std::min(grid_for(h, h->n, STEP_BLOCKS_PER_CU), MAX_PARTIALS); }
// the rejuvenation kernels: rounds 1-2 found them FASTER with fewer workgroups per CU (8 per CU 105 / 123 us, 2: 52 / 61, bearings MH / SV
// move-reweight) -- the reason was one same-address atomic per WAVE for the accept count (~10 ns each, serialised: 20-40 us).  With
// per-workgroup counts (round 3): 2 per CU 36.7 / 44.2 us, 3: 32.1 / 44.6, 4: 31.3 / 44.6, 6: 31.2 / 47.7, 8: 31.5 / 47.0.
#ifndef MOVE_BLOCKS_PER_CU
#define MOVE_BLOCKS_PER_CU 4
#endif
inline int move_grid(const gpf_filter* h) { return std::min(grid_for(h, h->n, MOVE_BLOCKS_PER_CU), MAX_PARTIALS); }

#define DISPATCH_MODEL(h, CALL)                                                                  \
    switch ((h)->cfg.model) {                                                                    \
        case MODEL_LGSSM2: { constexpr int MM = MODEL_LGSSM2; CALL; } break;                     \
        case MODEL_BEARINGS4: { constexpr int MM = MODEL_BEARINGS4; CALL; } break;               \
        case MODEL_SV1: { constexpr int MM = MODEL_SV1; CALL; } break;                           \
        case MODEL_OBJECT_MOTION: { constexpr int MM = MODEL_OBJECT_MOTION; CALL; } break;       \
        case MODEL_LINE: { constexpr int MM = MODEL_LINE; CALL; } break;                         \
    }

inline PrioView raw_view(const gpf_filter* h) { return PrioView{h->lw, nullptr, 0.0, 0}; }

// ------------------------------------------------------------------ weight summary = (max) + scan
// The scan's inter-workgroup protocol needs every workgroup of the launch resident at once (block b owns tiles b, b + G, ...
// and waits for lower tiles of its round): at most scan_blocks_per_cu per CU, from the occupancy query at gpf_create.
inline int scan_grid(const gpf_filter* h) { return (int)std::max<int64_t>(1, std::min<int64_t>(h->ntiles, (int64_t)h->scan_blocks_per_cu * h->n_cu)); }
// the weight scans (k_scan<InFixQ, *>): up to WSCAN_MAX workgroups per CU, so that filters of up to WSCAN_MAX x 0.52 M particles scan in ONE round
inline int wscan_grid(const gpf_filter* h) { return (int)std::max<int64_t>(1, std::min<int64_t>(h->ntiles, (int64_t)h->wscan_blocks_per_cu * h->n_cu)); }

inline void normalise_Q(const WSum& w, uint64_t& hi, uint64_t& lo)
{
    unsigned __int128 Q = (unsigned __int128)w.Ql[0] + ((unsigned __int128)w.Ql[1] << 32) +
                          ((unsigned __int128)w.Ql[2] << 64) + ((unsigned __int128)w.Ql[3] << 96);
    hi = (uint64_t)(Q >> 64);
    lo = (uint64_t)Q;
}


// ---- defined in libgpf_core.hip
Bufs take_particle_buffers(gpf_filter* h);
void free_bufs(Bufs& b);
gpf_status alloc_particle_buffers(gpf_filter* h);
void launch_gather_ex(gpf_filter* h, const int32_t* anc, const double* in, double* out, const PrioView& pv, double* lw_out, int64_t n);
void launch_gather(gpf_filter* h, const PrioView& pv, double* lw_out);
void launch_gather_rows_lw(gpf_filter* h, const int32_t* anc, const double* rows_in, const double* lw_in, double* rows_out, double* lw_out, int64_t n);
gpf_status materialize(gpf_filter* h);
gpf_status finish_move(gpf_filter* h);
void hist_clear(gpf_filter* h);
gpf_status hist_snapshot(gpf_filter* h);
gpf_status hist_on_resample(gpf_filter* h);
gpf_status hist_begin_step(gpf_filter* h, bool first);
void mutated(gpf_filter* h);
gpf_status view_enter(gpf_filter* v);
gpf_status view_exit(gpf_filter* v);
gpf_status check_ready(gpf_handle h, bool keep_pending_move = false);
gpf_status set_obs(gpf_filter* h, const double* obs, int n_obs);
gpf_status copy_out(gpf_handle h, const void* dsrc, void* out, size_t bytes);
gpf_status set_strata(gpf_handle h, const double* values, int32_t n_strata, int32_t interleaved);
gpf_status speculative_step(gpf_filter* h, const GateIn& gate);
void speculative_step_done(gpf_filter* h, bool ran);
// ---- defined in libgpf_resample.hip
// one scan launch on descriptor channel `ch` (0 weights, 1 residual counts, 2 residual weights): the sharded weight scans (MODE 3: pushes
// the shard's total itself; 4: also the limbs of sum q^2) and the scans of pf_optimal_resize!
gpf_status scan_launch_shard(gpf_filter* h, int mode, const InFixQ& in, int np, WSum* slot, bool want_cdf, uint64_t* total_out, const double* mf_all, const ScanExtras& ex);
gpf_status scan_launch_optimal(gpf_filter* h, int ch, const InOptimal& in, uint64_t* total_out);
gpf_status resample_device_setup(gpf_filter* h);      // occupancy of the scan kernels, LDS attributes of the searches (gpf_create)
gpf_status ensure_max(gpf_filter* h, const PrioView& pv, bool use_producer_max);
gpf_status summarize(gpf_filter* h, const PrioView& pv, WSum* slot, bool want_cdf, const int32_t* order, bool use_producer_max,
                     bool want_q = false, bool publish_flags = false, bool max_ready = false);
gpf_status ensure_raw(gpf_filter* h, bool want_q = false);
gpf_status ensure_raw_summary(gpf_filter* h, bool want_q, bool* done);
gpf_status read_published_summary(gpf_filter* h, WSum& w);
bool sum_host_ok(const gpf_filter* h);
gpf_status sum_host_launch(gpf_filter* h, const double* thr);
gpf_status sum_host_fold(gpf_filter* h, const double* thr, int* go_out = nullptr);
gpf_status sum_gate_check(gpf_filter* h, int host_go, int64_t ticket);
gpf_status shard_sum_launch(gpf_filter* h, const ShardSum& ss, bool* ok);
bool shard_sum_collect();                                        // GPF_SHARD_SUM=collect: k_sum_reduce<SHARD> instead of k_sum_shard
gpf_status wait_ticket(gpf_filter* h, volatile int64_t* tk, int64_t want, const char* what);
gpf_status check_scan_timeout(gpf_filter* h);
gpf_status fetch_scalars(gpf_filter* h, bool fold_raw_q = false);
gpf_status sort_desc(gpf_filter* h, const PrioView& pv, int64_t n);
gpf_status ensure_residual_buffers(gpf_filter* h);
CdfLevels levels(const gpf_filter* h, int ch);
gpf_status residual_scans(gpf_filter* h, const WSum* ws, int64_t n_slots_global, int32_t* head_anc = nullptr, const ResidDirect* direct = nullptr);
void launch_multinomial_search(gpf_filter* h, const SearchArgs& sa);
void launch_search_plain(gpf_filter* h, int which, int grid, size_t lds, const SearchArgs& sa);   // k_search<which>, which = 1 (residual) | 3 (systematic)
void launch_search_strat(gpf_filter* h, const SearchArgs& sa, int64_t n_slots, bool sorted_uniforms);   // k_search_strat<sorted_uniforms>
gpf_status sorted_job_prepare(gpf_filter* h, int64_t gid0, int64_t n_slots);
gpf_status sorted_gammas_finish(gpf_filter* h, bool with_tiles);
gpf_status finish_search(gpf_filter* h);
gpf_status resample_impl(gpf_filter* h, int method, PrioView pv, int sort_particles, int check, int32_t* invalid, bool local = false);
// ---- defined in libgpf_aux.hip
gpf_status weighted_tree_sum(gpf_filter* h, const double* values, int stride, int col, int pw, const double* center, double match, double* out_dev);
// ---- defined in libgpf_shard.hip
gpf_status shard_device_setup(gpf_filter* h);         // LDS attributes of the push kernels (gpf_create)

} // namespace gpfh
