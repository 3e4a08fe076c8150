// libgpf.hip -- C ABI (include/gpf.h) over the gfx950 kernels of gpf_kernels.hpp.
//
// Host orchestration only: which kernels run for each pf_* operation, on one HIP stream, with all
// scalars (max, sums, log-ML estimate, residual counts) kept in device memory so that the common
// path (check = false / :warn without reading the flag) never waits for the GPU.
//
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared libgpf.hip -o libgpf_hip.so
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
#include <string>
#include <vector>

using namespace gpf;

namespace {

// polite busy-wait on a pinned-memory ticket (x86 PAUSE; a plain compiler barrier elsewhere)
static inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    __asm__ __volatile__("" ::: "memory");
#endif
}
thread_local std::string g_err;   // errors before a handle exists

// Kernel timing (gpf_kernel_timing): inside timed() the launch carries a start/stop event pair that the runtime
// stamps at the kernel's own begin and end (hipExtLaunchKernel), so the elapsed time is the dispatch's duration, the
// same quantity rocprofv3 --kernel-trace reports -- not launch gap + kernel as with events recorded around the launch.
thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;
#define GPF_LAUNCH(kernel, grid, block, lds, stream, ...) \
    hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, g_ev_start, g_ev_stop, 0, __VA_ARGS__)

struct Timer {
    bool on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
};

} // namespace

struct gpf_filter {
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
    bool mb_engine = false;              // set by the library engine around its phase calls: they push / wait through the mailbox
    // what the shard phases summarise / pack on behalf of the engine (defaults: the raw log-weights, no extra field)
    PrioView sum_pv{nullptr, nullptr, 0.0, 0}; bool sum_pv_set = false; WSum* sum_slot = nullptr; bool sum_no_cdf = false;
    int push_extra = 0; PrioView push_pv{nullptr, nullptr, 0.0, 0};
    uint64_t sh_round = 0;               // summary rounds so far: the local / gathered arrays are rings of SH_RING rounds
    const double* cur_mf_all = nullptr; const int64_t* cur_tot_all = nullptr; const int64_t* cur_cr_all = nullptr;   // the gathered summaries of the current round
    uint64_t mb_seq[MB_KINDS] = {0, 0, 0};   // rounds so far per kind: the same on every rank (SPMD call order)
    uint64_t mb_cur[MB_KINDS] = {0, 0, 0};   // the round whose entries the current gathered pointers name
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
    ShardPlan* shard_plan = nullptr;     // sharded stratified resampling: the slot range this shard serves (k_strat_plan)
    int64_t push_cap = 0;
    bool push_counted = false;
    Timer timers[GPF_K_COUNT];
    std::string err;
};

namespace {

#define HIP_TRY(h, expr)                                                                        \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                       \
            return GPF_ERR_HIP;                                                                 \
        }                                                                                       \
    } while (0)

gpf_status materialize(gpf_filter* h);
gpf_status finish_search(gpf_filter* h);
gpf_status finish_move(gpf_filter* h);

gpf_status fail(gpf_handle h, gpf_status s, const std::string& msg)
{
    if (h) h->err = msg; else g_err = msg;
    return s;
}

constexpr int row_width(int D, bool keep) { return ((keep ? 2 * D : D) + 1) & ~1; }
// one buffer per scan channel holds the per-256 level (8 u64 per tile) followed by the 4-byte key level (64 u32 per tile)
// one buffer per scan channel: the per-256 level (8 u64 per tile), the 4-byte key level (64 u32 per tile), the 16-bit
// in-group offsets (2048 u16 per tile) and their coarse rows (<= 512 u16 per tile) -- gpf_kernels.hpp ScanOut
// ... and, beyond 2.5 M particles, the compact copy of every 4th / 8th / 16th key (<= 16 u32 per tile)
size_t t256_bytes(int64_t ntiles) { return (size_t)ntiles * ((TILE / 256) * sizeof(uint64_t) + (TILE / 32) * sizeof(uint32_t) + TILE * sizeof(uint16_t) + (TILE / 4) * sizeof(uint16_t) + 16 * sizeof(uint32_t)); }
uint32_t* k32_of(uint64_t* t256, int64_t ntiles) { return reinterpret_cast<uint32_t*>(t256 + ntiles * (TILE / 256)); }
uint16_t* off16_of(uint64_t* t256, int64_t ntiles) { return reinterpret_cast<uint16_t*>(k32_of(t256, ntiles) + ntiles * (TILE / 32)); }
uint16_t* coarse_of(uint64_t* t256, int64_t ntiles) { return off16_of(t256, ntiles) + ntiles * TILE; }
uint32_t* k32s_of(uint64_t* t256, int64_t ntiles) { return reinterpret_cast<uint32_t*>(coarse_of(t256, ntiles) + ntiles * (TILE / 4)); }
// channel 0 (the weights) carries the offset levels when k_search_multi can take the filter (multi_logg >= 0)
ScanOut scan_out(uint64_t* cdf, uint64_t* t16, uint64_t* t256, int64_t ntiles, bool with_offsets)
{
    int logg = with_offsets ? multi_logg(ntiles) : -1;
    const int sample = with_offsets && logg < 0 ? multi_sample(ntiles) : 0;       // beyond 2.5 M particles: 32-cell levels + sampled keys
    if (sample > 0) logg = 0;
    return ScanOut{cdf, t16, t256, k32_of(t256, ntiles), logg >= 0 ? off16_of(t256, ntiles) : nullptr, logg >= 0 ? coarse_of(t256, ntiles) : nullptr, logg,
                   sample > 0 ? k32s_of(t256, ntiles) : nullptr, sample};
}

int grid_for(const gpf_filter* h, int64_t work_items, int blocks_per_cu)
{
    const int64_t need = (work_items + BLOCK - 1) / BLOCK;
    const int64_t cap = (int64_t)h->n_cu * blocks_per_cu;
    return (int)std::max<int64_t>(1, std::min(need, cap));
}

// the slots the next producer of log-weights folds its maximum into (and the array it clears for the producer after it)
MaxSlots next_slots(gpf_filter* h) { h->mcur ^= 1; return MaxSlots{h->mslots[h->mcur], h->mslots[1 - h->mcur]}; }

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

// ------------------------------------------------------------------ shard mailboxes (host side)
// begin a new round of `kind` (the producing kernel of this call pushes it); no mailbox / not the library engine: push nowhere
MboxPush mb_begin(gpf_filter* h, int kind)
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
MboxWait mb_wait(const gpf_filter* h, int kind)
{
    MboxWait w{};
    if (!(h->mb_active && h->mb_engine)) return w;
    const uint64_t seq = h->mb_cur[kind];
    w.tags = h->mbox + mb_tag_off(kind, (int)(seq & (MB_SLOTS - 1))); w.want = seq; w.n = h->comm_world; w.timeout = h->h_timeout;
    return w;
}
// the gathered array of the current round of `kind` inside the own mailbox ([G][words], dense like the all-gather's output)
const void* mb_gathered(const gpf_filter* h, int kind)
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

Bufs take_particle_buffers(gpf_filter* h)
{
    Bufs b;
    b.n = h->n; b.ntiles = h->ntiles; b.cur = h->cur;
    b.rows[0] = h->rows[0]; b.rows[1] = h->rows[1]; h->rows[0] = h->rows[1] = nullptr;
    b.lw = h->lw; b.lws = h->lws; b.lp = h->lp; b.dtmp = h->dtmp; h->lw = h->lws = h->lp = h->dtmp = nullptr;
    for (int i = 0; i < 3; ++i) {
        b.cdf[i] = h->cdf[i]; b.t16[i] = h->t16[i]; b.t256[i] = h->t256[i]; h->cdf[i] = h->t16[i] = h->t256[i] = nullptr;
        for (int j = 0; j < 2; ++j) { b.desc[i][j] = h->desc[i][j]; h->desc[i][j] = nullptr; }
        h->table[i] = nullptr; h->dcur[i] = 0;
    }
    b.anc = h->anc; b.order = h->order; b.idx_in = h->idx_in; h->anc = h->order = h->idx_in = nullptr;
    b.keys = h->keys; b.keys_out = h->keys_out; b.sort_tmp = h->sort_tmp;
    h->keys = h->keys_out = nullptr; h->sort_tmp = nullptr; h->sort_tmp_bytes = 0;
    return b;
}

void free_bufs(Bufs& b)
{
    void* p[] = {b.rows[0], b.rows[1], b.lw, b.lws, b.lp, b.dtmp, b.cdf[0], b.cdf[1], b.cdf[2], b.t16[0], b.t16[1], b.t16[2],
                 b.t256[0], b.t256[1], b.t256[2], b.desc[0][0], b.desc[0][1], b.desc[1][0], b.desc[1][1], b.desc[2][0], b.desc[2][1],
                 b.anc, b.order, b.idx_in, b.keys, b.keys_out, b.sort_tmp};
    for (void* q : p) if (q) (void)hipFree(q);
    b = Bufs();
}

// allocate the per-N buffers for h->n particles (fields must be null); sets ntiles, K, logN
gpf_status alloc_particle_buffers(gpf_filter* h)
{
    h->ntiles = (h->n + TILE - 1) / TILE;
    h->K = fix_K(h->cfg.n_global);
    h->logN = log_((double)h->cfg.n_global);
    h->cur = 0;
    const size_t n = (size_t)h->n, rb = n * (size_t)h->W * sizeof(double);
    HIP_TRY(h, hipMalloc(&h->rows[0], rb));
    HIP_TRY(h, hipMalloc(&h->rows[1], rb));
    HIP_TRY(h, hipMalloc(&h->lw, n * sizeof(double)));
    HIP_TRY(h, hipMalloc(&h->lws, n * sizeof(double)));
    HIP_TRY(h, hipMalloc(&h->lp, n * sizeof(double)));
    HIP_TRY(h, hipMalloc(&h->dtmp, n * sizeof(double)));
    HIP_TRY(h, hipMalloc(&h->cdf[0], (size_t)h->ntiles * TILE * sizeof(uint64_t)));
    HIP_TRY(h, hipMalloc(&h->t16[0], (size_t)h->ntiles * (TILE / 16) * sizeof(uint64_t)));
    HIP_TRY(h, hipMalloc(&h->t256[0], t256_bytes(h->ntiles)));
    const size_t db = (((size_t)2 * h->ntiles * sizeof(uint64_t)) + 15) & ~(size_t)15;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 2; ++j) {
            HIP_TRY(h, hipMalloc(&h->desc[i][j], db));
            HIP_TRY(h, hipMemsetAsync(h->desc[i][j], 0, db, h->stream));     // descriptors start invalid; kernels keep them so
        }
    HIP_TRY(h, hipMalloc(&h->anc, n * sizeof(int32_t)));
    return GPF_OK;
}

// ------------------------------------------------------------------ model dispatch
#ifndef STEP_BLOCKS_PER_CU
#define STEP_BLOCKS_PER_CU 4
#endif
int step_grid(const gpf_filter* h) { return std::min(grid_for(h, h->n, STEP_BLOCKS_PER_CU), MAX_PARTIALS); }
// the rejuvenation kernels: rounds 1-2 found them FASTER with fewer workgroups per CU (8 per CU 105 / 123 us, 2: 52 / 61, bearings MH / SV
// move-reweight) -- the reason was one same-address atomic per WAVE for the accept count (~10 ns each, serialised: 20-40 us).  With
// per-workgroup counts (round 3): 2 per CU 36.7 / 44.2 us, 3: 32.1 / 44.6, 4: 31.3 / 44.6, 6: 31.2 / 47.7, 8: 31.5 / 47.0.
#ifndef MOVE_BLOCKS_PER_CU
#define MOVE_BLOCKS_PER_CU 4
#endif
int move_grid(const gpf_filter* h) { return std::min(grid_for(h, h->n, MOVE_BLOCKS_PER_CU), MAX_PARTIALS); }

// PROP 0: the model's own sampler; 1: native custom proposal; 2: stratified
template <int M, bool KEEP, int PROP = 0>
void launch_step_t(gpf_filter* h, int grid)
{
    constexpr int Wc = row_width(Model<M>::D, KEEP);
    if constexpr ((PROP == 1 && !Model<M>::HAS_PROPOSAL) || (PROP == 2 && !Model<M>::HAS_STRATA)) { (void)h; (void)grid; return; }
    else if (h->pending_packed && h->pend_own) {
        // own-direct commit: the shard's own hits through the ancestor array FIRST (it reads anc[j] >= 0 / -1; the packed entries'
        // launch behind it overwrites the -1 with the received ancestors), then the received entries; one weight vector, one slot array
        const MaxSlots ms = next_slots(h);
        PackedCommit pg{nullptr, h->anc, h->pend_mf, h->pend_tot, h->pend_G, h->K, h->logN, h->sc, nullptr, (int)h->pend_mailbox, h->pend_own_range ? 2 : 1, h->cfg.gid0,
                        h->pend_own_range ? h->shard_plan->own_range : nullptr};
        GPF_LAUNCH((k_step<M, Wc, KEEP, true, PROP>), dim3(grid), dim3(BLOCK), 0, h->stream, h->args, h->cfg.seed, h->epoch,
                           h->cfg.gid0, h->n, h->anc, h->rows[h->cur], h->rows[1 - h->cur], h->lw, ms, pg);
        if (h->pend_m > 0) {
            const PackedCommit pc{h->pend_packed, h->anc, h->pend_mf, h->pend_tot, h->pend_G, h->K, h->logN, nullptr, nullptr, (int)h->pend_mailbox, 0, 0, nullptr};
            g_ev_start = g_ev_stop = nullptr;                    // (timed(): the event pair belongs to the first launch)
            GPF_LAUNCH((k_step<M, Wc, KEEP, false, PROP, true>), dim3(grid_for(h, h->pend_m, STEP_BLOCKS_PER_CU)), dim3(BLOCK), 0, h->stream, h->args, h->cfg.seed, h->epoch,
                               h->cfg.gid0, h->pend_m, h->anc, h->rows[h->cur], h->rows[1 - h->cur], h->lw, ms, pc);
        }
    }
    else if (h->pending_packed) {
        const MaxSlots ms = next_slots(h);
        const PackedCommit pc{h->pend_packed, h->anc, h->pend_mf, h->pend_tot, h->pend_G, h->K, h->logN, h->sc, nullptr, (int)h->pend_mailbox, 0, 0, nullptr};
        GPF_LAUNCH((k_step<M, Wc, KEEP, false, PROP, true>), dim3(grid), dim3(BLOCK), 0, h->stream, h->args, h->cfg.seed, h->epoch,
                           h->cfg.gid0, h->n, h->anc, h->rows[h->cur], h->rows[1 - h->cur], h->lw, ms, pc);
    } else if (h->pending_gather && h->pending_search && PROP == 0) {
        // the pending multinomial search rides in the propagate (gpf_k_fused.hpp): one 1024-thread workgroup per CU like k_search_multi
        const MaxSlots ms = next_slots(h);
        const SearchArgs& sa = h->pend_sa;
        const int gsr = (int)std::max<int64_t>(1, std::min<int64_t>((sa.n + FCH - 1) / FCH, (int64_t)h->n_cu));
        const size_t tbytes = (multi_lds_bytes(sa.ntiles, sa.w.logg) + 15) & ~(size_t)15;
        if (sa.w.logg == 0)
            GPF_LAUNCH((k_step_search<M, Wc, KEEP, 0>), dim3(gsr), dim3(SBLOCK), tbytes + FUSED_LDS_EXTRA, h->stream, h->args, h->cfg.seed, h->epoch, sa,
                       h->rows[h->cur], h->rows[1 - h->cur], h->lw, ms, (uint32_t)(tbytes / 4));
        else
            GPF_LAUNCH((k_step_search<M, Wc, KEEP, 1>), dim3(gsr), dim3(SBLOCK), tbytes + FUSED_LDS_EXTRA, h->stream, h->args, h->cfg.seed, h->epoch, sa,
                       h->rows[h->cur], h->rows[1 - h->cur], h->lw, ms, (uint32_t)(tbytes / 4));
        h->pending_search = false;
    } else if (h->pending_gather) {
        const MaxSlots ms = next_slots(h);
        PackedCommit pc{};
        pc.lw_fill = h->pending_fill ? &h->sc->lw_fill : nullptr;
        GPF_LAUNCH((k_step<M, Wc, KEEP, true, PROP>), dim3(grid), dim3(BLOCK), 0, h->stream, h->args, h->cfg.seed, h->epoch,
                           h->cfg.gid0, h->n, h->anc, h->rows[h->cur], h->rows[1 - h->cur], h->lw, ms, pc);
    }
    else
        GPF_LAUNCH((k_step<M, Wc, KEEP, false, PROP>), dim3(grid), dim3(BLOCK), 0, h->stream, h->args, h->cfg.seed, h->epoch,
                           h->cfg.gid0, h->n, h->anc, h->rows[h->cur], h->rows[1 - h->cur], h->lw, next_slots(h), PackedCommit{});
}
template <int M, int PROP = 0>
void launch_init_t(gpf_filter* h, int grid)
{
    if constexpr ((PROP == 1 && !Model<M>::HAS_PROPOSAL) || (PROP == 2 && !Model<M>::HAS_STRATA) || (PROP == 3 && !Model<M>::HAS_STRATA_PROPOSAL)) { (void)h; (void)grid; return; }
    else
        GPF_LAUNCH((k_init<M, PROP>), dim3(grid), dim3(BLOCK), 0, h->stream, h->args, h->cfg.seed, h->epoch,
                           h->cfg.gid0, h->n, h->W, h->rows[h->cur], h->lw, next_slots(h));
}
bool model_has_proposal(int model)
{
    switch (model) {
        case MODEL_LGSSM2: return Model<MODEL_LGSSM2>::HAS_PROPOSAL;
        case MODEL_BEARINGS4: return Model<MODEL_BEARINGS4>::HAS_PROPOSAL;
        case MODEL_SV1: return Model<MODEL_SV1>::HAS_PROPOSAL;
        case MODEL_OBJECT_MOTION: return Model<MODEL_OBJECT_MOTION>::HAS_PROPOSAL;
        case MODEL_LINE: return Model<MODEL_LINE>::HAS_PROPOSAL;
    }
    return false;
}
bool model_has_strata(int model)
{
    switch (model) {
        case MODEL_LGSSM2: return Model<MODEL_LGSSM2>::HAS_STRATA;
        case MODEL_BEARINGS4: return Model<MODEL_BEARINGS4>::HAS_STRATA;
        case MODEL_SV1: return Model<MODEL_SV1>::HAS_STRATA;
        case MODEL_OBJECT_MOTION: return Model<MODEL_OBJECT_MOTION>::HAS_STRATA;
        case MODEL_LINE: return Model<MODEL_LINE>::HAS_STRATA;
    }
    return false;
}
template <int M>
void launch_move_prop_t(gpf_filter* h, int grid, int n_iters)
{
    constexpr int Wc = row_width(Model<M>::D, true);
    if constexpr (!Model<M>::HAS_MOVE_PROPOSAL) { (void)h; (void)grid; (void)n_iters; return; }
    else if (h->pending_gather)
        GPF_LAUNCH((k_move<M, Wc, true, true, true>), dim3(grid), dim3(BLOCK), 0, h->stream, h->args, h->cfg.seed, h->epoch,
                           h->cfg.gid0, h->n, (int)h->has_prev, n_iters, h->anc, h->rows[h->cur], h->rows[1 - h->cur], h->lw,
                           h->acc_part, next_slots(h));
    else
        GPF_LAUNCH((k_move<M, Wc, true, false, true>), dim3(grid), dim3(BLOCK), 0, h->stream, h->args, h->cfg.seed, h->epoch,
                           h->cfg.gid0, h->n, (int)h->has_prev, n_iters, h->anc, h->rows[h->cur], h->rows[1 - h->cur], h->lw,
                           h->acc_part, next_slots(h));
}
bool model_has_move_proposal(int model)
{
    switch (model) {
        case MODEL_LGSSM2: return Model<MODEL_LGSSM2>::HAS_MOVE_PROPOSAL;
        case MODEL_BEARINGS4: return Model<MODEL_BEARINGS4>::HAS_MOVE_PROPOSAL;
        case MODEL_SV1: return Model<MODEL_SV1>::HAS_MOVE_PROPOSAL;
        case MODEL_OBJECT_MOTION: return Model<MODEL_OBJECT_MOTION>::HAS_MOVE_PROPOSAL;
        case MODEL_LINE: return Model<MODEL_LINE>::HAS_MOVE_PROPOSAL;
    }
    return false;
}
template <int M, bool RW>
void launch_move_t(gpf_filter* h, int grid, int n_iters, const ModelArgs& args, uint32_t epoch)
{
    constexpr int Wc = row_width(Model<M>::D, true);
    if (h->pending_gather)           // the resample gather rides on the move (rows read through anc, incoming weights 0)
        GPF_LAUNCH((k_move<M, Wc, RW, true>), dim3(grid), dim3(BLOCK), 0, h->stream, args, h->cfg.seed, epoch,
                           h->cfg.gid0, h->n, (int)h->has_prev, n_iters, h->anc, h->rows[h->cur], h->rows[1 - h->cur], h->lw,
                           h->acc_part, RW ? next_slots(h) : MaxSlots{nullptr, nullptr});
    else
        GPF_LAUNCH((k_move<M, Wc, RW, false>), dim3(grid), dim3(BLOCK), 0, h->stream, args, h->cfg.seed, epoch,
                           h->cfg.gid0, h->n, (int)h->has_prev, n_iters, h->anc, h->rows[h->cur], h->rows[1 - h->cur], h->lw,
                           h->acc_part, RW ? next_slots(h) : MaxSlots{nullptr, nullptr});
}
// the pending move inside the propagate (k_move_step): old observation + the move's epoch, new observation (h->args) + the update's epoch
template <int M, bool RW>
void launch_move_step_t(gpf_filter* h, int grid)
{
    constexpr int Wc = row_width(Model<M>::D, true);
    ObsVec om;
    for (int i = 0; i < MAX_OBS; ++i) om.v[i] = h->pm_args.obs[i];
    const MaxSlots ms = next_slots(h);
    if (h->pending_gather)
        GPF_LAUNCH((k_move_step<M, Wc, RW, true>), dim3(grid), dim3(BLOCK), 0, h->stream, h->args, om, h->cfg.seed, h->pm_epoch, h->epoch,
                   h->cfg.gid0, h->n, (int)h->has_prev, h->pm_iters, h->anc, h->rows[h->cur], h->rows[1 - h->cur], h->lw, ms);
    else
        GPF_LAUNCH((k_move_step<M, Wc, RW, false>), dim3(grid), dim3(BLOCK), 0, h->stream, h->args, om, h->cfg.seed, h->pm_epoch, h->epoch,
                   h->cfg.gid0, h->n, (int)h->has_prev, h->pm_iters, h->anc, h->rows[h->cur], h->rows[1 - h->cur], h->lw, ms);
}

#define DISPATCH_MODEL(h, CALL)                                                                  \
    switch ((h)->cfg.model) {                                                                    \
        case MODEL_LGSSM2: { constexpr int MM = MODEL_LGSSM2; CALL; } break;                     \
        case MODEL_BEARINGS4: { constexpr int MM = MODEL_BEARINGS4; CALL; } break;               \
        case MODEL_SV1: { constexpr int MM = MODEL_SV1; CALL; } break;                           \
        case MODEL_OBJECT_MOTION: { constexpr int MM = MODEL_OBJECT_MOTION; CALL; } break;       \
        case MODEL_LINE: { constexpr int MM = MODEL_LINE; CALL; } break;                         \
    }

void launch_gather_ex(gpf_filter* h, const int32_t* anc, const double* in, double* out, const PrioView& pv, double* lw_out, int64_t n)
{
    const int grid = grid_for(h, n * (h->W / 2), 8);
    switch (h->W) {
        case 2: GPF_LAUNCH((k_gather<2>), dim3(grid), dim3(BLOCK), 0, h->stream, anc, in, out, pv, lw_out, n); break;
        case 4: GPF_LAUNCH((k_gather<4>), dim3(grid), dim3(BLOCK), 0, h->stream, anc, in, out, pv, lw_out, n); break;
        case 8: GPF_LAUNCH((k_gather<8>), dim3(grid), dim3(BLOCK), 0, h->stream, anc, in, out, pv, lw_out, n); break;
    }
}
void launch_gather(gpf_filter* h, const PrioView& pv, double* lw_out)
{
    launch_gather_ex(h, h->anc, h->rows[h->cur], h->rows[1 - h->cur], pv, lw_out, h->n);
}
void launch_gather_rows_lw(gpf_filter* h, const int32_t* anc, const double* rows_in, const double* lw_in, double* rows_out,
                           double* lw_out, int64_t n)
{
    const int grid = grid_for(h, n * (h->W / 2), 8);
    switch (h->W) {
        case 2: GPF_LAUNCH((k_gather_rows_lw<2>), dim3(grid), dim3(BLOCK), 0, h->stream, anc, rows_in, lw_in, rows_out, lw_out, n); break;
        case 4: GPF_LAUNCH((k_gather_rows_lw<4>), dim3(grid), dim3(BLOCK), 0, h->stream, anc, rows_in, lw_in, rows_out, lw_out, n); break;
        case 8: GPF_LAUNCH((k_gather_rows_lw<8>), dim3(grid), dim3(BLOCK), 0, h->stream, anc, rows_in, lw_in, rows_out, lw_out, n); break;
    }
}

PrioView raw_view(const gpf_filter* h) { return PrioView{h->lw, nullptr, 0.0, 0}; }

// a pending resample gather is executed now: rows[1-cur][j] = rows[cur][anc[j]], lw = 0
gpf_status materialize(gpf_filter* h)
{
    if (h->pending_packed) {                                     // scatter the received exchange buffer by slot (+ log-ML update)
        double* out = h->rows[1 - h->cur];
        const int64_t m = h->pend_own ? h->pend_m : h->n;
        if (h->pend_own) {                                       // the shard's own hits first (anc[j] >= 0; the scatter below overwrites the -1 of the others)
            const int go = grid_for(h, h->n * (h->W / 2), 8);
            const int64_t* own_rng = h->pend_own_range ? h->shard_plan->own_range : nullptr;
            switch (h->W) {
                case 2: GPF_LAUNCH((k_gather_own<2>), dim3(go), dim3(BLOCK), 0, h->stream, h->anc, h->cfg.gid0, h->rows[h->cur], out, h->lw, h->n, own_rng); break;
                case 4: GPF_LAUNCH((k_gather_own<4>), dim3(go), dim3(BLOCK), 0, h->stream, h->anc, h->cfg.gid0, h->rows[h->cur], out, h->lw, h->n, own_rng); break;
                case 8: GPF_LAUNCH((k_gather_own<8>), dim3(go), dim3(BLOCK), 0, h->stream, h->anc, h->cfg.gid0, h->rows[h->cur], out, h->lw, h->n, own_rng); break;
            }
        }
        const int grid = grid_for(h, std::max<int64_t>(m, 1), 8);
        switch (h->W) {
            case 2: GPF_LAUNCH((k_commit_packed<2>), dim3(grid), dim3(BLOCK), 0, h->stream, h->pend_packed, m, out, h->anc, h->lw, h->pend_mf, h->pend_tot, h->pend_G, h->K, h->logN, h->sc, (int)h->pend_mailbox); break;
            case 4: GPF_LAUNCH((k_commit_packed<4>), dim3(grid), dim3(BLOCK), 0, h->stream, h->pend_packed, m, out, h->anc, h->lw, h->pend_mf, h->pend_tot, h->pend_G, h->K, h->logN, h->sc, (int)h->pend_mailbox); break;
            case 8: GPF_LAUNCH((k_commit_packed<8>), dim3(grid), dim3(BLOCK), 0, h->stream, h->pend_packed, m, out, h->anc, h->lw, h->pend_mf, h->pend_tot, h->pend_G, h->K, h->logN, h->sc, (int)h->pend_mailbox); break;
        }
        HIP_TRY(h, hipGetLastError());
        h->cur ^= 1;
        h->pending_packed = false; h->pend_own = false;
        h->max_valid = false;
        return GPF_OK;
    }
    if (!h->pending_gather) return GPF_OK;
    { gpf_status fs = finish_search(h); if (fs) return fs; }         // (a lazy multinomial resample: its ancestors first)
    gpf_status s = timed(h, GPF_K_GATHER, [&] { launch_gather(h, raw_view(h), h->lw); });
    if (s) return s;
    if (h->pending_fill) GPF_LAUNCH(k_fill_from, dim3(grid_for(h, h->n, 8)), dim3(BLOCK), 0, h->stream, h->lw, h->n, &h->sc->lw_fill);
    HIP_TRY(h, hipGetLastError());
    h->cur ^= 1;                    // update_refs! (utils.jl:10-15)
    h->pending_gather = false; h->pending_fill = false;
    h->max_valid = false;           // log-weights are all 0 now
    return GPF_OK;
}

// ------------------------------------------------------------------ trajectory store
void hist_clear(gpf_filter* h)
{
    for (double* p : h->hist_x) if (p) (void)hipFree(p);
    for (int32_t* p : h->hist_map) if (p) (void)hipFree(p);
    h->hist_x.clear(); h->hist_map.clear(); h->hist_step = -1;
}
// snapshot the latent columns of the CURRENT step (final order: called when the step is over, or at query time)
gpf_status hist_snapshot(gpf_filter* h)
{
    if (!h->hist_on || h->hist_step < 0) return GPF_OK;
    gpf_status s = materialize(h);
    if (s) return s;
    double*& dst = h->hist_x[h->hist_step];
    if (!dst) HIP_TRY(h, hipMalloc(&dst, (size_t)h->n * h->d * sizeof(double)));
    GPF_LAUNCH(k_hist_snapshot, dim3(grid_for(h, h->n * h->d, 8)), dim3(BLOCK), 0, h->stream, h->rows[h->cur], h->W, h->d, h->n, dst);
    HIP_TRY(h, hipGetLastError());
    return GPF_OK;
}
// a resample happened during the current step: compose its ancestors into the step's map
gpf_status hist_on_resample(gpf_filter* h)
{
    if (!h->hist_on || h->hist_step < 0) return GPF_OK;
    int32_t* old = h->hist_map[h->hist_step];
    int32_t* neu = nullptr;
    HIP_TRY(h, hipMalloc(&neu, (size_t)h->n * sizeof(int32_t)));
    GPF_LAUNCH(k_hist_compose, dim3(grid_for(h, h->n, 8)), dim3(BLOCK), 0, h->stream, h->anc, old, h->n, neu);
    HIP_TRY(h, hipGetLastError());
    if (old) { HIP_TRY(h, hipStreamSynchronize(h->stream)); (void)hipFree(old); }
    h->hist_map[h->hist_step] = neu;
    return GPF_OK;
}
gpf_status hist_begin_step(gpf_filter* h, bool first)
{
    if (!h->hist_on) return GPF_OK;
    if (first) hist_clear(h);
    else { gpf_status s = hist_snapshot(h); if (s) return s; }     // the step that ends now, in its final order
    if ((int)h->hist_x.size() >= h->hist_cap) return fail(h, GPF_ERR_STATE, "trajectory store full: raise max_steps of gpf_history_enable");
    h->hist_x.push_back(nullptr); h->hist_map.push_back(nullptr);
    h->hist_step = (int)h->hist_x.size() - 1;
    return GPF_OK;
}

// ------------------------------------------------------------------ weight summary = (max) + scan
// The scan's inter-workgroup protocol needs every workgroup of the launch resident at once (block b owns tiles b, b + G, ...
// and waits for lower tiles of its round): at most scan_blocks_per_cu per CU, from the occupancy query at gpf_create.
int scan_grid(const gpf_filter* h) { return (int)std::max<int64_t>(1, std::min<int64_t>(h->ntiles, (int64_t)h->scan_blocks_per_cu * h->n_cu)); }
// the weight scans (k_scan<InFixQ, *>): up to WSCAN_MAX workgroups per CU, so that filters of up to WSCAN_MAX x 0.52 M particles scan in ONE round
int wscan_grid(const gpf_filter* h) { return (int)std::max<int64_t>(1, std::min<int64_t>(h->ntiles, (int64_t)h->wscan_blocks_per_cu * h->n_cu)); }

// one scan launch on descriptor channel `ch` (0 weights, 1 residual counts, 2 residual weights)
template <class In, int FIXQ>
gpf_status scan_launch(gpf_filter* h, int ch, const In& in, int np, WSum* slot, bool want_cdf, uint64_t* total_out,
                       const double* mf_all = nullptr, ScanExtras ex = ScanExtras{nullptr, nullptr, 0, 0})
{
    uint64_t* dc = h->desc[ch][h->dcur[ch]];
    uint64_t* dn = h->desc[ch][1 - h->dcur[ch]];
    int gs = std::is_same<In, InFixQ>::value ? wscan_grid(h) : scan_grid(h);
    // a sorted multinomial resample is waiting for its tile totals: they ride in this launch as extra workgroups behind the scan's own
    if (h->sp_job_set && std::is_same<In, InFixQ>::value) { ex.sp = h->sp_job; ex.sp.blocks = (int)((h->sp_job.ntl + SCAN_BLOCK - 1) / SCAN_BLOCK); h->sp_job_set = false; }
    const int g_launch = gs + ex.sp.blocks;
    const bool offsets = ch == 0 && h->want_offsets && want_cdf;
    const ScanOut so = scan_out(want_cdf ? h->cdf[ch] : nullptr, h->t16[ch], h->t256[ch], h->ntiles, offsets);
    if (ch == 0 && want_cdf) h->ch0_offsets = offsets && so.off16 != nullptr;
    gpf_status s = timed(h, GPF_K_SCAN, [&] {
        GPF_LAUNCH((k_scan<In, FIXQ>), dim3(g_launch), dim3(SCAN_BLOCK), 0, h->stream, in, h->n, h->ntiles, mf_all, h->mslots[h->mcur], np, slot,
                           so, dc, dn, total_out, h->blockQ, h->h_timeout, ex);
    });
    if (s) return s;
    h->table[ch] = dc + h->ntiles;
    h->dcur[ch] ^= 1;
    return GPF_OK;
}

// the maximum slots (MaxSlots) describe pv: the slots left by the kernel that produced the log-weights when pv is the raw weights and
// they are current (use_producer_max), else one k_max_partial pass over pv
gpf_status ensure_max(gpf_filter* h, const PrioView& pv, bool use_producer_max)
{
    if (use_producer_max && h->max_valid) return GPF_OK;
    const int gp = (int)std::min<int64_t>(MAX_PARTIALS, (h->n + BLOCK - 1) / BLOCK);
    gpf_status s = timed(h, GPF_K_MAX, [&] {
        GPF_LAUNCH(k_max_partial, dim3(gp), dim3(BLOCK), 0, h->stream, pv, h->n, next_slots(h));
    });
    if (s) return s;
    h->max_valid = use_producer_max;           // (otherwise the slots describe pv, which may not be the raw log-weights)
    return GPF_OK;
}
// pv: the weights to summarise; max_ready: ensure_max(pv) has run already (the sorted resample needs the maximum for its sort keys)
gpf_status summarize(gpf_filter* h, const PrioView& pv, WSum* slot, bool want_cdf, const int32_t* order, bool use_producer_max,
                     bool want_q = false, bool publish_flags = false, bool max_ready = false)
{
    ScanExtras ex{nullptr, nullptr, 0, h->cfg.n_global};
    if (publish_flags) {
        if (!h->h_flags) { HIP_TRY(h, hipHostMalloc(&h->h_flags, 2 * sizeof(int64_t))); h->h_flags[0] = h->h_flags[1] = 0; }
        h->flag_ticket += 1;
        ex.host_flags = h->h_flags; ex.ticket = h->flag_ticket;
    }
    const int np = 0;               // (only the sharded scans fold gathered pairs)
    gpf_status s;
    h->q_published = false;
    static const bool q_publish_off = getenv("GPF_ESS_PUBLISH") && !strcmp(getenv("GPF_ESS_PUBLISH"), "kernel");   // (A/B: the separate publish launch)
    // (tag << 48 | limb sum: only while a workgroup folds <= Q_TAG_MAX_TILES tiles -- beyond, e.g. N > 2^26 at 4 x 256 workgroups, the
    // untagged partials + k_publish_scalars)
    const int64_t tiles_per_wg = (h->ntiles + wscan_grid(h) - 1) / wscan_grid(h);
    if (want_q && slot == &h->sc->raw && !q_publish_off && tiles_per_wg <= Q_TAG_MAX_TILES) {
        // the ESS getter's scan: the workgroup of its last tile folds sum q^2 and publishes {flags, S, limbs} to pinned memory itself
        if (!h->h_qpub) { HIP_TRY(h, hipHostMalloc(&h->h_qpub, 8 * sizeof(int64_t))); for (int i = 0; i < 8; ++i) h->h_qpub[i] = 0; }
        h->q_ticket += 1;
        ex.q_host = h->h_qpub; ex.q_ticket = h->q_ticket;
        h->q_published = true;
    }
    if (!max_ready && (s = ensure_max(h, pv, use_producer_max))) return s;
    InFixQ in{pv, order, order ? h->keys : nullptr, h->K, 0.0, 0};     // (after sort_desc the sorted keys are in h->keys)
    if (want_q) s = scan_launch<InFixQ, 2>(h, 0, in, np, slot, want_cdf, &slot->S, nullptr, ex);
    else        s = scan_launch<InFixQ, 1>(h, 0, in, np, slot, want_cdf, &slot->S, nullptr, ex);
    if (s) return s;
    HIP_TRY(h, hipGetLastError());
    return GPF_OK;
}

// want_q: also accumulate sum q^2 (only the ESS needs it)
gpf_status ensure_raw(gpf_filter* h, bool want_q = false)
{
    gpf_status s = materialize(h);
    if (s) return s;
    if (h->raw_valid && (!want_q || h->raw_has_q)) return GPF_OK;
    h->want_offsets = h->offsets_hint;                           // a resample that follows may reuse this CDF
    s = summarize(h, raw_view(h), &h->sc->raw, true, nullptr, true, want_q);
    h->want_offsets = true;
    if (s) return s;
    h->raw_valid = true;
    h->raw_has_q = want_q;
    h->raw_q_folded = h->q_published;                            // (a publishing scan folds the limbs itself; its partials are tagged words)
    return GPF_OK;
}

gpf_status wait_ticket(gpf_filter* h, volatile int64_t* tk, int64_t want, const char* what);
gpf_status check_scan_timeout(gpf_filter* h);
// {flags, S, limbs} as published by the ESS scan or by k_sum_reduce (8 words, unordered on their way to pinned memory: re-read until the
// check word -- ticket ^ payload -- agrees)
gpf_status read_published_summary(gpf_filter* h, WSum& w)
{
    gpf_status s;
    if ((s = wait_ticket(h, h->h_qpub + 6, h->q_ticket, "weight summary"))) return s;
    if ((s = check_scan_timeout(h))) return s;
    for (uint64_t spins = 0;; ++spins) {
        int64_t v[8];
        for (int k = 0; k < 8; ++k) v[k] = __atomic_load_n(h->h_qpub + k, __ATOMIC_ACQUIRE);
        uint64_t chk = (uint64_t)v[6];
        for (int k = 0; k < 6; ++k) chk ^= (uint64_t)v[k];
        if (v[6] == h->q_ticket && chk == (uint64_t)v[7]) {
            w.flags = (int32_t)v[0]; w.S = (uint64_t)v[1];
            for (int k = 0; k < 4; ++k) w.Ql[k] = (uint64_t)v[2 + k];
            return GPF_OK;
        }
        cpu_relax();
        if (spins > (1ull << 26)) return fail(h, GPF_ERR_HIP, "weight summary: the published words never became consistent");
    }
}
// The summary of the raw log-weights WITHOUT the CDF (the ESS and log-ML getters): reuses a valid scan, else ONE reduction launch
// (k_sum_reduce) instead of the scan -- no inter-workgroup chain, no 10 MB of CDF and levels.  false in *done: the filter is too large
// for the tagged partials (a workgroup would fold more than Q_TAG_MAX_TILES tiles): the caller takes the scan.
gpf_status ensure_raw_summary(gpf_filter* h, bool want_q, bool* done)
{
    *done = false;
    gpf_status s = materialize(h);
    if (s) return s;
    static const bool off = getenv("GPF_SUM_REDUCE") && !strcmp(getenv("GPF_SUM_REDUCE"), "0");          // (A/B: always the scan)
    if (off || (h->raw_valid && (!want_q || h->raw_has_q))) return GPF_OK;                              // (a scan's summary is there: use it)
    if (h->raw_sum_valid) { *done = true; return GPF_OK; }
    // GPF_SUM_REDUCE=device: the reduction whose workgroup 0 folds the partials on the device (k_sum_reduce) instead of the host (k_sum_host)
    static const bool device_fold = getenv("GPF_SUM_REDUCE") && !strcmp(getenv("GPF_SUM_REDUCE"), "device");
    const int hgrid = (int)std::max<int64_t>(1, std::min<int64_t>((h->n + SH_TILE - 1) / SH_TILE, (int64_t)h->n_cu));
    if (!device_fold && (h->n + hgrid - 1) / hgrid <= (int64_t)Q_TAG_MAX_TILES * TILE) {
        // every workgroup's partial sums go straight to pinned memory; this thread adds them up
        if (!h->h_spart) {
            HIP_TRY(h, hipHostMalloc(&h->h_spart, (size_t)8 * h->n_cu * sizeof(int64_t)));
            memset(h->h_spart, 0, (size_t)8 * h->n_cu * sizeof(int64_t));
        }
        if ((s = ensure_max(h, raw_view(h), true))) return s;
        h->q_ticket += 1;
        InFixQ in{raw_view(h), nullptr, nullptr, h->K, 0.0, 0};
        s = timed(h, GPF_K_SCAN, [&] {
            GPF_LAUNCH(k_sum_host, dim3(hgrid), dim3(SH_BLOCK), 0, h->stream, in, h->n, h->mslots[h->mcur], h->h_spart, h->q_ticket);
        });
        if (s) return s;
        HIP_TRY(h, hipGetLastError());
        const uint64_t tag = (uint64_t)((h->q_ticket & 0x7fff) + 1);
        uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        uint64_t flags_m[3] = {0, 0, 0};
        for (int b = 0; b < hgrid; ++b) {
            volatile int64_t* line = h->h_spart + (size_t)b * 8;
            uint64_t v[8];
            for (int k = 0; k < 8; ++k) {
                uint64_t spins = 0;
                while (((v[k] = (uint64_t)__atomic_load_n(line + k, __ATOMIC_ACQUIRE)) >> 48) != tag) {
                    cpu_relax();
                    if ((++spins & 0x3fff) != 0) continue;
                    const hipError_t q = hipStreamQuery(h->stream);
                    if (q == hipErrorNotReady) continue;
                    if (((uint64_t)__atomic_load_n(line + k, __ATOMIC_ACQUIRE) >> 48) == tag) continue;
                    return fail(h, GPF_ERR_HIP, q == hipSuccess ? "weight summary: the stream drained without the partial sums being published" : hipGetErrorString(q));
                }
                v[k] &= 0xffffffffffffull;
            }
            t[0] += v[0]; t[1] += v[1] & 0xffffffffffull;             // (S can be 2^62 itself: the high part takes 32 bits; the flags sit above bit 40)
            for (int k = 2; k < 6; ++k) t[k] += v[k];
            if (b == 0) { flags_m[0] = v[1] >> 40; flags_m[1] = v[6]; flags_m[2] = v[7]; }
        }
        WSum w{};
        w.flags = (int32_t)flags_m[0];
        w.S = t[0] + (t[1] << 31);
        for (int k = 0; k < 4; ++k) w.Ql[k] = t[2 + k];
        const uint64_t mb = flags_m[1] | (flags_m[2] << 32);
        memcpy(&w.m, &mb, sizeof(double));
        h->sum_cache = w;                                        // (sc->raw on the device is NOT updated: the getters read this copy)
        h->sum_on_host = true;
        h->raw_sum_valid = true;
        *done = true;
        return GPF_OK;
    }
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(h->ntiles, (int64_t)h->n_cu * 4));
    if ((h->ntiles + grid - 1) / grid > Q_TAG_MAX_TILES) return GPF_OK;
    if (!h->sum_part) {
        HIP_TRY(h, hipMalloc(&h->sum_part, (size_t)6 * 4 * h->n_cu * sizeof(uint64_t)));
        HIP_TRY(h, hipMemsetAsync(h->sum_part, 0, (size_t)6 * 4 * h->n_cu * sizeof(uint64_t), h->stream));
    }
    if (!h->h_qpub) { HIP_TRY(h, hipHostMalloc(&h->h_qpub, 8 * sizeof(int64_t))); for (int i = 0; i < 8; ++i) h->h_qpub[i] = 0; }
    if ((s = ensure_max(h, raw_view(h), true))) return s;
    h->q_ticket += 1;
    InFixQ in{raw_view(h), nullptr, nullptr, h->K, 0.0, 0};
    s = timed(h, GPF_K_SCAN, [&] {
        GPF_LAUNCH(k_sum_reduce, dim3(grid), dim3(SCAN_BLOCK), 0, h->stream, in, h->n, h->ntiles, h->mslots[h->mcur], &h->sc->raw, h->sum_part, h->h_qpub,
                   h->q_ticket, h->h_timeout);
    });
    if (s) return s;
    HIP_TRY(h, hipGetLastError());
    WSum w{};
    if ((s = read_published_summary(h, w))) return s;
    h->sum_cache = w;                                            // (m is not published by this kernel: the log-ML getter reads the device block)
    h->sum_on_host = false;
    h->raw_sum_valid = true;
    *done = true;
    return GPF_OK;
}

// Poll a pinned ticket that a kernel on h->stream publishes.  A failed kernel never writes it: any stream status other than
// "not ready" is terminal (re-read once, then report), so a faulting kernel cannot hang the host -- or, in a multi-rank job,
// its peers in the next collective.
gpf_status wait_ticket(gpf_filter* h, volatile int64_t* tk, int64_t want, const char* what)
{
    uint64_t spins = 0;
    while (__atomic_load_n(tk, __ATOMIC_ACQUIRE) != want) {
        cpu_relax();
        if ((++spins & 0x3fff) != 0) continue;
        const hipError_t q = hipStreamQuery(h->stream);
        if (q == hipErrorNotReady) continue;
        if (__atomic_load_n(tk, __ATOMIC_ACQUIRE) == want) break;
        if (q == hipSuccess) return fail(h, GPF_ERR_HIP, std::string(what) + ": the stream drained without the ticket being published");
        return fail(h, GPF_ERR_HIP, std::string(what) + ": " + hipGetErrorString(q));
    }
    return GPF_OK;
}
// a scan whose bounded inter-workgroup wait gave up leaves garbage prefixes behind: fail loudly at the next host touch point
gpf_status check_scan_timeout(gpf_filter* h)
{
    if (h->h_timeout && __atomic_load_n(h->h_timeout, __ATOMIC_ACQUIRE) == 2)
        return fail(h, GPF_ERR_HIP, "sharded resample: a peer's summary did not arrive in its mailbox in time (a rank is down or far behind); results are invalid");
    if (h->h_timeout && __atomic_load_n(h->h_timeout, __ATOMIC_ACQUIRE) != 0)
        return fail(h, GPF_ERR_HIP, "scan kernel: bounded inter-workgroup wait timed out (workgroups not co-resident?); results are invalid");
    return GPF_OK;
}

gpf_status fetch_scalars(gpf_filter* h, bool fold_raw_q = false)
{
    if (!h->h_sc_ticket) { HIP_TRY(h, hipHostMalloc(&h->h_sc_ticket, sizeof(long long))); *h->h_sc_ticket = 0; }
    h->sc_ticket += 1;
    GPF_LAUNCH(k_publish_scalars, dim3(1), dim3(WAVE), 0, h->stream, h->sc, h->h_sc, h->h_sc_ticket, h->sc_ticket,
               fold_raw_q ? h->blockQ : nullptr, fold_raw_q ? wscan_grid(h) : 0);
    HIP_TRY(h, hipGetLastError());
    { gpf_status w = wait_ticket(h, reinterpret_cast<volatile int64_t*>(h->h_sc_ticket), (int64_t)h->sc_ticket, "scalar block"); if (w) return w; }
    return check_scan_timeout(h);
}

void normalise_Q(const WSum& w, uint64_t& hi, uint64_t& lo)
{
    unsigned __int128 Q = (unsigned __int128)w.Ql[0] + ((unsigned __int128)w.Ql[1] << 32) +
                          ((unsigned __int128)w.Ql[2] << 64) + ((unsigned __int128)w.Ql[3] << 96);
    hi = (uint64_t)(Q >> 64);
    lo = (uint64_t)Q;
}

// Every change of a filter's rows / log-weights -- through the filter itself or through any view of it -- bumps the ROOT's
// mutation counter.  A view's cached summaries (raw CDF, sum q^2, producer maxima) describe the weights at the value it last
// saw; view_enter drops them when the counter has moved (the reference's SubArray views are live, src/view.jl:35-48).
void mutated(gpf_filter* h)
{
    gpf_filter* root = h->parent ? h->parent : h;
    root->mutations += 1;
    if (h->parent) h->seen_mutations = root->mutations;          // its own change: this view's bookkeeping is already current
}

// A view re-derives its aliased pointers from the parent on every call (the parent may have swapped its row buffers),
// shares the parent's epoch counter, and forces a pending gather of the parent first.
gpf_status view_enter(gpf_filter* v)
{
    gpf_filter* p = v->parent;
    if (p->generation != v->parent_generation) return fail(v, GPF_ERR_STATE, "stale view: the parent filter was resized or re-created");
    if (!p->initialized) return fail(v, GPF_ERR_STATE, "parent filter not initialised");
    if (p->pending_move) { gpf_status ms = finish_move(p); if (ms) { v->err = p->err; return ms; } }
    gpf_status s = materialize(p);
    if (s) { v->err = p->err; return s; }
    const int64_t o = v->view_start;
    if (v->view_step == 1) {
        v->rows[0] = p->rows[p->cur] + o * v->W;
        v->rows[1] = p->rows[1 - p->cur] + o * v->W;
        v->lw = p->lw + o;
        v->anc = p->anc + o;
    } else {                                                     // strided: a compact copy of particles o + i * step
        v->rows[0] = v->vrows[0]; v->rows[1] = v->vrows[1]; v->lw = v->vlw; v->anc = v->vanc;
        if (v->view_step == 0)                                   // state[idxs]: an arbitrary index vector
            GPF_LAUNCH(k_view_index_copy, dim3(grid_for(v, v->n * (v->W / 2), 8)), dim3(BLOCK), 0, v->stream, p->rows[p->cur], p->lw, p->anc,
                       v->vrows[0], v->vlw, v->vanc, v->W, v->vidx, v->n, 1);
        else
        GPF_LAUNCH(k_view_strided_copy, dim3(grid_for(v, v->n * (v->W / 2), 8)), dim3(BLOCK), 0, v->stream, p->rows[p->cur] + o * v->W, p->lw + o, p->anc + o,
                   v->vrows[0], v->vlw, v->vanc, v->W, v->view_step, v->n, 1);
        HIP_TRY(v, hipGetLastError());
    }
    v->cur = 0;
    v->epoch = p->epoch;
    v->has_prev = p->has_prev;
    v->initialized = true;
    if (v->seen_mutations != p->mutations) {                     // the aliased weights changed behind this view's back
        v->raw_valid = false; v->raw_sum_valid = false; v->raw_has_q = false; v->raw_q_folded = false; v->max_valid = false;
        v->seen_mutations = p->mutations;
    }
    return GPF_OK;
}
// after a mutating call on a view: update_refs! for sub-states copies back (utils.jl:17-20); parent caches are stale
gpf_status view_exit(gpf_filter* v)
{
    if (!v->parent) return GPF_OK;
    gpf_filter* p = v->parent;
    if (v->view_step != 1) {                                     // strided: scatter the compact copy back into the source
        const int64_t o = v->view_start;
        if (v->view_step == 0)
            GPF_LAUNCH(k_view_index_copy, dim3(grid_for(v, v->n * (v->W / 2), 8)), dim3(BLOCK), 0, v->stream, p->rows[p->cur], p->lw, p->anc,
                       v->rows[v->cur], v->vlw, v->vanc, v->W, v->vidx, v->n, 0);
        else
        GPF_LAUNCH(k_view_strided_copy, dim3(grid_for(v, v->n * (v->W / 2), 8)), dim3(BLOCK), 0, v->stream, p->rows[p->cur] + o * v->W, p->lw + o, p->anc + o,
                   v->rows[v->cur], v->vlw, v->vanc, v->W, v->view_step, v->n, 0);
        HIP_TRY(v, hipGetLastError());
        v->cur = 0;
    } else if (v->cur == 1) {
        HIP_TRY(v, hipMemcpyAsync(v->rows[0], v->rows[1], (size_t)v->n * v->W * sizeof(double), hipMemcpyDeviceToDevice, v->stream));
        v->cur = 0;
    }
    p->epoch = v->epoch;
    p->has_prev = p->has_prev || v->has_prev;
    p->raw_valid = false; p->raw_sum_valid = false; p->max_valid = false; p->raw_has_q = false; p->raw_q_folded = false;
    return GPF_OK;
}

gpf_status check_ready(gpf_handle h, bool keep_pending_move = false)
{
    if (!h) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null handle");
    if (h->pending_move && !keep_pending_move) { gpf_status ms = finish_move(h); if (ms) return ms; }   // (a lazy move: only the plain pf_update! carries it)
    if (h->parent) { gpf_status vs = view_enter(h); if (vs) return vs; }
    if (!h->initialized) return fail(h, GPF_ERR_STATE, "filter not initialised: call gpf_initialize (pf_initialize) first");
    HIP_TRY(h, hipSetDevice(h->cfg.device));       // launches go to the calling thread's current device
    return GPF_OK;
}

gpf_status set_obs(gpf_filter* h, const double* obs, int n_obs)
{
    if (n_obs < 0 || n_obs > MAX_OBS || (n_obs > 0 && !obs)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad observation vector");
    // a native model's step is defined by its full data vector: an empty choicemap() (no constraint, weight 0) has no
    // device meaning and must not silently become "observed zeros"
    if (n_obs != model_obs_dim(h->cfg.model))
        return fail(h, GPF_ERR_INVALID_ARGUMENT, "this model takes " + std::to_string(model_obs_dim(h->cfg.model)) + " observation values per step");
    for (int i = 0; i < MAX_OBS; ++i) h->args.obs[i] = i < n_obs ? obs[i] : 0.0;
    return GPF_OK;
}

gpf_status ensure_sort_buffers(gpf_filter* h)
{
    if (h->order) return GPF_OK;
    const size_t n = (size_t)h->n;
    HIP_TRY(h, hipMalloc(&h->order, n * sizeof(int32_t)));
    HIP_TRY(h, hipMalloc(&h->idx_in, n * sizeof(int32_t)));
    HIP_TRY(h, hipMalloc(&h->keys, n * sizeof(uint64_t)));
    HIP_TRY(h, hipMalloc(&h->keys_out, n * sizeof(uint64_t)));
    // TWO workspaces (histograms, tickets, descriptor planes), used in turn: every sort clears the other one for the next sort
    h->sort_tmp_bytes = (sort_ws_bytes(h->n) + 15) & ~(size_t)15;
    HIP_TRY(h, hipMalloc(&h->sort_tmp, 2 * h->sort_tmp_bytes));
    HIP_TRY(h, hipMemsetAsync(h->sort_tmp, 0, 2 * h->sort_tmp_bytes, h->stream));
    h->sort_ws_cur = 0;
    return GPF_OK;
}

// order = sortperm(log_priorities, rev=true) (resample.jl:156-157) into h->order, the sorted keys into h->keys (gpf_k_sort.hpp K10):
// keys + digit histograms in one pass, then one onesweep kernel per 8-bit digit.
//   coarse = true : three passes over the 24-bit coarse key (sort_coarse: distance from the maximum, which the maximum slots must hold
//                   -- ensure_max); the caller finishes the runs of equal coarse keys (k_sort_finish).  keys -> keys_out -> keys ->
//                   keys_out, payload index -> idx_in -> order -> idx_in: the finish brings both back to h->keys / h->order.
//   coarse = false: all eight passes over the 64-bit key; the eighth leaves keys / permutation in h->keys / h->order.
gpf_status sort_passes(gpf_filter* h, const PrioView& pv, int64_t n, bool coarse, uint32_t** ws_used = nullptr, bool buckets = false)
{
    gpf_status s = ensure_sort_buffers(h);
    if (s) return s;
    static_assert(SORT_BINS == BLOCK, "one thread per digit bin");
    // this sort's workspace starts zeroed (by the previous sort, or by the allocation); the key pass zeroes the other one
    char* ws = static_cast<char*>(h->sort_tmp) + (size_t)h->sort_ws_cur * h->sort_tmp_bytes;
    char* other = static_cast<char*>(h->sort_tmp) + (size_t)(1 - h->sort_ws_cur) * h->sort_tmp_bytes;
    h->sort_ws_cur ^= 1;
    if (ws_used) *ws_used = reinterpret_cast<uint32_t*>(ws);
    uint32_t* hist = reinterpret_cast<uint32_t*>(ws);
    uint32_t* const ticket_words = hist + SORT_PASSES * SORT_BINS;
    double* m_ptr = reinterpret_cast<double*>(ticket_words + SORT_M_WORD);
    uint32_t* fine = reinterpret_cast<uint32_t*>(ws + sort_ws_fine_offset());
    uint32_t* bbase = fine + SORT_FINE;
    uint64_t* desc = reinterpret_cast<uint64_t*>(ws + sort_ws_desc_offset());
    const int64_t nt = (n + SORT_TILE - 1) / SORT_TILE;
    uint32_t* const ticket = ticket_words;
    const int64_t clear16 = (int64_t)(h->sort_tmp_bytes / 16);
    // (ONE workgroup per CU: every workgroup ends with up to 256 global atomic adds per sorted digit into the same counters; with 2 / 4
    //  workgroups per CU a four-digit kernel took 16.0 / 24.2 us against 13.3)
    const unsigned long long* slots = h->mslots[h->mcur];
    if (buckets) {
        // K10d: keys + fine-bin histogram, ONE partition pass (keys -> keys_out, payload index -> idx_in); the caller runs k_sort_buckets
        const int64_t kf_grid = std::max<int64_t>(1, std::min<int64_t>((n + 4 * KF_BLOCK - 1) / (4 * KF_BLOCK), h->n_cu));
        GPF_LAUNCH(k_sort_keys_fine, dim3((unsigned)kf_grid), dim3(KF_BLOCK), 0, h->stream, pv, n, h->keys, fine, reinterpret_cast<uint4*>(other), clear16, slots, m_ptr);
        GPF_LAUNCH(k_sort_pass<2>, dim3((unsigned)nt), dim3(SORT_BLOCK), 0, h->stream, h->keys, nullptr, h->keys_out, h->idx_in, n, 0, hist, ticket, desc, h->h_timeout, m_ptr, fine, bbase);
        HIP_TRY(h, hipGetLastError());
        return GPF_OK;
    }
    const int64_t kh_grid = std::max<int64_t>(1, std::min<int64_t>((n + 4 * KF_BLOCK - 1) / (4 * KF_BLOCK), h->n_cu));
    if (coarse) GPF_LAUNCH((k_sort_keys_hist<0, true>), dim3((unsigned)kh_grid), dim3(KF_BLOCK), 0, h->stream, pv, n, h->keys, hist, reinterpret_cast<uint4*>(other), clear16, slots, m_ptr);
    else        GPF_LAUNCH((k_sort_keys_hist<0, false>), dim3((unsigned)kh_grid), dim3(KF_BLOCK), 0, h->stream, pv, n, h->keys, hist, reinterpret_cast<uint4*>(other), clear16, slots, m_ptr);
    for (int p = 0; p < (coarse ? 3 : SORT_PASSES); ++p) {
        const uint64_t* kin = (p & 1) ? h->keys_out : h->keys;
        uint64_t* kout = (p & 1) ? h->keys : h->keys_out;
        const int32_t* vin = p == 0 ? nullptr : ((p & 1) ? h->idx_in : h->order);
        int32_t* vout = (p & 1) ? h->order : h->idx_in;
        if (coarse) GPF_LAUNCH(k_sort_pass<1>, dim3((unsigned)nt), dim3(SORT_BLOCK), 0, h->stream, kin, vin, kout, vout, n, p, hist, ticket, desc, h->h_timeout, m_ptr, nullptr, nullptr);
        else        GPF_LAUNCH(k_sort_pass<0>, dim3((unsigned)nt), dim3(SORT_BLOCK), 0, h->stream, kin, vin, kout, vout, n, p, hist, ticket, desc, h->h_timeout, m_ptr, nullptr, nullptr);
    }
    HIP_TRY(h, hipGetLastError());
    return GPF_OK;
}
// three coarse passes + k_sort_finish; all eight passes when a run of equal coarse keys was too long for the finish (the host learns
// it from pinned memory), or with GPF_SORT=radix8 in the environment (A/B measurements; GPF_SORT=fallback: always both, for the tests).
// The maximum slots must describe pv (ensure_max).
//   sort_desc_begin enqueues the sort; *pending = the finish's verdict is still out: the caller may enqueue the work that consumes
//   the order behind it and asks sort_desc_flagged AFTERWARDS (no host wait between the sort and its consumers) -- when that says
//   "flagged" the order was wrong: sort_passes(..., false) and the consumers again.
// GPF_SORT: radix8 = always the eight passes; fallback = the fast path AND the eight passes (tests); coarse3 = the three coarse passes +
// k_sort_finish also where the bucket sort (K10d, n <= BK_MAX_N) would run (A/B measurements, tests of that path at small n)
int sort_mode() { static const int mode = [] { const char* e = getenv("GPF_SORT"); return e && strstr(e, "radix8") ? 1 : (e && strstr(e, "fallback") ? 2 : 0); }(); return mode; }
bool sort_buckets_ok(int64_t n) { static const bool off = getenv("GPF_SORT") && strstr(getenv("GPF_SORT"), "coarse3"); return !off && n <= BK_MAX_N; }
gpf_status sort_desc_begin(gpf_filter* h, const PrioView& pv, int64_t n, bool* pending)
{
    *pending = false;
    if (sort_mode() == 1) return sort_passes(h, pv, n, false);
    const bool buckets = sort_buckets_ok(n);
    uint32_t* ws = nullptr;
    gpf_status s = sort_passes(h, pv, n, true, &ws, buckets);    // (either form leaves keys / payload in h->keys_out / h->idx_in)
    if (s) return s;
    if (!h->h_sort_flag) { HIP_TRY(h, hipHostMalloc(&h->h_sort_flag, 2 * sizeof(int64_t))); h->h_sort_flag[0] = h->h_sort_flag[1] = 0; }
    uint32_t* done = ws + SORT_PASSES * SORT_BINS + 64;                                  // (behind this sort's zeroed tickets)
    const double* m_ptr = reinterpret_cast<const double*>(ws + SORT_PASSES * SORT_BINS + SORT_M_WORD);
    h->sort_ticket += 1;
    if (buckets) {
        const uint32_t* bbase = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(ws) + sort_ws_fine_offset()) + SORT_FINE;
        // every bucket is ordered in place (the dead keys' bucket is left as the partition wrote it): the partition's output buffers
        // become the sorted keys / the permutation
        GPF_LAUNCH(k_sort_buckets, dim3(SORT_BINS), dim3(BK_BLOCK), 0, h->stream, h->keys_out, h->idx_in, n, bbase,
                   done, h->h_sort_flag, h->sort_ticket, m_ptr);
        std::swap(h->keys, h->keys_out);
        std::swap(h->order, h->idx_in);
    } else {
        GPF_LAUNCH(k_sort_finish, dim3((unsigned)((n + FIN_TILE - 1) / FIN_TILE)), dim3(FIN_BLOCK), 0, h->stream, h->keys_out, h->idx_in, h->keys, h->order, n,
                   done, h->h_sort_flag, h->sort_ticket, m_ptr);
    }
    HIP_TRY(h, hipGetLastError());
    *pending = true;
    return GPF_OK;
}
gpf_status sort_desc_flagged(gpf_filter* h, bool* flagged)
{
    // (the finish publishes ticket << 1 | verdict as one word)
    volatile int64_t* tk = h->h_sort_flag + 1;
    uint64_t spins = 0;
    int64_t v;
    while (((v = __atomic_load_n(tk, __ATOMIC_ACQUIRE)) >> 1) != h->sort_ticket) {
        cpu_relax();
        if ((++spins & 0x3fff) != 0) continue;
        const hipError_t q = hipStreamQuery(h->stream);
        if (q == hipErrorNotReady) continue;
        if ((__atomic_load_n(tk, __ATOMIC_ACQUIRE) >> 1) == h->sort_ticket) continue;
        return fail(h, GPF_ERR_HIP, q == hipSuccess ? "sort finish: the stream drained without the ticket being published" : hipGetErrorString(q));
    }
    *flagged = (v & 1) != 0 || sort_mode() == 2;
    return GPF_OK;
}
gpf_status sort_desc(gpf_filter* h, const PrioView& pv, int64_t n)
{
    bool pending = false, flagged = false;
    gpf_status s = sort_desc_begin(h, pv, n, &pending);
    if (s || !pending) return s;
    if ((s = sort_desc_flagged(h, &flagged))) return s;
    return flagged ? sort_passes(h, pv, n, false) : GPF_OK;
}

gpf_status ensure_residual_buffers(gpf_filter* h)
{
    for (int i = 1; i < 3; ++i) {
        if (h->cdf[i]) continue;
        HIP_TRY(h, hipMalloc(&h->cdf[i], (size_t)h->ntiles * TILE * sizeof(uint64_t)));
        HIP_TRY(h, hipMalloc(&h->t16[i], (size_t)h->ntiles * (TILE / 16) * sizeof(uint64_t)));
        HIP_TRY(h, hipMalloc(&h->t256[i], t256_bytes(h->ntiles)));
    }
    return GPF_OK;
}

CdfLevels levels(const gpf_filter* h, int ch)
{
    int logg = ch == 0 && h->ch0_offsets ? multi_logg(h->ntiles) : -1;
    const int sample = ch == 0 && h->ch0_offsets && logg < 0 ? multi_sample(h->ntiles) : 0;
    if (sample > 0) logg = 0;
    return CdfLevels{h->cdf[ch], h->t16[ch], h->t256[ch], h->table[ch], k32_of(h->t256[ch], h->ntiles),
                     logg >= 0 ? off16_of(h->t256[ch], h->ntiles) : nullptr, logg >= 0 ? coarse_of(h->t256[ch], h->ntiles) : nullptr, logg,
                     sample > 0 ? k32s_of(h->t256[ch], h->ntiles) : nullptr, sample};
}

// residual: copy-count and residual-weight CDFs from the weight CDF (resample.jl:99,109); ws->S must be the GLOBAL sum
// head_anc: the plain resample hands over its ancestor array -- the scan writes the deterministic head into it (k_scan_residual2), the
// search then covers the tail only (SearchArgs::head_done) and the copy-count CDF is not stored
gpf_status residual_scans(gpf_filter* h, const WSum* ws, int64_t n_slots_global, int32_t* head_anc = nullptr, const ResidDirect* direct = nullptr)
{
    gpf_status s = ensure_residual_buffers(h);
    if (s) return s;
    // both prefix sums in one pass (one read of the weight CDF, one division per element)
    Scan2Chan ch[2];
    for (int c = 0; c < 2; ++c) {
        const int id = 1 + c;
        uint64_t* dc = h->desc[id][h->dcur[id]];
        ch[c].out = scan_out(h->cdf[id], h->t16[id], h->t256[id], h->ntiles, false);
        ch[c].dcur = dc; ch[c].dnext = h->desc[id][1 - h->dcur[id]];
        ch[c].total_out = c == 0 ? &h->sc->Ctot : &h->sc->Rs;
        h->table[id] = dc + h->ntiles;
        h->dcur[id] ^= 1;
    }
    const int gs = scan_grid(h);
    s = timed(h, GPF_K_SCAN, [&] {
        if (direct) GPF_LAUNCH(k_scan_residual2<true>, dim3(gs), dim3(SCAN_BLOCK), 0, h->stream, h->cdf[0], ws, n_slots_global, h->n, h->ntiles, ch[0], ch[1], h->h_timeout, head_anc, &h->sc->giants, h->epoch & 0xffffffu, *direct);
        else        GPF_LAUNCH(k_scan_residual2<false>, dim3(gs), dim3(SCAN_BLOCK), 0, h->stream, h->cdf[0], ws, n_slots_global, h->n, h->ntiles, ch[0], ch[1], h->h_timeout, head_anc, &h->sc->giants, h->epoch & 0xffffffu, ResidDirect{});
    });
    if (s) return s;
    HIP_TRY(h, hipGetLastError());
    return GPF_OK;
}

// block-wise propagate / move (ModelArgs::blk_*): the default proposal only, no fused gather (a block resample gathers eagerly)
template <int M>
void launch_init_blk(gpf_filter* h, int grid)
{
    GPF_LAUNCH((k_init<M, 0, true>), dim3(grid), dim3(BLOCK), 0, h->stream, h->args, h->cfg.seed, h->epoch,
                       h->cfg.gid0, h->n, h->W, h->rows[h->cur], h->lw, next_slots(h));
}
template <int M, bool KEEP>
void launch_step_blk(gpf_filter* h, int grid)
{
    constexpr int Wc = row_width(Model<M>::D, KEEP);
    GPF_LAUNCH((k_step<M, Wc, KEEP, false, 0, false, true>), dim3(grid), dim3(BLOCK), 0, h->stream, h->args, h->cfg.seed, h->epoch,
                       h->cfg.gid0, h->n, h->anc, h->rows[h->cur], h->rows[1 - h->cur], h->lw, next_slots(h), PackedCommit{});
}
template <int M, bool RW>
void launch_move_blk(gpf_filter* h, int grid, int n_iters)
{
    constexpr int Wc = row_width(Model<M>::D, true);
    GPF_LAUNCH((k_move<M, Wc, RW, false, false, true>), dim3(grid), dim3(BLOCK), 0, h->stream, h->args, h->cfg.seed, h->epoch,
                       h->cfg.gid0, h->n, (int)h->has_prev, n_iters, h->anc, h->rows[h->cur], h->rows[1 - h->cur], h->lw,
                       h->acc_part, RW ? next_slots(h) : MaxSlots{nullptr, nullptr});
}
// a block of <= 128 / <= 512 particles is the work of one wave (2 / 8 particles per lane, four blocks per workgroup), a larger one of a workgroup
template <int METHOD, int Wc, bool PRIO>
void launch_block_resample_w(gpf_filter* h, const BlockArgs& a)
{
    if (a.nb <= 2 * WAVE)      GPF_LAUNCH((k_block_resample<METHOD, Wc, WAVE, 2, PRIO>), dim3((unsigned)((a.nblocks + 3) / 4)), dim3(BLOCK), 0, h->stream, a);
    else if (a.nb <= 8 * WAVE) GPF_LAUNCH((k_block_resample<METHOD, Wc, WAVE, 8, PRIO>), dim3((unsigned)((a.nblocks + 3) / 4)), dim3(BLOCK), 0, h->stream, a);
    else                       GPF_LAUNCH((k_block_resample<METHOD, Wc, BLOCK, 8, PRIO>), dim3((unsigned)a.nblocks), dim3(BLOCK), 0, h->stream, a);
}
template <int METHOD>
void launch_block_resample(gpf_filter* h, const BlockArgs& a, bool prio)
{
    switch (h->W) {
        case 2: if (prio) launch_block_resample_w<METHOD, 2, true>(h, a); else launch_block_resample_w<METHOD, 2, false>(h, a); break;
        case 4: if (prio) launch_block_resample_w<METHOD, 4, true>(h, a); else launch_block_resample_w<METHOD, 4, false>(h, a); break;
        case 8: if (prio) launch_block_resample_w<METHOD, 8, true>(h, a); else launch_block_resample_w<METHOD, 8, false>(h, a); break;
    }
}
// ancestors of i.i.d. targets: k_search_multi (4-byte keys of every 32 / 64 cells in LDS) while the key table fits, else k_search<0>
void launch_multinomial_search(gpf_filter* h, const SearchArgs& sa)
{
    const int logg = sa.w.off16 && sa.w.sample == 0 ? sa.w.logg : -1;   // the offset levels exist for channel 0 only
    if (sa.w.off16 && sa.w.sample > 0) {                         // 2.5 M .. 5 M particles: sampled key table
        const int gss = (int)std::max<int64_t>(1, std::min<int64_t>((sa.n + 2 * SBLOCK - 1) / (2 * SBLOCK), (int64_t)h->n_cu));
        const size_t lds = multi_lds_bytes(sa.ntiles, sa.w.sample);
        static_assert(MULTI_SAMPLE_MAX == 2, "one instantiation");
        GPF_LAUNCH((k_search_multi_s<2>), dim3(gss), dim3(SBLOCK), lds, h->stream, sa);
        return;
    }
    const int gsr = (int)std::max<int64_t>(1, std::min<int64_t>((sa.n + 2 * SBLOCK - 1) / (2 * SBLOCK), (int64_t)h->n_cu));   // (k_search_multi strides by its own slots per lane)
    if (logg == 0)      GPF_LAUNCH((k_search_multi<0>), dim3(gsr), dim3(SBLOCK), multi_lds_bytes(sa.ntiles, 0), h->stream, sa);
    else if (logg == 1) GPF_LAUNCH((k_search_multi<1>), dim3(gsr), dim3(SBLOCK), multi_lds_bytes(sa.ntiles, 1), h->stream, sa);
    else                GPF_LAUNCH((k_search<0>), dim3(gsr), dim3(SBLOCK), search_lds_bytes(sa.ntiles, 1), h->stream, sa);
}

// the ancestors of a pending multinomial resample are wanted as an array after all (getters, views, rejuvenation, a second resample,
// an update that is not the plain propagate): the stand-alone search, as the resample itself would have run it
gpf_status finish_search(gpf_filter* h)
{
    if (!h->pending_search) return GPF_OK;
    h->pending_search = false;
    gpf_status s = timed(h, GPF_K_SEARCH, [&] { launch_multinomial_search(h, h->pend_sa); });
    if (s) return s;
    HIP_TRY(h, hipGetLastError());
    return GPF_OK;
}
// a pending lazy move is wanted as a state after all: the stand-alone k_move with the arguments and the epoch of its pf_rejuvenate! call
gpf_status finish_move(gpf_filter* h)
{
    if (!h->pending_move) return GPF_OK;
    h->pending_move = false;
    const bool fused_gather = h->pending_gather;
    const int grid = move_grid(h);
    const int n_iters = h->pm_iters;
    gpf_status s = timed(h, GPF_K_MOVE, [&] {
        if (h->pm_method == GPF_REJUVENATE_REWEIGHT) { DISPATCH_MODEL(h, (launch_move_t<MM, true>(h, grid, n_iters, h->pm_args, h->pm_epoch))); }
        else                                         { DISPATCH_MODEL(h, (launch_move_t<MM, false>(h, grid, n_iters, h->pm_args, h->pm_epoch))); }
    });
    if (s) return s;
    HIP_TRY(h, hipGetLastError());
    h->cur ^= 1;                                                 // (the epoch was consumed at the call)
    if (fused_gather) { h->pending_gather = false; h->pending_fill = false; h->max_valid = false; }
    if (h->pm_method == GPF_REJUVENATE_REWEIGHT) { h->raw_valid = false; h->raw_sum_valid = false; h->max_valid = true; }
    mutated(h);
    return GPF_OK;
}
// which models / sizes k_step_search covers: the key-table regime of the search (up to 2.5 M particles)
bool lazy_search_ok(const gpf_filter* h)
{
    const bool off = !h->lazy_search;                            // off by default: no faster than the two kernels (gpf_k_fused.hpp), gpf_set_lazy_search
    const int logg = multi_logg(h->ntiles);
    // (the ancestor ring sits behind the key table: both must fit the CU's LDS with the kernel's static words)
    return !off && !h->parent && !h->hist_on && h->cfg.n_global == h->n && logg >= 0 &&
           multi_lds_bytes(h->ntiles, logg) + 16 + FUSED_LDS_EXTRA + 1024 <= (size_t)160 * 1024;
}

gpf_status resample_impl(gpf_filter* h, int method, PrioView pv, int sort_particles, int check, int32_t* invalid, bool local = false)
{
    if (method != GPF_RESAMPLE_MULTINOMIAL && method != GPF_RESAMPLE_RESIDUAL && method != GPF_RESAMPLE_STRATIFIED && method != GPF_RESAMPLE_MULTINOMIAL_SORTED)
        return fail(h, GPF_ERR_UNKNOWN_METHOD, "Resampling method not recognized.");          // resample.jl:28
    if (h->cfg.n_global != h->n)
        return fail(h, GPF_ERR_STATE, "sharded filters resample through the shard-level API (sharded.py)");
    const bool sorted = method == GPF_RESAMPLE_STRATIFIED && sort_particles;
    const bool need_sync = check == GPF_CHECK_TRUE || invalid != nullptr;
    // only the multinomial search reads the offset levels: the scans of this call write them for it alone
    const bool need_off = method == GPF_RESAMPLE_MULTINOMIAL && (multi_logg(h->ntiles) >= 0 || multi_sample(h->ntiles) > 0);
    struct OffScope { gpf_filter* h; ~OffScope() { h->want_offsets = true; } } off_scope{h};
    h->want_offsets = need_off;
    h->offsets_hint = need_off;
    gpf_status s;
    if ((s = check_scan_timeout(h))) return s;                   // an earlier scan gave up: do not build on its CDF
    if ((s = materialize(h))) return s;                          // two resamples in a row: finish the first one
    // sortperm(log_priorities, rev=true)  (resample.jl:156-157)
    // with priorities the log-ML update needs the summary of the RAW weights (cdf[0] is overwritten later; only S, m matter).  First:
    // it may recompute the maximum slots for the raw weights, and from here on they must describe the priorities (sort keys, scan)
    if (pv.mode != 0 && (s = ensure_raw(h))) return s;
    bool sort_pending = false;                                   // the sort's verdict (k_sort_finish) is asked for after the search is enqueued
    if (sorted) {
        if ((s = ensure_max(h, pv, pv.mode == 0))) return s;     // (the coarse sort keys are distances from the maximum)
        if ((s = sort_desc_begin(h, pv, h->n, &sort_pending))) return s;
    }
    if (method == GPF_RESAMPLE_MULTINOMIAL_SORTED) {
        // the gamma total of every tile of SP_TILE slots: one lane per tile, as extra workgroups of the weight scan below;
        // beyond SP_DIRECT_TILES tiles one more small launch turns them into the tiles' starting points, else the merge kernel does
        // that for its own tile
        const int64_t ntl = (h->n + SP_TILE - 1) / SP_TILE;
        if (h->sp_cap < ntl + 1) {
            if (h->sp_g) { HIP_TRY(h, hipStreamSynchronize(h->stream)); (void)hipFree(h->sp_g); (void)hipFree(h->sp_vlo); h->sp_g = h->sp_vlo = nullptr; h->sp_cap = 0; }
            HIP_TRY(h, hipMalloc(&h->sp_g, (size_t)(ntl + 1) * sizeof(uint64_t)));
            HIP_TRY(h, hipMalloc(&h->sp_vlo, (size_t)(ntl + 1) * sizeof(uint64_t)));
            h->sp_cap = ntl + 1;
        }
        h->sp_job = SortedGammaJob{h->cfg.seed, h->sp_g, h->cfg.gid0, h->n, ntl, h->epoch, gamma_E(ntl), 0};
        h->sp_job_set = true;
    }
    struct SpScope { gpf_filter* h; ~SpScope() { h->sp_job_set = false; } } sp_scope{h};
    // safe_softmax(log_priorities) (resample.jl:54) and logsumexp(log_weights) (resample.jl:180)
    WSum* ws;
    bool published = false;                                      // the scan of THIS call publishes the flags to pinned memory
    // :residual right after an ESS / log-ML read (README.md:68-70): the summary is with the host (k_sum_host) and the residual scan needs
    // nothing else of the weight scan -- it converts the weights itself (k_scan_residual2<DIRECT>) and the weight scan is not run
    static const bool no_direct = getenv("GPF_RESIDUAL_DIRECT") && !strcmp(getenv("GPF_RESIDUAL_DIRECT"), "0");
    ResidDirect rdirect{};
    const bool resid_direct = !no_direct && method == GPF_RESAMPLE_RESIDUAL && pv.mode == 0 && !h->raw_valid && h->raw_sum_valid && h->sum_on_host;
    int direct_flags = 0;
    if (resid_direct) {
        rdirect = ResidDirect{h->lw, h->sum_cache.m, h->sum_cache.flags, h->K, h->sum_cache.S, &h->sc->raw};
        direct_flags = h->sum_cache.flags;
        ws = &h->sc->raw;
    } else
    if (pv.mode == 0) {
        ws = &h->sc->raw;
        if (!h->raw_valid || sorted || (need_off && !h->ch0_offsets)) {
            if ((s = summarize(h, pv, ws, true, sorted ? h->order : nullptr, true, false, need_sync, sorted))) return s;
            published = need_sync;
        }
    } else {
        h->want_offsets = need_off;
        ws = &h->sc->prio;
        if ((s = summarize(h, pv, ws, true, sorted ? h->order : nullptr, false, false, need_sync, sorted))) return s;
        published = need_sync;
    }
    h->raw_valid = false; h->raw_sum_valid = false;                                        // cdf[0] no longer the plain raw CDF / lw about to change
    h->raw_q_folded = false;
    if (need_sync) {
        int flags;
        if (published) {
            // safe_softmax's flags are known when the scan STARTS (it folds the per-block maxima first): poll the ticket; the
            // scan keeps running and the search below is enqueued behind it without a gap
            if ((s = wait_ticket(h, h->h_flags + 1, h->flag_ticket, "weight scan flags"))) return s;
            flags = (int)h->h_flags[0];
        } else if (resid_direct) flags = direct_flags;           // (known since the getter)
        else {
            if ((s = fetch_scalars(h))) return s;
            flags = (pv.mode == 0 ? h->h_sc->raw : h->h_sc->prio).flags;
        }
        const bool inv = flags != 0;
        if (invalid) *invalid = inv ? 1 : 0;
        if (flags & (FLAG_NAN | FLAG_POSINF)) return fail(h, GPF_ERR_INVALID_WEIGHTS, "Invalid weights (NaN).");
        if (check == GPF_CHECK_TRUE && inv) return fail(h, GPF_ERR_INVALID_WEIGHTS, "Invalid weights.");   // resample.jl:55
    }
    // ancestors (+ update_lml_est!, resample.jl:57,178-182, inside the search kernel)
    SearchArgs sa{};
    sa.w = levels(h, 0); sa.c = levels(h, 0); sa.ntiles = h->ntiles;
    sa.order = sorted ? h->order : nullptr; sa.sc = h->sc; sa.ws = ws; sa.raw = &h->sc->raw; sa.n = h->n; sa.n_cells = h->n;
    sa.n_global = h->cfg.n_global; sa.gid0 = h->cfg.gid0; sa.seed = h->cfg.seed; sa.epoch = h->epoch;
    sa.K = h->K; sa.logN = h->logN; sa.anc = h->anc; sa.invN = 1.0 / (double)h->cfg.n_global;
    sa.update_lml = local ? 2 : (h->parent ? 0 : 1);             // sub-states do not track the estimate (resample.jl:185-187)

    if (method == GPF_RESAMPLE_RESIDUAL) {
        static const bool head_in_search = getenv("GPF_RESIDUAL_HEAD") && !strcmp(getenv("GPF_RESIDUAL_HEAD"), "search");   // (A/B measurements)
        if ((s = residual_scans(h, ws, h->cfg.n_global, head_in_search ? nullptr : h->anc, resid_direct ? &rdirect : nullptr))) return s;
        sa.w = levels(h, 2); sa.c = levels(h, 1);
        sa.head_done = head_in_search ? 0 : 1;
    }
    if (method == GPF_RESAMPLE_MULTINOMIAL_SORTED) {
        const int64_t ntl = (h->n + SP_TILE - 1) / SP_TILE;
        if (h->sp_job_set) {                                     // no weight scan ran in this call (the CDF of an earlier getter is reused): a launch of its own
            h->sp_job_set = false;
            GPF_LAUNCH(k_sorted_gammas, dim3((unsigned)((ntl + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, h->stream, h->sp_job);
        }
        sa.sp_g = h->sp_g; sa.sp_vlo = nullptr;
        if (ntl > SP_DIRECT_TILES) {
            GPF_LAUNCH(k_sorted_tiles, dim3(1), dim3(STILES_BLOCK), 0, h->stream, h->sp_g, ntl, h->sp_vlo);
            sa.sp_vlo = h->sp_vlo;
        }
    }
    const int64_t nt = method == GPF_RESAMPLE_RESIDUAL && !sa.head_done ? 2 : 1;   // (top tables the search keeps in LDS: its shape depends on their number)
    const size_t lds = search_lds_bytes(h->ntiles, (int)nt);
    // every block first copies the top level of the CDF into LDS: keep the grid small (persistent blocks)
    // one 1024-thread workgroup per CU, two slots per lane and iteration
    const int gsr = (int)std::max<int64_t>(1, std::min<int64_t>((h->n + 2 * SBLOCK - 1) / (2 * SBLOCK), (int64_t)h->n_cu * SEARCH_BLOCKS_PER_CU));
    auto search = [&]() {
        return timed(h, GPF_K_SEARCH, [&] {
            switch (method) {
                case GPF_RESAMPLE_MULTINOMIAL: launch_multinomial_search(h, sa); break;
                case GPF_RESAMPLE_RESIDUAL:    GPF_LAUNCH((k_search<1>), dim3(gsr), dim3(SBLOCK), lds, h->stream, sa); break;
                case GPF_RESAMPLE_MULTINOMIAL_SORTED:   // sorted uniforms: the same streaming merge (the spacing sums were enqueued above)
                    GPF_LAUNCH((k_search_strat<true>), dim3((unsigned)((h->n + MJB - 1) / MJB)), dim3(MBLOCK), 0, h->stream, sa); break;
                default:                       // monotone targets: a streaming merge, MJB_STRAT slots per workgroup
                    GPF_LAUNCH((k_search_strat<false>), dim3((unsigned)((h->n + MJB_STRAT - 1) / MJB_STRAT)), dim3(MBLOCK), 0, h->stream, sa); break;
            }
        });
    };
    // lazy search: a plain multinomial resample of a whole filter leaves its search to the pf_update! that follows (k_step_search)
    const bool lazy = method == GPF_RESAMPLE_MULTINOMIAL && pv.mode == 0 && !local && lazy_search_ok(h) && sa.w.off16 && sa.w.sample == 0;
    if (lazy) { h->pend_sa = sa; h->pending_search = true; }
    else if ((s = search())) return s;
    if (sort_pending) {
        // the scan and the search above ran behind the sort without a host wait; if the finish met a run it could not order (equal
        // or nearly equal priorities) they worked on a wrong order: eight passes over the full key, then both again.  The weight
        // sums are order-independent (integers): the log-ML update of the first search stands.
        bool flagged = false;
        if ((s = sort_desc_flagged(h, &flagged))) return s;
        if (flagged) {
            if ((s = sort_passes(h, pv, h->n, false))) return s;
            h->want_offsets = need_off;
            if ((s = summarize(h, pv, ws, true, h->order, pv.mode == 0, false, false, true))) return s;
            sa.w = levels(h, 0); sa.c = levels(h, 0); sa.update_lml = 0;
            if ((s = search())) return s;
        }
    }
    if ((s = hist_on_resample(h))) return s;
    if (h->parent) {
        // sub-state (resample.jl:205-218): eager gather; weights keep the block's total mass
        s = timed(h, GPF_K_GATHER, [&] { launch_gather(h, pv, pv.mode == 0 ? h->lw : h->lws); });
        if (s) return s;
        h->cur ^= 1;
        if (pv.mode == 0) {
            GPF_LAUNCH(k_view_fill_weights, dim3(grid_for(h, h->n, 8)), dim3(BLOCK), 0, h->stream, h->lw, h->n, &h->sc->raw, h->K, h->logN);
        } else {
            PrioView post{h->lws, nullptr, 0.0, 0};
            if ((s = summarize(h, post, &h->sc->post, false, nullptr, false))) return s;
            GPF_LAUNCH(k_view_apply_post, dim3(grid_for(h, h->n, 8)), dim3(BLOCK), 0, h->stream, h->sc, h->K, h->lws, h->lw, h->n);
        }
        h->max_valid = false;
        HIP_TRY(h, hipGetLastError());
        h->epoch += 1;
        mutated(h);
        return view_exit(h);
    }
    if (pv.mode == 0) {
        // new_traces .= view(traces, parents) is deferred: the next pf_update! reads rows through anc (fused
        // gather), any other consumer calls materialize().  Log-weights are 0 (resample.jl:195).
        h->pending_gather = true;
        h->pending_fill = local;
        h->max_valid = false;
    } else {
        // gather + update_weights! with priorities (resample.jl:60,198-200), update_refs! (utils.jl:10-15)
        s = timed(h, GPF_K_GATHER, [&] { launch_gather(h, pv, h->lws); });
        if (s) return s;
        h->cur ^= 1;
        PrioView post{h->lws, nullptr, 0.0, 0};
        if ((s = summarize(h, post, &h->sc->post, false, nullptr, false))) return s;
        GPF_LAUNCH(k_apply_post, dim3(grid_for(h, h->n, 8)), dim3(BLOCK), 0, h->stream, h->sc, h->K, h->logN, h->lws, h->lw, h->n);
        h->max_valid = false;
    }
    HIP_TRY(h, hipGetLastError());
    h->epoch += 1;
    mutated(h);
    return GPF_OK;
}

} // namespace

// =================================================================================== C ABI
template <int METHOD>
static void launch_push(gpf_filter* h, const PushArgs& a, int grid, size_t lds, const CdfLevels& lw_, const CdfLevels& lc_, int64_t capacity, double* out)
{
    switch (h->W) {
        case 2: GPF_LAUNCH((k_push<METHOD, 2>), dim3(grid), dim3(SBLOCK), lds, h->stream, a, lw_, lc_, h->n, h->ntiles, h->cfg.gid0, h->rows[h->cur], capacity, out); break;
        case 4: GPF_LAUNCH((k_push<METHOD, 4>), dim3(grid), dim3(SBLOCK), lds, h->stream, a, lw_, lc_, h->n, h->ntiles, h->cfg.gid0, h->rows[h->cur], capacity, out); break;
        case 8: GPF_LAUNCH((k_push<METHOD, 8>), dim3(grid), dim3(SBLOCK), lds, h->stream, a, lw_, lc_, h->n, h->ntiles, h->cfg.gid0, h->rows[h->cur], capacity, out); break;
    }
}

static void launch_push_multi(gpf_filter* h, const PushArgs& a, int grid, const CdfLevels& lw_, int64_t capacity, double* out)
{
#define GPF_PN(LG, WW) GPF_LAUNCH((k_push_multi<LG, WW>), dim3(grid), dim3(SBLOCK), multi_lds_bytes(h->ntiles, LG), h->stream, a, lw_, h->n, h->ntiles, h->cfg.gid0, h->rows[h->cur], capacity, out)
    if (lw_.logg == 0) { switch (h->W) { case 2: GPF_PN(0, 2); break; case 4: GPF_PN(0, 4); break; case 8: GPF_PN(0, 8); break; } }
    else               { switch (h->W) { case 2: GPF_PN(1, 2); break; case 4: GPF_PN(1, 4); break; case 8: GPF_PN(1, 8); break; } }
#undef GPF_PN
}

extern "C" {

int gpf_abi_version(void) { return GPF_ABI_VERSION; }

const char* gpf_last_error(gpf_handle h) { return h ? h->err.c_str() : g_err.c_str(); }

gpf_status gpf_create(const gpf_config* cfg, gpf_handle* out)
{
    if (!cfg || !out) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null config/out");
    *out = nullptr;
    if (cfg->abi_version != GPF_ABI_VERSION) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "ABI version mismatch");
    const int d = model_dim(cfg->model);
    if (d == 0) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "unknown model id");
    if (cfg->n_params < 0 || cfg->n_params > MAX_PARAMS || (cfg->n_params > 0 && !cfg->params))
        return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "bad parameter vector");
    if (cfg->n_particles < 1 || cfg->n_global < cfg->n_particles || cfg->gid0 < 0 ||
        cfg->gid0 + cfg->n_particles > cfg->n_global || cfg->n_global >= ((int64_t)1 << 31))
        return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "bad particle counts (need 1 <= n <= n_global < 2^31)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, GPF_ERR_NO_DEVICE, "no HIP device: libgpf_hip has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "bad device ordinal");

    gpf_filter* h = new gpf_filter();
    h->lazy_search = getenv("GPF_LAZY_SEARCH") && !strcmp(getenv("GPF_LAZY_SEARCH"), "1");
    h->lazy_move = !(getenv("GPF_LAZY_MOVE") && !strcmp(getenv("GPF_LAZY_MOVE"), "0"));
    h->cfg = *cfg;
    h->cfg.params = nullptr;
    for (int i = 0; i < cfg->n_params; ++i) h->args.P[i] = cfg->params[i];
    h->args.gstride = 1;
    h->d = d;
    h->W = row_width(d, cfg->keep_prev != 0);
    h->n = cfg->n_particles;
    gpf_status st = GPF_OK;
    auto body = [&]() -> gpf_status {
        HIP_TRY(h, hipSetDevice(cfg->device));
        hipDeviceProp_t prop;
        HIP_TRY(h, hipGetDeviceProperties(&prop, cfg->device));
        h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (cfg->stream) { h->stream = (hipStream_t)cfg->stream; h->own_stream = false; }
        else { HIP_TRY(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)); h->own_stream = true; }
        { gpf_status a_ = alloc_particle_buffers(h); if (a_) return a_; }
        const size_t n = (size_t)h->n, rb = n * (size_t)h->W * sizeof(double);
        for (int b = 0; b < 2; ++b) {
            HIP_TRY(h, hipMalloc(&h->mslots[b], (size_t)MAX_SLOTS * SLOT_WORDS * sizeof(unsigned long long)));
            HIP_TRY(h, hipMemsetAsync(h->mslots[b], 0, (size_t)MAX_SLOTS * SLOT_WORDS * sizeof(unsigned long long), h->stream));
        }
        HIP_TRY(h, hipMalloc(&h->blockQ, (size_t)4 * 8 * h->n_cu * sizeof(uint64_t) + 64));   // (4 limbs per scan workgroup, <= 8 workgroups per CU)
        // (zeroed: the ESS scan recognises this launch's partials by a tag in their top bits -- recycled memory may hold another filter's)
        HIP_TRY(h, hipMemsetAsync(h->blockQ, 0, (size_t)4 * 8 * h->n_cu * sizeof(uint64_t) + 64, h->stream));
        HIP_TRY(h, hipMalloc(&h->partial, MAX_PARTIALS * sizeof(double)));
        HIP_TRY(h, hipMalloc(&h->acc_part, MAX_PARTIALS * sizeof(unsigned long long)));
        HIP_TRY(h, hipMalloc(&h->dscal, 4 * sizeof(double)));
        HIP_TRY(h, hipMalloc(&h->sc, sizeof(Scalars)));
        HIP_TRY(h, hipHostMalloc(&h->h_sc, sizeof(Scalars)));
        HIP_TRY(h, hipHostMalloc(&h->h_timeout, sizeof(int32_t)));
        *h->h_timeout = 0;
        {   // resident scan workgroups per CU: the smallest answer over the scan kernels, never more than 2 (what the tile
            // schedule was tuned for), one fewer than the API says when it says more (the API can over-count by one)
            int nb = 2;
            const void* scans[] = {reinterpret_cast<const void*>(&k_scan<InFixQ, 1>), reinterpret_cast<const void*>(&k_scan<InFixQ, 2>),
                                   reinterpret_cast<const void*>(&k_scan<InFixQ, 3>), reinterpret_cast<const void*>(&k_scan<InFixQ, 4>),
                                   reinterpret_cast<const void*>(&k_scan<InOptimal, 0>), reinterpret_cast<const void*>(&k_scan_residual2<false>),
                                   reinterpret_cast<const void*>(&k_scan_residual2<true>)};
            for (const void* f : scans) {
                int q = 0;
                HIP_TRY(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, f, SCAN_BLOCK, 0));
                if (q < 1) return fail(h, GPF_ERR_HIP, "a scan kernel cannot be resident on this device");
                nb = std::min(nb, q > 2 ? q - 1 : q);
            }
            h->scan_blocks_per_cu = std::max(1, std::min(nb, 2));
            int wb = 64;
            for (int k = 0; k < 4; ++k) {
                int q = 0;
                HIP_TRY(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, scans[k], SCAN_BLOCK, 0));
                wb = std::min(wb, q > 2 ? q - 1 : q);
            }
            // (measured at 2 x 10^6 particles, 977 tiles: 2 per CU = two rounds 23.2 us, 4 per CU = one round 21.3 us; GPF_WSCAN_BLOCKS for A/B)
            static const int wscan_max = getenv("GPF_WSCAN_BLOCKS") ? atoi(getenv("GPF_WSCAN_BLOCKS")) : 4;
            h->wscan_blocks_per_cu = std::max(1, std::min(std::min(wb, wscan_max), 8));
        }
        HIP_TRY(h, hipMemsetAsync(h->sc, 0, sizeof(Scalars), h->stream));
        HIP_TRY(h, hipMemsetAsync(h->lw, 0, n * sizeof(double), h->stream));
        HIP_TRY(h, hipMemsetAsync(h->rows[0], 0, rb, h->stream));
        GPF_LAUNCH(k_iota, dim3(grid_for(h, h->n, 8)), dim3(BLOCK), 0, h->stream, h->anc, h->n);   // parents = 1:N
        // k_search keeps up to LDS_TILE_TABLE top-level entries (64 KiB) + 32 KiB of cooperation strips in LDS
        const int max_dyn = (int)((lds_pad(LDS_TILE_TABLE) + 4) * sizeof(uint64_t));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_search<0>), hipFuncAttributeMaxDynamicSharedMemorySize, max_dyn));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_search<1>), hipFuncAttributeMaxDynamicSharedMemorySize, max_dyn));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_search<3>), hipFuncAttributeMaxDynamicSharedMemorySize, max_dyn));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_search_multi<0>), hipFuncAttributeMaxDynamicSharedMemorySize, MULTI_LDS_BUDGET));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_search_multi<1>), hipFuncAttributeMaxDynamicSharedMemorySize, MULTI_LDS_BUDGET));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_search_multi_s<2>), hipFuncAttributeMaxDynamicSharedMemorySize, MULTI_LDS_BUDGET));
#define GPF_PUSH_ATTR(M, W) HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_push<M, W>), hipFuncAttributeMaxDynamicSharedMemorySize, max_dyn))
        GPF_PUSH_ATTR(0, 2); GPF_PUSH_ATTR(0, 4); GPF_PUSH_ATTR(0, 8);
        GPF_PUSH_ATTR(1, 2); GPF_PUSH_ATTR(1, 4); GPF_PUSH_ATTR(1, 8);
#define GPF_PN_ATTR(LG, W) HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_push_multi<LG, W>), hipFuncAttributeMaxDynamicSharedMemorySize, MULTI_LDS_BUDGET))
        GPF_PN_ATTR(0, 2); GPF_PN_ATTR(0, 4); GPF_PN_ATTR(0, 8); GPF_PN_ATTR(1, 2); GPF_PN_ATTR(1, 4); GPF_PN_ATTR(1, 8);
#undef GPF_PUSH_ATTR
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        return GPF_OK;
    };
    st = body();
    if (st != GPF_OK) { g_err = h->err; gpf_destroy(h); return st; }
    *out = h;
    return GPF_OK;
}

gpf_status gpf_destroy(gpf_handle h)
{
    if (!h) return GPF_OK;
    hipSetDevice(h->cfg.device);
    if (h->stream) hipStreamSynchronize(h->stream);
    for (auto& t : h->timers) for (auto& e : t.ev) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    h->pending_packed = false;                                   // the filter goes away: nothing to scatter a deferred commit into
    for (gpf_filter* v : h->blk_views) gpf_destroy(v);
    h->blk_views.clear();
    gpf_comm_destroy(h);
    hist_clear(h);
    if (h->hist_dev_maps) (void)hipFree(h->hist_dev_maps);
    if (h->parent) { h->rows[0] = h->rows[1] = nullptr; h->lw = nullptr; h->anc = nullptr; }   // aliases of the parent's buffers (or of the compact copies below)
    for (void* q : {(void*)h->vrows[0], (void*)h->vrows[1], (void*)h->vlw, (void*)h->vanc, (void*)h->vidx, (void*)h->vgid}) if (q) (void)hipFree(q);
    { Bufs b = take_particle_buffers(h); free_bufs(b); }
    void* bufs[] = {h->mslots[0], h->mslots[1], h->blockQ, h->partial, h->dscal, h->sc, h->push_stage, h->shard_counts, h->shard_plan, h->tree_buf, h->acc_part,
                    h->pull_req, h->pull_counts, h->pull_pc, h->pull_pc_all, h->blk_words, h->blk_mask, h->blk_stats, h->blk_obs};
    for (void* b : bufs) if (b) hipFree(b);
    if (h->h_sc) hipHostFree(h->h_sc);
    if (h->h_sc_ticket) hipHostFree(h->h_sc_ticket);
    if (h->h_shard_counts) hipHostFree(h->h_shard_counts);
    if (h->h_pull_pc_all) hipHostFree(h->h_pull_pc_all);
    if (h->h_qpub) hipHostFree(h->h_qpub);
    if (h->h_spart) hipHostFree(h->h_spart);
    if (h->sp_g) { (void)hipFree(h->sp_g); (void)hipFree(h->sp_vlo); }
    if (h->sum_part) (void)hipFree(h->sum_part);
    for (int k = 0; k < gpf_filter::BLK_STAGE; ++k) if (h->h_blk_obs[k]) hipHostFree(h->h_blk_obs[k]);
    if (h->h_blk_done) hipHostFree(h->h_blk_done);
    if (h->blk_stage_counter) (void)hipFree(h->blk_stage_counter);
    if (h->h_flags) hipHostFree(h->h_flags);
    if (h->h_sort_flag) hipHostFree(h->h_sort_flag);
    if (h->h_timeout) hipHostFree(h->h_timeout);
    if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
    delete h;
    return GPF_OK;
}

gpf_status gpf_set_lazy_search(gpf_handle h, int32_t enable)
{
    if (!h) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null handle");
    if (!enable) { gpf_status s = finish_search(h); if (s) return s; }
    h->lazy_search = enable != 0;
    return GPF_OK;
}

gpf_status gpf_synchronize(gpf_handle h)
{
    // (work that was left for a later call to pick up is enqueued now: a timed loop that ends in pf_rejuvenate! pays for its move)
    if (h && h->pending_move) { gpf_status ms = finish_move(h); if (ms) return ms; }
    if (!h) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null handle");
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return check_scan_timeout(h);
}

static gpf_status initialize_impl(gpf_handle h, const double* obs, int32_t n_obs, int prop)
{
    if (!h) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null handle");
    if (prop == 1 && !model_has_proposal(h->cfg.model)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "this model has no native proposal");
    if (prop == 2 && !model_has_strata(h->cfg.model)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "this model has no discrete latent to stratify over");
    if (prop == 3 && h->cfg.model != MODEL_LINE) return fail(h, GPF_ERR_INVALID_ARGUMENT, "stratified initialisation with a native proposal: line_model only");
    if (h->parent) return fail(h, GPF_ERR_STATE, "gpf_initialize on a sub-state view");
    h->generation += 1;
    gpf_status s = set_obs(h, obs, n_obs);
    if (s) return s;
    h->blk_obs_size = 0;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if ((s = hist_begin_step(h, true))) return s;
    const int grid = step_grid(h);
    s = timed(h, GPF_K_STEP, [&] {
        if (prop == 1)      { DISPATCH_MODEL(h, (launch_init_t<MM, 1>(h, grid))); }
        else if (prop == 2) { DISPATCH_MODEL(h, (launch_init_t<MM, 2>(h, grid))); }
        else if (prop == 3) { DISPATCH_MODEL(h, (launch_init_t<MM, 3>(h, grid))); }
        else                { DISPATCH_MODEL(h, (launch_init_t<MM, 0>(h, grid))); }
    });
    if (s) return s;
    h->pending_gather = false; h->pending_fill = false; h->pending_search = false; h->pending_move = false;
    h->pending_packed = false;
    h->max_valid = true;
    GPF_LAUNCH(k_iota, dim3(grid), dim3(BLOCK), 0, h->stream, h->anc, h->n);            // parents = 1:N (initialize.jl:43)
    HIP_TRY(h, hipMemsetAsync(&h->sc->lml_est, 0, sizeof(double), h->stream));                   // log_ml_est = 0.
    HIP_TRY(h, hipGetLastError());
    h->epoch += 1;
    h->initialized = true;
    h->has_prev = false;
    h->raw_valid = false; h->raw_sum_valid = false;
    mutated(h);
    return GPF_OK;
}

gpf_status gpf_initialize(gpf_handle h, const double* obs, int32_t n_obs) { return initialize_impl(h, obs, n_obs, 0); }
// the native proposal a model has: the locally optimal one (lgssm2), the reference tests' fixed proposals (line_model)
static bool proposal_matches(gpf_handle h, int32_t proposal)
{
    if (!h) return true;                                          // reported by the callee
    if (proposal == GPF_PROPOSAL_LOCALLY_OPTIMAL) return h->cfg.model != MODEL_LINE;
    if (proposal == GPF_PROPOSAL_LINE_FIXED) return h->cfg.model == MODEL_LINE;
    return false;
}
gpf_status gpf_initialize_proposal(gpf_handle h, const double* obs, int32_t n_obs, int32_t proposal)
{
    if (!proposal_matches(h, proposal)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "unknown proposal id for this model");
    return initialize_impl(h, obs, n_obs, 1);
}

static gpf_status update_impl(gpf_handle h, const double* obs, int32_t n_obs, int prop)
{
    gpf_status s = check_ready(h, prop == 0);                    // (the plain propagate carries a pending lazy move)
    if (s) return s;
    if (prop == 1 && !model_has_proposal(h->cfg.model)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "this model has no native proposal");
    if (prop == 2 && !model_has_strata(h->cfg.model)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "this model has no discrete latent to stratify over");
    if ((s = set_obs(h, obs, n_obs))) return s;
    h->blk_obs_size = 0;                                         // one observation for all particles again
    if ((s = hist_begin_step(h, false))) return s;
    if (prop != 0 && (s = finish_search(h))) return s;           // (only the plain propagate carries a pending search)
    const int grid = step_grid(h);
    const bool keep = h->cfg.keep_prev != 0;
    if (h->pending_move) {
        // gather (if pending) -> move -> propagate in one launch (k_move_step): the old observation and epoch travel with the move
        const bool rw = h->pm_method == GPF_REJUVENATE_REWEIGHT;
        s = timed(h, GPF_K_STEP, [&] {
            if (rw) { DISPATCH_MODEL(h, (launch_move_step_t<MM, true>(h, grid))); }
            else    { DISPATCH_MODEL(h, (launch_move_step_t<MM, false>(h, grid))); }
        });
        if (s) return s;
        HIP_TRY(h, hipGetLastError());
        h->pending_move = false;
        h->pending_gather = false; h->pending_fill = false;
        h->max_valid = true;
        h->cur ^= 1;                // (read rows[cur], wrote the other buffer once: the move's and the update's swaps cancel to one)
        h->epoch += 1;
        h->has_prev = true;
        h->raw_valid = false; h->raw_sum_valid = false;
        mutated(h);
        return GPF_OK;
    }
    s = timed(h, GPF_K_STEP, [&] {
        if (prop == 1) {
            if (keep) { DISPATCH_MODEL(h, (launch_step_t<MM, true, 1>(h, grid))); }
            else      { DISPATCH_MODEL(h, (launch_step_t<MM, false, 1>(h, grid))); }
        } else if (prop == 2) {
            if (keep) { DISPATCH_MODEL(h, (launch_step_t<MM, true, 2>(h, grid))); }
            else      { DISPATCH_MODEL(h, (launch_step_t<MM, false, 2>(h, grid))); }
        } else {
            if (keep) { DISPATCH_MODEL(h, (launch_step_t<MM, true>(h, grid))); }
            else      { DISPATCH_MODEL(h, (launch_step_t<MM, false>(h, grid))); }
        }
    });
    if (s) return s;
    HIP_TRY(h, hipGetLastError());
    h->pending_gather = false; h->pending_fill = false;      // a pending resample gather was fused into this step
    h->pending_packed = false; h->pend_own = false;      // ... or a pending sharded commit
    h->max_valid = true;
    h->cur ^= 1;                    // update_refs! (utils.jl:10-15)
    h->epoch += 1;
    h->has_prev = true;
    h->raw_valid = false; h->raw_sum_valid = false;
    mutated(h);
    return view_exit(h);            // sub-state: copy back (utils.jl:17-20)
}

gpf_status gpf_update(gpf_handle h, const double* obs, int32_t n_obs) { return update_impl(h, obs, n_obs, 0); }
gpf_status gpf_update_proposal(gpf_handle h, const double* obs, int32_t n_obs, int32_t proposal)
{
    if (!proposal_matches(h, proposal)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "unknown proposal id for this model");
    return update_impl(h, obs, n_obs, 1);
}

// stratified initialisation / update: the strata are values of the model's discrete latent
static gpf_status set_strata(gpf_handle h, const double* values, int32_t n_strata, int32_t interleaved)
{
    if (!h) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null handle");
    if (!values || n_strata < 1 || n_strata > MAX_STRATA) return fail(h, GPF_ERR_INVALID_ARGUMENT, "need 1..8 strata");
    if (h->cfg.n_global != h->n) return fail(h, GPF_ERR_STATE, "stratified initialisation / update of a sharded filter is not supported");
    for (int k = 0; k < MAX_STRATA; ++k) h->args.strata[k] = k < n_strata ? values[k] : 0.0;
    h->args.n_strata = n_strata; h->args.interleaved = interleaved != 0;
    h->args.logK = log_((double)n_strata);
    return GPF_OK;
}
gpf_status gpf_initialize_strata(gpf_handle h, const double* obs, int32_t n_obs, const double* values, int32_t n_strata, int32_t interleaved)
{
    gpf_status s = set_strata(h, values, n_strata, interleaved);
    return s ? s : initialize_impl(h, obs, n_obs, 2);
}
gpf_status gpf_initialize_strata_proposal(gpf_handle h, const double* obs, int32_t n_obs, const double* values, int32_t n_strata, int32_t interleaved,
                                          int32_t proposal)
{
    if (!proposal_matches(h, proposal) || proposal != GPF_PROPOSAL_LINE_FIXED) return fail(h, GPF_ERR_INVALID_ARGUMENT, "unknown proposal id for this model");
    gpf_status s = set_strata(h, values, n_strata, interleaved);
    return s ? s : initialize_impl(h, obs, n_obs, 3);
}
gpf_status gpf_update_strata(gpf_handle h, const double* obs, int32_t n_obs, const double* values, int32_t n_strata, int32_t interleaved)
{
    gpf_status s = set_strata(h, values, n_strata, interleaved);
    return s ? s : update_impl(h, obs, n_obs, 2);
}

gpf_status gpf_resample(gpf_handle h, int32_t method, double priority_alpha, int32_t sort_particles, int32_t check,
                        int32_t* invalid)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    PrioView pv = raw_view(h);
    if (priority_alpha == priority_alpha) { pv.alpha = priority_alpha; pv.mode = 1; }
    return resample_impl(h, method, pv, sort_particles, check, invalid);
}

// pf_resample!(state[1:n], method) on a whole filter or shard (src/resample.jl:185-187,205-218) without the view's copies: it
// normalises over its OWN n particles (strata, fixed-point scale and log n of n, not of n_global), leaves log_ml_est alone and
// every particle keeps the log-weight logsumexp - log n.  The gather stays deferred like gpf_resample's.
gpf_status gpf_resample_local(gpf_handle h, int32_t method, int32_t sort_particles, int32_t check, int32_t* invalid)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (h->parent) return fail(h, GPF_ERR_STATE, "gpf_resample_local on a sub-state view: resample the view itself");
    if ((s = materialize(h))) return s;
    struct Scope {                                               // the filter as its own population for the duration of the call
        gpf_filter* h; int K; double logN; int64_t ng;
        explicit Scope(gpf_filter* f) : h(f), K(f->K), logN(f->logN), ng(f->cfg.n_global)
        { h->K = fix_K(h->n); h->logN = log_((double)h->n); h->cfg.n_global = h->n; h->raw_valid = false; h->raw_sum_valid = false; h->raw_has_q = false; h->raw_q_folded = false; }
        ~Scope() { h->K = K; h->logN = logN; h->cfg.n_global = ng; h->raw_valid = false; h->raw_sum_valid = false; h->raw_has_q = false; h->raw_q_folded = false; }
    } scope(h);
    return resample_impl(h, method, raw_view(h), sort_particles, check, invalid, true);
}

// ------------------------------------------------------------------ block-wise resampling: many small filters in one launch (K11)
static gpf_status block_buffers(gpf_filter* h, int64_t nblocks)
{
    if (!h->blk_words) HIP_TRY(h, hipMalloc(&h->blk_words, 2 * sizeof(int32_t)));
    if (h->blk_cap < nblocks) {
        if (h->blk_mask) { HIP_TRY(h, hipStreamSynchronize(h->stream)); (void)hipFree(h->blk_mask); (void)hipFree(h->blk_stats); h->blk_mask = nullptr; h->blk_stats = nullptr; h->blk_cap = 0; }
        h->blk_last = 0;                                         // the new mask is uninitialised: no block resample to refer to
        HIP_TRY(h, hipMalloc(&h->blk_mask, (size_t)nblocks * sizeof(int32_t)));
        HIP_TRY(h, hipMalloc(&h->blk_stats, (size_t)nblocks * 2 * sizeof(double)));
        h->blk_cap = nblocks;
    }
    return GPF_OK;
}
static gpf_status block_checks(gpf_handle h, int64_t block_size, const char* who)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (h->parent) return fail(h, GPF_ERR_STATE, std::string(who) + " on a sub-state view: call it on the filter");
    if (h->cfg.n_global != h->n) return fail(h, GPF_ERR_STATE, std::string(who) + " on a shard of a sharded filter");
    if (block_size < 1) return fail(h, GPF_ERR_INVALID_ARGUMENT, "block_size < 1");
    return GPF_OK;
}
// Blocks of more than BLK_MAX = 2048 particles do not fit the one-workgroup-per-block kernels (gpf_k_block.hpp keeps a block's weights, CDF
// and order in LDS).  Their loop over sub-states (for b in blocks; pf_resample!(state[b], ...); end -- test/resample.jl:130-162 has no
// size limit) runs on the host over view handles of the blocks, with the full-size kernels: the same results as the views give, the
// same single epoch for all blocks, no size cliff.  At these sizes a block fills the chip by itself.
static gpf_status big_block_views(gpf_filter* h, int64_t block_size)
{
    const int64_t nblocks = (h->n + block_size - 1) / block_size;
    if (h->blk_views_size == block_size && h->blk_views_gen == h->generation && (int64_t)h->blk_views.size() == nblocks) return GPF_OK;
    for (gpf_filter* v : h->blk_views) gpf_destroy(v);
    h->blk_views.clear();
    for (int64_t b = 0; b < nblocks; ++b) {
        gpf_handle v = nullptr;
        const int64_t start = b * block_size, cnt = std::min(block_size, h->n - start);
        gpf_status s = gpf_view_create(h, start, cnt, &v);
        if (s) return s;
        h->blk_views.push_back(v);
    }
    h->blk_views_size = block_size; h->blk_views_gen = h->generation;
    return GPF_OK;
}
static gpf_status resample_big_blocks(gpf_handle h, int32_t method, int64_t block_size, double priority_alpha, int32_t sort_particles,
                                      double ess_frac, int32_t check, int32_t* invalid, int64_t* n_resampled)
{
    gpf_status s = big_block_views(h, block_size);
    if (s) return s;
    const int64_t nblocks = (int64_t)h->blk_views.size();
    if ((s = block_buffers(h, nblocks))) return s;
    const uint32_t E = h->epoch;                                 // every block resamples under the call's ONE epoch (like the batched kernel)
    std::vector<int32_t> words((size_t)nblocks, 0);
    bool any_invalid = false, any_nan = false, any_neginf_err = false;
    int64_t count = 0;
    const bool gate = ess_frac == ess_frac && ess_frac >= 0.0;
    for (int64_t b = 0; b < nblocks; ++b) {
        gpf_filter* v = h->blk_views[(size_t)b];
        h->epoch = E;
        if (gate) {
            double ess = 0.0;
            if ((s = gpf_effective_sample_size(v, &ess))) { h->err = v->err; h->epoch = E; return s; }
            if (!(ess < ess_frac * (double)v->n)) continue;      // (an invalid block: ESS NaN -- it does not resample, nothing is reported)
        }
        int32_t inv = 0;
        s = gpf_resample(v, method, priority_alpha, sort_particles, check == GPF_CHECK_TRUE ? GPF_CHECK_TRUE : GPF_CHECK_WARN, &inv);
        if (s == GPF_ERR_INVALID_WEIGHTS) {                      // the block is left as it stands; the others go on
            any_invalid = true;
            const bool nan_block = v->err.find("NaN") != std::string::npos;
            if (nan_block) any_nan = true; else any_neginf_err = true;
            words[(size_t)b] = (nan_block ? FLAG_NAN : FLAG_ALL_NEGINF) << 8;      // (the word layout of the batched kernel: flags << 8 | resampled)
            continue;
        }
        if (s) { h->err = v->err; h->epoch = E; return s; }
        if (inv) { any_invalid = true; words[(size_t)b] |= FLAG_ALL_NEGINF << 8; }
        words[(size_t)b] |= 1;
        ++count;
    }
    h->epoch = E + 1;
    HIP_TRY(h, hipMemcpyAsync(h->blk_mask, words.data(), (size_t)nblocks * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));                // (the host vector goes out of scope)
    h->blk_last = nblocks;
    h->raw_valid = false; h->raw_sum_valid = false; h->raw_has_q = false; h->raw_q_folded = false; h->max_valid = false;
    mutated(h);
    if (invalid) *invalid = any_invalid ? 1 : 0;
    if (n_resampled) *n_resampled = count;
    if (check != GPF_CHECK_FALSE || invalid || n_resampled) {
        if (any_nan) return fail(h, GPF_ERR_INVALID_WEIGHTS, "Invalid weights (NaN).");
        if (check == GPF_CHECK_TRUE && (any_neginf_err || any_invalid)) return fail(h, GPF_ERR_INVALID_WEIGHTS, "Invalid weights.");   // resample.jl:55
    }
    return GPF_OK;
}
gpf_status gpf_resample_blocks(gpf_handle h, int32_t method, int64_t block_size, double priority_alpha, int32_t sort_particles,
                               double ess_frac, int32_t check, int32_t* invalid, int64_t* n_resampled)
{
    gpf_status s = block_checks(h, block_size, "gpf_resample_blocks");
    if (s) return s;
    if (method != GPF_RESAMPLE_MULTINOMIAL && method != GPF_RESAMPLE_RESIDUAL && method != GPF_RESAMPLE_STRATIFIED)
        return fail(h, GPF_ERR_UNKNOWN_METHOD, "Resampling method not recognized.");          // resample.jl:28
    if (h->hist_on) return fail(h, GPF_ERR_STATE, "gpf_resample_blocks on a filter with a trajectory store");
    if (h->W != 2 && h->W != 4 && h->W != 8) return fail(h, GPF_ERR_STATE, "row width");
    if ((s = materialize(h))) return s;
    if (block_size > BLK_MAX) return resample_big_blocks(h, method, block_size, priority_alpha, sort_particles, ess_frac, check, invalid, n_resampled);
    const int64_t nblocks = (h->n + block_size - 1) / block_size;
    if ((s = block_buffers(h, nblocks))) return s;
    BlockArgs a{};
    a.rows_in = h->rows[h->cur]; a.rows_out = h->rows[1 - h->cur]; a.lw = h->lw; a.anc = h->anc;
    a.n = h->n; a.nb = block_size; a.nblocks = nblocks; a.gid0 = h->cfg.gid0; a.seed = h->cfg.seed; a.epoch = h->epoch;
    a.sorted = method == GPF_RESAMPLE_STRATIFIED && sort_particles ? 1 : 0;
    const bool prio = priority_alpha == priority_alpha;
    a.alpha = prio ? priority_alpha : 1.0;
    a.ess_frac = ess_frac == ess_frac ? ess_frac : -1.0;
    a.check_true = check == GPF_CHECK_TRUE ? 1 : 0;
    a.resampled = h->blk_mask;
    s = timed(h, GPF_K_SEARCH, [&] {
        if (method == GPF_RESAMPLE_MULTINOMIAL)   launch_block_resample<0>(h, a, prio);
        else if (method == GPF_RESAMPLE_RESIDUAL) launch_block_resample<1>(h, a, prio);
        else                                      launch_block_resample<2>(h, a, prio);
    });
    if (s) return s;
    HIP_TRY(h, hipGetLastError());
    h->cur ^= 1;
    h->blk_last = nblocks;
    h->pending_gather = false; h->pending_fill = false;
    h->raw_valid = false; h->raw_sum_valid = false; h->raw_has_q = false; h->raw_q_folded = false; h->max_valid = false;
    h->epoch += 1;
    mutated(h);
    if (check != GPF_CHECK_FALSE || invalid || n_resampled) {
        int32_t words[2] = {0, 0};
        GPF_LAUNCH(k_block_summary, dim3(1), dim3(BLOCK), 0, h->stream, h->blk_mask, nblocks, h->blk_words);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipMemcpyAsync(words, h->blk_words, sizeof(words), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (invalid) *invalid = words[0] != 0;
        if (n_resampled) *n_resampled = (int64_t)(uint32_t)words[1];
        if (words[0] & (FLAG_NAN | FLAG_POSINF)) return fail(h, GPF_ERR_INVALID_WEIGHTS, "Invalid weights (NaN).");
        if (check == GPF_CHECK_TRUE && words[0]) return fail(h, GPF_ERR_INVALID_WEIGHTS, "Invalid weights.");   // resample.jl:55
    }
    return GPF_OK;
}
gpf_status gpf_block_resampled(gpf_handle h, int32_t* out)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (!out) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null out");
    if (!h->blk_mask || h->blk_last < 1) return fail(h, GPF_ERR_STATE, "gpf_block_resampled needs gpf_resample_blocks first");
    HIP_TRY(h, hipMemcpyAsync(out, h->blk_mask, (size_t)h->blk_last * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int64_t i = 0; i < h->blk_last; ++i) out[i] &= 1;       // (the words also carry the blocks' validity flags)
    return GPF_OK;
}
gpf_status gpf_block_stats(gpf_handle h, int64_t block_size, double* ess_out, double* lml_out)
{
    gpf_status s = block_checks(h, block_size, "gpf_block_stats");
    if (s) return s;
    if ((s = materialize(h))) return s;
    const int64_t nblocks = (h->n + block_size - 1) / block_size;
    if (block_size > BLK_MAX) {                                  // the loop over sub-states (big_block_views)
        if ((s = big_block_views(h, block_size))) return s;
        for (int64_t b = 0; b < nblocks; ++b) {
            gpf_filter* v = h->blk_views[(size_t)b];
            if (ess_out && (s = gpf_effective_sample_size(v, ess_out + b))) { h->err = v->err; return s; }
            if (lml_out && (s = gpf_log_ml_estimate(v, lml_out + b))) { h->err = v->err; return s; }
        }
        return GPF_OK;
    }
    if ((s = block_buffers(h, nblocks))) return s;
    if (block_size <= 2 * WAVE)      GPF_LAUNCH((k_block_stats<WAVE, 2>), dim3((unsigned)((nblocks + 3) / 4)), dim3(BLOCK), 0, h->stream, h->lw, h->n, block_size, nblocks, &h->sc->lml_est, h->blk_stats, h->blk_stats + nblocks);
    else if (block_size <= 8 * WAVE) GPF_LAUNCH((k_block_stats<WAVE, 8>), dim3((unsigned)((nblocks + 3) / 4)), dim3(BLOCK), 0, h->stream, h->lw, h->n, block_size, nblocks, &h->sc->lml_est, h->blk_stats, h->blk_stats + nblocks);
    else                             GPF_LAUNCH((k_block_stats<BLOCK, 8>), dim3((unsigned)nblocks), dim3(BLOCK), 0, h->stream, h->lw, h->n, block_size, nblocks, &h->sc->lml_est, h->blk_stats, h->blk_stats + nblocks);
    HIP_TRY(h, hipGetLastError());
    if (ess_out) HIP_TRY(h, hipMemcpyAsync(ess_out, h->blk_stats, (size_t)nblocks * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (lml_out) HIP_TRY(h, hipMemcpyAsync(lml_out, h->blk_stats + nblocks, (size_t)nblocks * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return GPF_OK;
}

// the blocks' observation vectors -> device ([n_blocks][MAX_OBS], zero-padded), ModelArgs::blk_* set
static gpf_status set_block_obs(gpf_filter* h, const double* obs, int32_t n_obs, int64_t block_size)
{
    if (!obs || n_obs != model_obs_dim(h->cfg.model))
        return fail(h, GPF_ERR_INVALID_ARGUMENT, "this model takes " + std::to_string(model_obs_dim(h->cfg.model)) + " observation values per step and block");
    const int64_t nblocks = (h->n + block_size - 1) / block_size;
    if (h->blk_obs_cap < nblocks) {
        if (h->blk_obs) {
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            (void)hipFree(h->blk_obs); h->blk_obs = nullptr; h->blk_obs_cap = 0;
            for (int k = 0; k < gpf_filter::BLK_STAGE; ++k) { (void)hipHostFree(h->h_blk_obs[k]); h->h_blk_obs[k] = nullptr; }
        }
        HIP_TRY(h, hipMalloc(&h->blk_obs, (size_t)nblocks * MAX_OBS * sizeof(double)));
        for (int k = 0; k < gpf_filter::BLK_STAGE; ++k) HIP_TRY(h, hipHostMalloc(&h->h_blk_obs[k], (size_t)nblocks * MAX_OBS * sizeof(double)));
        if (!h->h_blk_done) {
            HIP_TRY(h, hipHostMalloc(&h->h_blk_done, sizeof(int64_t))); *h->h_blk_done = 0;
            HIP_TRY(h, hipMalloc(&h->blk_stage_counter, sizeof(unsigned int)));
            HIP_TRY(h, hipMemsetAsync(h->blk_stage_counter, 0, sizeof(unsigned int), h->stream));
        }
        h->blk_obs_cap = nblocks;
    }
    // the staging buffers are used in turn: wait only until the copy that last read THIS buffer (four calls ago) has finished -- its
    // kernel publishes a ticket to pinned memory -- not for the stream
    const int k = (int)(h->blk_stage_next % gpf_filter::BLK_STAGE);
    if (h->blk_stage_next >= gpf_filter::BLK_STAGE) {
        const int64_t need = h->blk_stage_next - gpf_filter::BLK_STAGE + 1;
        uint64_t spins = 0;
        while (__atomic_load_n(h->h_blk_done, __ATOMIC_ACQUIRE) < need) {
            cpu_relax();
            if ((++spins & 0x3fff) != 0) continue;
            const hipError_t q = hipStreamQuery(h->stream);
            if (q == hipErrorNotReady) continue;
            if (__atomic_load_n(h->h_blk_done, __ATOMIC_ACQUIRE) >= need) break;
            return fail(h, GPF_ERR_HIP, q == hipSuccess ? "observation staging: the stream drained without the copy's ticket" : hipGetErrorString(q));
        }
    }
    h->blk_stage_next += 1;
    double* const stage = h->h_blk_obs[k];
    for (int64_t b = 0; b < nblocks; ++b)
        for (int i = 0; i < MAX_OBS; ++i) stage[b * MAX_OBS + i] = i < n_obs ? obs[b * n_obs + i] : 0.0;
    const int64_t n_words = nblocks * MAX_OBS;
    GPF_LAUNCH(k_stage_obs, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(64, (n_words + BLOCK - 1) / BLOCK))), dim3(BLOCK), 0, h->stream,
               stage, h->blk_obs, n_words, h->blk_stage_counter, h->h_blk_done, h->blk_stage_next);
    HIP_TRY(h, hipGetLastError());
    h->args.blk_obs = h->blk_obs; h->args.blk_mask = nullptr; h->args.blk_size = (int32_t)block_size;
    h->blk_obs_size = block_size;
    return GPF_OK;
}
static gpf_status block_step_checks(gpf_handle h, int64_t block_size, const char* who)
{
    if (!h) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null handle");
    if (h->parent) return fail(h, GPF_ERR_STATE, std::string(who) + " on a sub-state view: call it on the filter");
    if (h->cfg.n_global != h->n) return fail(h, GPF_ERR_STATE, std::string(who) + " on a shard of a sharded filter");
    if (h->hist_on) return fail(h, GPF_ERR_STATE, std::string(who) + " on a filter with a trajectory store");
    if (block_size < 1) return fail(h, GPF_ERR_INVALID_ARGUMENT, "block_size < 1");      // (the per-block steps index observations by i / block_size: any size)
    return GPF_OK;
}
gpf_status gpf_initialize_blocks(gpf_handle h, const double* obs, int32_t n_obs, int64_t block_size)
{
    gpf_status s = block_step_checks(h, block_size, "gpf_initialize_blocks");
    if (s) return s;
    h->generation += 1;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if ((s = set_block_obs(h, obs, n_obs, block_size))) return s;
    const int grid = step_grid(h);
    s = timed(h, GPF_K_STEP, [&] { DISPATCH_MODEL(h, (launch_init_blk<MM>(h, grid))); });
    if (s) return s;
    h->pending_gather = false; h->pending_fill = false; h->pending_packed = false; h->pending_search = false; h->pending_move = false;
    h->max_valid = true;
    GPF_LAUNCH(k_iota, dim3(grid), dim3(BLOCK), 0, h->stream, h->anc, h->n);            // parents = 1:N (initialize.jl:43)
    HIP_TRY(h, hipMemsetAsync(&h->sc->lml_est, 0, sizeof(double), h->stream));
    HIP_TRY(h, hipGetLastError());
    h->epoch += 1;
    h->initialized = true; h->has_prev = false; h->raw_valid = false; h->raw_sum_valid = false;
    h->blk_last = 0;                                             // only_resampled refers to a gpf_resample_blocks of the CURRENT step
    mutated(h);
    return GPF_OK;
}
gpf_status gpf_update_blocks(gpf_handle h, const double* obs, int32_t n_obs, int64_t block_size)
{
    gpf_status s = block_step_checks(h, block_size, "gpf_update_blocks");
    if (s) return s;
    if ((s = check_ready(h))) return s;
    if ((s = materialize(h))) return s;                          // (no fused gather in the block-wise step)
    if ((s = set_block_obs(h, obs, n_obs, block_size))) return s;
    const int grid = step_grid(h);
    const bool keep = h->cfg.keep_prev != 0;
    s = timed(h, GPF_K_STEP, [&] {
        if (keep) { DISPATCH_MODEL(h, (launch_step_blk<MM, true>(h, grid))); }
        else      { DISPATCH_MODEL(h, (launch_step_blk<MM, false>(h, grid))); }
    });
    if (s) return s;
    HIP_TRY(h, hipGetLastError());
    h->max_valid = true;
    h->cur ^= 1;
    h->epoch += 1;
    h->has_prev = true;
    h->raw_valid = false; h->raw_sum_valid = false;
    h->blk_last = 0;                                             // (as in gpf_initialize_blocks)
    mutated(h);
    return GPF_OK;
}
gpf_status gpf_rejuvenate_blocks(gpf_handle h, int32_t method, int32_t n_iters, int32_t only_resampled, uint64_t* n_accepted)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (h->blk_obs_size < 1) return fail(h, GPF_ERR_STATE, "gpf_rejuvenate_blocks needs gpf_initialize_blocks / gpf_update_blocks first (per-block observations)");
    if ((s = block_step_checks(h, h->blk_obs_size, "gpf_rejuvenate_blocks"))) return s;
    if (method != GPF_REJUVENATE_MOVE && method != GPF_REJUVENATE_REWEIGHT) return fail(h, GPF_ERR_UNKNOWN_METHOD, "Method not recognized.");   // rejuvenate.jl:25
    if (!h->cfg.keep_prev) return fail(h, GPF_ERR_STATE, "gpf_rejuvenate needs keep_prev = 1 (x_{t-1} must travel with the particle)");
    if (n_iters < 0) return fail(h, GPF_ERR_INVALID_ARGUMENT, "n_iters < 0");
    if (only_resampled) {
        const int64_t nblocks = (h->n + h->blk_obs_size - 1) / h->blk_obs_size;
        if (!h->blk_mask || h->blk_last != nblocks) return fail(h, GPF_ERR_STATE, "only_resampled needs a gpf_resample_blocks with the same block size first");
    }
    if ((s = materialize(h))) return s;
    h->args.blk_mask = only_resampled ? h->blk_mask : nullptr;
    const int grid = move_grid(h);
    s = timed(h, GPF_K_MOVE, [&] {
        if (method == GPF_REJUVENATE_REWEIGHT) { DISPATCH_MODEL(h, (launch_move_blk<MM, true>(h, grid, n_iters))); }
        else                                   { DISPATCH_MODEL(h, (launch_move_blk<MM, false>(h, grid, n_iters))); }
    });
    h->args.blk_mask = nullptr;
    if (s) return s;
    HIP_TRY(h, hipGetLastError());
    h->cur ^= 1;
    h->epoch += 1;
    if (method == GPF_REJUVENATE_REWEIGHT) { h->raw_valid = false; h->raw_sum_valid = false; h->max_valid = true; }
    mutated(h);
    if (n_accepted) {
        // (move-reweight: every particle of a participating block moves; the per-workgroup counts cover both cases)
        if (method == GPF_REJUVENATE_REWEIGHT && !only_resampled) *n_accepted = (uint64_t)h->n * (uint64_t)n_iters;
        else if (method == GPF_REJUVENATE_REWEIGHT) {
            int64_t nres = 0;
            GPF_LAUNCH(k_block_summary, dim3(1), dim3(BLOCK), 0, h->stream, h->blk_mask, h->blk_last, h->blk_words);
            int32_t words[2] = {0, 0};
            HIP_TRY(h, hipMemcpyAsync(words, h->blk_words, sizeof(words), hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            nres = (int64_t)(uint32_t)words[1];
            // (all blocks have block_size particles except possibly the last)
            const int64_t bs = h->blk_obs_size, last = h->n - (h->blk_last - 1) * bs;
            int32_t last_word = 0;
            HIP_TRY(h, hipMemcpy(&last_word, h->blk_mask + (h->blk_last - 1), sizeof(int32_t), hipMemcpyDeviceToHost));
            *n_accepted = (uint64_t)((nres - (last_word & 1)) * bs + (last_word & 1) * last) * (uint64_t)n_iters;
        } else {
            GPF_LAUNCH(k_sum_accepts, dim3(1), dim3(BLOCK), 0, h->stream, h->acc_part, grid, reinterpret_cast<unsigned long long*>(&h->sc->n_accept));
            HIP_TRY(h, hipGetLastError());
            if ((s = fetch_scalars(h))) return s;
            *n_accepted = h->h_sc->n_accept;
        }
    }
    return GPF_OK;
}

gpf_status gpf_resample_with_priorities(gpf_handle h, int32_t method, const double* log_priorities, int32_t sort_particles,
                                        int32_t check, int32_t* invalid)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (!log_priorities) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null log_priorities");
    HIP_TRY(h, hipMemcpyAsync(h->lp, log_priorities, (size_t)h->n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    PrioView pv{h->lw, h->lp, 0.0, 2};
    return resample_impl(h, method, pv, sort_particles, check, invalid);
}

static gpf_status rejuvenate_impl(gpf_handle h, int32_t method, int32_t n_iters, uint64_t* n_accepted, bool with_proposal);
gpf_status gpf_rejuvenate(gpf_handle h, int32_t method, int32_t n_iters, uint64_t* n_accepted)
{
    return rejuvenate_impl(h, method, n_iters, n_accepted, false);
}
gpf_status gpf_rejuvenate_proposal(gpf_handle h, int32_t proposal, const double* params, int32_t n_params, int32_t n_iters)
{
    if (!h) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null handle");
    if (n_params < 0 || n_params > 4 || (n_params > 0 && !params)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad proposal parameters");
    const bool ok = (proposal == GPF_MOVE_PROPOSAL_LOCALLY_OPTIMAL && h->cfg.model == MODEL_LGSSM2 && n_params == 0) ||
                    (proposal == GPF_MOVE_PROPOSAL_LINE_OUTLIER && h->cfg.model == MODEL_LINE && n_params == 3);
    if (!ok || !model_has_move_proposal(h->cfg.model)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "unknown move proposal for this model (or wrong parameter count)");
    for (int i = 0; i < 4; ++i) h->args.q[i] = i < n_params ? params[i] : 0.0;
    return rejuvenate_impl(h, GPF_REJUVENATE_REWEIGHT, n_iters, nullptr, true);
}
static gpf_status rejuvenate_impl(gpf_handle h, int32_t method, int32_t n_iters, uint64_t* n_accepted, bool with_proposal)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (h->parent && h->parent->blk_obs_size != 0)
        return fail(h, GPF_ERR_STATE, "rejuvenation of a view after a block-wise update of its filter: use gpf_rejuvenate_blocks on the filter");
    if (h->blk_obs_size < 0)
        return fail(h, GPF_ERR_STATE, "the filter was resized after a block-wise update: no current observation until the next update");
    if (h->blk_obs_size > 0 && !h->parent) {                     // the latest observations are per block (gpf_update_blocks)
        if (with_proposal) return fail(h, GPF_ERR_STATE, "proposal moves are not available after a block-wise update");
        return gpf_rejuvenate_blocks(h, method, n_iters, 0, n_accepted);
    }
    if (method != GPF_REJUVENATE_MOVE && method != GPF_REJUVENATE_REWEIGHT)
        return fail(h, GPF_ERR_UNKNOWN_METHOD, "Method not recognized.");                        // rejuvenate.jl:25
    if (!h->cfg.keep_prev) return fail(h, GPF_ERR_STATE, "gpf_rejuvenate needs keep_prev = 1 (x_{t-1} must travel with the particle)");
    if (n_iters < 0) return fail(h, GPF_ERR_INVALID_ARGUMENT, "n_iters < 0");
    if (h->pending_packed && (s = materialize(h))) return s;     // sharded deferred commit: scatter first
    if (h->pending_fill && (s = materialize(h))) return s;       // (the move kernel's fused gather assumes incoming weights 0)
    if ((s = finish_search(h))) return s;                        // (a lazy multinomial resample: the move kernel reads the ancestor array)
    // lazy move: a plain selection move of a whole, unsharded filter whose acceptance count nobody asked for waits for the pf_update! that
    // follows (k_move_step); its epoch is consumed now
    if (h->lazy_move && !with_proposal && !n_accepted && !h->parent && !h->hist_on && h->cfg.n_global == h->n && !h->pending_packed) {
        h->pending_move = true; h->pm_method = method; h->pm_iters = n_iters; h->pm_epoch = h->epoch; h->pm_args = h->args;
        h->epoch += 1;
        return GPF_OK;
    }
    const bool fused_gather = h->pending_gather;                 // a pending resample gather rides on the move kernel
    const int grid = move_grid(h);
    s = timed(h, GPF_K_MOVE, [&] {
        if (with_proposal)                          { DISPATCH_MODEL(h, (launch_move_prop_t<MM>(h, grid, n_iters))); }
        else if (method == GPF_REJUVENATE_REWEIGHT) { DISPATCH_MODEL(h, (launch_move_t<MM, true>(h, grid, n_iters, h->args, h->epoch))); }
        else                                        { DISPATCH_MODEL(h, (launch_move_t<MM, false>(h, grid, n_iters, h->args, h->epoch))); }
    });
    if (s) return s;
    HIP_TRY(h, hipGetLastError());
    h->cur ^= 1;
    h->epoch += 1;
    if (fused_gather) { h->pending_gather = false; h->pending_fill = false; h->max_valid = false; }   // log-weights are all 0 now (resample.jl:195)
    if (method == GPF_REJUVENATE_REWEIGHT) { h->raw_valid = false; h->raw_sum_valid = false; h->max_valid = true; }
    mutated(h);
    if ((s = view_exit(h))) return s;
    if (n_accepted) {
        if (method == GPF_REJUVENATE_REWEIGHT) *n_accepted = (uint64_t)h->n * (uint64_t)n_iters;     // every particle moves (rejuvenate.jl:81-86)
        else {
            GPF_LAUNCH(k_sum_accepts, dim3(1), dim3(BLOCK), 0, h->stream, h->acc_part, grid, reinterpret_cast<unsigned long long*>(&h->sc->n_accept));
            HIP_TRY(h, hipGetLastError());
            if ((s = fetch_scalars(h))) return s;
            *n_accepted = h->h_sc->n_accept;
        }
    }
    return GPF_OK;
}

gpf_status gpf_effective_sample_size(gpf_handle h, double* out)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (!out) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null out");
    WSum w{};
    bool reduced = false;
    if ((s = ensure_raw_summary(h, true, &reduced))) return s;
    if (reduced) w = h->sum_cache;                                // S and sum q^2 from the reduction: no CDF was written
    else {
        h->q_published = false;
        if ((s = ensure_raw(h, true))) return s;
        if (h->q_published) {
            // the scan of this call publishes {flags, S, limbs of sum q^2} itself: wait for its ticket, no publish launch
            h->q_published = false;
            if ((s = read_published_summary(h, w))) return s;
        } else {
            const bool fold = !h->raw_q_folded;                      // the scan blocks' limb partials of sum q^2: folded by the publish kernel
            if ((s = fetch_scalars(h, fold))) return s;
            h->raw_q_folded = true;
            w = h->h_sc->raw;
        }
    }
    if (w.flags) { *out = std::nan(""); return GPF_OK; }
    uint64_t hi, lo;
    normalise_Q(w, hi, lo);
    *out = ess_from(w.S, hi, lo);
    return GPF_OK;
}

gpf_status gpf_log_ml_estimate(gpf_handle h, double* out)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (!out) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null out");
    bool reduced = false;
    if ((s = ensure_raw_summary(h, false, &reduced))) return s;  // (S and the maximum are all it needs: no CDF)
    if (!reduced && (s = ensure_raw(h))) return s;
    if ((s = fetch_scalars(h))) return s;
    const WSum& w = reduced && h->sum_on_host ? h->sum_cache : h->h_sc->raw;   // (k_sum_host leaves {m, flags, S} with the host, not in the device block)
    double base = h->h_sc->lml_est;
    if (h->parent) {                                             // source.log_ml_est (utils.jl:174-178)
        if ((s = fetch_scalars(h->parent))) { h->err = h->parent->err; return s; }
        base = h->parent->h_sc->lml_est;
    }
    *out = base + lse_from(w.m, w.S, h->K, w.flags) - h->logN;
    return GPF_OK;
}

static gpf_status copy_out(gpf_handle h, const void* dsrc, void* out, size_t bytes)
{
    HIP_TRY(h, hipMemcpyAsync(out, dsrc, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return GPF_OK;
}

gpf_status gpf_get_log_weights(gpf_handle h, double* out, int64_t n)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (!out || n != h->n) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad output array");
    if ((s = materialize(h))) return s;
    return copy_out(h, h->lw, out, (size_t)n * sizeof(double));
}

static gpf_status norm_weights(gpf_handle h, double* out, int64_t n, int want_log)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (!out || n != h->n) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad output array");
    if ((s = ensure_raw(h))) return s;
    GPF_LAUNCH(k_norm_weights, dim3(grid_for(h, h->n, 8)), dim3(BLOCK), 0, h->stream, h->lw, &h->sc->raw, h->K, h->n, want_log,
                       h->dtmp);
    return copy_out(h, h->dtmp, out, (size_t)n * sizeof(double));
}
gpf_status gpf_get_log_norm_weights(gpf_handle h, double* out, int64_t n) { return norm_weights(h, out, n, 1); }
gpf_status gpf_get_norm_weights(gpf_handle h, double* out, int64_t n) { return norm_weights(h, out, n, 0); }

gpf_status gpf_get_parents(gpf_handle h, int64_t* out, int64_t n)
{
    gpf_status s0 = check_ready(h);                             // views: generation check + pointers; device; initialised
    if (s0) return s0;
    if (!out || n != h->n) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad output array");
    if (h->pending_packed) { gpf_status s = materialize(h); if (s) return s; }   // a deferred sharded commit also carries the parents
    { gpf_status s = finish_search(h); if (s) return s; }                         // a lazy multinomial resample: its ancestors now
    GPF_LAUNCH(k_parents, dim3(grid_for(h, h->n, 8)), dim3(BLOCK), 0, h->stream, h->anc, h->n, reinterpret_cast<int64_t*>(h->dtmp));
    return copy_out(h, h->dtmp, out, (size_t)n * sizeof(int64_t));
}

gpf_status gpf_state_dim(gpf_handle h, int32_t* dim, int32_t* row_width_out)
{
    if (!h) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null handle");
    if (dim) *dim = h->d;
    if (row_width_out) *row_width_out = h->W;
    return GPF_OK;
}

gpf_status gpf_get_column(gpf_handle h, int32_t column, double* out, int64_t n)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (!out || n != h->n || column < 0 || column >= h->W) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad column/output");
    if ((s = materialize(h))) return s;
    GPF_LAUNCH(k_extract_column, dim3(grid_for(h, h->n, 8)), dim3(BLOCK), 0, h->stream, h->rows[h->cur], h->W, column, h->n, h->dtmp);
    return copy_out(h, h->dtmp, out, (size_t)n * sizeof(double));
}

gpf_status gpf_get_rows(gpf_handle h, double* out, int64_t n_doubles)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (!out || n_doubles != h->n * h->W) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad output array");
    if ((s = materialize(h))) return s;
    return copy_out(h, h->rows[h->cur], out, (size_t)n_doubles * sizeof(double));
}

gpf_status gpf_set_rows(gpf_handle h, const double* rows, int64_t n_doubles)
{
    if (!h) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null handle");
    if (!rows || n_doubles != h->n * h->W) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad input array");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (h->pending_move) { gpf_status s = finish_move(h); if (s) return s; }
    if (h->parent) { gpf_status s = view_enter(h); if (s) return s; }
    { gpf_status s = materialize(h); if (s) return s; }
    HIP_TRY(h, hipMemcpyAsync(h->rows[h->cur], rows, (size_t)n_doubles * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (h->parent && h->view_step != 1) { gpf_status s = view_exit(h); if (s) return s; }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->initialized = true;
    mutated(h);
    return GPF_OK;
}

gpf_status gpf_set_log_weights(gpf_handle h, const double* lw, int64_t n)
{
    if (!h) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null handle");
    if (!lw || n != h->n) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad input array");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (h->pending_move) { gpf_status s = finish_move(h); if (s) return s; }
    if (h->parent) { gpf_status s = view_enter(h); if (s) return s; h->parent->raw_valid = false; h->parent->raw_sum_valid = false; h->parent->max_valid = false; }
    { gpf_status s = materialize(h); if (s) return s; }
    h->max_valid = false;
    HIP_TRY(h, hipMemcpyAsync(h->lw, lw, (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (h->parent && h->view_step != 1) { gpf_status s = view_exit(h); if (s) return s; }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->raw_valid = false; h->raw_sum_valid = false;
    h->initialized = true;
    mutated(h);
    return GPF_OK;
}

// sum_i w_i f(values[i * stride + col]) by the binary tree of DESIGN.md §3.5 (one workgroup per 2048 terms, then the same tree over
// the partials) into *out_dev (device)
static gpf_status weighted_tree_sum(gpf_filter* h, const double* values, int stride, int col, int pw, const double* center, double match, double* out_dev)
{
    const int64_t nb = (h->n + TREE_CHUNK - 1) / TREE_CHUNK;
    const int64_t need = nb + (nb + TREE_CHUNK - 1) / TREE_CHUNK + 1;
    if (h->tree_cap < need) {
        if (h->tree_buf) { HIP_TRY(h, hipStreamSynchronize(h->stream)); (void)hipFree(h->tree_buf); h->tree_buf = nullptr; h->tree_cap = 0; }
        HIP_TRY(h, hipMalloc(&h->tree_buf, (size_t)need * sizeof(double)));
        h->tree_cap = need;
    }
    double *in = h->tree_buf, *out = h->tree_buf + nb;
    GPF_LAUNCH(k_wsum_tree, dim3((unsigned)nb), dim3(BLOCK), 0, h->stream, h->lw, &h->sc->raw, h->K, values, stride, col, h->n, pw, center, match, in);
    for (int64_t np = nb; np > 1;) {
        const int64_t g = (np + TREE_CHUNK - 1) / TREE_CHUNK;
        GPF_LAUNCH(k_tree_partials, dim3((unsigned)g), dim3(BLOCK), 0, h->stream, in, np, out);
        np = g; std::swap(in, out);
    }
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(out_dev, in, sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    return GPF_OK;
}

static gpf_status wstat(gpf_handle h, int32_t column, double* out, bool variance)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (!out || column < 0 || column >= h->W) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad column/output");
    if ((s = ensure_raw(h))) return s;
    if ((s = weighted_tree_sum(h, h->rows[h->cur], h->W, column, 1, nullptr, 0.0, h->dscal))) return s;
    if (variance && (s = weighted_tree_sum(h, h->rows[h->cur], h->W, column, 2, h->dscal, 0.0, h->dscal + 1))) return s;
    double tmp[2];
    if ((s = copy_out(h, h->dscal, tmp, sizeof(tmp)))) return s;
    *out = variance ? tmp[1] : tmp[0];
    return GPF_OK;
}
gpf_status gpf_mean(gpf_handle h, int32_t column, double* out) { return wstat(h, column, out, false); }
gpf_status gpf_var(gpf_handle h, int32_t column, double* out) { return wstat(h, column, out, true); }

gpf_status gpf_kernel_timing(gpf_handle h, int32_t id, int32_t enable)
{
    if (!h || id < 0 || id >= GPF_K_COUNT) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad kernel id");
    Timer& t = h->timers[id];
    for (auto& e : t.ev) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    t.ev.clear();
    t.on = enable != 0;
    return GPF_OK;
}

gpf_status gpf_kernel_time(gpf_handle h, int32_t id, double* total_ms, int64_t* launches)
{
    if (!h || id < 0 || id >= GPF_K_COUNT) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad kernel id");
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    double tot = 0.0;
    for (auto& e : h->timers[id].ev) {
        float ms = 0.f;
        HIP_TRY(h, hipEventElapsedTime(&ms, e.first, e.second));
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = (int64_t)h->timers[id].ev.size();
    return GPF_OK;
}

gpf_status gpf_debug_math(gpf_handle h, int32_t which, const double* a, const double* b, int64_t n, double* out, double* out2)
{
    if (!h) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null handle");
    if (!a || !out || n < 1) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad arrays");
    double *da = nullptr, *db = nullptr, *d1 = nullptr, *d2 = nullptr;
    const size_t bytes = (size_t)n * sizeof(double);
    HIP_TRY(h, hipMalloc(&da, bytes)); HIP_TRY(h, hipMalloc(&db, bytes));
    HIP_TRY(h, hipMalloc(&d1, bytes)); HIP_TRY(h, hipMalloc(&d2, bytes));
    HIP_TRY(h, hipMemcpyAsync(da, a, bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(db, b ? b : a, bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemsetAsync(d2, 0, bytes, h->stream));
    GPF_LAUNCH(k_debug_math, dim3(grid_for(h, n, 8)), dim3(BLOCK), 0, h->stream, which, da, db, n, h->cfg.seed, h->epoch,
                       (uint32_t)TAG_UPDATE, d1, d2);
    HIP_TRY(h, hipMemcpyAsync(out, d1, bytes, hipMemcpyDeviceToHost, h->stream));
    if (out2) HIP_TRY(h, hipMemcpyAsync(out2, d2, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    hipFree(da); hipFree(db); hipFree(d1); hipFree(d2);
    return GPF_OK;
}

gpf_status gpf_debug_levels(gpf_handle h, int32_t which, void* out, int64_t* n_bytes)
{
    if (!h || !out || !n_bytes) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null argument");
    const int64_t nt = h->ntiles;
    const int logg = multi_logg(nt);
    const void* src = nullptr; int64_t bytes = 0;
    switch (which) {
        case 0: src = h->cdf[0]; bytes = nt * TILE * 8; break;
        case 1: src = h->t16[0]; bytes = nt * (TILE / 16) * 8; break;
        case 2: src = h->t256[0]; bytes = nt * (TILE / 256) * 8; break;
        case 3: src = k32_of(h->t256[0], nt); bytes = nt * (TILE / 32) * 4; break;
        case 4: src = off16_of(h->t256[0], nt); bytes = logg >= 0 ? nt * TILE * 2 : 0; break;
        case 5: src = coarse_of(h->t256[0], nt); bytes = logg >= 0 ? (nt * TILE / (4 << logg)) * 2 : 0; break;
        default: return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad level");
    }
    if (bytes > *n_bytes) return fail(h, GPF_ERR_INVALID_ARGUMENT, "output too small");
    *n_bytes = bytes;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    return bytes ? copy_out(h, src, out, (size_t)bytes) : GPF_OK;
}

// Gen.sample_unweighted_traces(state, n_samples) (reference src/utils.jl:7,189-194): n i.i.d. draws from the normalised
// weights, WITHOUT touching the filter (no log-ML update, weights unchanged).  Same CDF + search kernels as a resample.
gpf_status gpf_sample_unweighted(gpf_handle h, int64_t n_samples, double* rows_out, int64_t* idx_out)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (n_samples < 1 || n_samples >= ((int64_t)1 << 31) || !rows_out) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad arguments");
    if (h->cfg.n_global != h->n) return fail(h, GPF_ERR_STATE, "not available on shards");
    if ((s = ensure_raw(h))) return s;                            // CDF of state.log_weights in cdf[0]
    if ((s = fetch_scalars(h))) return s;
    if (h->h_sc->raw.flags & (FLAG_NAN | FLAG_POSINF)) return fail(h, GPF_ERR_INVALID_WEIGHTS, "Invalid weights (NaN).");
    int32_t* anc = nullptr; double* rows = nullptr; int64_t* idx64 = nullptr;
    HIP_TRY(h, hipMalloc(&anc, (size_t)n_samples * sizeof(int32_t)));
    HIP_TRY(h, hipMalloc(&rows, (size_t)n_samples * h->W * sizeof(double)));
    SearchArgs sa{};
    sa.w = levels(h, 0); sa.c = levels(h, 0); sa.ntiles = h->ntiles; sa.order = nullptr; sa.sc = h->sc; sa.ws = &h->sc->raw;
    sa.raw = &h->sc->raw; sa.n = n_samples; sa.n_cells = h->n; sa.n_global = n_samples; sa.gid0 = h->cfg.gid0; sa.seed = h->cfg.seed;
    sa.epoch = h->epoch; sa.K = h->K; sa.logN = h->logN; sa.update_lml = 0; sa.anc = anc;
    launch_multinomial_search(h, sa);
    launch_gather_rows_lw(h, anc, h->rows[h->cur], h->lw, rows, nullptr, n_samples);
    HIP_TRY(h, hipMemcpyAsync(rows_out, rows, (size_t)n_samples * h->W * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (idx_out) {
        HIP_TRY(h, hipMalloc(&idx64, (size_t)n_samples * sizeof(int64_t)));
        GPF_LAUNCH(k_parents, dim3(grid_for(h, n_samples, 8)), dim3(BLOCK), 0, h->stream, anc, n_samples, idx64);
        HIP_TRY(h, hipMemcpyAsync(idx_out, idx64, (size_t)n_samples * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    (void)hipFree(anc); (void)hipFree(rows); if (idx64) (void)hipFree(idx64);
    h->epoch += 1;
    if (h->parent) h->parent->epoch = h->epoch;
    return GPF_OK;
}

// =================================================================================== sub-state views (src/view.jl)
gpf_status gpf_view_create(gpf_handle parent, int64_t start, int64_t count, gpf_handle* out)
{
    return gpf_view_create_strided(parent, start, 1, count, out);
}

static gpf_status view_create_impl(gpf_handle parent, int64_t start, int64_t step, int64_t count, const int64_t* index, gpf_handle* out);
gpf_status gpf_view_create_strided(gpf_handle parent, int64_t start, int64_t step, int64_t count, gpf_handle* out)
{
    if (!parent || !out) return fail(parent, GPF_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (step < 1 || step >= ((int64_t)1 << 31)) return fail(parent, GPF_ERR_INVALID_ARGUMENT, "view step must be >= 1");
    return view_create_impl(parent, start, step, count, nullptr, out);
}
// state[idxs] / view(state, idxs) for any vector of DISTINCT indices (src/view.jl:35-48): the strided view's compact-copy mechanism with
// an index array.  index: HOST, 0-based, count entries.
gpf_status gpf_view_create_indexed(gpf_handle parent, const int64_t* index, int64_t count, gpf_handle* out)
{
    if (!parent || !out || !index) return fail(parent, GPF_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (count < 1) return fail(parent, GPF_ERR_INVALID_ARGUMENT, "empty index vector");
    std::vector<int64_t> sorted(index, index + count);
    std::sort(sorted.begin(), sorted.end());
    if (sorted.front() < 0 || sorted.back() >= parent->n) return fail(parent, GPF_ERR_INVALID_ARGUMENT, "view index out of bounds");
    if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end())
        return fail(parent, GPF_ERR_INVALID_ARGUMENT, "view indices must be distinct (a particle written through two slots of a view has no defined value)");
    return view_create_impl(parent, index[0], 0, count, index, out);
}
static gpf_status view_create_impl(gpf_handle parent, int64_t start, int64_t step, int64_t count, const int64_t* index, gpf_handle* out)
{
    if (parent->parent) return fail(parent, GPF_ERR_STATE, "views of views are not supported");
    // the trajectory store keeps ONE ancestor map and one set of columns per time step for the whole filter: a sub-state that
    // resamples or advances only its own particles would leave it describing something else -- refuse instead of going stale
    if (parent->hist_on) return fail(parent, GPF_ERR_STATE, "a filter with a trajectory store has no sub-state views");
    if (!index && (start < 0 || count < 1 || start + (count - 1) * step >= parent->n)) return fail(parent, GPF_ERR_INVALID_ARGUMENT, "view range out of bounds");
    gpf_filter* v = new gpf_filter();
    v->cfg = parent->cfg;
    v->cfg.n_particles = count; v->cfg.n_global = count;          // a sub-state normalises over its own particles
    v->cfg.gid0 = parent->cfg.gid0 + start;                        // ... but RNG counters keep the global particle id
    v->args = parent->args;
    v->d = parent->d; v->W = parent->W; v->n = count; v->n_cu = parent->n_cu;
    v->stream = parent->stream; v->own_stream = false;
    v->parent = parent; v->view_start = start; v->view_step = step; v->parent_generation = parent->generation;
    v->args.gstride = (int32_t)(step ? step : 1);                  // per-particle RNG counters stay the source's particle ids (index views: ModelArgs::gid_map)
    auto body = [&]() -> gpf_status {
        HIP_TRY(v, hipSetDevice(v->cfg.device));
        // scratch of its own (weight levels, descriptors, partials, scalars); rows / lw / anc alias the parent
        v->ntiles = (v->n + TILE - 1) / TILE;
        v->K = fix_K(count);
        v->logN = log_((double)count);
        const size_t n = (size_t)count;
        if (index) {                                             // the particles' indices in the parent; their ids relative to the first
            std::vector<int32_t> ix((size_t)count), rel((size_t)count);
            for (int64_t i = 0; i < count; ++i) { ix[i] = (int32_t)index[i]; rel[i] = (int32_t)(index[i] - index[0]); }
            HIP_TRY(v, hipMalloc(&v->vidx, n * sizeof(int32_t)));
            HIP_TRY(v, hipMalloc(&v->vgid, n * sizeof(int32_t)));
            HIP_TRY(v, hipMemcpy(v->vidx, ix.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
            HIP_TRY(v, hipMemcpy(v->vgid, rel.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
            v->args.gid_map = v->vgid;
        }
        if (step != 1) {
            HIP_TRY(v, hipMalloc(&v->vrows[0], n * (size_t)v->W * sizeof(double)));
            HIP_TRY(v, hipMalloc(&v->vrows[1], n * (size_t)v->W * sizeof(double)));
            HIP_TRY(v, hipMalloc(&v->vlw, n * sizeof(double)));
            HIP_TRY(v, hipMalloc(&v->vanc, n * sizeof(int32_t)));
        }
        HIP_TRY(v, hipMalloc(&v->lws, n * sizeof(double)));
        HIP_TRY(v, hipMalloc(&v->lp, n * sizeof(double)));
        HIP_TRY(v, hipMalloc(&v->dtmp, n * sizeof(double)));
        HIP_TRY(v, hipMalloc(&v->cdf[0], (size_t)v->ntiles * TILE * sizeof(uint64_t)));
        HIP_TRY(v, hipMalloc(&v->t16[0], (size_t)v->ntiles * (TILE / 16) * sizeof(uint64_t)));
        HIP_TRY(v, hipMalloc(&v->t256[0], t256_bytes(v->ntiles)));
        const size_t db = (((size_t)2 * v->ntiles * sizeof(uint64_t)) + 15) & ~(size_t)15;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 2; ++j) {
                HIP_TRY(v, hipMalloc(&v->desc[i][j], db));
                HIP_TRY(v, hipMemsetAsync(v->desc[i][j], 0, db, v->stream));
            }
        for (int b = 0; b < 2; ++b) {
            HIP_TRY(v, hipMalloc(&v->mslots[b], (size_t)MAX_SLOTS * SLOT_WORDS * sizeof(unsigned long long)));
            HIP_TRY(v, hipMemsetAsync(v->mslots[b], 0, (size_t)MAX_SLOTS * SLOT_WORDS * sizeof(unsigned long long), v->stream));
        }
        HIP_TRY(v, hipMalloc(&v->blockQ, (size_t)4 * 8 * v->n_cu * sizeof(uint64_t) + 64));
        HIP_TRY(v, hipMemsetAsync(v->blockQ, 0, (size_t)4 * 8 * v->n_cu * sizeof(uint64_t) + 64, v->stream));
        HIP_TRY(v, hipMalloc(&v->partial, MAX_PARTIALS * sizeof(double)));
        HIP_TRY(v, hipMalloc(&v->acc_part, MAX_PARTIALS * sizeof(unsigned long long)));
        HIP_TRY(v, hipMalloc(&v->dscal, 4 * sizeof(double)));
        HIP_TRY(v, hipMalloc(&v->sc, sizeof(Scalars)));
        HIP_TRY(v, hipHostMalloc(&v->h_sc, sizeof(Scalars)));
        HIP_TRY(v, hipHostMalloc(&v->h_timeout, sizeof(int32_t)));
        *v->h_timeout = 0;
        v->scan_blocks_per_cu = parent->scan_blocks_per_cu; v->wscan_blocks_per_cu = parent->wscan_blocks_per_cu;
        HIP_TRY(v, hipMemsetAsync(v->sc, 0, sizeof(Scalars), v->stream));
        return GPF_OK;
    };
    gpf_status st = body();
    if (st != GPF_OK) { parent->err = v->err; gpf_destroy(v); return st; }
    *out = v;
    return GPF_OK;
}

// =================================================================================== resize family (src/resize.jl)
static gpf_status resize_ready(gpf_handle h)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (h->cfg.n_global != h->n) return fail(h, GPF_ERR_STATE, "resizing a sharded filter is not supported");
    if (h->hist_on) return fail(h, GPF_ERR_STATE, "resizing a filter with a trajectory store is not supported");
    if (h->parent) return fail(h, GPF_ERR_STATE, "a sub-state view cannot be resized");
    h->generation += 1;                  // views of this filter become stale
    return materialize(h);
}
// after the particle count changed: unsharded bookkeeping
static void set_count(gpf_filter* h, int64_t n_new)
{
    h->n = n_new; h->cfg.n_particles = n_new; h->cfg.n_global = n_new; h->cfg.gid0 = 0;
    if (h->blk_obs_size != 0) h->blk_obs_size = -1;              // per-block observations do not survive a change of the particle count
    h->blk_last = 0;
    h->raw_valid = false; h->raw_sum_valid = false; h->raw_has_q = false; h->raw_q_folded = false; h->max_valid = false; h->pending_gather = false; h->pending_fill = false;
    h->pending_packed = false; h->pending_search = false;
}

gpf_status gpf_n_particles(gpf_handle h, int64_t* out)
{
    if (!h || !out) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null argument");
    *out = h->n;
    return GPF_OK;
}

// pf_optimal_resize! (resize.jl:149-200): keep every particle with c w_i >= 1, resample the rest by systematic
// sampling, in exact fixed point.  n_new <= n_old.
static gpf_status resize_optimal(gpf_handle h, int64_t n_new, int32_t check, int32_t* invalid)
{
    const int64_t n_old = h->n;
    if (n_new < 1 || n_new > n_old) return fail(h, GPF_ERR_INVALID_ARGUMENT, "optimal resize: need 1 <= n_particles <= current count");   // resize.jl:185
    gpf_status s;
    // sort(weights) (resize.jl:204), descending; safe_softmax + logsumexp (resize.jl:152,190) over that order
    if ((s = ensure_residual_buffers(h))) return s;
    const PrioView pv = raw_view(h);
    if ((s = ensure_max(h, pv, true))) return s;
    if ((s = sort_desc(h, pv, n_old))) return s;
    WSum* ws = &h->sc->raw;
    h->raw_valid = false; h->raw_sum_valid = false;
    if ((s = summarize(h, pv, ws, true, h->order, true, false, false, true))) return s;
    HIP_TRY(h, hipMemsetAsync(&h->sc->opt_d, 0xff, sizeof(long long), h->stream));
    GPF_LAUNCH(k_opt_threshold, dim3(grid_for(h, n_new, 8)), dim3(BLOCK), 0, h->stream, h->cdf[0], ws, n_new, n_old, h->sc);
    GPF_LAUNCH(k_opt_params, dim3(1), dim3(1), 0, h->stream, h->cdf[0], ws, n_new, h->sc);
    // keep flags -> compaction offsets (channel 1); weights of the others -> their CDF (channel 2)
    InOptimal ik{h->lw, ws, h->sc, h->K, 0}, iw{h->lw, ws, h->sc, h->K, 1};
    if ((s = scan_launch<InOptimal, 0>(h, 1, ik, 0, nullptr, true, &h->sc->Ctot))) return s;
    if ((s = scan_launch<InOptimal, 0>(h, 2, iw, 0, nullptr, true, &h->sc->Rs))) return s;
    if ((s = fetch_scalars(h))) return s;
    const WSum& w = h->h_sc->raw;
    const int64_t n_keep = (int64_t)h->h_sc->Ctot, n_res = n_new - n_keep;
    bool inv = w.flags != 0;
    if ((w.flags & (FLAG_NAN | FLAG_POSINF)) || (check == GPF_CHECK_TRUE && inv)) {
        if (invalid) *invalid = 1;
        return fail(h, GPF_ERR_INVALID_WEIGHTS, "Invalid weights.");                              // resize.jl:153
    }
    if (n_res > 0 && h->h_sc->Rs == 0) {
        // every particle that is not kept has weight 0 at this resolution: uniform among them (safe_softmax, resize.jl:166-168)
        inv = true;
        if (check == GPF_CHECK_TRUE) { if (invalid) *invalid = 1; return fail(h, GPF_ERR_INVALID_WEIGHTS, "Invalid weights."); }
        InOptimal iu{h->lw, ws, h->sc, h->K, 2};
        if ((s = scan_launch<InOptimal, 0>(h, 2, iu, 0, nullptr, true, &h->sc->Rs))) return s;
    }
    if (invalid) *invalid = inv ? 1 : 0;
    const CdfLevels lv = levels(h, 2);
    const uint64_t* keepcdf = h->cdf[1];
    const int64_t ntiles_old = h->ntiles;
    const int K = h->K;
    Bufs old = take_particle_buffers(h);
    set_count(h, n_new);
    if ((s = alloc_particle_buffers(h))) { free_bufs(old); return s; }
    GPF_LAUNCH(k_opt_keep_scatter, dim3(grid_for(h, n_old, 8)), dim3(BLOCK), 0, h->stream, keepcdf, n_old, h->anc);
    if (n_res > 0) {
        SearchArgs sa{};
        sa.w = lv; sa.c = lv; sa.ntiles = ntiles_old; sa.order = nullptr; sa.sc = h->sc; sa.ws = ws; sa.raw = ws;
        sa.n = n_res; sa.n_cells = n_old; sa.n_global = n_res; sa.gid0 = 0; sa.seed = h->cfg.seed; sa.epoch = h->epoch;
        sa.K = K; sa.logN = 0.0; sa.update_lml = 0; sa.anc = h->anc + n_keep;
        const size_t lds = search_lds_bytes(ntiles_old, 1);
        const int gsr = (int)std::max<int64_t>(1, std::min<int64_t>((n_res + 2 * SBLOCK - 1) / (2 * SBLOCK), h->n_cu));
        GPF_LAUNCH((k_search<3>), dim3(gsr), dim3(SBLOCK), lds, h->stream, sa);
    }
    // new_traces .= view(traces, parents); log_weights (resize.jl:189-197)
    launch_gather_rows_lw(h, h->anc, old.rows[old.cur], old.lw, h->rows[0], h->lw, n_new);
    const double ratio = log_((double)n_new) - log_((double)n_old);
    GPF_LAUNCH(k_opt_weights, dim3(grid_for(h, n_new, 8)), dim3(BLOCK), 0, h->stream, h->lw, n_new, h->sc, ws, K, ratio);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    free_bufs(old);
    HIP_TRY(h, hipGetLastError());
    h->epoch += 1;
    return GPF_OK;
}

gpf_status gpf_resize(gpf_handle h, int64_t n_new, int32_t method, double priority_alpha, int32_t check, int32_t* invalid)
{
    gpf_status s = resize_ready(h);
    if (s) return s;
    if (method == GPF_RESAMPLE_OPTIMAL) return resize_optimal(h, n_new, check, invalid);          // resize.jl:22-23
    if (method != GPF_RESAMPLE_MULTINOMIAL && method != GPF_RESAMPLE_RESIDUAL)
        return fail(h, GPF_ERR_UNKNOWN_METHOD, "Resampling method not recognized.");             // resize.jl:26
    if (n_new < 1 || n_new >= ((int64_t)1 << 31)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad n_particles");
    const int64_t n_old = h->n;
    PrioView pv = raw_view(h);
    if (priority_alpha == priority_alpha) { pv.alpha = priority_alpha; pv.mode = 1; }
    // fixed-point scale for max(n_old, n_new): both N_old 2^K and n_new 2^K must stay below 2^62
    h->K = fix_K(std::max(n_old, n_new));
    h->raw_valid = false; h->raw_sum_valid = false;
    WSum* ws = &h->sc->raw;
    if ((s = summarize(h, raw_view(h), &h->sc->raw, true, nullptr, true))) return s;          // logsumexp(log_weights), resize.jl:58
    if (pv.mode != 0) { ws = &h->sc->prio; if ((s = summarize(h, pv, ws, true, nullptr, false))) return s; }
    if (check == GPF_CHECK_TRUE || invalid) {
        if ((s = fetch_scalars(h))) return s;
        const WSum& w = pv.mode == 0 ? h->h_sc->raw : h->h_sc->prio;
        if (invalid) *invalid = w.flags != 0;
        if ((w.flags & (FLAG_NAN | FLAG_POSINF)) || (check == GPF_CHECK_TRUE && w.flags)) {
            h->K = fix_K(n_old);
            return fail(h, GPF_ERR_INVALID_WEIGHTS, "Invalid weights.");                          // resize.jl:56,97
        }
    }
    SearchArgs sa{};
    sa.w = levels(h, 0); sa.c = levels(h, 0); sa.ntiles = h->ntiles; sa.order = nullptr; sa.sc = h->sc; sa.ws = ws;
    sa.raw = &h->sc->raw; sa.n = n_new; sa.n_cells = n_old; sa.n_global = n_new; sa.gid0 = 0; sa.seed = h->cfg.seed;
    sa.epoch = h->epoch; sa.K = h->K; sa.logN = log_((double)n_old); sa.update_lml = 1;
    if (method == GPF_RESAMPLE_RESIDUAL) {
        if ((s = residual_scans(h, ws, n_new))) return s;                                        // floor(n_particles * w), resize.jl:106
        sa.w = levels(h, 2); sa.c = levels(h, 1);
    }
    const int64_t ntiles_old = h->ntiles;
    const int Kp = h->K;
    Bufs old = take_particle_buffers(h);                                                          // resize!(...), resize.jl:60-61
    set_count(h, n_new);
    if ((s = alloc_particle_buffers(h))) { free_bufs(old); return s; }
    sa.anc = h->anc;
    const int64_t nt = method == GPF_RESAMPLE_RESIDUAL ? 2 : 1;
    const size_t lds = search_lds_bytes(ntiles_old, (int)nt);
    const int gsr = (int)std::max<int64_t>(1, std::min<int64_t>((n_new + 2 * SBLOCK - 1) / (2 * SBLOCK), h->n_cu));
    if (method == GPF_RESAMPLE_RESIDUAL) GPF_LAUNCH((k_search<1>), dim3(gsr), dim3(SBLOCK), lds, h->stream, sa);
    else                                 launch_multinomial_search(h, sa);
    // new_traces .= view(traces, parents) + update_weights!(state, n_particles, log_priorities)   resize.jl:64-66,424-438
    launch_gather_ex(h, h->anc, old.rows[old.cur], h->rows[0], pv, pv.mode == 0 ? h->lw : h->lws, n_new);
    if (pv.mode != 0) {
        PrioView post{h->lws, nullptr, 0.0, 0};
        if ((s = summarize(h, post, &h->sc->post, false, nullptr, false))) { free_bufs(old); return s; }
        GPF_LAUNCH(k_apply_post, dim3(grid_for(h, n_new, 8)), dim3(BLOCK), 0, h->stream, h->sc, h->K, h->logN, h->lws, h->lw, n_new);
        h->max_valid = false;
    }
    (void)Kp;
    HIP_TRY(h, hipStreamSynchronize(h->stream));                 // the old buffers are read by the kernels above
    free_bufs(old);
    HIP_TRY(h, hipGetLastError());
    h->epoch += 1;
    return GPF_OK;
}

gpf_status gpf_replicate(gpf_handle h, int32_t n_replicates, int32_t interleaved)
{
    gpf_status s = resize_ready(h);
    if (s) return s;
    if (n_replicates < 1 || h->n * (int64_t)n_replicates >= ((int64_t)1 << 31)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad n_replicates");
    const int64_t n_old = h->n, n_new = n_old * n_replicates;
    Bufs old = take_particle_buffers(h);
    set_count(h, n_new);
    if ((s = alloc_particle_buffers(h))) { free_bufs(old); return s; }
    GPF_LAUNCH(k_replicate_anc, dim3(grid_for(h, n_new, 8)), dim3(BLOCK), 0, h->stream, n_new, n_old, (int)n_replicates,
                       (int)(interleaved != 0), 0, h->anc);
    launch_gather_rows_lw(h, h->anc, old.rows[old.cur], old.lw, h->rows[0], h->lw, n_new);     // resize.jl:240-242
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    free_bufs(old);
    HIP_TRY(h, hipGetLastError());
    return GPF_OK;
}

gpf_status gpf_dereplicate(gpf_handle h, int32_t n_replicates, int32_t interleaved, int32_t sample)
{
    gpf_status s = resize_ready(h);
    if (s) return s;
    if (n_replicates < 1 || h->n % n_replicates != 0)
        return fail(h, GPF_ERR_INVALID_ARGUMENT, "the number of particles must be a multiple of n_replicates");   // resize.jl:270
    const int64_t n_old = h->n, n_new = n_old / n_replicates;
    Bufs old = take_particle_buffers(h);
    set_count(h, n_new);
    if ((s = alloc_particle_buffers(h))) { free_bufs(old); return s; }
    const int grid = grid_for(h, n_new, 8);
    if (sample) {                                                                                // resize.jl:281-293
        GPF_LAUNCH(k_dereplicate_sample, dim3(grid), dim3(BLOCK), 0, h->stream, old.lw, n_new, n_old, (int)n_replicates,
                           (int)(interleaved != 0), h->cfg.seed, h->epoch, fix_K(n_replicates), log_((double)n_replicates), h->anc, h->lw);
        launch_gather_rows_lw(h, h->anc, old.rows[old.cur], old.lw, h->rows[0], nullptr, n_new);
        h->epoch += 1;
    } else {                                                                                     // :keepfirst, resize.jl:274-279
        GPF_LAUNCH(k_replicate_anc, dim3(grid), dim3(BLOCK), 0, h->stream, n_new, n_old, (int)n_replicates,
                           (int)(interleaved != 0), 1, h->anc);
        launch_gather_rows_lw(h, h->anc, old.rows[old.cur], old.lw, h->rows[0], h->lw, n_new);
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    free_bufs(old);
    HIP_TRY(h, hipGetLastError());
    return GPF_OK;
}

// =================================================================================== trajectory store
gpf_status gpf_history_enable(gpf_handle h, int32_t max_steps)
{
    if (!h) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null handle");
    if (max_steps < 1) return fail(h, GPF_ERR_INVALID_ARGUMENT, "max_steps < 1");
    if (h->cfg.n_global != h->n) return fail(h, GPF_ERR_STATE, "the trajectory store is not available for sharded filters");
    if (h->initialized) return fail(h, GPF_ERR_STATE, "enable the trajectory store before gpf_initialize");
    h->hist_on = true; h->hist_cap = max_steps;
    if (h->hist_dev_maps) (void)hipFree(h->hist_dev_maps);
    HIP_TRY(h, hipMalloc(&h->hist_dev_maps, (size_t)max_steps * sizeof(int32_t*)));
    return GPF_OK;
}

gpf_status gpf_history_steps(gpf_handle h, int32_t* n_steps)
{
    if (!h || !n_steps) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null argument");
    *n_steps = h->hist_on ? (int32_t)h->hist_x.size() : 0;
    return GPF_OK;
}

// column `column` of time step `step` (1-based, like the t of the Julia address t => :name) into h->dtmp
static gpf_status history_values(gpf_handle h, int32_t step, int32_t column)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    if (!h->hist_on) return fail(h, GPF_ERR_STATE, "trajectory store not enabled (gpf_history_enable)");
    const int T = (int)h->hist_x.size();
    if (step < 1 || step > T || column < 0 || column >= h->d) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad step/column");
    if ((s = hist_snapshot(h))) return s;                         // the current step, in its current order
    // maps of steps T, T-1, ..., step+1 (0-based indices T-1 ... step), applied in that order
    std::vector<const int32_t*> maps;
    for (int q = T - 1; q >= step; --q) maps.push_back(h->hist_map[q]);
    if (!maps.empty())
        HIP_TRY(h, hipMemcpyAsync(h->hist_dev_maps, maps.data(), maps.size() * sizeof(int32_t*), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));                   // `maps` is a host temporary
    GPF_LAUNCH(k_hist_column, dim3(grid_for(h, h->n, 8)), dim3(BLOCK), 0, h->stream, h->hist_dev_maps, (int)maps.size(),
                       h->hist_x[step - 1], h->d, (int)column, h->n, h->dtmp);
    HIP_TRY(h, hipGetLastError());
    return GPF_OK;
}

gpf_status gpf_history_column(gpf_handle h, int32_t step, int32_t column, double* out, int64_t n)
{
    if (!h) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null handle");
    if (!out || n != h->n) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad output array");
    gpf_status s = history_values(h, step, column);
    if (s) return s;
    return copy_out(h, h->dtmp, out, (size_t)n * sizeof(double));
}

static gpf_status history_stat(gpf_handle h, int32_t step, int32_t column, double* out, bool variance)
{
    if (!h || !out) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null argument");
    gpf_status s = history_values(h, step, column);
    if (s) return s;
    if ((s = ensure_raw(h))) return s;
    if ((s = weighted_tree_sum(h, h->dtmp, 1, 0, 1, nullptr, 0.0, h->dscal))) return s;
    if (variance && (s = weighted_tree_sum(h, h->dtmp, 1, 0, 2, h->dscal, 0.0, h->dscal + 1))) return s;
    double tmp[2];
    if ((s = copy_out(h, h->dscal, tmp, sizeof(tmp)))) return s;
    *out = variance ? tmp[1] : tmp[0];
    return GPF_OK;
}
// proportionmap(state, addr)[value] (statistics.jl:91-101): normalised weight of the particles whose column equals `value`;
// step = 0: the current step's column, step >= 1: a past choice along the ancestry (trajectory store)
gpf_status gpf_proportion(gpf_handle h, int32_t step, int32_t column, double value, double* out)
{
    if (!h || !out) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null argument");
    gpf_status s;
    if (step > 0) { if ((s = history_values(h, step, column))) return s; }
    else {
        if ((s = check_ready(h))) return s;
        if (column < 0 || column >= h->W) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad column");
        if ((s = materialize(h))) return s;
        GPF_LAUNCH(k_extract_column, dim3(grid_for(h, h->n, 8)), dim3(BLOCK), 0, h->stream, h->rows[h->cur], h->W, column, h->n, h->dtmp);
    }
    if ((s = ensure_raw(h))) return s;
    if ((s = weighted_tree_sum(h, h->dtmp, 1, 0, 3, nullptr, value, h->dscal))) return s;
    return copy_out(h, h->dscal, out, sizeof(double));
}
gpf_status gpf_history_mean(gpf_handle h, int32_t step, int32_t column, double* out) { return history_stat(h, step, column, out, false); }
gpf_status gpf_history_var(gpf_handle h, int32_t step, int32_t column, double* out) { return history_stat(h, step, column, out, true); }

// =================================================================================== shard-level ABI
static gpf_status shard_ready(gpf_handle h)
{
    gpf_status s = check_ready(h);
    if (s) return s;
    return GPF_OK;
}

static gpf_status shard_max_slots(gpf_handle h);
gpf_status gpf_shard_weight_max(gpf_handle h, double* out2)
{
    gpf_status s = shard_ready(h);
    if (s) return s;
    if (!out2) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null out");
    if ((s = shard_max_slots(h))) return s;
    GPF_LAUNCH(k_pack_mflags, dim3(1), dim3(BLOCK), 0, h->stream, h->mslots[h->mcur], out2, mb_begin(h, MB_MF));
    HIP_TRY(h, hipGetLastError());
    return GPF_OK;
}
// the maximum slots describe the weights to summarise (the producer's slots, or one k_max_partial pass)
static gpf_status shard_max_slots(gpf_handle h)
{
    gpf_status s;
    if ((s = materialize(h))) return s;
    const PrioView pv = h->sum_pv_set ? h->sum_pv : raw_view(h);  // (the engine's prioritised resample summarises alpha lw and log_ws too)
    if (!h->max_valid || h->sum_pv_set) {
        const int gp = (int)std::min<int64_t>(MAX_PARTIALS, (h->n + BLOCK - 1) / BLOCK);
        s = timed(h, GPF_K_MAX, [&] {
            GPF_LAUNCH(k_max_partial, dim3(gp), dim3(BLOCK), 0, h->stream, pv, h->n, next_slots(h));
        });
        if (s) return s;
        h->max_valid = !h->sum_pv_set;
    }
    return GPF_OK;
}

gpf_status gpf_shard_weight_scan(gpf_handle h, const double* mf_all, int32_t G, int32_t want_q, int64_t* out5)
{
    gpf_status s = shard_ready(h);
    if (s) return s;
    if (!mf_all || !out5 || G < 1) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad arguments");
    if (!h->shard_counts) {
        HIP_TRY(h, hipMalloc(&h->shard_counts, (size_t)2 * MAX_SHARDS * COUNT_STRIDE * sizeof(int64_t)));
        HIP_TRY(h, hipMemsetAsync(h->shard_counts, 0, (size_t)2 * MAX_SHARDS * COUNT_STRIDE * sizeof(int64_t), h->stream));
        HIP_TRY(h, hipHostMalloc(&h->h_shard_counts, (size_t)(2 * MAX_SHARDS + 1) * sizeof(int64_t)));
        h->h_shard_counts[2 * MAX_SHARDS] = 0;
    }
    InFixQ in{h->sum_pv_set ? h->sum_pv : raw_view(h), nullptr, nullptr, h->K, 0.0, 0};
    WSum* const slot = h->sum_slot ? h->sum_slot : &h->sc->raw;
    const bool want_cdf = !h->sum_no_cdf;
    const int gs = wscan_grid(h);
    // the scan folds the gathered (max, flags) pairs itself and writes the shard total straight into out5[0]; it also
    // publishes the global validity flags to pinned host memory (gpf_shard_flags)
    if (!h->h_flags) { HIP_TRY(h, hipHostMalloc(&h->h_flags, 2 * sizeof(int64_t))); h->h_flags[0] = h->h_flags[1] = 0; }
    h->flag_ticket += 1;
    // shard mailboxes: the scan waits for the ranks' (max, flags) entries itself and the kernel that ends up with the shard's
    // {S, limbs} stores them into every peer's mailbox (the scan's last workgroup, or k_export_q when the limbs are wanted)
    ScanExtras ex{h->shard_counts, h->h_flags, h->flag_ticket, 0};
    ex.zero_stride = COUNT_STRIDE;
    if (h->fuse_mf_out) {                                        // (library engine: the scan produces and pushes the first summary itself)
        ex.fuse_mf = 1; ex.mf_me = h->comm_rank; ex.mf_out = h->fuse_mf_out; ex.mf_push = mb_begin(h, MB_MF);
        if (h->mb_active && h->mb_engine) mf_all = static_cast<const double*>(mb_gathered(h, MB_MF));   // the round that has just begun
    }
    ex.wait = mb_wait(h, MB_MF);
    const MboxPush tot_push = mb_begin(h, MB_TOT);
    if (want_q) {
        if ((s = scan_launch<InFixQ, 4>(h, 0, in, (int)G, slot, want_cdf, reinterpret_cast<uint64_t*>(out5), mf_all, ex))) return s;
        GPF_LAUNCH(k_export_q, dim3(1), dim3(BLOCK), 0, h->stream, h->blockQ, gs, out5, tot_push);
    } else {
        ex.push = tot_push;
        if ((s = scan_launch<InFixQ, 3>(h, 0, in, (int)G, slot, want_cdf, reinterpret_cast<uint64_t*>(out5), mf_all, ex))) return s;
    }
    HIP_TRY(h, hipGetLastError());
    h->raw_valid = false; h->raw_sum_valid = false;            // sc->raw holds the GLOBAL max but no sum: not the unsharded summary
    return GPF_OK;
}

gpf_status gpf_shard_flags(gpf_handle h, int32_t* flags_out)
{
    gpf_status s = shard_ready(h);
    if (s) return s;
    if (!flags_out || !h->h_flags || h->flag_ticket == 0) return fail(h, GPF_ERR_STATE, "gpf_shard_flags needs gpf_shard_weight_scan first");
    if ((s = wait_ticket(h, h->h_flags + 1, h->flag_ticket, "weight scan flags"))) return s;
    *flags_out = (int32_t)h->h_flags[0];
    return GPF_OK;
}

gpf_status gpf_shard_residual_scan(gpf_handle h, const int64_t* tot_all, int32_t G, int64_t* out2)
{
    gpf_status s = shard_ready(h);
    if (s) return s;
    if (!tot_all || !out2 || G < 1) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad arguments");
    // global S into sc->prio (the local CDF in cdf[0] stays local)
    GPF_LAUNCH(k_set_global, dim3(1), dim3(64), 0, h->stream, tot_all, (int)G, &h->sc->prio, mb_wait(h, MB_TOT));
    if ((s = residual_scans(h, &h->sc->prio, h->cfg.n_global))) return s;
    GPF_LAUNCH(k_export_residual, dim3(1), dim3(64), 0, h->stream, h->sc, out2, mb_begin(h, MB_CR));
    HIP_TRY(h, hipGetLastError());
    h->residual_scanned = true;
    return GPF_OK;
}

// fill the argument block of the push kernels; bounds: HOST int64[G+1], first global slot of every shard
static gpf_status push_args(gpf_handle h, int32_t method, const int64_t* tot_all, const int64_t* cr_all, int32_t G, int32_t me,
                            const int64_t* bounds, PushArgs& a)
{
    if (method < 0 || method > 2) return fail(h, GPF_ERR_UNKNOWN_METHOD, "Resampling method not recognized.");
    if (!tot_all || !bounds || G < 1 || G > MAX_SHARDS || me < 0 || me >= G || (method == GPF_RESAMPLE_RESIDUAL && !cr_all))
        return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad arguments");
    if (bounds[0] != 0 || bounds[G] != h->cfg.n_global || bounds[me] != h->cfg.gid0 || bounds[me + 1] != h->cfg.gid0 + h->n)
        return fail(h, GPF_ERR_INVALID_ARGUMENT, "shard bounds do not match this filter's global range");
    a.seed = h->cfg.seed; a.epoch = h->epoch; a.n_global = h->cfg.n_global; a.G = G; a.me = me;
    int64_t c = 0;
    for (int g = 0; g < G; ++g) {
        if (bounds[g + 1] < bounds[g]) return fail(h, GPF_ERR_INVALID_ARGUMENT, "shard bounds must be non-decreasing");
        a.bounds[g] = bounds[g]; a.chunk0[g] = c;
        c += (bounds[g + 1] - bounds[g] + PUSH_CHUNK - 1) / PUSH_CHUNK;
    }
    a.bounds[G] = bounds[G]; a.chunk0[G] = c; a.nchunks = c;
    if (h->cfg.n_global > h->push_cap) {                       // staging list: one 16-byte entry per GLOBAL output slot at most
        if (h->push_stage) (void)hipFree(h->push_stage);
        h->push_stage = nullptr; h->push_cap = 0;
        HIP_TRY(h, hipMalloc(&h->push_stage, (size_t)h->cfg.n_global * sizeof(ulonglong2)));
        h->push_cap = h->cfg.n_global;
    }
    a.extra = h->push_extra; a.pv = h->push_pv; a.skip_own = 0;
    a.wait_tot = mb_wait(h, MB_TOT);
    a.wait_cr = method == GPF_RESAMPLE_RESIDUAL ? mb_wait(h, MB_CR) : MboxWait{};
    a.tot_all = tot_all; a.cr_all = method == GPF_RESAMPLE_RESIDUAL ? cr_all : nullptr; a.stage = h->push_stage; a.counts = h->shard_counts; a.host_counts = h->h_shard_counts; a.ticket = h->push_ticket;
    return GPF_OK;
}

gpf_status gpf_shard_push_count(gpf_handle h, int32_t method, const int64_t* tot_all, const int64_t* cr_all, int32_t G, int32_t me,
                                const int64_t* bounds)
{
    gpf_status s = shard_ready(h);
    if (s) return s;
    if (!h->shard_counts) return fail(h, GPF_ERR_STATE, "gpf_shard_push_count needs gpf_shard_weight_scan of the same resample first");
    PushArgs a;
    if ((s = push_args(h, method, tot_all, cr_all, G, me, bounds, a))) return s;    // the counters were cleared by the weight scan
    h->counts_published = false;
    if (method == GPF_RESAMPLE_STRATIFIED) {
        // contiguous strata x contiguous shard ranges: the plan (served slot range, exchange counts) is closed-form; the
        // counts go to the host right away
        if (!h->shard_plan) HIP_TRY(h, hipMalloc(&h->shard_plan, sizeof(ShardPlan)));
        h->push_ticket += 1;
        a.ticket = h->push_ticket;
        s = timed(h, GPF_K_SEARCH, [&] { GPF_LAUNCH(k_strat_plan, dim3(1), dim3(128), 0, h->stream, a, h->shard_plan); });
        if (s) return s;
        HIP_TRY(h, hipGetLastError());
        h->counts_published = true;
        h->push_counted = true;
        return GPF_OK;
    }
#ifndef PUSH_SCAN_BLOCKS_PER_CU
#define PUSH_SCAN_BLOCKS_PER_CU 8
#endif
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(a.nchunks, (int64_t)h->n_cu * PUSH_SCAN_BLOCKS_PER_CU));
    a.skip_own = h->own_direct ? 1 : 0;
    if (h->own_direct) {
        // the shard's own slots: ancestors in place (k_search_own), nothing staged or packed for them; pass 1 walks the other shards' slots
        // only -- with one shard there are none
        const int gso = (int)std::max<int64_t>(1, std::min<int64_t>((h->n + GPF_MULTI_NS * SBLOCK - 1) / (GPF_MULTI_NS * SBLOCK), (int64_t)h->n_cu));
        if (method == GPF_RESAMPLE_RESIDUAL) {
            if (!h->residual_scanned) return fail(h, GPF_ERR_STATE, "residual push needs gpf_shard_residual_scan first");
            const CdfLevels lw_ = levels(h, 2), lc_ = levels(h, 1);
            const size_t lds = search_lds_bytes(h->ntiles, 2);
            s = timed(h, GPF_K_SEARCH, [&] {
                GPF_LAUNCH(k_search_own_res, dim3(gso), dim3(SBLOCK), lds, h->stream, a, lw_, lc_, h->n, h->ntiles, h->cfg.gid0, h->anc);
            });
            if (s) return s;
            if (G > 1) GPF_LAUNCH((k_push_scan<1>), dim3(grid), dim3(PUSH_SCAN_BLOCK), 0, h->stream, a);
            HIP_TRY(h, hipGetLastError());
            h->push_counted = true;
            return GPF_OK;
        }
        const CdfLevels lw_ = levels(h, 0);
        const size_t lds = multi_lds_bytes(h->ntiles, lw_.logg);
        s = timed(h, GPF_K_SEARCH, [&] {
            if (lw_.logg == 0) GPF_LAUNCH((k_search_own<0>), dim3(gso), dim3(SBLOCK), lds, h->stream, a, lw_, h->n, h->ntiles, h->cfg.gid0, h->anc);
            else               GPF_LAUNCH((k_search_own<1>), dim3(gso), dim3(SBLOCK), lds, h->stream, a, lw_, h->n, h->ntiles, h->cfg.gid0, h->anc);
        });
        if (s) return s;
        if (G > 1) GPF_LAUNCH((k_push_scan<0>), dim3(grid), dim3(PUSH_SCAN_BLOCK), 0, h->stream, a);
        HIP_TRY(h, hipGetLastError());
        h->push_counted = true;
        return GPF_OK;
    }
    s = timed(h, GPF_K_SEARCH, [&] {
        if (method == GPF_RESAMPLE_MULTINOMIAL) GPF_LAUNCH((k_push_scan<0>), dim3(grid), dim3(PUSH_SCAN_BLOCK), 0, h->stream, a);
        else                                    GPF_LAUNCH((k_push_scan<1>), dim3(grid), dim3(PUSH_SCAN_BLOCK), 0, h->stream, a);
    });
    if (s) return s;
    HIP_TRY(h, hipGetLastError());
    h->push_counted = true;
    return GPF_OK;
}

gpf_status gpf_shard_counts(gpf_handle h, int32_t G, int64_t* host_counts)
{
    gpf_status s = shard_ready(h);
    if (s) return s;
    if (!host_counts || G < 1 || G > MAX_SHARDS || !h->shard_counts || !h->push_counted) return fail(h, GPF_ERR_STATE, "no counted resample");
    if (h->counts_published) {
        // k_push publishes the counts to pinned host memory when it STARTS: poll the ticket (the kernel keeps running)
        if ((s = wait_ticket(h, h->h_shard_counts + 2 * MAX_SHARDS, h->push_ticket, "push counts"))) return s;
    } else {
        for (int k = 0; k < 2 * G; ++k)                          // (the counters sit COUNT_STRIDE words apart on the device, densely in the mirror)
            HIP_TRY(h, hipMemcpyAsync(h->h_shard_counts + k, h->shard_counts + (size_t)k * COUNT_STRIDE, sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    for (int g = 0; g < G; ++g) { host_counts[g] = h->h_shard_counts[g]; host_counts[G + g] = h->h_shard_counts[G + g]; }
    return GPF_OK;
}

gpf_status gpf_shard_push(gpf_handle h, int32_t method, const int64_t* tot_all, const int64_t* cr_all, int32_t G, int32_t me,
                          const int64_t* bounds, int64_t capacity, double* packed_out)
{
    gpf_status s = shard_ready(h);
    if (s) return s;
    if (!h->push_counted) return fail(h, GPF_ERR_STATE, "gpf_shard_push needs gpf_shard_push_count of the same resample first");
    if (capacity < 0 || (capacity > 0 && !packed_out)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad arguments");
    PushArgs a;
    if (method == GPF_RESAMPLE_STRATIFIED) {
        if (!h->shard_plan) return fail(h, GPF_ERR_STATE, "gpf_shard_push needs gpf_shard_push_count of the same resample first");
        if ((s = push_args(h, method, tot_all, cr_all, G, me, bounds, a))) return s;
        if (capacity == 0) return GPF_OK;
        // ancestors of the served slots (a streaming merge over the shard's own CDF) and, in the same kernel, their rows packed in
        // slot order; the grid is sized for the send buffer and stops at the served count, which only the device knows
        const int64_t cap = std::min<int64_t>(capacity, h->cfg.n_global);
        SearchArgs sa{};
        sa.w = levels(h, 0); sa.c = sa.w; sa.ntiles = h->ntiles;
        sa.order = nullptr; sa.sc = h->sc; sa.ws = &h->shard_plan->ws; sa.raw = &h->sc->raw; sa.plan = h->shard_plan;
        sa.n = cap; sa.n_cells = h->n; sa.n_global = h->cfg.n_global; sa.gid0 = h->cfg.gid0; sa.seed = h->cfg.seed; sa.epoch = h->epoch;
        sa.K = h->K; sa.logN = h->logN; sa.anc = nullptr; sa.invN = 1.0 / (double)h->cfg.n_global;
        sa.update_lml = 0;                                            // the commit carries the log-ML update
        sa.pack = PackOut{h->rows[h->cur], packed_out, capacity, h->cfg.gid0, h->W, h->push_extra, h->push_pv, h->own_direct ? h->anc : nullptr, (int)me};
        s = timed(h, GPF_K_GATHER, [&] {
            GPF_LAUNCH((k_search_strat<false>), dim3((unsigned)((cap + MJB_STRAT - 1) / MJB_STRAT)), dim3(MBLOCK), 0, h->stream, sa);
        });
        if (s) return s;
        HIP_TRY(h, hipGetLastError());
        return GPF_OK;
    }
    h->push_ticket += 1;
    if ((s = push_args(h, method, tot_all, cr_all, G, me, bounds, a))) return s;
    if (capacity == 0) return GPF_OK;
    h->counts_published = true;
    const bool two = method == GPF_RESAMPLE_RESIDUAL;
    if (two && !h->residual_scanned) return fail(h, GPF_ERR_STATE, "residual push needs gpf_shard_residual_scan first");
    const int64_t nt = two ? 2 : 1;
    const size_t lds = search_lds_bytes(h->ntiles, (int)nt);
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((capacity + 2 * SBLOCK - 1) / (2 * SBLOCK), (int64_t)h->n_cu));
    const CdfLevels lw_ = levels(h, two ? 2 : 0);
    const CdfLevels lc_ = levels(h, two ? 1 : 0);
    const bool narrow = method == GPF_RESAMPLE_MULTINOMIAL && lw_.off16 != nullptr && lw_.sample == 0;
    const int gridn = (int)std::max<int64_t>(1, std::min<int64_t>((capacity + 4 * SBLOCK - 1) / (4 * SBLOCK), (int64_t)h->n_cu));
    s = timed(h, GPF_K_GATHER, [&] {
        if (narrow)                                  launch_push_multi(h, a, gridn, lw_, capacity, packed_out);
        else if (method == GPF_RESAMPLE_MULTINOMIAL) launch_push<0>(h, a, grid, lds, lw_, lc_, capacity, packed_out);
        else                                         launch_push<1>(h, a, grid, lds, lw_, lc_, capacity, packed_out);
    });
    if (s) return s;
    HIP_TRY(h, hipGetLastError());
    return GPF_OK;
}

gpf_status gpf_shard_commit(gpf_handle h, const double* packed, int64_t m, const double* mf_all, const int64_t* tot_all, int32_t G)
{
    gpf_status s = shard_ready(h);
    if (s) return s;
    if (!packed || !mf_all || !tot_all || G < 1) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad arguments");
    if (m != h->n && !(h->own_direct && m >= 0 && m <= h->n)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "a shard must receive exactly one entry per output slot");
    h->pend_own = h->own_direct; h->pend_m = m;                  // (own-direct engine: m entries from the other shards, the rest through h->anc)
    h->pend_own_range = h->own_direct && h->own_direct_range;
    // Deferred like the single-GPU gather: the next gpf_update propagates the entries straight out of
    // the exchange buffer into their slots (k_step<PACKED>); any other consumer scatters first (materialize()).
    // packed / mf_all / tot_all must stay alive and unchanged until then (the caller keeps them until the next commit).
    h->pending_packed = true;
    h->pend_packed = packed; h->pend_mf = mf_all; h->pend_tot = tot_all; h->pend_G = G;
    h->pend_mailbox = h->mb_active && h->mb_engine;
    h->epoch += 1;
    h->raw_valid = false; h->raw_sum_valid = false;
    h->max_valid = false;
    h->residual_scanned = false;
    h->push_counted = false;
    mutated(h);
    return GPF_OK;
}

gpf_status gpf_shard_lml_est(gpf_handle h, double* out)
{
    gpf_status s = shard_ready(h);
    if (s) return s;
    if (!out) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null out");
    if ((s = materialize(h))) return s;                          // a deferred commit also carries the log-ML update
    if ((s = fetch_scalars(h))) return s;
    *out = h->h_sc->lml_est;
    return GPF_OK;
}

} // extern "C"

// =================================================================================== the sharded resample in one call
namespace {
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
// librccl, once per process: the copy that is already mapped (a host such as PyTorch brings its own) or the ROCm one
bool rccl_load(std::string& err)
{
    if (g_rccl.lib) return true;
    const char* names[] = {"librccl.so", "librccl.so.1"};
    void* L = nullptr;
    // tests: GPF_RCCL_LIBRARY names a library with the same nine entry points (tests/loopback_rccl: several ranks on one GPU)
    if (const char* over = getenv("GPF_RCCL_LIBRARY")) {
        if (!(L = dlopen(over, RTLD_NOW | RTLD_LOCAL))) { err = std::string("GPF_RCCL_LIBRARY: ") + dlerror(); return false; }
    }
    if (!L) for (const char* n : names) if ((L = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!L) for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) if ((L = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!L) { err = std::string("librccl not found: ") + dlerror(); return false; }
#define GPF_RCCL_SYM(field, name) *reinterpret_cast<void**>(&g_rccl.field) = dlsym(L, name); if (!g_rccl.field) { err = std::string("librccl lacks ") + name; return false; }
    GPF_RCCL_SYM(GetUniqueId, "ncclGetUniqueId") GPF_RCCL_SYM(CommInitRank, "ncclCommInitRank") GPF_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    GPF_RCCL_SYM(AllGather, "ncclAllGather") GPF_RCCL_SYM(Send, "ncclSend") GPF_RCCL_SYM(Recv, "ncclRecv")
    GPF_RCCL_SYM(GroupStart, "ncclGroupStart") GPF_RCCL_SYM(GroupEnd, "ncclGroupEnd") GPF_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef GPF_RCCL_SYM
    g_rccl.lib = L;
    return true;
}
#define NCCL_TRY(h, expr)                                                                       \
    do {                                                                                        \
        ncclResult_t r_ = (expr);                                                               \
        if (r_ != ncclSuccess) return fail(h, GPF_ERR_HIP, std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); \
    } while (0)

// all-gather of `count` elements per rank on the handle's stream; a 1-rank communicator still goes through RCCL when the
// environment asks for it (GPF_SHARD_FORCE_COLLECTIVES=1: exercises the call path on a 1-GPU box), else it is a copy
gpf_status shard_all_gather(gpf_filter* h, const void* src, void* dst, size_t count, ncclDataType_t dt, size_t elem)
{
    static const bool force = getenv("GPF_SHARD_FORCE_COLLECTIVES") && !strcmp(getenv("GPF_SHARD_FORCE_COLLECTIVES"), "1");
    if (h->comm_world == 1 && !(force && h->comm)) {
        if (dst != src) HIP_TRY(h, hipMemcpyAsync(dst, src, count * elem, hipMemcpyDeviceToDevice, h->stream));
        return GPF_OK;
    }
    NCCL_TRY(h, g_rccl.AllGather(src, dst, count, dt, h->comm, h->stream));
    return GPF_OK;
}
// The local summaries and (for the RCCL all-gathers) the gathered arrays are rings of SH_RING rounds: a prioritised resample runs
// three summary rounds (raw weights, priorities, log_ws) and its commit still reads the first.
constexpr int SH_RING = 4;
gpf_status shard_scratch(gpf_filter* h)
{
    if (h->sh_mf) return GPF_OK;
    const size_t G = (size_t)h->comm_world;
    HIP_TRY(h, hipMalloc(&h->sh_mf, SH_RING * 2 * sizeof(double)));
    HIP_TRY(h, hipMalloc(&h->sh_tot, SH_RING * 5 * sizeof(int64_t)));
    HIP_TRY(h, hipMalloc(&h->sh_cr, 2 * sizeof(int64_t)));
    if (h->comm_world == 1 && !h->comm) {                        // one shard, no communicator: the "gathered" arrays ARE the local ones
        h->sh_mf_all = h->sh_mf; h->sh_tot_all = h->sh_tot; h->sh_cr_all = h->sh_cr;
    } else {
        HIP_TRY(h, hipMalloc(&h->sh_mf_all, SH_RING * 2 * G * sizeof(double)));
        HIP_TRY(h, hipMalloc(&h->sh_tot_all, SH_RING * 5 * G * sizeof(int64_t)));
        HIP_TRY(h, hipMalloc(&h->sh_cr_all, 2 * G * sizeof(int64_t)));
    }
    HIP_TRY(h, hipMemsetAsync(h->sh_tot, 0, SH_RING * 5 * sizeof(int64_t), h->stream));
    return GPF_OK;
}
// phases 1 + 2 of DESIGN.md §6: (max, flags) and {S, sum q^2 limbs} of every shard, gathered on every rank -- through the shard
// mailboxes (peer stores from the producing kernels, waits in the consuming ones: no collective) or two RCCL all-gathers.
// Leaves h->cur_mf_all / cur_tot_all naming the gathered arrays of this round.
struct EngineScope { gpf_filter* h; explicit EngineScope(gpf_filter* f) : h(f) { h->mb_engine = true; } ~EngineScope() { h->mb_engine = false; } };
gpf_status shard_summary(gpf_filter* h, int want_q)
{
    gpf_status s = shard_scratch(h);
    if (s) return s;
    const bool mb = h->mb_active;
    const size_t G = (size_t)h->comm_world;
    const int r = (int)(h->sh_round++ % SH_RING);
    const bool alias = h->sh_mf_all == h->sh_mf;                 // one shard without communicator
    double* mf = h->sh_mf + 2 * r; int64_t* tot = h->sh_tot + 5 * r;
    double* mf_all = alias ? mf : h->sh_mf_all + 2 * G * r; int64_t* tot_all = alias ? tot : h->sh_tot_all + 5 * G * r;
    // One shard without a communicator: the first summary -- (max, flags) -- is folded inside the scan's launch (no k_pack_mflags launch).
    // With the mailboxes the same fusion is possible (GPF_SHARD_FUSE_MF=1: the scan's workgroup 0 pushes, every workgroup waits) and was
    // measured SLOWER on one rank (+6 us per step: the push's system-scope stores land on the critical path of every scan workgroup instead
    // of in an earlier launch), so the separate launch stays; as an RCCL all-gather the summary needs its own launch ahead of the collective.
    static const bool fuse_mb = getenv("GPF_SHARD_FUSE_MF") && !strcmp(getenv("GPF_SHARD_FUSE_MF"), "1");
    const bool fuse = alias || (mb && fuse_mb);
    if (fuse) { if ((s = shard_ready(h)) || (s = shard_max_slots(h))) return s; }
    else if ((s = gpf_shard_weight_max(h, mf))) return s;
    if (!mb && !alias && (s = shard_all_gather(h, mf, mf_all, 2, ncclDouble, sizeof(double)))) return s;
    struct FuseScope { gpf_filter* h; ~FuseScope() { h->fuse_mf_out = nullptr; } } fuse_scope{h};
    h->fuse_mf_out = fuse ? mf : nullptr;
    // (fused: MB_MF's round begins inside gpf_shard_weight_scan -- name the gathered array after it)
    if (!fuse) h->cur_mf_all = mb ? static_cast<const double*>(mb_gathered(h, MB_MF)) : mf_all;
    if ((s = gpf_shard_weight_scan(h, fuse ? mf_all : h->cur_mf_all, h->comm_world, want_q, tot))) return s;      // (fused + mailboxes: the callee names the gathered array)
    if (fuse) h->cur_mf_all = mb ? static_cast<const double*>(mb_gathered(h, MB_MF)) : mf_all;
    if (!mb && !alias && (s = shard_all_gather(h, tot, tot_all, 5, ncclInt64, sizeof(int64_t)))) return s;
    h->cur_tot_all = mb ? static_cast<const int64_t*>(mb_gathered(h, MB_TOT)) : tot_all;
    return GPF_OK;
}
// the same over another view of the weights (a prioritised resample: alpha lw, then log_ws), into summary slot `slot`, with or
// without the CDF levels
gpf_status shard_summary_of(gpf_filter* h, const PrioView& pv, WSum* slot, bool want_cdf)
{
    struct Scope { gpf_filter* h; ~Scope() { h->sum_pv_set = false; h->sum_slot = nullptr; h->sum_no_cdf = false; } } scope{h};
    h->sum_pv = pv; h->sum_pv_set = true; h->sum_slot = slot; h->sum_no_cdf = !want_cdf;
    return shard_summary(h, 0);
}
// the gathered summaries on the host: global max, flags, S and sum q^2
gpf_status shard_scalars(gpf_filter* h, double& m, int& flags, uint64_t& S, uint64_t& Qhi, uint64_t& Qlo)
{
    const int G = h->comm_world;
    std::vector<double> mf(2 * (size_t)G); std::vector<int64_t> tot(5 * (size_t)G);
    if (h->mb_active) {                                          // wait for the peers' entries, then out of the mailbox into plain device memory
        GPF_LAUNCH(k_mbox_collect, dim3(1), dim3(BLOCK), 0, h->stream, mb_wait(h, MB_MF), reinterpret_cast<const uint64_t*>(h->cur_mf_all),
                   reinterpret_cast<uint64_t*>(h->sh_mf_all), 2 * G, mb_wait(h, MB_TOT), reinterpret_cast<const uint64_t*>(h->cur_tot_all),
                   reinterpret_cast<uint64_t*>(h->sh_tot_all), 5 * G);
        HIP_TRY(h, hipGetLastError());
    }
    HIP_TRY(h, hipMemcpyAsync(mf.data(), h->mb_active ? h->sh_mf_all : h->cur_mf_all, mf.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(tot.data(), h->mb_active ? h->sh_tot_all : h->cur_tot_all, tot.size() * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    m = -HUGE_VAL; flags = 0; S = 0;
    unsigned __int128 Q = 0;
    for (int g = 0; g < G; ++g) {
        m = std::max(m, mf[2 * g]); flags |= (int)mf[2 * g + 1]; S += (uint64_t)tot[5 * g];
        for (int k = 0; k < 4; ++k) Q += (unsigned __int128)(uint64_t)tot[5 * g + 1 + k] << (32 * k);
    }
    if (!(flags & FLAG_NAN) && m == -HUGE_VAL) flags |= FLAG_ALL_NEGINF;
    Qhi = (uint64_t)(Q >> 64); Qlo = (uint64_t)Q;
    return check_scan_timeout(h);
}
// ---- shard mailboxes: allocation, hipIpc exchange of the handles over the communicator that was just created, peer mapping.
// Every decision is taken from data all ranks hold identically (the all-gathered packets), so either every rank ends with the
// mailboxes up or every rank stays on the RCCL all-gathers.  Any failure is soft: the collectives remain.
void mailbox_teardown(gpf_filter* h)
{
    for (void* p : h->mb_opened) (void)hipIpcCloseMemHandle(p);
    h->mb_opened.clear();
    if (h->mb_peers) (void)hipFree(h->mb_peers);
    if (h->mbox) (void)hipFree(h->mbox);
    h->mb_peers = nullptr; h->mbox = nullptr; h->mb_active = false;
}
struct MboxPacket { hipIpcMemHandle_t handle; int64_t ok; int64_t pid; };
gpf_status mailbox_setup(gpf_filter* h)
{
    const char* mode = getenv("GPF_SHARD_SUMMARY");               // "rccl": keep the all-gathers (A/B measurements, fallback drills)
    if (mode && !strcmp(mode, "rccl")) return GPF_OK;
    const int G = h->comm_world, me = h->comm_rank;
    if (!h->comm) return GPF_OK;                                  // a single shard without communicator aliases its own summaries
    const size_t bytes = (size_t)MB_TOTAL_WORDS * sizeof(uint64_t);
    int64_t ok = 1;
    // uncached device memory (peers write it, system-scope loads read it); plain device memory serves as well
    if (hipExtMallocWithFlags(reinterpret_cast<void**>(&h->mbox), bytes, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        if (hipMalloc(&h->mbox, bytes) != hipSuccess) { (void)hipGetLastError(); h->mbox = nullptr; ok = 0; }
    }
    if (h->mbox) { HIP_TRY(h, hipMemsetAsync(h->mbox, 0, bytes, h->stream)); HIP_TRY(h, hipStreamSynchronize(h->stream)); }
    std::vector<MboxPacket> all((size_t)G);
    MboxPacket mine{};
    mine.pid = (int64_t)getpid();
    if (G > 1 && ok) {
        if (hipIpcGetMemHandle(&mine.handle, h->mbox) != hipSuccess) { (void)hipGetLastError(); ok = 0; }
    }
    mine.ok = ok;
    auto gather = [&](const void* src, void* dst_host, size_t each) -> gpf_status {   // all-gather of `each` bytes per rank, host to host
        if (G == 1) { memcpy(dst_host, src, each); return GPF_OK; }
        char *dsrc = nullptr, *ddst = nullptr;
        HIP_TRY(h, hipMalloc(&dsrc, each)); HIP_TRY(h, hipMalloc(&ddst, each * G));
        HIP_TRY(h, hipMemcpyAsync(dsrc, src, each, hipMemcpyHostToDevice, h->stream));
        NCCL_TRY(h, g_rccl.AllGather(dsrc, ddst, each, ncclInt8, h->comm, h->stream));
        HIP_TRY(h, hipMemcpyAsync(dst_host, ddst, each * G, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        (void)hipFree(dsrc); (void)hipFree(ddst);
        return GPF_OK;
    };
    gpf_status s = gather(&mine, all.data(), sizeof(MboxPacket));
    if (s) { mailbox_teardown(h); return s; }                     // (a failed collective is not soft: the communicator is unusable)
    bool all_ok = true;
    for (int r = 0; r < G; ++r) {
        all_ok = all_ok && all[r].ok != 0;
        if (r != me && all[r].pid == mine.pid) all_ok = false;    // two shards in one process: hipIpc cannot map a handle of its own process
    }
    if (!all_ok) { mailbox_teardown(h); return GPF_OK; }
    std::vector<uint64_t*> peers((size_t)G, nullptr);
    int64_t opened = 1;
    for (int r = 0; r < G && opened; ++r) {
        if (r == me) { peers[r] = h->mbox; continue; }
        void* ptr = nullptr;
        if (hipIpcOpenMemHandle(&ptr, all[r].handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); opened = 0; break; }
        h->mb_opened.push_back(ptr);
        peers[r] = static_cast<uint64_t*>(ptr);
    }
    std::vector<int64_t> oks((size_t)G, 0);
    if ((s = gather(&opened, oks.data(), sizeof(int64_t)))) { mailbox_teardown(h); return s; }
    for (int r = 0; r < G; ++r) if (!oks[r]) { mailbox_teardown(h); return GPF_OK; }
    HIP_TRY(h, hipMalloc(&h->mb_peers, (size_t)G * sizeof(uint64_t*)));
    HIP_TRY(h, hipMemcpy(h->mb_peers, peers.data(), (size_t)G * sizeof(uint64_t*), hipMemcpyHostToDevice));
    for (int k = 0; k < MB_KINDS; ++k) h->mb_seq[k] = h->mb_cur[k] = 0;
    h->mb_active = true;
    return GPF_OK;
}
} // namespace

extern "C" {

/* 1: the handle's sharded resamples exchange their summaries through the shard mailboxes (peer stores, no collective);
 * 0: through RCCL all-gathers (or there is nothing to exchange: one shard) */
gpf_status gpf_comm_summary_mode(gpf_handle h, int32_t* mailbox)
{
    if (!h || !mailbox) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null argument");
    *mailbox = h->mb_active ? 1 : 0;
    return GPF_OK;
}

gpf_status gpf_comm_set_plan(gpf_handle h, int32_t plan)
{
    if (!h) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null handle");
    if (plan != GPF_SHARD_PLAN_PUSH && plan != GPF_SHARD_PLAN_PULL) return fail(h, GPF_ERR_INVALID_ARGUMENT, "exchange plan: GPF_SHARD_PLAN_PUSH or GPF_SHARD_PLAN_PULL");
    if (!h->sh_mf) return fail(h, GPF_ERR_STATE, "gpf_comm_set_plan needs gpf_comm_create first");
    h->shard_plan_kind = plan;
    return GPF_OK;
}
gpf_status gpf_comm_plan(gpf_handle h, int32_t* plan)
{
    if (!h || !plan) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null argument");
    if (!h->sh_mf) return fail(h, GPF_ERR_STATE, "gpf_comm_plan needs gpf_comm_create first");
    *plan = h->shard_plan_kind;
    return GPF_OK;
}

gpf_status gpf_comm_unique_id(void* id128)
{
    if (!id128) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null id");
    std::string err;
    if (!rccl_load(err)) return fail(nullptr, GPF_ERR_HIP, err);
    static_assert(sizeof(ncclUniqueId) == 128, "the ABI hands the id over as 128 bytes");
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return fail(nullptr, GPF_ERR_HIP, "ncclGetUniqueId failed");
    memcpy(id128, &id, sizeof(id));
    return GPF_OK;
}

gpf_status gpf_comm_create(gpf_handle h, const void* id128, int32_t rank, int32_t world)
{
    if (!h) return fail(nullptr, GPF_ERR_INVALID_ARGUMENT, "null handle");
    if (world < 1 || world > MAX_SHARDS || rank < 0 || rank >= world) return fail(h, GPF_ERR_INVALID_ARGUMENT, "bad rank / world (<= 64 shards)");
    if (h->parent) return fail(h, GPF_ERR_STATE, "a sub-state view has no communicator");
    if (h->comm || h->sh_mf) return fail(h, GPF_ERR_STATE, "the handle already has a communicator");
    if (world > 1 && !id128) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null id");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    h->comm_rank = rank; h->comm_world = world;
    if (id128) {                                                 // (world == 1 without an id: no RCCL at all)
        std::string err;
        if (!rccl_load(err)) return fail(h, GPF_ERR_HIP, err);
        ncclUniqueId id;
        memcpy(&id, id128, sizeof(id));
        NCCL_TRY(h, g_rccl.CommInitRank(&h->comm, world, id, rank));
    }
    gpf_status s = shard_scratch(h);
    if (s) return s;
    h->shard_plan_kind = GPF_SHARD_PLAN_PUSH;
    if (const char* e = getenv("GPF_SHARD_PLAN")) {
        if (!strcmp(e, "pull")) h->shard_plan_kind = GPF_SHARD_PLAN_PULL;
        else if (strcmp(e, "push")) return fail(h, GPF_ERR_INVALID_ARGUMENT, "GPF_SHARD_PLAN: push or pull");
    }
    return mailbox_setup(h);
}

gpf_status gpf_comm_destroy(gpf_handle h)
{
    if (!h) return GPF_OK;
    hipSetDevice(h->cfg.device);
    // a deferred commit (gpf_shard_resample leaves the new population in the exchange buffers: pend_packed / pend_mf / pend_tot
    // point into sh_recv / sh_send / sh_mf_all / sh_tot_all) is scattered into the filter's own rows BEFORE those buffers go
    gpf_status ms = GPF_OK;
    if (h->pending_packed && h->initialized && h->rows[0]) ms = materialize(h);
    h->pending_packed = false; h->pend_packed = nullptr; h->pend_mf = nullptr; h->pend_tot = nullptr; h->pend_G = 0;
    if (h->stream) hipStreamSynchronize(h->stream);
    mailbox_teardown(h);
    h->cur_mf_all = nullptr; h->cur_tot_all = h->cur_cr_all = nullptr;
    if (h->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
    h->comm = nullptr; h->comm_world = 1; h->comm_rank = 0;
    void* bufs[] = {h->sh_mf, h->sh_mf_all != h->sh_mf ? h->sh_mf_all : nullptr, h->sh_tot, h->sh_tot_all != h->sh_tot ? h->sh_tot_all : nullptr,
                    h->sh_cr, h->sh_cr_all != h->sh_cr ? h->sh_cr_all : nullptr, h->sh_send, h->sh_recv};
    for (void* b : bufs) if (b) (void)hipFree(b);
    h->sh_mf = h->sh_mf_all = nullptr; h->sh_tot = h->sh_tot_all = h->sh_cr = h->sh_cr_all = nullptr;
    h->sh_send = h->sh_recv = nullptr; h->sh_send_cap = h->sh_recv_cap = 0;
    return ms;
}

// Phase 3 of the PULL plan (gpf_k_shard.hpp, k_pull_scan): this shard's requests grouped by owner, the request matrix gathered on
// every rank (one all-gather of G counts + the host wait the split sizes need), the requests exchanged straight into the owners'
// staging lists, the exchange counters set from the matrix.  Leaves the handle where gpf_shard_push_count leaves it; counts[0..G) =
// entries to serve per shard, counts[G..2G) = entries to receive.  Buffers are allocated by pull_buffers before any collective.
static gpf_status pull_buffers(gpf_filter* h, int G)
{
    if (!h->pull_counts) {
        HIP_TRY(h, hipMalloc(&h->pull_counts, (size_t)MAX_SHARDS * COUNT_STRIDE * sizeof(int64_t)));
        HIP_TRY(h, hipMalloc(&h->pull_pc, (size_t)MAX_SHARDS * sizeof(int64_t)));
        HIP_TRY(h, hipMalloc(&h->pull_pc_all, (size_t)MAX_SHARDS * MAX_SHARDS * sizeof(int64_t)));
        HIP_TRY(h, hipHostMalloc(&h->h_pull_pc_all, (size_t)MAX_SHARDS * MAX_SHARDS * sizeof(int64_t)));
    }
    const int64_t want = (int64_t)G * std::max<int64_t>(h->n, 1);
    if (h->pull_req_cap < want) {
        if (h->pull_req) { HIP_TRY(h, hipStreamSynchronize(h->stream)); (void)hipFree(h->pull_req); h->pull_req = nullptr; h->pull_req_cap = 0; }
        HIP_TRY(h, hipMalloc(&h->pull_req, (size_t)want * sizeof(ulonglong2)));
        h->pull_req_cap = want;
    }
    return GPF_OK;
}
static gpf_status pull_requests(gpf_filter* h, int32_t method, const int64_t* tot_all, const int64_t* cr_all, int G, int me, const int64_t* bounds,
                                bool exchange, bool force_self, std::vector<int64_t>& counts)
{
    PushArgs a;
    gpf_status s = push_args(h, method, tot_all, cr_all, G, me, bounds, a);
    if (s) return s;
    const int64_t n = h->n;
    HIP_TRY(h, hipMemsetAsync(h->pull_counts, 0, (size_t)MAX_SHARDS * COUNT_STRIDE * sizeof(int64_t), h->stream));
    const int64_t nch = (n + PUSH_CHUNK - 1) / PUSH_CHUNK;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(nch, (int64_t)h->n_cu * PUSH_SCAN_BLOCKS_PER_CU));
    s = timed(h, GPF_K_SEARCH, [&] {
        if (method == GPF_RESAMPLE_MULTINOMIAL) GPF_LAUNCH((k_pull_scan<0>), dim3(grid), dim3(PUSH_SCAN_BLOCK), 0, h->stream, a, h->pull_req, n, h->pull_counts);
        else                                    GPF_LAUNCH((k_pull_scan<1>), dim3(grid), dim3(PUSH_SCAN_BLOCK), 0, h->stream, a, h->pull_req, n, h->pull_counts);
    });
    if (s) return s;
    GPF_LAUNCH(k_pull_counts, dim3(1), dim3(WAVE), 0, h->stream, h->pull_counts, G, h->pull_pc);
    HIP_TRY(h, hipGetLastError());
    if ((s = shard_all_gather(h, h->pull_pc, h->pull_pc_all, (size_t)G, ncclInt64, sizeof(int64_t)))) return s;
    HIP_TRY(h, hipMemcpyAsync(h->h_pull_pc_all, h->pull_pc_all, (size_t)G * G * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));                  // the host wait of this plan: the split sizes of BOTH exchanges
    const int64_t* M = h->h_pull_pc_all;                          // M[requester][owner]
    int64_t asked = 0;
    for (int g = 0; g < G; ++g) { counts[g] = M[(size_t)g * G + me]; counts[G + g] = M[(size_t)me * G + g]; asked += counts[G + g]; }
    // (as in the push plan: from the first exchange on, a local failure is remembered and the rank still joins the exchanges)
    gpf_status late = GPF_OK; std::string late_msg;
    auto remember = [&](gpf_status st) { if (st && !late) { late = st; late_msg = h->err; } };
    if (asked != n) remember(fail(h, GPF_ERR_STATE, "request counts do not add up to the shard's slots"));
    ncclResult_t first = ncclSuccess; const char* where = "";
    auto note = [&](ncclResult_t r, const char* w) { if (r != ncclSuccess && first == ncclSuccess) { first = r; where = w; } };
    if (exchange) note(g_rccl.GroupStart(), "ncclGroupStart");
    for (int g = 0; g < G; ++g) {
        ulonglong2* from = h->pull_req + (size_t)g * n;
        ulonglong2* to = h->push_stage + bounds[g];
        if (g == me && !force_self) continue;
        if (counts[G + g]) note(g_rccl.Send(from, (size_t)counts[G + g] * 2, ncclUint64, g, h->comm, h->stream), "ncclSend");
        if (counts[g] && counts[g] <= bounds[g + 1] - bounds[g]) note(g_rccl.Recv(to, (size_t)counts[g] * 2, ncclUint64, g, h->comm, h->stream), "ncclRecv");
    }
    if (exchange) note(g_rccl.GroupEnd(), "ncclGroupEnd");
    if (first != ncclSuccess) remember(fail(h, GPF_ERR_HIP, std::string(where) + ": " + g_rccl.GetErrorString(first)));
    if (!force_self && counts[me]) {                              // the shard's own requests never touch RCCL
        const hipError_t ce = hipMemcpyAsync(h->push_stage + bounds[me], h->pull_req + (size_t)me * n, (size_t)counts[me] * sizeof(ulonglong2), hipMemcpyDeviceToDevice, h->stream);
        if (ce != hipSuccess) remember(fail(h, GPF_ERR_HIP, std::string("self copy: ") + hipGetErrorString(ce)));
    }
    GPF_LAUNCH(k_pull_set_counts, dim3(1), dim3(WAVE), 0, h->stream, h->pull_pc_all, G, me, h->shard_counts);
    if (hipGetLastError() != hipSuccess) remember(fail(h, GPF_ERR_HIP, "k_pull_set_counts launch"));
    if (late) { h->err = late_msg; return late; }
    h->counts_published = false;
    h->push_counted = true;
    return GPF_OK;
}

static gpf_status shard_resample_impl(gpf_handle h, int32_t method, double priority_alpha, int32_t check, int32_t* invalid)
{
    gpf_status s = shard_ready(h);
    if (s) return s;
    if (method != GPF_RESAMPLE_MULTINOMIAL && method != GPF_RESAMPLE_RESIDUAL && method != GPF_RESAMPLE_STRATIFIED)
        return fail(h, GPF_ERR_UNKNOWN_METHOD, "Resampling method not recognized.");             // resample.jl:28
    if (!h->sh_mf) return fail(h, GPF_ERR_STATE, "gpf_shard_resample needs gpf_comm_create first");
    const int G = h->comm_world, me = h->comm_rank;
    // priority_fn = w -> alpha w (resample.jl:51-52): ancestors from the priorities' CDF, the log-ML update from the RAW weights
    // (:57), new log-weights log_ws + (log N - logsumexp(log_ws)) with log_ws = lw[a] - lp[a] (:198-200).  Across shards that is
    // three summary rounds instead of one (raw weights, priorities, log_ws) and one more double per exchanged entry (log_ws: the
    // receiver does not hold its ancestors' weights); the commit cannot be deferred (the weights need the third round).
    const bool prio = priority_alpha == priority_alpha;
    const int64_t n = h->n, E = h->W + 1 + (prio ? 1 : 0);
    EngineScope engine(h);                                        // the phases below push / wait through the shard mailboxes when they are up
    // shard bounds from the contiguous-range rule every rank applies to its own gpf_config (ranks ordered by gid0)
    std::vector<int64_t> bounds((size_t)G + 1);
    {
        const int64_t base = h->cfg.n_global / G, extra = h->cfg.n_global % G;
        for (int g = 0; g <= G; ++g) bounds[g] = (int64_t)g * base + std::min<int64_t>(g, extra);
        if (bounds[me] != h->cfg.gid0 || bounds[me + 1] != h->cfg.gid0 + n)
            return fail(h, GPF_ERR_STATE, "this shard's (gid0, n_particles) is not rank's contiguous share of n_global");
    }
    // Everything that can fail for reasons of THIS rank alone (allocations) happens before the first collective of the call: a
    // rank that returned early would leave its peers blocked in a collective it never joins.  The send buffer holds a balanced
    // exchange with slack -- or, when that is cheap against the HBM at hand (<= 1/16 of the free memory), one entry per GLOBAL slot,
    // the most any shard can ever serve, so that the overflow re-push below never has to allocate.
    auto ensure = [&](double*& buf, int64_t& cap, int64_t want) -> gpf_status {
        if (cap >= want) return GPF_OK;
        if (buf) { HIP_TRY(h, hipStreamSynchronize(h->stream)); (void)hipFree(buf); buf = nullptr; cap = 0; }
        HIP_TRY(h, hipMalloc(&buf, (size_t)want * (h->W + 2) * sizeof(double)));   // (room for the widest entry: a prioritised resample's [row | meta | log_ws])
        cap = want;
        return GPF_OK;
    };
    int64_t cap = std::min<int64_t>(h->cfg.n_global, 2 * n + 65536);
    if (h->sh_send_cap < h->cfg.n_global) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (size_t)h->cfg.n_global * (h->W + 2) * sizeof(double) <= free_b / 16) cap = h->cfg.n_global;
    } else cap = h->cfg.n_global;
    if (const char* e = getenv("GPF_PUSH_CAPACITY")) cap = atoll(e);                                  // tests: force the overflow path
    if ((s = ensure(h->sh_send, h->sh_send_cap, std::max<int64_t>(cap, 1)))) return s;
    static const bool force = getenv("GPF_SHARD_FORCE_COLLECTIVES") && !strcmp(getenv("GPF_SHARD_FORCE_COLLECTIVES"), "1");
    const bool exchange = G > 1 || (force && h->comm);            // one shard: what it "sends" is what it "receives"
    if (exchange && (s = ensure(h->sh_recv, h->sh_recv_cap, n))) return s;
    const bool pull = h->shard_plan_kind == GPF_SHARD_PLAN_PULL && method != GPF_RESAMPLE_STRATIFIED;
    if (pull && (s = pull_buffers(h, G))) return s;
    // Own-direct (multinomial, push plan, no priorities): a slot of this shard whose target falls into this shard's own part of the CDF is
    // resolved in place -- its ancestor goes into h->anc and the next propagate gathers the row through it, as on an unsharded filter;
    // only the slots other shards serve travel as packed entries.  On one rank nothing is staged, packed, counted or waited for.
    static const bool own_off = getenv("GPF_SHARD_OWN") && !strcmp(getenv("GPF_SHARD_OWN"), "0");        // (A/B measurements; tests of the packed path)
    const bool own = !own_off && !prio && ((method == GPF_RESAMPLE_MULTINOMIAL && !pull && multi_logg(h->ntiles) >= 0 &&
                                            multi_lds_bytes(h->ntiles, multi_logg(h->ntiles)) + 4096 <= (size_t)160 * 1024) ||
                                           (method == GPF_RESAMPLE_RESIDUAL && !pull && search_lds_bytes(h->ntiles, 2) + 40 * 1024 <= (size_t)160 * 1024) ||
                                           method == GPF_RESAMPLE_STRATIFIED);
    struct OwnScope { gpf_filter* h; ~OwnScope() { h->own_direct = false; h->own_direct_range = false; } } own_scope{h};
    h->own_direct = own; h->own_direct_range = own && method == GPF_RESAMPLE_STRATIFIED;

    const double* raw_mf = nullptr; const int64_t* raw_tot = nullptr;
    struct PushScope { gpf_filter* h; ~PushScope() { h->push_extra = 0; } } push_scope{h};
    if (prio) {
        if ((s = materialize(h))) return s;
        h->want_offsets = false;
        s = shard_summary_of(h, raw_view(h), &h->sc->raw, false);  // logsumexp(log_weights), every shard (resample.jl:180)
        h->want_offsets = true;
        if (s) return s;
        raw_mf = h->cur_mf_all; raw_tot = h->cur_tot_all;
        h->push_extra = 1; h->push_pv = PrioView{h->lw, nullptr, priority_alpha, 1};
    }
    h->want_offsets = method == GPF_RESAMPLE_MULTINOMIAL;         // (the offset levels serve k_push_multi only)
    s = prio ? shard_summary_of(h, h->push_pv, &h->sc->prio, true) : shard_summary(h, 0);   // phases 1, 2 (safe_softmax of the priorities, :54)
    h->want_offsets = true;
    if (s) return s;
    if (check != GPF_CHECK_FALSE || invalid) {                    // safe_softmax validity (utils.jl:117-140): pinned flags, no stream sync
        // (the flags describe the GLOBAL weights: every rank takes the same branch here)
        int32_t flags = 0;
        if ((s = gpf_shard_flags(h, &flags))) return s;
        if (invalid) *invalid = flags != 0;
        if (flags & (FLAG_NAN | FLAG_POSINF)) return fail(h, GPF_ERR_INVALID_WEIGHTS, "Invalid weights (NaN).");
        if (check == GPF_CHECK_TRUE && flags) return fail(h, GPF_ERR_INVALID_WEIGHTS, "Invalid weights.");   // resample.jl:55
    }
    const int64_t* cr_all = nullptr;
    const int64_t* tot_all = h->cur_tot_all;
    if (method == GPF_RESAMPLE_RESIDUAL) {                        // phase 2b
        if ((s = gpf_shard_residual_scan(h, tot_all, G, h->sh_cr))) return s;
        if (!h->mb_active && (s = shard_all_gather(h, h->sh_cr, h->sh_cr_all, 2, ncclInt64, sizeof(int64_t)))) return s;
        cr_all = h->cur_cr_all = h->mb_active ? static_cast<const int64_t*>(mb_gathered(h, MB_CR)) : h->sh_cr_all;
    }
    std::vector<int64_t> counts(2 * (size_t)G);
    int64_t pushed_cap = std::min(cap, h->sh_send_cap);
    int64_t n_send = 0, n_recv = 0;
    if (pull) {
        // phase 3 of the pull plan: requests out, counts known on the host BEFORE pass 2 is enqueued (no speculative capacity)
        if ((s = pull_requests(h, method, tot_all, cr_all, G, me, bounds.data(), exchange, force && G == 1, counts))) return s;
        for (int g = 0; g < G; ++g) { n_send += counts[g]; n_recv += counts[G + g]; }
        pushed_cap = std::min(n_send, h->sh_send_cap);
        if ((s = gpf_shard_push(h, method, tot_all, cr_all, G, me, bounds.data(), pushed_cap, h->sh_send))) return s;
    } else {
        if ((s = gpf_shard_push_count(h, method, tot_all, cr_all, G, me, bounds.data()))) return s;   // phase 3
        if (own && G == 1) {
            // one shard, own-direct: every slot is an own hit -- nothing to exchange, so no split sizes to wait for (the host wait left a
            // gap in the queue on every resample); the i.i.d. methods have nothing to push either, stratified writes its ancestors in
            // place from the merge kernel
            counts[0] = method == GPF_RESAMPLE_STRATIFIED ? n : 0; counts[1] = n;
            if (method == GPF_RESAMPLE_STRATIFIED && (s = gpf_shard_push(h, method, tot_all, cr_all, G, me, bounds.data(), pushed_cap, h->sh_send))) return s;
        } else {
        // phase 4 is enqueued before the host learns the counts; the kernel stops at the capacity and the push is repeated if the
        // counts say it overflowed
        if ((s = gpf_shard_push(h, method, tot_all, cr_all, G, me, bounds.data(), pushed_cap, h->sh_send))) return s;
        if ((s = gpf_shard_counts(h, G, counts.data()))) return s;   // ONE host wait (the exchange's split sizes), behind phase 4
        }
        for (int g = 0; g < G; ++g) { n_send += counts[g]; n_recv += counts[G + g]; }
    }
    // From here on a local failure is REMEMBERED and the rank still joins the exchange with the counts its peers expect (they
    // worked out their receive counts themselves and will wait for exactly that many entries): the error is returned after the
    // group has closed.  A failed gpf_shard_resample leaves the communicator and the filter unusable on every rank that sees
    // one (the entries a failing rank sends are undefined): the host must tear the job down.
    gpf_status late = GPF_OK;
    std::string late_msg;
    auto remember = [&](gpf_status st) { if (st && !late) { late = st; late_msg = h->err; } };
    if (n_recv != n) remember(fail(h, GPF_ERR_STATE, "exchange counts do not add up to the shard's slots"));
    if (n_send > h->sh_send_cap) {                                // skewed weights: this shard serves more than its buffer held
        gpf_status es = ensure(h->sh_send, h->sh_send_cap, n_send);
        if (es) {                                                 // cannot hold what the peers expect: nothing sane can be sent
            if (exchange) remember(fail(h, GPF_ERR_HIP, "send buffer for a skewed exchange could not be allocated; the communicator is poisoned (peers are waiting)"));
            return es;
        }
    }
    if (n_send > pushed_cap)
        remember(gpf_shard_push(h, method, tot_all, cr_all, G, me, bounds.data(), n_send, h->sh_send));
    // the exchange: [row | slot | ancestor id], grouped point-to-point sends and receives (one pair per PEER; the shard's own
    // entries never touch RCCL: one device-to-device copy on the same stream)
    const double* commit_from = h->sh_send;
    if (exchange) {
        commit_from = h->sh_recv;
        ncclResult_t first = ncclSuccess;
        const char* where = "";
        auto note = [&](ncclResult_t r, const char* w) { if (r != ncclSuccess && first == ncclSuccess) { first = r; where = w; } };
        const bool recv_fits = n_recv <= h->sh_recv_cap;
        int64_t so = 0, ro = 0, self_so = -1, self_ro = -1;
        note(g_rccl.GroupStart(), "ncclGroupStart");
        for (int g = 0; g < G; ++g) {
            if (g == me && own) { so += counts[g]; continue; }    // own hits never enter the exchange (their places in the send buffer stay unused; counts[G + me] of them sit in h->anc)
            if (g == me && !(force && G == 1)) { self_so = so; self_ro = ro; }
            else {
                if (counts[g]) note(g_rccl.Send(h->sh_send + so * E, (size_t)(counts[g] * E), ncclDouble, g, h->comm, h->stream), "ncclSend");
                if (counts[G + g] && recv_fits) note(g_rccl.Recv(h->sh_recv + ro * E, (size_t)(counts[G + g] * E), ncclDouble, g, h->comm, h->stream), "ncclRecv");
            }
            so += counts[g]; ro += counts[G + g];
        }
        note(g_rccl.GroupEnd(), "ncclGroupEnd");                  // ALWAYS closed: librccl is shared with the host (PyTorch); an open group
                                                                  // would swallow every later RCCL call of this thread
        if (first != ncclSuccess) remember(fail(h, GPF_ERR_HIP, std::string(where) + ": " + g_rccl.GetErrorString(first)));
        if (self_so >= 0 && counts[me] && recv_fits) {
            const hipError_t ce = hipMemcpyAsync(h->sh_recv + self_ro * E, h->sh_send + self_so * E, (size_t)counts[me] * E * sizeof(double), hipMemcpyDeviceToDevice, h->stream);
            if (ce != hipSuccess) remember(fail(h, GPF_ERR_HIP, std::string("self copy: ") + hipGetErrorString(ce)));
        }
    }
    if (late) { h->err = late_msg; return late; }
    if (!prio) return gpf_shard_commit(h, commit_from, own ? n - counts[(size_t)G + me] : n, h->cur_mf_all, tot_all, G);   // phase 5 (deferred)
    // phase 5 of a prioritised resample, at once: scatter rows / parents / log_ws, log-ML from the raw summary ...
    {
        const int grid = grid_for(h, n, 8);
        double* out = h->rows[1 - h->cur];
        const int mbx = (int)h->mb_active;
        switch (h->W) {
            case 2: GPF_LAUNCH((k_commit_packed_ws<2>), dim3(grid), dim3(BLOCK), 0, h->stream, commit_from, n, out, h->anc, h->lws, raw_mf, raw_tot, G, h->K, h->logN, h->sc, mbx); break;
            case 4: GPF_LAUNCH((k_commit_packed_ws<4>), dim3(grid), dim3(BLOCK), 0, h->stream, commit_from, n, out, h->anc, h->lws, raw_mf, raw_tot, G, h->K, h->logN, h->sc, mbx); break;
            case 8: GPF_LAUNCH((k_commit_packed_ws<8>), dim3(grid), dim3(BLOCK), 0, h->stream, commit_from, n, out, h->anc, h->lws, raw_mf, raw_tot, G, h->K, h->logN, h->sc, mbx); break;
        }
        HIP_TRY(h, hipGetLastError());
        h->cur ^= 1;
    }
    // ... then logsumexp(log_ws) over all shards and lw = log_ws + (log N - logsumexp) (resample.jl:200)
    h->want_offsets = false;
    s = shard_summary_of(h, PrioView{h->lws, nullptr, 0.0, 0}, &h->sc->post, false);
    h->want_offsets = true;
    if (s) return s;
    GPF_LAUNCH(k_shard_apply_post, dim3(grid_for(h, n, 8)), dim3(BLOCK), 0, h->stream, h->cur_mf_all, h->cur_tot_all, G, h->K, h->logN, h->lws, h->lw, n,
               mb_wait(h, MB_TOT));
    HIP_TRY(h, hipGetLastError());
    h->epoch += 1;
    h->raw_valid = false; h->raw_sum_valid = false; h->max_valid = false; h->residual_scanned = false; h->push_counted = false;
    mutated(h);
    return GPF_OK;
}

gpf_status gpf_shard_resample(gpf_handle h, int32_t method, int32_t check, int32_t* invalid)
{
    return shard_resample_impl(h, method, std::nan(""), check, invalid);
}
gpf_status gpf_shard_resample_tempered(gpf_handle h, int32_t method, double priority_alpha, int32_t check, int32_t* invalid)
{
    if (!(priority_alpha == priority_alpha)) return fail(h, GPF_ERR_INVALID_ARGUMENT, "priority_alpha is NaN: use gpf_shard_resample for priority_fn = nothing");
    return shard_resample_impl(h, method, priority_alpha, check, invalid);
}


gpf_status gpf_shard_effective_sample_size(gpf_handle h, double* out)
{
    gpf_status s = shard_ready(h);
    if (s) return s;
    if (!out) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null out");
    if (!h->sh_mf) return fail(h, GPF_ERR_STATE, "needs gpf_comm_create first");
    EngineScope engine(h);
    if ((s = shard_summary(h, 1))) return s;
    double m; int flags; uint64_t S, Qhi, Qlo;
    if ((s = shard_scalars(h, m, flags, S, Qhi, Qlo))) return s;
    *out = flags ? std::nan("") : ess_from(S, Qhi, Qlo);
    return GPF_OK;
}

gpf_status gpf_shard_log_ml_estimate(gpf_handle h, double* out)
{
    gpf_status s = shard_ready(h);
    if (s) return s;
    if (!out) return fail(h, GPF_ERR_INVALID_ARGUMENT, "null out");
    if (!h->sh_mf) return fail(h, GPF_ERR_STATE, "needs gpf_comm_create first");
    EngineScope engine(h);
    if ((s = shard_summary(h, 0))) return s;
    double m; int flags; uint64_t S, Qhi, Qlo;
    if ((s = shard_scalars(h, m, flags, S, Qhi, Qlo))) return s;
    double base;
    if ((s = gpf_shard_lml_est(h, &base))) return s;
    *out = base + lse_from(m, S, h->K, flags) - h->logN;
    return GPF_OK;
}

} // extern "C"

// =================================================================================== host scalar spec
extern "C" {
int32_t gpf_host_fix_K(int64_t n_global) { return fix_K(n_global); }
int32_t gpf_host_gamma_E(int64_t n_tiles) { return gamma_E(n_tiles); }
uint64_t gpf_host_div128(uint64_t P, uint64_t den) { return div128(P, div128_setup(den)); }
uint64_t gpf_host_muldiv128(uint64_t p, uint64_t W, uint64_t den) { return muldiv128(p, W, div128_setup(den)); }
uint64_t gpf_host_gamma_tile(uint64_t seed, uint32_t gid, uint32_t epoch, int64_t shape, int32_t Eg) { return gamma_tile(seed, gid, epoch, shape, Eg); }
double gpf_host_log(double x) { return log_(x); }
double gpf_host_lse(double m, uint64_t S, int32_t K, int32_t flags)
{
    int f = flags;
    if (!(f & FLAG_NAN) && m == -__builtin_huge_val()) f |= FLAG_ALL_NEGINF;
    return lse_from(m, S, K, f);
}
double gpf_host_ess(uint64_t S, uint64_t Q_hi, uint64_t Q_lo) { return ess_from(S, Q_hi, Q_lo); }
void gpf_host_math(int32_t which, const double* a, const double* b, int64_t n, double* out, double* out2)
{
    for (int64_t i = 0; i < n; ++i) {
        switch (which) {
            case 0: out[i] = exp_(a[i]); break;
            case 1: out[i] = log_(a[i]); break;
            case 2: sincos2pi(a[i], out[i], out2[i]); break;
            case 3: out[i] = atan2_(a[i], b[i]); break;
            case 4: out[i] = sqrt_(a[i]); break;
            case 5: out[i] = a[i] / b[i]; break;
            case 7: out[i] = neglog_u52(d2u(a[i])); break;
            default: out[i] = 0.0;
        }
    }
}

} // extern "C"

#ifdef GPF_DBG_STRAT
extern "C" int gpf_debug_strat(unsigned long long* out, int n_words)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gpf::g_dbg_strat), (size_t)n_words * sizeof(unsigned long long));
}
#endif

#ifdef GPF_DBG_SORT
extern "C" int gpf_debug_sort_buckets(unsigned long long* out, int n_words)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gpf::g_dbg_bk), (size_t)n_words * sizeof(unsigned long long));
}
extern "C" int gpf_debug_sort(unsigned long long* out, int n_words)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gpf::g_dbg_sort), (size_t)n_words * sizeof(unsigned long long));
}
#endif
