// K5a + K6 + K2 in one launch: the i.i.d. ancestor search fused into the propagate ("lazy search")  (part of gpf_kernels.hpp; include that header, not this file)
#pragma once

namespace gpf {
// pf_resample!(state, :multinomial) followed by pf_update! (README.md:60-79; BASELINE configs[1]): the resample enqueues only the
// weight scan and leaves (CDF levels, slot stream) pending; this kernel draws every slot's target (resample.jl:59), finds its ancestor
// (the key-table search of k_search_multi: LDS key table, two narrow reads), reads row `a` (new_traces .= view(traces, parents),
// resample.jl:60), propagates it (update.jl:15-22), writes the new row, the log-weight (incoming weights are 0, resample.jl:195) and
// parents[j] = a.  Against k_search_multi + k_step<GATHER>: one launch boundary and the 4-byte ancestor round trip are gone, and the
// Philox / Box-Muller work of a slot -- which does not depend on its ancestor -- overlaps the search's dependent round trips.
// Any other consumer of the resampled population (getters, views, rejuvenation, a second resample) runs the stand-alone search first
// (libgpf_resample.hip finish_search), exactly as materialize() runs the stand-alone gather.
//
// One 1024-thread workgroup per CU (the key table is copied into LDS once per CU), NS slots per lane and iteration.
template <int NW>
__device__ __forceinline__ void block_max_store_n(double m, int f, MaxSlots ms)
{
    m = wave_max_f64(m);
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) f |= __shfl_xor(f, s, WAVE);
    __shared__ double sm_[NW];
    __shared__ int sf_[NW];
    if (lane_id() == 0) { sm_[wave_id()] = m; sf_[wave_id()] = f; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < NW; ++w) { m = sm_[w] > m ? sm_[w] : m; f |= sf_[w]; }
        unsigned long long* slot = ms.cur + (blockIdx.x % MAX_SLOTS) * SLOT_WORDS;
        atomicMax(slot, max_key(m));
        if (f) atomicOr(slot + 1, (unsigned long long)f);
    }
    if (blockIdx.x == 0 && threadIdx.x < MAX_SLOTS) { ms.clear[threadIdx.x * SLOT_WORDS] = 0; ms.clear[threadIdx.x * SLOT_WORDS + 1] = 0; }
}

// Software pipeline, every wave the same program: trip i looks up the lane's slots of chunk i + 1 while the rows of chunk i are in
// flight, issues the row reads of chunk i + 1, then propagates chunk i -- the fabric (the gather of i.i.d. 16-byte rows drags a
// 128-byte line each: the 21 us floor of k_step<GATHER>) always has one or two chunks of rows in flight, behind the searches' latency
// chains and the propagations' ALU work.  Measured alternatives (N = 10^6, profiles/r04_lazy_search.txt): search then propagate in
// every wave without the pipeline 38.7 us; 8 searcher + 8 propagator waves with an ancestor ring in LDS and a barrier per chunk 42.1 us;
// the two separate kernels 36.9 us (+ a launch boundary).
#ifndef GPF_FUSED_NS
#define GPF_FUSED_NS 1
#endif
constexpr int FNS = GPF_FUSED_NS;                  // slots per lane and chunk
constexpr int FCH = SBLOCK * FNS;                  // slots per chunk
constexpr int FUSED_LDS_EXTRA = 0;
template <int M, int W, bool KEEP_PREV, int LOGG>
__global__ __launch_bounds__(SBLOCK, 4) void k_step_search(ModelArgs a, uint64_t seed, uint32_t epoch, SearchArgs sa,
                                                           const double* __restrict__ rows_in, double* __restrict__ rows_out,
                                                           double* __restrict__ lw, MaxSlots ms, uint32_t table_words)
{
    using Mo = Model<M>;
    constexpr int D = Mo::D;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    (void)table_words;
    // update_lml_est! (resample.jl:57,178-182) of the pending resample, once
    if (sa.update_lml && blockIdx.x == 0 && threadIdx.x == 0) resample_bookkeeping(sa);
    const uint64_t S = sa.ws->S;
    auto targets = [&](int64_t base, uint64_t (&T)[FNS]) {             // one Philox block per aligned slot pair (gpf_math.hpp resample_u64_run)
        if constexpr (FNS % 2 == 0) {
            uint64_t U[FNS];
            resample_u64_run<FNS>(sa.seed, (uint32_t)(sa.gid0 + base + FNS * (int64_t)threadIdx.x), sa.epoch, U);
#pragma unroll
            for (int u = 0; u < FNS; ++u) T[u] = mulhi64(U[u], S);   // resample.jl:59
        } else {
#pragma unroll
            for (int u = 0; u < FNS; ++u) T[u] = mulhi64(resample_u64(sa.seed, (uint32_t)(sa.gid0 + base + FNS * (int64_t)threadIdx.x + u), sa.epoch), S);   // resample.jl:59
        }
    };
    auto fetch = [&](const uint32_t (&idx)[FNS], double (&r)[FNS][W]) {   // new_traces .= view(traces, parents): issue, do not wait
#pragma unroll
        for (int u = 0; u < FNS; ++u) {
            const double2* src = reinterpret_cast<const double2*>(rows_in + (int64_t)idx[u] * W);
#pragma unroll
            for (int q = 0; q < (D + 1) / 2; ++q) { const double2 v = src[q]; r[u][2 * q] = v.x; r[u][2 * q + 1] = v.y; }
        }
    };
    const int64_t stride = (int64_t)gridDim.x * FCH;
    const int64_t base0 = (int64_t)blockIdx.x * FCH;
    uint64_t T[FNS];
    const MultiTable tb = multi_table_load<LOGG>(sa.w, sa.ntiles, S, reinterpret_cast<uint32_t*>(smem), [&]() { targets(base0, T); });
    double bm = -__builtin_huge_val(); int bf = 0;
    uint32_t idx[FNS]; double r[FNS][W];
    if (base0 < sa.n) {
        multi_lookup<LOGG, FNS>(tb, sa.w, sa.n_cells, T, idx);         // wave-collective: every lane takes part, slots beyond n discard theirs
        fetch(idx, r);
    }
    for (int64_t base = base0; base < sa.n; base += stride) {
        uint32_t idn[FNS]; double rn[FNS][W];
        const bool more = base + stride < sa.n;                        // workgroup-uniform
        if (more) {
            targets(base + stride, T);
            multi_lookup<LOGG, FNS>(tb, sa.w, sa.n_cells, T, idn);     // (the rows of this chunk are in flight meanwhile)
            fetch(idn, rn);
        }
        const int64_t j0 = base + FNS * (int64_t)threadIdx.x;
#pragma unroll
        for (int u = 0; u < FNS; ++u) {
            const int64_t j = j0 + u;
            if (j >= sa.n) break;
            double xn[MAX_DIM];
            Mo::sample(a.P, false, r[u], a.obs, seed, (uint32_t)(sa.gid0 + j), 0, epoch, TAG_UPDATE, xn);   // update.jl:15-22
            const double ll = Mo::loglik(a.P, xn, a.obs);
            double o[W];
#pragma unroll
            for (int k = 0; k < W; ++k) o[k] = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) o[k] = xn[k];
            if (KEEP_PREV) {
#pragma unroll
                for (int k = 0; k < D; ++k) o[D + k] = r[u][k];
            }
            double2* dst = reinterpret_cast<double2*>(rows_out + j * W);
#pragma unroll
            for (int q = 0; q < W / 2; ++q) dst[q] = make_double2(o[2 * q], o[2 * q + 1]);
            lw[j] = ll;                                                // the incoming log-weights are 0 (resample.jl:195)
            sa.anc[j] = (int32_t)idx[u];                               // state.parents
            track_max(ll, bm, bf);
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < FNS; ++u) {
                idx[u] = idn[u];
#pragma unroll
                for (int k = 0; k < W; ++k) r[u][k] = rn[u][k];
            }
        }
    }
    block_max_store_n<SBLOCK / WAVE>(bm, bf, ms);
}

} // namespace gpf
