// K6/K9: stand-alone gather + reweight, statistics, small utilities  (part of gpf_kernels.hpp; include that header, not this file)
#pragma once

namespace gpf {
// ----------------------------------------------------------------------------- K6: gather + reweight
// new_traces .= view(traces, parents) (resample.jl:60 / :103,114 / :169) as a real row copy, fused
// with update_weights! (resample.jl:190-202): no priorities -> lw = 0; priorities -> log_ws = lw[a] - lp[a].
// One lane per 16-byte row chunk: W/2 consecutive lanes move one row.
template <int W>
__global__ __launch_bounds__(BLOCK) void k_gather(const int32_t* __restrict__ anc, const double* __restrict__ rows_in,
                                                  double* __restrict__ rows_out, PrioView pv,
                                                  double* __restrict__ lw_out, int64_t n)
{
    constexpr int C = W / 2;
    const int64_t total = n * C;
    for (int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x; t < total; t += (int64_t)gridDim.x * BLOCK) {
        const int64_t j = t / C;
        const int c = (int)(t - j * C);
        const int64_t a = anc[j];
        const double2 v = reinterpret_cast<const double2*>(rows_in)[a * C + c];
        reinterpret_cast<double2*>(rows_out)[t] = v;
        if (c == 0) lw_out[j] = pv.mode == 0 ? 0.0 : pv.lw[a] - pv.at(a);
    }
}

// lw = log_ws + (log N - logsumexp(log_ws))   (resample.jl:200)
static __global__ __launch_bounds__(BLOCK) void k_apply_post(const Scalars* sc, int K, double logN, const double* __restrict__ lws,
                                                      double* __restrict__ lw, int64_t n)
{
    const double off = logN - lse_from(sc->post.m, sc->post.S, K, sc->post.flags);
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK)
        lw[i] = lws[i] + off;
}

// ----------------------------------------------------------------------------- K9: statistics
// sum_i w_i f(x_i), w_i = q_i / S (statistics.jl:13-14, 48-50, 91-101).  The reference adds the terms one after the other; a
// parallel machine cannot, and Float64 addition is not associative -- so the ORDER is part of the spec: the
// terms are summed by the perfect binary tree over their indices (neighbours first: t0 + t1, t2 + t3, then pairs of pairs, ...),
// evaluated in chunks of 2048 terms (missing terms are +0.0) whose partials are summed by the same tree again.  Every level of
// it maps onto the machine -- 8 consecutive terms per lane, an xor-butterfly over the 64 lanes (lane l holds terms 8 l .. 8 l + 7),
// the 4 waves of a workgroup, one workgroup per chunk -- and onto two nested loops in the oracle (o_tree_sum): the same bits on
// any number of threads, and an error bound of O(log n) ulps instead of the sequential sum's O(n).
constexpr int TREE_CHUNK = BLOCK * 8;              // terms per workgroup
static_assert(TREE_CHUNK == 2048, "the oracle's o_tree_sum uses the same chunk");
__device__ __forceinline__ double tree8(const double (&t)[8]) { return ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7])); }
// block-collective: the tree sum of the workgroup's 2048 terms (thread x holds terms 8x .. 8x+7 as their subtree's sum); valid in thread 0
__device__ __forceinline__ double tree_block_sum(double v)
{
#pragma unroll
    for (int m = 1; m < WAVE; m <<= 1) v += u2d(shfl_xor_u64(d2u(v), m));       // neighbours first
    __shared__ double s_tree[NWAVES];
    if (lane_id() == 0) s_tree[wave_id()] = v;
    __syncthreads();
    static_assert(NWAVES == 4, "two more levels");
    return (s_tree[0] + s_tree[1]) + (s_tree[2] + s_tree[3]);
}
// the terms: values[i * stride + col] (a column of the particle rows, or a plain array with stride 1, col 0)
// pw = 1: w v;  2: w (v - *center)^2;  3: w [v == match]  (proportionmap)
static __global__ __launch_bounds__(BLOCK) void k_wsum_tree(const double* __restrict__ lw, const WSum* ws, int K,
                                                     const double* __restrict__ values, int stride, int col, int64_t n,
                                                     int pw, const double* center, double match, double* __restrict__ partial)
{
    const double m = ws->m;
    const double Sd = (double)ws->S;
    const bool uniform = (ws->flags & FLAG_ALL_NEGINF) != 0;
    const double c = center ? *center : 0.0;
    double t[8];
    const int64_t i0 = (int64_t)blockIdx.x * TREE_CHUNK + (int64_t)threadIdx.x * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t i = i0 + j;
        t[j] = 0.0;
        if (i < n) {
            const uint64_t q = uniform ? 1 : exp_fix(lw[i] - m, K);
            double v = values[i * stride + col];
            if (pw == 2) { v = v - c; v = v * v; }
            if (pw == 3) v = (v == match) ? 1.0 : 0.0;
            t[j] = ((double)q / Sd) * v;
        }
    }
    const double r = tree_block_sum(tree8(t));
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}
// the next level of the tree: np partials -> ceil(np / 2048)
static __global__ __launch_bounds__(BLOCK) void k_tree_partials(const double* __restrict__ in, int64_t np, double* __restrict__ out)
{
    double t[8];
    const int64_t i0 = (int64_t)blockIdx.x * TREE_CHUNK + (int64_t)threadIdx.x * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = i0 + j < np ? in[i0 + j] : 0.0;
    const double r = tree_block_sum(tree8(t));
    if (threadIdx.x == 0) out[blockIdx.x] = r;
}

// ----------------------------------------------------------------------------- strided sub-state views
// state[start:step:stop] (reference src/view.jl:35-48; test/initialize.jl:60, test/update.jl:33): a strided view works on a
// compact copy of its particles -- view_enter gathers rows / log-weights / parents of particles start + i*step, view_exit
// scatters them back (update_refs! for sub-states copies back as well, utils.jl:17-20).  to_view: parent -> compact, else back.
static __global__ __launch_bounds__(BLOCK) void k_view_strided_copy(double* __restrict__ prow, double* __restrict__ plw, int32_t* __restrict__ panc,
                                                             double* __restrict__ vrow, double* __restrict__ vlw, int32_t* __restrict__ vanc,
                                                             int W, int64_t step, int64_t n, int to_view)
{
    const int C = W / 2;
    const int64_t total = n * C;
    for (int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x; t < total; t += (int64_t)gridDim.x * BLOCK) {
        const int64_t j = t / C;
        const int c = (int)(t - j * C);
        double2* pp = reinterpret_cast<double2*>(prow) + (j * step) * C + c;
        double2* vp = reinterpret_cast<double2*>(vrow) + t;
        if (to_view) *vp = *pp; else *pp = *vp;
        if (c == 0) {
            if (to_view) { vlw[j] = plw[j * step]; vanc[j] = panc[j * step]; }
            else { plw[j * step] = vlw[j]; panc[j * step] = vanc[j]; }
        }
    }
}

// state[idxs] for an arbitrary vector of distinct indices (src/view.jl:35-48): the same compact-copy mechanism with an index array
static __global__ __launch_bounds__(BLOCK) void k_view_index_copy(double* __restrict__ prow, double* __restrict__ plw, int32_t* __restrict__ panc,
                                                           double* __restrict__ vrow, double* __restrict__ vlw, int32_t* __restrict__ vanc,
                                                           int W, const int32_t* __restrict__ idx, int64_t n, int to_view)
{
    const int C = W / 2;
    const int64_t total = n * C;
    for (int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x; t < total; t += (int64_t)gridDim.x * BLOCK) {
        const int64_t j = t / C;
        const int c = (int)(t - j * C);
        const int64_t pj = idx[j];
        double2* pp = reinterpret_cast<double2*>(prow) + pj * C + c;
        double2* vp = reinterpret_cast<double2*>(vrow) + t;
        if (to_view) *vp = *pp; else *pp = *vp;
        if (c == 0) {
            if (to_view) { vlw[j] = plw[pj]; vanc[j] = panc[pj]; }
            else { plw[pj] = vlw[j]; panc[pj] = vanc[j]; }
        }
    }
}

// ----------------------------------------------------------------------------- small utilities
static __global__ void k_iota(int32_t* v, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) v[i] = (int32_t)i;
}

} // namespace gpf
