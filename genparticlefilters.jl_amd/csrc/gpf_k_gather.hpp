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
__global__ __launch_bounds__(BLOCK) void k_apply_post(const Scalars* sc, int K, double logN, const double* __restrict__ lws,
                                                      double* __restrict__ lw, int64_t n)
{
    const double off = logN - lse_from(sc->post.m, sc->post.S, K, sc->post.flags);
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK)
        lw[i] = lws[i] + off;
}

// ----------------------------------------------------------------------------- K9: statistics
// sum_i w_i f(x_i), w_i = q_i / S (statistics.jl:13-14, 48-50); per-block partials in Float64
__global__ __launch_bounds__(BLOCK) void k_wsum(const double* __restrict__ lw, const WSum* ws, int K,
                                                const double* __restrict__ rows, int W, int col, int64_t n,
                                                int pw, const double* center, double* __restrict__ partial)
{
    const double m = ws->m;
    const double Sd = (double)ws->S;
    const bool uniform = (ws->flags & FLAG_ALL_NEGINF) != 0;
    const double c = center ? *center : 0.0;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        const uint64_t q = uniform ? 1 : exp_fix(lw[i] - m, K);
        double v = rows[i * W + col];
        if (pw == 2) { v = v - c; v = v * v; }
        acc += ((double)q / Sd) * v;
    }
    acc = wave_sum_f64(acc);
    __shared__ double s[NWAVES];
    if (lane_id() == 0) s[wave_id()] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int w = 0; w < NWAVES; ++w) t += s[w]; partial[blockIdx.x] = t; }
}
__global__ void k_sum_partials(const double* __restrict__ partial, int np, double* out)
{
    double acc = 0.0;
    for (int i = threadIdx.x; i < np; i += BLOCK) acc += partial[i];
    acc = wave_sum_f64(acc);
    __shared__ double s[NWAVES];
    if (lane_id() == 0) s[wave_id()] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int w = 0; w < NWAVES; ++w) t += s[w]; *out = t; }
}

// ----------------------------------------------------------------------------- strided sub-state views
// state[start:step:stop] (reference src/view.jl:35-48; test/initialize.jl:60, test/update.jl:33): a strided view works on a
// compact copy of its particles -- view_enter gathers rows / log-weights / parents of particles start + i*step, view_exit
// scatters them back (update_refs! for sub-states copies back as well, utils.jl:17-20).  to_view: parent -> compact, else back.
__global__ __launch_bounds__(BLOCK) void k_view_strided_copy(double* __restrict__ prow, double* __restrict__ plw, int32_t* __restrict__ panc,
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

// ----------------------------------------------------------------------------- small utilities
__global__ void k_iota(int32_t* v, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) v[i] = (int32_t)i;
}

} // namespace gpf
