// resize family, trajectory store, sub-state views  (part of gpf_kernels.hpp; include that header, not this file)
#pragma once

namespace gpf {
// ----------------------------------------------------------------------------- resize family (reference src/resize.jl)
// pf_replicate! (resize.jl:236-244): parents = repeat(1:N, inner=k) (contiguous) or repeat(1:N, k) (interleaved);
// pf_dereplicate! :keepfirst (resize.jl:267-280): parents = 1:k:N (contiguous) or 1:N/k (interleaved)
static __global__ void k_replicate_anc(int64_t n_new, int64_t n_old, int k, int interleaved, int shrink, int32_t* __restrict__ anc)
{
    for (int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x; j < n_new; j += (int64_t)gridDim.x * BLOCK) {
        int64_t a;
        if (!shrink) a = interleaved ? j % n_old : j / k;
        else         a = interleaved ? j : j * k;
        anc[j] = (int32_t)a;
    }
}
// rows_out[j] = rows_in[anc[j]], lw_out[j] = lw_in[anc[j]]  (traces and weights of the selected parents)
template <int W>
__global__ __launch_bounds__(BLOCK) void k_gather_rows_lw(const int32_t* __restrict__ anc, const double* __restrict__ rows_in,
                                                         const double* __restrict__ lw_in, double* __restrict__ rows_out,
                                                         double* __restrict__ lw_out, int64_t n)
{
    constexpr int C = W / 2;
    const int64_t total = n * C;
    for (int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x; t < total; t += (int64_t)gridDim.x * BLOCK) {
        const int64_t j = t / C;
        const int c = (int)(t - j * C);
        const int64_t a = anc[j];
        reinterpret_cast<double2*>(rows_out)[t] = reinterpret_cast<const double2*>(rows_in)[a * C + c];
        if (c == 0 && lw_out) lw_out[j] = lw_in[a];
    }
}
// ---- pf_optimal_resize! (resize.jl:149-219) in exact fixed point
// find_inv_w_threshold (resize.jl:203-219) on the DESCENDING order: position d holds kappa = q_(d), A = d weights
// before it and B = S - C[d-1] from it on; the reference's first kappa (ascending) with B / kappa + A <= n is the
// LARGEST such d.  The condition is constant over ties, and d < n is necessary.
static __global__ __launch_bounds__(BLOCK) void k_opt_threshold(const uint64_t* __restrict__ cdf_desc, const WSum* ws, int64_t n_new,
                                                         int64_t n_old, Scalars* sc)
{
    const uint64_t S = ws->S;
    const int64_t lim = n_new < n_old ? n_new : n_old;
    long long best = -1;
    for (int64_t d = (int64_t)blockIdx.x * BLOCK + threadIdx.x; d < lim; d += (int64_t)gridDim.x * BLOCK) {
        const uint64_t prev = d > 0 ? cdf_desc[d - 1] : 0;
        const uint64_t kappa = cdf_desc[d] - prev;
        if (kappa > 0 && le_mul(S - prev, (uint64_t)(n_new - d), kappa)) best = d;
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { const long long o = __shfl_xor(best, s, WAVE); best = o > best ? o : best; }
    if (lane_id() == 0 && best >= 0) atomicMax(&sc->opt_d, best);
}
// c = (n - A) / B, or float(n) when no kappa qualifies (resize.jl:215,218), as the pair (a, B): c w_i >= 1 <=> a q_i >= B
static __global__ void k_opt_params(const uint64_t* __restrict__ cdf_desc, const WSum* ws, int64_t n_new, Scalars* sc)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const long long d = sc->opt_d;
    sc->opt_a = d < 0 ? (uint64_t)n_new : (uint64_t)(n_new - d);
    sc->opt_B = d <= 0 ? ws->S : ws->S - cdf_desc[d - 1];
}
// parents[1:n_keep] .= findall(keep_idxs) (resize.jl:159,180) from the inclusive scan of the keep flags
static __global__ __launch_bounds__(BLOCK) void k_opt_keep_scatter(const uint64_t* __restrict__ keepcdf, int64_t n_old, int32_t* __restrict__ anc)
{
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n_old; i += (int64_t)gridDim.x * BLOCK) {
        const uint64_t c = keepcdf[i], p = i > 0 ? keepcdf[i - 1] : 0;
        if (c != p) anc[p] = (int32_t)i;
    }
}
// log_weights (resize.jl:189-195): kept particles keep theirs, the others get logsumexp - log c; all + log(n / n_old)
static __global__ __launch_bounds__(BLOCK) void k_opt_weights(double* __restrict__ lw, int64_t n, const Scalars* sc, const WSum* ws, int K,
                                                       double log_n_ratio)
{
    const int64_t n_keep = (int64_t)sc->Ctot;
    const double rw = lse_from(ws->m, sc->opt_B, K, ws->flags) - log_((double)sc->opt_a);
    for (int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x; j < n; j += (int64_t)gridDim.x * BLOCK)
        lw[j] = (j < n_keep ? lw[j] : rw) + log_n_ratio;
}

// pf_dereplicate! method = :sample (resize.jl:281-293): one categorical draw per block of k replicates, with the
// block's softmax in K_b-bit fixed point (same spec as §3.3 of DESIGN.md, N = k); new weight = logsumexp(block) - log k
static __global__ void k_dereplicate_sample(const double* __restrict__ lw, int64_t n_new, int64_t n_old, int k, int interleaved,
                                     uint64_t seed, uint32_t epoch, int Kb, double logk, int32_t* __restrict__ anc,
                                     double* __restrict__ lw_out)
{
    const int64_t stride = interleaved ? n_new : 1;
    for (int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x; j < n_new; j += (int64_t)gridDim.x * BLOCK) {
        const int64_t first = interleaved ? j : j * k;
        double m = -__builtin_huge_val();
        bool nan = false;
        for (int e = 0; e < k; ++e) { const double v = lw[first + e * stride]; if (v != v) nan = true; else m = v > m ? v : m; }
        const bool uniform = !nan && m == -__builtin_huge_val();
        uint64_t S = 0;
        for (int e = 0; e < k; ++e) S += uniform ? 1 : exp_fix(lw[first + e * stride] - m, Kb);
        const Philox b = rng(seed, (uint32_t)j, 0, epoch, TAG_RESAMPLE);
        const uint64_t T = mulhi64(u64(b.w0, b.w1), S);
        uint64_t acc = 0;
        int pick = k - 1;
        for (int e = 0; e < k; ++e) {
            acc += uniform ? 1 : exp_fix(lw[first + e * stride] - m, Kb);
            if (acc > T) { pick = e; break; }
        }
        anc[j] = (int32_t)(first + pick * stride);
        const int f = nan ? FLAG_NAN : (uniform ? FLAG_ALL_NEGINF : 0);
        lw_out[j] = lse_from(m, S, Kb, f) - logk;
    }
}

// ----------------------------------------------------------------------------- trajectory store
// Gen traces are persistent: mean(state, 5 => :moving) (reference README.md:97) reads a PAST choice of every
// surviving particle.  The device keeps, per time step, the step's latent columns (final particle order of that
// step) and the composed ancestor map of the resamples that happened during the step.
static __global__ void k_hist_snapshot(const double* __restrict__ rows, int W, int d, int64_t n, double* __restrict__ out)
{
    for (int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x; t < n * d; t += (int64_t)gridDim.x * BLOCK) {
        const int64_t i = t / d;
        out[t] = rows[i * W + (t - i * d)];
    }
}
// B[j] = first resample of the step ? anc[j] : B_old[anc[j]]
static __global__ void k_hist_compose(const int32_t* __restrict__ anc, const int32_t* __restrict__ b_old, int64_t n, int32_t* __restrict__ b_new)
{
    for (int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x; j < n; j += (int64_t)gridDim.x * BLOCK)
        b_new[j] = b_old ? b_old[anc[j]] : anc[j];
}
// value of column `col` of step `t` along the ancestry of every current particle: follow B_T, B_{T-1}, ..., B_{t+1}
static __global__ void k_hist_column(const int32_t* const* __restrict__ maps, int n_maps, const double* __restrict__ hx, int d, int col,
                              int64_t n, double* __restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        int64_t idx = i;
        for (int s = 0; s < n_maps; ++s) { const int32_t* m = maps[s]; if (m) idx = m[idx]; }
        out[i] = hx[idx * d + col];
    }
}
// ----------------------------------------------------------------------------- sub-state views (src/view.jl, resample.jl:205-218)
// after resampling a view: every log-weight = logsumexp(view) - log n (the block keeps its total mass, resample.jl:210)
static __global__ void k_fill_from(double* __restrict__ lw, int64_t n, const double* __restrict__ value)
{
    const double v = *value;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) lw[i] = v;
}
static __global__ void k_view_fill_weights(double* __restrict__ lw, int64_t n, const WSum* ws, int K, double logN)
{
    const double v = lse_from(ws->m, ws->S, K, ws->flags) - logN;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) lw[i] = v;
}
// with priorities: lw = log_ws + (logsumexp(view weights) - logsumexp(log_ws))   (resample.jl:213-216)
static __global__ void k_view_apply_post(const Scalars* sc, int K, const double* __restrict__ lws, double* __restrict__ lw, int64_t n)
{
    const double off = lse_from(sc->raw.m, sc->raw.S, K, sc->raw.flags) - lse_from(sc->post.m, sc->post.S, K, sc->post.flags);
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) lw[i] = lws[i] + off;
}

} // namespace gpf
