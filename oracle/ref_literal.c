/*
 * ref_literal.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Statement-by-statement Float64 restatement of the weight/resampling arithmetic of
 * GenParticleFilters.jl v0.2.3 (paths relative to /root/reference), with NO fixed-point
 * tricks: sequential Float64 sums, true division, the same branches in the same order.
 * It exists to pin oracle/gpf_oracle.c (the exact-integer spec the GPU must match
 * bit-for-bit) to the reference's own arithmetic: tests/test_oracle_vs_literal.py checks that
 * both give the same ancestors / weights on seeded inputs (they can only differ when a
 * uniform lands within ~2^-40 of a CDF edge, or where Float64 rounding of N*(1/N) differs from
 * the exact rational -- both documented in DESIGN.md §3.4).
 *
 * Randomness: the reference calls the global RNG (resample.jl:59,113,162); here every call
 * site takes its uniforms from a caller-supplied array u[] indexed by output slot, so results
 * do not depend on how many draws earlier slots consumed.
 * Categorical sampling: the reference builds a Distributions.jl alias table
 * (rand!(Categorical(weights), parents), resample.jl:59); any exact sampler of the same
 * categorical distribution is distributionally identical, and this restatement (like the spec)
 * uses inverse-CDF with one uniform per slot.
 *
 * libm (exp/log) is used here on purpose: this file mirrors Julia's Float64 math, not the
 * deterministic spec.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define L_EXPORT __attribute__((visibility("default")))

/* Gen.logsumexp [Gen, SURVEY App. B]: m = maximum(v); m == -Inf ? -Inf : m + log(sum(exp.(v .- m))) */
L_EXPORT double lit_logsumexp(const double *v, int64_t n)
{
    double m = -INFINITY;
    int nan = 0;
    for (int64_t i = 0; i < n; ++i) { if (v[i] != v[i]) nan = 1; if (v[i] > m) m = v[i]; }
    if (nan) return NAN;
    if (m == -INFINITY) return -INFINITY;
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) s += exp(v[i] - m);
    return m + log(s);
}

/* utils.jl:100  lognorm(vs) = vs .- logsumexp(vs) */
L_EXPORT void lit_lognorm(const double *v, int64_t n, double *out)
{
    double l = lit_logsumexp(v, n);
    for (int64_t i = 0; i < n; ++i) out[i] = v[i] - l;
}

/* utils.jl:103-107 softmax */
L_EXPORT void lit_softmax(const double *v, int64_t n, double *ws)
{
    if (n == 0) return;
    double m = -INFINITY;
    for (int64_t i = 0; i < n; ++i) if (v[i] > m) m = v[i];
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) { ws[i] = exp(v[i] - m); s += ws[i]; }
    for (int64_t i = 0; i < n; ++i) ws[i] = ws[i] / s;
}

/* utils.jl:117-140 safe_softmax; returns invalid flag */
L_EXPORT int lit_safe_softmax(const double *v, int64_t n, double *ws)
{
    if (n == 0) return 0;                                        /* :118 */
    int any_nan = 0, all_ninf = 1;
    for (int64_t i = 0; i < n; ++i) { if (v[i] != v[i]) any_nan = 1; if (v[i] != -INFINITY) all_ninf = 0; }
    if (any_nan) { for (int64_t i = 0; i < n; ++i) ws[i] = NAN; return 1; }          /* :119-122 */
    if (all_ninf) { for (int64_t i = 0; i < n; ++i) ws[i] = 1.0 / (double)n; return 1; } /* :123-126 */
    double m = -INFINITY;
    for (int64_t i = 0; i < n; ++i) if (v[i] > m) m = v[i];
    double total = 0.0;
    for (int64_t i = 0; i < n; ++i) { ws[i] = exp(v[i] - m); total += ws[i]; }       /* :128-129 */
    if (total == 0.0) { for (int64_t i = 0; i < n; ++i) ws[i] = 1.0 / (double)n; return 1; } /* :130-133 */
    if (total != total) { for (int64_t i = 0; i < n; ++i) ws[i] = NAN; return 1; }   /* :134-137 */
    for (int64_t i = 0; i < n; ++i) ws[i] = ws[i] / total;                           /* :139 */
    return 0;
}

/* Gen.effective_sample_size(lnw) = exp(-logsumexp(2 .* lnw)) applied to lognorm(lw), utils.jl:163-164 */
L_EXPORT double lit_ess(const double *lw, int64_t n)
{
    double *t = (double *)malloc((size_t)n * sizeof(double));
    lit_lognorm(lw, n, t);
    for (int64_t i = 0; i < n; ++i) t[i] = 2.0 * t[i];
    double e = exp(-lit_logsumexp(t, n));
    free(t);
    return e;
}

/* inverse-CDF categorical draw: first i with cumsum(w)[i] > u (sequential Float64 accumulation) */
static int64_t categorical(const double *w, int64_t n, double u)
{
    double acc = 0.0;
    for (int64_t i = 0; i < n; ++i) { acc += w[i]; if (acc > u) return i; }
    return n - 1;
}

/* resample.jl:59  rand!(Categorical(weights), parents) -- 0-based parents */
L_EXPORT void lit_multinomial(const double *w, int64_t n, const double *u, int64_t *parents)
{
    for (int64_t j = 0; j < n; ++j) parents[j] = categorical(w, n, u[j]);
}

/* resample.jl:96-115 residual; u[j] is used only by tail slots j >= n_resampled */
L_EXPORT int64_t lit_residual(const double *w, int64_t n, const double *u, int64_t *parents)
{
    int64_t n_resampled = 0;
    for (int64_t i = 0; i < n; ++i) {                            /* :98 */
        int64_t n_copies = (int64_t)floor((double)n * w[i]);     /* :99 */
        if (n_copies == 0) continue;
        for (int64_t j = 0; j < n_copies && n_resampled + j < n; ++j) parents[n_resampled + j] = i; /* :101 */
        n_resampled += n_copies;
    }
    if (n_resampled < n) {                                       /* :108 */
        double *r = (double *)malloc((size_t)n * sizeof(double));
        double s = 0.0;
        for (int64_t i = 0; i < n; ++i) { r[i] = (double)n * w[i] - floor((double)n * w[i]); s += r[i]; } /* :109 */
        for (int64_t i = 0; i < n; ++i) r[i] = r[i] / s;         /* :110 */
        for (int64_t j = n_resampled; j < n; ++j) parents[j] = categorical(r, n, u[j]); /* :113 */
        free(r);
    }
    return n_resampled;
}

/* resample.jl:159-170 stratified; order = sortperm(log_priorities, rev=true) or 1:n (0-based here) */
L_EXPORT void lit_stratified(const double *w, const int64_t *order, int64_t n, const double *u,
                             int64_t *parents)
{
    int64_t i_old = 0;
    double step = 1.0 / (double)n, accum = 0.0;                  /* :159 */
    for (int64_t j = 0; j < n; ++j) {
        double lower = (double)j / (double)n;                    /* range element of 0.0:step:1.0-step */
        if (lower + step > accum) {                              /* :161 */
            double uu = u[j] * step + lower;                     /* :162 */
            while (accum < uu && i_old < n) {                    /* :163 (+ bound guard) */
                accum += w[order[i_old]];                        /* :164 */
                i_old += 1;                                      /* :165 */
            }
        }
        parents[j] = order[i_old > 0 ? i_old - 1 : 0];           /* :168 */
    }
}

/* resample.jl:190-202 update_weights! with priorities */
L_EXPORT void lit_update_weights(const double *lw, const double *lp, const int64_t *parents, int64_t n,
                                 double *lw_out)
{
    double *t = (double *)malloc((size_t)n * sizeof(double));
    for (int64_t j = 0; j < n; ++j) t[j] = lw[parents[j]] - lp[parents[j]];   /* :198 */
    double l = lit_logsumexp(t, n);
    for (int64_t j = 0; j < n; ++j) lw_out[j] = t[j] + (log((double)n) - l);  /* :200 */
    free(t);
}

/* resize.jl:203-219 find_inv_w_threshold: the Float64 formulation (ascending sort, running A and B) */
static int cmp_f64(const void *a, const void *b)
{
    const double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}
L_EXPORT double lit_inv_w_threshold(const double *w, int64_t n_old, int64_t n_particles)
{
    double *s = (double *)malloc((size_t)n_old * sizeof(double));
    memcpy(s, w, (size_t)n_old * sizeof(double));
    qsort(s, (size_t)n_old, sizeof(double), cmp_f64);            /* :204 */
    int64_t A = n_old;                                           /* :206 */
    double B = 0.0, c = (double)n_particles;                     /* :207, :218 */
    for (int64_t k = 0; k < n_old; ++k) {
        const double kappa = s[k];
        A -= 1;                                                  /* :209 */
        B += kappa;                                              /* :210 */
        const double n_check = B / kappa + (double)A;            /* :212 */
        const double eps = nextafter(fabs(n_check), INFINITY) - fabs(n_check);
        if (n_check <= (double)n_particles + eps) {              /* :213 */
            c = ((double)n_particles - (double)A) / B;           /* :215 */
            break;
        }
    }
    free(s);
    return c;
}
/* resize.jl:170-178: systematic sampling among the particles that are not kept; returns the number of picks */
L_EXPORT int64_t lit_systematic(const double *w_norm, int64_t n_strat, int64_t n_resample, double rand01, int64_t *picks)
{
    const double step = 1.0 / (double)n_resample;                /* :170 */
    double u = rand01 * step;                                    /* :171 */
    int64_t c = 0;
    for (int64_t i = 0; i < n_strat; ++i) {
        u = u - w_norm[i];                                       /* :173 */
        if (u < 0) { picks[c++] = i; u += step; }                /* :174-177 */
    }
    return c;
}
