/*
 * examples/lgssm_filter.c -- the C ABI (include/gpf.h) used directly from a compiled host, no Python:
 * a bootstrap particle filter (pf_initialize, then pf_resample! + pf_update! per observation, with an ESS-triggered or
 * unconditional resample) on one of the compiled models.  This is the call sequence a GenParticleFilters.jl maintainer's
 * ccall glue produces (julia/GenParticleFiltersAMD.jl); tests/test_c_example.py builds it with gcc, runs it on the GPU and
 * checks its output against the Python host bit for bit.
 *
 *   cc -O2 -Iinclude examples/lgssm_filter.c -o lgssm_filter genparticlefilters.jl_amd/libgpf_hip.so -Wl,-rpath,'$ORIGIN'
 *   ./lgssm_filter input.txt n_particles seed method(0 multinomial | 1 residual | 2 stratified) ess_fraction [rejuvenate [timed_from]]
 *
 * rejuvenate: 0 none (default) | 1 one MH sweep | 2 one move-reweight sweep after every resample (pf_rejuvenate!, README.md:72-75).
 * timed_from > 0: a second output line "us_per_step X steps K" -- wall clock from step timed_from (after a synchronize) to the end:
 * what an ESS-triggered loop costs per step from a compiled host (tools/bench_configs.py prints it beside the Python host's number).
 *
 * input.txt: model id, n_params, the parameters, obs_dim, T, then T x obs_dim observations (text, %.17g round-trips).
 */
#define _POSIX_C_SOURCE 199309L
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "gpf.h"

#define CHECK(call)                                                                   \
    do {                                                                              \
        gpf_status st_ = (call);                                                      \
        if (st_ != GPF_OK) {                                                          \
            fprintf(stderr, "%s failed (%d): %s\n", #call, (int)st_, gpf_last_error(h)); \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

int main(int argc, char** argv)
{
    gpf_handle h = NULL;
    if (argc < 6) { fprintf(stderr, "usage: %s input.txt n_particles seed method ess_fraction [rejuvenate [timed_from [one_call]]]\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "r");
    if (!f) { perror(argv[1]); return 2; }
    int model, n_params, obs_dim, T;
    double params[24];
    if (fscanf(f, "%d %d", &model, &n_params) != 2 || n_params > 24) return 2;
    for (int i = 0; i < n_params; ++i) if (fscanf(f, "%lf", &params[i]) != 1) return 2;
    if (fscanf(f, "%d %d", &obs_dim, &T) != 2) return 2;
    double* ys = (double*)malloc(sizeof(double) * (size_t)obs_dim * (size_t)T);
    for (int i = 0; i < obs_dim * T; ++i) if (fscanf(f, "%lf", &ys[i]) != 1) return 2;
    fclose(f);

    const int64_t n = atoll(argv[2]);
    const int method = atoi(argv[4]);
    const double ess_fraction = atof(argv[5]);
    const int rejuvenate = argc > 6 ? atoi(argv[6]) : 0;
    const int timed_from = argc > 7 ? atoi(argv[7]) : 0;
    const int one_call = argc > 8 ? atoi(argv[8]) : 0;     /* 1: the loop body as ONE call (gpf_step_ess) instead of the four below -- same results */

    gpf_config cfg = {0};
    cfg.abi_version = GPF_ABI_VERSION;
    cfg.model = model; cfg.n_params = n_params; cfg.params = params; cfg.keep_prev = rejuvenate != 0;
    cfg.n_particles = n; cfg.n_global = n; cfg.gid0 = 0;
    cfg.seed = strtoull(argv[3], NULL, 10);
    cfg.device = 0; cfg.stream = NULL;
    CHECK(gpf_create(&cfg, &h));

    CHECK(gpf_initialize(h, ys, obs_dim));                                   /* pf_initialize, src/initialize.jl:31-44 */
    int n_resamples = 0;
    struct timespec t0 = {0, 0}, t1 = {0, 0};
    for (int t = 1; t < T; ++t) {
        double ess;
        if (t == timed_from) { CHECK(gpf_synchronize(h)); clock_gettime(CLOCK_MONOTONIC, &t0); }
        if (one_call) {
            /* if effective_sample_size(state) < tau N; pf_resample!; pf_rejuvenate!; end; pf_update!  (README.md:66-77) */
            int32_t res = 0;
            CHECK(gpf_step_ess(h, ys + (size_t)t * obs_dim, obs_dim, ess_fraction, method, 0, GPF_CHECK_FALSE,
                               rejuvenate ? (rejuvenate == 2 ? GPF_REJUVENATE_REWEIGHT : GPF_REJUVENATE_MOVE) : -1, 1, &res, NULL, NULL));
            n_resamples += res;
            continue;
        }
        CHECK(gpf_effective_sample_size(h, &ess));                           /* get_ess, src/utils.jl:171 */
        if (ess < ess_fraction * (double)n) {
            /* pf_resample!(state, method), src/resample.jl:19-30; sort_particles = false, check = :warn without the print */
            CHECK(gpf_resample(h, method, 0.0 / 0.0 /* priority_fn = nothing */, 0, GPF_CHECK_FALSE, NULL));
            ++n_resamples;
            /* pf_rejuvenate!(state, kern, (), 1; method), src/rejuvenate.jl:18-27 (the model's own move kernel) */
            if (rejuvenate) CHECK(gpf_rejuvenate(h, rejuvenate == 2 ? GPF_REJUVENATE_REWEIGHT : GPF_REJUVENATE_MOVE, 1, NULL));
        }
        CHECK(gpf_update(h, ys + (size_t)t * obs_dim, obs_dim));             /* pf_update!, src/update.jl:12-25 */
    }
    if (timed_from > 0) { CHECK(gpf_synchronize(h)); clock_gettime(CLOCK_MONOTONIC, &t1); }
    double lml, ess, mean0, var0;
    CHECK(gpf_log_ml_estimate(h, &lml));
    CHECK(gpf_effective_sample_size(h, &ess));
    CHECK(gpf_mean(h, 0, &mean0));
    CHECK(gpf_var(h, 0, &var0));
    printf("%.17g %.17g %.17g %.17g %d\n", lml, ess, mean0, var0, n_resamples);
    if (timed_from > 0 && timed_from < T)
        printf("us_per_step %.3f steps %d\n", ((double)(t1.tv_sec - t0.tv_sec) * 1e6 + (double)(t1.tv_nsec - t0.tv_nsec) * 1e-3) / (double)(T - timed_from), T - timed_from);
    CHECK(gpf_destroy(h));
    free(ys);
    return 0;
}
