# reference_twins.jl -- @gen twins of the five native models and the script that pins the CPU oracle to Gen's own arithmetic.
#
# UNTESTED in this repository (the build image and the GPU boxes have no Julia; SURVEY.md F6).  Run on any machine with
# Julia >= 1.6, Gen 0.4.x and (for the timing part) GenParticleFilters 0.2.x:
#
#     julia julia/reference_twins.jl weights     # tests/golden/twin_inputs.txt -> tests/golden/ref_twin_weights.txt
#     julia julia/reference_twins.jl bench       # particle-steps/s of the REAL reference CPU path on the LG-SSM twin
#
# `weights` is deterministic (no RNG): every latent choice is constrained to the value listed in twin_inputs.txt and Gen is
# asked for the score of the observation given those latents (Gen.project on the selection of the observed addresses) and
# for the joint score of the step (Gen.assess).  tests/test_reference_twins.py compares the file with the oracle's
# log p(y_t | x_t) (column `loglik_oracle` of twin_inputs.txt) at rtol 1e-12; it skips while the file is absent.
# Written against the public API of Gen / GenParticleFilters; nothing is copied from the reference's sources.
using Gen

# ---------------------------------------------------------------- single-step twins (what pf_update! adds per particle)
# Each twin takes the previous latent state and the step's covariates, samples the new latents at addresses :x1.. and the
# observation at :y1.. -- the same choices, in the same order, as csrc/gpf_models.hpp / oracle/gpf_oracle.c.
const LG_A = 0.99 .* [cos(0.1) -sin(0.1); sin(0.1) cos(0.1)]
@gen function lgssm2_step(xp1::Float64, xp2::Float64)
    x1 ~ normal(LG_A[1, 1] * xp1 + LG_A[1, 2] * xp2, 0.1)
    x2 ~ normal(LG_A[2, 1] * xp1 + LG_A[2, 2] * xp2, 0.1)
    y1 ~ normal(x1, 0.5)
    y2 ~ normal(x2, 0.5)
end

wrap_pi(r) = r > pi ? r - 2pi : (r <= -pi ? r + 2pi : r)
# y = atan2(py, px) + N(0, 0.005^2) with the residual wrapped to (-pi, pi]: score the wrapped residual
@gen function bearings4_step(px::Float64, py::Float64, vx::Float64, vy::Float64, yobs::Float64)
    x1 ~ normal(px + vx, 0.001)
    x2 ~ normal(py + vy, 0.001)
    x3 ~ normal(vx, 0.001)
    x4 ~ normal(vy, 0.001)
    y1 ~ normal(wrap_pi(yobs - atan(x2, x1)), 0.005)          # constrained to 0.0: logpdf(normal, 0, r, s) == logpdf(normal, r, 0, s)
end

@gen function sv1_step(hp::Float64)
    x1 ~ normal(-1.0 + 0.97 * (hp + 1.0), 0.15)
    y1 ~ normal(0.0, exp(x1 / 2))
end

# README.md:43-55 of GenParticleFilters.jl, one step: x1 = moving (Bool), x2 = y; covariate sin(t)
@gen function object_motion_step(moving_prev::Bool, y_prev::Float64, sint::Float64)
    x1 ~ bernoulli(moving_prev ? 0.75 : 0.25)
    x2 ~ normal(y_prev + (x1 ? sint : 0.0), 0.01)
    y1 ~ normal(x2, 0.25)
end

# test/runtests.jl:3-8 of GenParticleFilters.jl (line_step), slope carried in the state: x1 = slope, x2 = outlier
@gen function line_step_twin(slope::Float64, x::Float64)
    x2 ~ bernoulli(0.1)
    y1 ~ normal(x * slope, x2 ? 10.0 : 1.0)
end

fmt(v) = repr(Float64(v))

function weights(inpath, outpath)
    out = String[]
    for ln in eachline(inpath)
        f = split(ln)
        model, t, nobs = f[1], parse(Int, f[2]), parse(Int, f[3])
        obs = parse.(Float64, f[4:3 + nobs])
        d = parse(Int, f[4 + nobs])
        prev = parse.(Float64, f[5 + nobs:4 + nobs + d])
        cur = parse.(Float64, f[5 + nobs + d:4 + nobs + 2d])
        cm = choicemap()
        if model == "lgssm2"
            fn, args = lgssm2_step, (prev[1], prev[2])
            cm[:x1] = cur[1]; cm[:x2] = cur[2]; cm[:y1] = obs[1]; cm[:y2] = obs[2]
            ysel = select(:y1, :y2)
        elseif model == "bearings4"
            fn, args = bearings4_step, (prev[1], prev[2], prev[3], prev[4], obs[1])
            for k in 1:4; cm[Symbol("x", k)] = cur[k]; end
            cm[:y1] = 0.0
            ysel = select(:y1)
        elseif model == "sv1"
            fn, args = sv1_step, (prev[1],)
            cm[:x1] = cur[1]; cm[:y1] = obs[1]
            ysel = select(:y1)
        elseif model == "object_motion"
            fn, args = object_motion_step, (prev[1] != 0.0, prev[2], obs[2])
            cm[:x1] = cur[1] != 0.0; cm[:x2] = cur[2]; cm[:y1] = obs[1]
            ysel = select(:y1)
        else                                           # line_model: slope = cur[1], outlier = cur[2], covariate x = obs[2]
            fn, args = line_step_twin, (cur[1], obs[2])
            cm[:x2] = cur[2] != 0.0; cm[:y1] = obs[1]
            ysel = select(:y1)
        end
        trace, joint = generate(fn, args, cm)          # everything constrained: joint = log p(x_t, y_t | x_{t-1})
        lobs = project(trace, ysel)                    # log p(y_t | x_t): what pf_update! adds to log_weights[i] (update.jl:21)
        push!(out, join([model, string(t), fmt(lobs), fmt(joint)], " "))
    end
    open(outpath, "w") do io; foreach(l -> println(io, l), out); end
    println(length(out), " lines -> ", outpath)
end

# ---------------------------------------------------------------- the real reference, timed (SURVEY.md 8d (ii))
# Full LG-SSM twin for GenParticleFilters' own pf_initialize / pf_update! / pf_resample!: state chained through Unfold.
@gen (static) function lgssm_kernel(t::Int, x::Vector{Float64})
    x1 ~ normal(LG_A[1, 1] * x[1] + LG_A[1, 2] * x[2], 0.1)
    x2 ~ normal(LG_A[2, 1] * x[1] + LG_A[2, 2] * x[2], 0.1)
    y1 ~ normal(x1, 0.5)
    y2 ~ normal(x2, 0.5)
    return [x1, x2]
end
lgssm_unfold = Unfold(lgssm_kernel)
@gen (static) function lgssm_model(T::Int)
    x01 ~ normal(0.0, 1.0)
    x02 ~ normal(0.0, 1.0)
    steps ~ lgssm_unfold(T, [x01, x02])
    return steps
end

function bench(n_particles::Int=10_000, T::Int=100)
    @eval using GenParticleFilters
    Gen.@load_generated_functions()
    ys = randn(T, 2)                                   # timing only: any data will do
    obs(t) = choicemap((:steps => t => :y1, ys[t, 1]), (:steps => t => :y2, ys[t, 2]))
    run() = begin
        state = Base.invokelatest(pf_initialize, lgssm_model, (1,), obs(1), n_particles)
        for t in 2:T
            Base.invokelatest(pf_resample!, state, :multinomial)
            Base.invokelatest(pf_update!, state, (t,), (UnknownChange(),), obs(t))
        end
        state
    end
    run()                                              # compile
    el = @elapsed run()
    println("reference CPU path (GenParticleFilters on ", Threads.nthreads(), " thread): N = ", n_particles, ", T = ", T, ": ",
            round(n_particles * T / el, digits=1), " particle-steps/s")
end

if abspath(PROGRAM_FILE) == @__FILE__
    root = normpath(joinpath(@__DIR__, ".."))
    mode = isempty(ARGS) ? "weights" : ARGS[1]
    mode == "weights" && weights(joinpath(root, "tests", "golden", "twin_inputs.txt"), joinpath(root, "tests", "golden", "ref_twin_weights.txt"))
    mode == "bench" && bench()
end
