# GenParticleFiltersAMD.jl -- Julia binding of libgpf_hip.so (include/gpf.h).
#
# UNTESTED in this repository's CI: the build image has no Julia (SURVEY.md F6). It is the thin `ccall`
# layer a maintainer of GenParticleFilters.jl would add so that the device state type participates in the
# package's generic functions.  Every method mirrors the signature of the reference method it stands in
# for (file:line of GenParticleFilters.jl v0.2.3 in the comments).
module GenParticleFiltersAMD

import GenParticleFilters: pf_resize!, pf_multinomial_resize!, pf_residual_resize!, pf_optimal_resize!, pf_replicate!, pf_dereplicate!
import GenParticleFilters: pf_initialize, pf_update!, pf_resample!, pf_multinomial_resample!,
    pf_residual_resample!, pf_stratified_resample!, pf_rejuvenate!, pf_move_accept!, pf_move_reweight!,
    get_log_norm_weights, get_norm_weights, get_ess, get_lml_est
import Gen: effective_sample_size, log_ml_estimate, get_log_weights, sample_unweighted_traces
import Statistics: mean, var

const libgpf = get(ENV, "LIBGPF_HIP", "libgpf_hip.so")
const GPF_ABI_VERSION = Cint(1)

"Native model descriptor: replaces `model::GenerativeFunction` (src/initialize.jl:31-35)."
struct NativeModel
    id::Cint                 # gpf_model
    params::Vector{Float64}  # layout of csrc/gpf_models.hpp
    dim::Int
end

struct GpfConfig
    abi_version::Cint; model::Cint; n_params::Cint; keep_prev::Cint
    params::Ptr{Cdouble}
    n_particles::Int64; n_global::Int64; gid0::Int64
    seed::UInt64
    device::Cint; reserved::Cint
    stream::Ptr{Cvoid}
end

"Device-resident counterpart of Gen.ParticleFilterState (traces/new_traces/log_weights/log_ml_est/parents)."
mutable struct DeviceParticleFilterState
    handle::Ptr{Cvoid}
    model::NativeModel
    n_particles::Int
    function DeviceParticleFilterState(model::NativeModel, n::Int; seed::Integer=1, keep_prev::Bool=false, device::Integer=0)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve model begin
            cfg = Ref(GpfConfig(GPF_ABI_VERSION, model.id, length(model.params), keep_prev, pointer(model.params),
                                n, n, 0, UInt64(seed), device, 0, C_NULL))
            st = ccall((:gpf_create, libgpf), Cint, (Ref{GpfConfig}, Ref{Ptr{Cvoid}}), cfg, h)
        end
        st == 0 || error(unsafe_string(ccall((:gpf_last_error, libgpf), Cstring, (Ptr{Cvoid},), C_NULL)))
        state = new(h[], model, n)
        finalizer(s -> ccall((:gpf_destroy, libgpf), Cint, (Ptr{Cvoid},), getfield(s, :handle)), state)
        return state
    end
    # wrap an existing handle (a sub-state view created by gpf_view_create)
    function DeviceParticleFilterState(handle::Ptr{Cvoid}, model::NativeModel, n::Int)
        state = new(handle, model, n)
        finalizer(s -> ccall((:gpf_destroy, libgpf), Cint, (Ptr{Cvoid},), getfield(s, :handle)), state)
        return state
    end
end

# state[a:b] / view(state, a:b): ParticleFilterSubState over a contiguous range (src/view.jl:35-48); every method of this
# file accepts the returned object (it is a DeviceParticleFilterState whose handle aliases the source's particles)
function Base.getindex(s::DeviceParticleFilterState, r::UnitRange{Int})
    h = Ref{Ptr{Cvoid}}(C_NULL)
    _status(s, ccall((:gpf_view_create, libgpf), Cint, (Ptr{Cvoid}, Int64, Int64, Ref{Ptr{Cvoid}}), getfield(s, :handle), first(r) - 1, length(r), h))
    return DeviceParticleFilterState(h[], getfield(s, :model), length(r))
end
Base.view(s::DeviceParticleFilterState, r::UnitRange{Int}) = s[r]
# state[k:5:100] (test/initialize.jl:60, test/update.jl:33): a strided sub-state; particle i of the view is particle first(r) + (i-1) step(r)
function Base.getindex(s::DeviceParticleFilterState, r::StepRange{Int,Int})
    step(r) >= 1 || error("sub-state ranges need a positive step")
    h = Ref{Ptr{Cvoid}}(C_NULL)
    _status(s, ccall((:gpf_view_create_strided, libgpf), Cint, (Ptr{Cvoid}, Int64, Int64, Int64, Ref{Ptr{Cvoid}}), getfield(s, :handle), first(r) - 1, step(r), length(r), h))
    return DeviceParticleFilterState(h[], getfield(s, :model), length(r))
end
Base.view(s::DeviceParticleFilterState, r::StepRange{Int,Int}) = s[r]
# state[idxs] for ANY vector of distinct indices (src/view.jl:35-48 takes idxs::AbstractVector): the same compact-copy view over an index array
function Base.getindex(s::DeviceParticleFilterState, idxs::AbstractVector{<:Integer})
    ix = Int64.(idxs) .- 1
    h = Ref{Ptr{Cvoid}}(C_NULL)
    _status(s, ccall((:gpf_view_create_indexed, libgpf), Cint, (Ptr{Cvoid}, Ptr{Int64}, Int64, Ref{Ptr{Cvoid}}), getfield(s, :handle), ix, length(ix), h))
    return DeviceParticleFilterState(h[], getfield(s, :model), length(ix))
end
Base.getindex(s::DeviceParticleFilterState, mask::AbstractVector{Bool}) = s[findall(mask)]
Base.view(s::DeviceParticleFilterState, idxs::AbstractVector{<:Integer}) = s[idxs]

# status helper; NOT called `check`: the resamplers take a keyword of that name (src/resample.jl:43-46) which would shadow it
_status(state, st) = st == 0 ? nothing :
    error(unsafe_string(ccall((:gpf_last_error, libgpf), Cstring, (Ptr{Cvoid},), state.handle)))   # ErrorException

# src/initialize.jl:31-44
function pf_initialize(model::NativeModel, model_args::Tuple, observations::Vector{Float64}, n_particles::Int; kw...)
    state = DeviceParticleFilterState(model, n_particles; kw...)
    _status(state, ccall((:gpf_initialize, libgpf), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint),
                       state.handle, observations, length(observations)))
    return state
end

# src/update.jl:12-25
function pf_update!(state::DeviceParticleFilterState, new_args::Tuple, argdiffs::Tuple, observations::Vector{Float64})
    _status(state, ccall((:gpf_update, libgpf), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint),
                       state.handle, observations, length(observations)))
    return state
end

"Native proposal for pf_initialize / pf_update! (src/initialize.jl:46-62, src/update.jl:79-96): exact conditional of the LG-SSM"
struct LocallyOptimal end
function pf_initialize(model::NativeModel, model_args::Tuple, observations::Vector{Float64}, proposal::LocallyOptimal, proposal_args::Tuple, n_particles::Int; kw...)
    state = DeviceParticleFilterState(model, n_particles; kw...)
    _status(state, ccall((:gpf_initialize_proposal, libgpf), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cint), state.handle, observations, length(observations), 1))
    return state
end
function pf_update!(state::DeviceParticleFilterState, new_args::Tuple, argdiffs::Tuple, observations::Vector{Float64}, proposal::LocallyOptimal, proposal_args::Tuple)
    _status(state, ccall((:gpf_update_proposal, libgpf), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cint), state.handle, observations, length(observations), 1))
    return state
end

"priority_fn = w -> alpha*w evaluated on the GPU (test/resample.jl:15 uses alpha = 1/2)"
struct Tempering; alpha::Float64; end
(t::Tempering)(w) = t.alpha * w

function _resample!(state, method::Int, priority_fn, check_kw, sort_particles::Bool)
    chk = check_kw === true ? 2 : (check_kw === :warn ? 1 : 0)
    invalid = Ref{Cint}(0)
    inv_ptr = chk == 0 ? Ptr{Cint}(C_NULL) : Base.unsafe_convert(Ptr{Cint}, invalid)
    GC.@preserve invalid begin
        if priority_fn === nothing || priority_fn isa Tempering
            alpha = priority_fn === nothing ? NaN : priority_fn.alpha
            st = ccall((:gpf_resample, libgpf), Cint, (Ptr{Cvoid}, Cint, Cdouble, Cint, Cint, Ptr{Cint}),
                       state.handle, method, alpha, sort_particles, chk, inv_ptr)
        else    # arbitrary closure: priority_fn.(log_weights) on the host (src/resample.jl:51-52)
            lp = priority_fn.(get_log_weights(state))
            st = ccall((:gpf_resample_with_priorities, libgpf), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Cint, Cint, Ptr{Cint}),
                       state.handle, method, lp, sort_particles, chk, inv_ptr)
        end
    end
    _status(state, st)                                            # error("Invalid weights."), src/resample.jl:55
    check_kw === :warn && invalid[] != 0 && @warn("Invalid weights: resampled with uniform weights.")   # utils.jl:120-135
    return state
end

"""
    pf_step_ess!(state, new_args, argdiffs, observations; ess_threshold=0.5, method=:multinomial, rejuvenate=nothing, n_iters=1, check=:warn, sort_particles=true)

One iteration of the README loop (README.md:66-77) in one ccall (gpf.h gpf_step_ess): `if effective_sample_size(state) < ess_threshold * N;
pf_resample!(state, method); pf_rejuvenate!(state, ...; method=rejuvenate); end; pf_update!(state, new_args, argdiffs, observations)` -- the
same results, without the host's round trip on the steps that do not resample.  Returns whether it resampled.
"""
function pf_step_ess!(s::DeviceParticleFilterState, new_args::Tuple, argdiffs::Tuple, observations::Vector{Float64}; ess_threshold::Real=0.5,
                      method::Symbol=:multinomial, rejuvenate::Union{Nothing,Symbol}=nothing, n_iters::Int=1, check=:warn, sort_particles::Bool=true)
    mid = method == :multinomial ? 0 : method == :residual ? 1 : method == :stratified ? 2 : method == :multinomial_sorted ? 4 : error("Resampling method $method not recognized.")
    rid = rejuvenate === nothing ? -1 : rejuvenate == :move ? 0 : rejuvenate == :reweight ? 1 : error("Method $rejuvenate not recognized.")
    chk = check === true ? 2 : (check === :warn ? 1 : 0)
    resampled = Ref{Cint}(0); invalid = Ref{Cint}(0)
    GC.@preserve resampled invalid begin
        st = ccall((:gpf_step_ess, libgpf), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cdouble, Cint, Cint, Cint, Cint, Cint, Ptr{Cint}, Ptr{Cint}, Ptr{Cdouble}),
                   s.handle, observations, length(observations), Float64(ess_threshold), mid, sort_particles, chk, rid, n_iters,
                   Base.unsafe_convert(Ptr{Cint}, resampled), chk == 0 ? Ptr{Cint}(C_NULL) : Base.unsafe_convert(Ptr{Cint}, invalid), Ptr{Cdouble}(C_NULL))
    end
    _status(s, st)
    check === :warn && invalid[] != 0 && @warn("Invalid weights: resampled with uniform weights.")
    return resampled[] != 0
end

"opt-in: pf_resample!(state, :multinomial) leaves its ancestor search to the pf_update! that follows (one fused kernel; gpf.h gpf_set_lazy_search)"
set_lazy_search!(s::DeviceParticleFilterState, enable::Bool=true) =
    (_status(s, ccall((:gpf_set_lazy_search, libgpf), Cint, (Ptr{Cvoid}, Cint), s.handle, enable ? 1 : 0)); s)
"checkpoint / resume (gpf.h gpf_checkpoint_*): the whole state as one byte vector; `restore!` on a state created with the same arguments continues bit for bit"
function checkpoint(s::DeviceParticleFilterState)
    nb = Ref{Int64}(0)
    _status(s, ccall((:gpf_checkpoint_size, libgpf), Cint, (Ptr{Cvoid}, Ptr{Int64}), s.handle, nb))
    blob = Vector{UInt8}(undef, nb[])
    _status(s, ccall((:gpf_checkpoint_save, libgpf), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64), s.handle, blob, nb[]))
    return blob
end
restore!(s::DeviceParticleFilterState, blob::Vector{UInt8}) =
    (_status(s, ccall((:gpf_checkpoint_load, libgpf), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64), s.handle, blob, length(blob))); s)
"opt-in extension (gpf.h GPF_RESAMPLE_MULTINOMIAL_SORTED): multinomial resampling with the uniforms drawn already sorted -- state.parents non-decreasing"
pf_multinomial_sorted_resample!(s::DeviceParticleFilterState; priority_fn=nothing, check=:warn) = _resample!(s, 4, priority_fn, check, true)

# src/resample.jl:48-65, 85-120, 143-175, 19-30
pf_multinomial_resample!(s::DeviceParticleFilterState; priority_fn=nothing, check=:warn) = _resample!(s, 0, priority_fn, check, true)
pf_residual_resample!(s::DeviceParticleFilterState; priority_fn=nothing, check=:warn) = _resample!(s, 1, priority_fn, check, true)
pf_stratified_resample!(s::DeviceParticleFilterState; priority_fn=nothing, check=:warn, sort_particles::Bool=true) =
    _resample!(s, 2, priority_fn, check, sort_particles)
function pf_resample!(s::DeviceParticleFilterState, method::Symbol=:multinomial; kwargs...)
    method == :multinomial && return pf_multinomial_resample!(s; kwargs...)
    method == :residual && return pf_residual_resample!(s; kwargs...)
    method == :stratified && return pf_stratified_resample!(s; kwargs...)
    method == :multinomial_sorted && return pf_multinomial_sorted_resample!(s; kwargs...)      # (extension)
    error("Resampling method $method not recognized.")
end

# Many small filters in one state: the batched form of
#     for b in blocks; (ess_frac === nothing || get_ess(state[b]) < ess_frac * length(b)) && pf_resample!(state[b], method; ...); end
# over consecutive blocks of block_size particles (<= 2048) in ONE launch (gpf.h gpf_resample_blocks; sub-state semantics of
# src/resample.jl:185-187,205-218, README.md:60-79 per block).  Returns the number of blocks that resampled.
function pf_resample_blocks!(s::DeviceParticleFilterState, block_size::Int, method::Symbol=:multinomial;
                             priority_fn::Union{Nothing,Tempering}=nothing, ess_frac=nothing, sort_particles::Bool=true, check=:warn)
    m = method == :multinomial ? 0 : method == :residual ? 1 : method == :stratified ? 2 : error("Resampling method $method not recognized.")
    chk = check === true ? 2 : (check === :warn ? 1 : 0)
    invalid = Ref{Cint}(0); count = Ref{Int64}(0)
    st = ccall((:gpf_resample_blocks, libgpf), Cint, (Ptr{Cvoid}, Cint, Int64, Cdouble, Cint, Cdouble, Cint, Ptr{Cint}, Ptr{Int64}),
               s.handle, m, block_size, priority_fn === nothing ? NaN : priority_fn.alpha, sort_particles ? 1 : 0,
               ess_frac === nothing ? NaN : Float64(ess_frac), chk, invalid, count)
    _status(s, st)
    check === :warn && invalid[] != 0 && @warn("Invalid weights in some block: resampled with uniform weights.")
    return Int(count[])
end
"(effective_sample_size(state[b]), log_ml_estimate(state[b])) of every block of block_size particles (src/utils.jl:163-178), one launch"
function block_stats(s::DeviceParticleFilterState, block_size::Int)
    nb = cld(s.n_particles, block_size)
    ess = Vector{Float64}(undef, nb); lml = Vector{Float64}(undef, nb)
    _status(s, ccall((:gpf_block_stats, libgpf), Cint, (Ptr{Cvoid}, Int64, Ptr{Cdouble}, Ptr{Cdouble}), s.handle, block_size, ess, lml))
    return ess, lml
end
# Every block a filter on ITS OWN data: per-block initialisation / update / rejuvenation, one launch each (gpf.h gpf_initialize_blocks,
# gpf_update_blocks, gpf_rejuvenate_blocks).  observations: a (n_obs, n_blocks) Matrix -- column b for block b.
function pf_initialize_blocks(model::NativeModel, model_args::Tuple, observations::Matrix{Float64}, n_particles::Int, block_size::Int; kw...)
    state = DeviceParticleFilterState(model, n_particles; kw...)
    size(observations, 2) == cld(n_particles, block_size) || error("one observation column per block expected")
    _status(state, ccall((:gpf_initialize_blocks, libgpf), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Int64), state.handle, observations, size(observations, 1), block_size))
    return state
end
function pf_update_blocks!(s::DeviceParticleFilterState, new_args::Tuple, argdiffs::Tuple, observations::Matrix{Float64}, block_size::Int)
    size(observations, 2) == cld(s.n_particles, block_size) || error("one observation column per block expected")
    _status(s, ccall((:gpf_update_blocks, libgpf), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Int64), s.handle, observations, size(observations, 1), block_size)); s
end
# every block stratified by itself (src/initialize.jl:92-109 / src/update.jl:193-210 on each sub-state), one launch; strata: the values of the model's discrete latent
function pf_initialize_blocks(model::NativeModel, model_args::Tuple, observations::Matrix{Float64}, strata::Vector{Float64}, n_particles::Int, block_size::Int; layout::Symbol=:contiguous, kw...)
    state = DeviceParticleFilterState(model, n_particles; kw...)
    size(observations, 2) == cld(n_particles, block_size) || error("one observation column per block expected")
    _status(state, ccall((:gpf_initialize_blocks_strata, libgpf), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Int64, Ptr{Cdouble}, Cint, Cint),
                         state.handle, observations, size(observations, 1), block_size, strata, length(strata), layout == :interleaved ? 1 : 0))
    return state
end
function pf_update_blocks!(s::DeviceParticleFilterState, new_args::Tuple, argdiffs::Tuple, observations::Matrix{Float64}, strata::Vector{Float64}, block_size::Int; layout::Symbol=:interleaved)
    size(observations, 2) == cld(s.n_particles, block_size) || error("one observation column per block expected")
    _status(s, ccall((:gpf_update_blocks_strata, libgpf), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Int64, Ptr{Cdouble}, Cint, Cint),
                     s.handle, observations, size(observations, 1), block_size, strata, length(strata), layout == :interleaved ? 1 : 0)); s
end
# "Update with different proposals per view" (test/update.jl:179-189) in one launch: use_proposal[b] = true extends block b with the model's native
# proposal (LocallyOptimal for the LG-SSM: id 1; line_model's fixed proposals: id 2), false with the default one
function pf_update_blocks!(s::DeviceParticleFilterState, new_args::Tuple, argdiffs::Tuple, observations::Matrix{Float64}, block_size::Int, use_proposal::Vector{Bool}, proposal_id::Int=1)
    size(observations, 2) == cld(s.n_particles, block_size) == length(use_proposal) || error("one observation column and one flag per block expected")
    flags = Int32.(use_proposal)
    _status(s, ccall((:gpf_update_blocks_proposal, libgpf), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Int64, Ptr{Int32}, Cint), s.handle, observations, size(observations, 1), block_size, flags, proposal_id)); s
end
function pf_rejuvenate_blocks!(s::DeviceParticleFilterState, kern=nothing, kern_args::Tuple=(), n_iters::Int=1; method::Symbol=:move, only_resampled::Bool=false)
    m = method == :move ? 0 : method == :reweight ? 1 : error("Method $method not recognized.")
    _status(s, ccall((:gpf_rejuvenate_blocks, libgpf), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{UInt64}), s.handle, m, n_iters, only_resampled ? 1 : 0, C_NULL)); s
end
"which blocks the last pf_resample_blocks! resampled"
function block_resampled(s::DeviceParticleFilterState, block_size::Int)
    out = Vector{Cint}(undef, cld(s.n_particles, block_size))
    _status(s, ccall((:gpf_block_resampled, libgpf), Cint, (Ptr{Cvoid}, Ptr{Cint}), s.handle, out))
    return out .!= 0
end

# src/rejuvenate.jl:18-90 with the native kernels (Gen.mh / move_reweight on the current step's latent)
function pf_rejuvenate!(s::DeviceParticleFilterState, kern=nothing, kern_args::Tuple=(), n_iters::Int=1; method::Symbol=:move)
    m = method == :move ? 0 : method == :reweight ? 1 : error("Method $method not recognized.")
    _status(s, ccall((:gpf_rejuvenate, libgpf), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{UInt64}), s.handle, m, n_iters, C_NULL))
    return s
end
# move_reweight(trace, proposal, proposal_args) (src/rejuvenate.jl:134-148) with a native proposal:
#   MoveProposal(1, Float64[])                      the LG-SSM's locally optimal proposal of x_t
#   MoveProposal(2, [q, log(q), log1p(-q)])         line_model: outlier ~ bernoulli(q), the outlier_propose of test/rejuvenate.jl:19-27
struct MoveProposal; id::Int; params::Vector{Float64}; end
function pf_move_reweight!(s::DeviceParticleFilterState, kern, kern_args::Tuple{MoveProposal,Vararg}, n_iters::Int=1)
    mp = kern_args[1]
    _status(s, ccall((:gpf_rejuvenate_proposal, libgpf), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Cint, Cint), s.handle, mp.id, mp.params, length(mp.params), n_iters))
    return s
end
# pf_move_accept!(state, mh, (proposal, proposal_args...), n_iters) (src/rejuvenate.jl:40-53 with Gen.mh(trace, proposal, proposal_args)): the same
# native proposals, accepted iff log(rand()) < weight - fwd_score + bwd_score
function pf_move_accept!(s::DeviceParticleFilterState, kern, kern_args::Tuple{MoveProposal,Vararg}, n_iters::Int=1)
    mp = kern_args[1]
    _status(s, ccall((:gpf_rejuvenate_with_proposal, libgpf), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cdouble}, Cint, Cint, Ptr{UInt64}),
                     s.handle, 0, mp.id, mp.params, length(mp.params), n_iters, Ptr{UInt64}(C_NULL)))
    return s
end
pf_move_accept!(s::DeviceParticleFilterState, kern=nothing, kern_args::Tuple=(), n_iters::Int=1) = pf_rejuvenate!(s, kern, kern_args, n_iters; method=:move)
pf_move_reweight!(s::DeviceParticleFilterState, kern=nothing, kern_args::Tuple=(), n_iters::Int=1) = pf_rejuvenate!(s, kern, kern_args, n_iters; method=:reweight)

# src/resize.jl:16-124, 236-297 -- the handle stays valid, its buffers are reallocated
function _refresh!(s)
    n = Ref{Int64}(0); _status(s, ccall((:gpf_n_particles, libgpf), Cint, (Ptr{Cvoid}, Ref{Int64}), s.handle, n)); s.n_particles = n[]; s
end
function _resize!(s, n::Int, method::Int, priority_fn, check_kw)
    chk = check_kw === true ? 2 : (check_kw === :warn ? 1 : 0)
    alpha = priority_fn === nothing ? NaN : (priority_fn::Tempering).alpha
    invalid = Ref{Cint}(0)
    _status(s, ccall((:gpf_resize, libgpf), Cint, (Ptr{Cvoid}, Int64, Cint, Cdouble, Cint, Ref{Cint}), s.handle, n, method, alpha, chk, invalid))
    check_kw === :warn && invalid[] != 0 && @warn("Invalid weights: resampled with uniform weights.")
    _refresh!(s)
end
pf_multinomial_resize!(s::DeviceParticleFilterState, n::Int; priority_fn=nothing, check=:warn) = _resize!(s, n, 0, priority_fn, check)
pf_residual_resize!(s::DeviceParticleFilterState, n::Int; priority_fn=nothing, check=:warn) = _resize!(s, n, 1, priority_fn, check)
pf_optimal_resize!(s::DeviceParticleFilterState, n::Int; check=:warn) = _resize!(s, n, 3, nothing, check)        # src/resize.jl:149-200
function pf_resize!(s::DeviceParticleFilterState, n::Int, method::Symbol=:multinomial; kwargs...)
    method == :multinomial && return pf_multinomial_resize!(s, n; kwargs...)
    method == :residual && return pf_residual_resize!(s, n; kwargs...)
    method == :optimal && return pf_optimal_resize!(s, n; kwargs...)
    error("Resampling method $method not recognized.")
end
function pf_replicate!(s::DeviceParticleFilterState, k::Int; layout::Symbol=:contiguous)
    _status(s, ccall((:gpf_replicate, libgpf), Cint, (Ptr{Cvoid}, Cint, Cint), s.handle, k, layout != :contiguous)); _refresh!(s)
end
function pf_dereplicate!(s::DeviceParticleFilterState, k::Int; layout::Symbol=:contiguous, method::Symbol=:keepfirst)
    _status(s, ccall((:gpf_dereplicate, libgpf), Cint, (Ptr{Cvoid}, Cint, Cint, Cint), s.handle, k, layout != :contiguous, method == :sample)); _refresh!(s)
end

# src/utils.jl:148-186
function _scalar(s, sym)
    out = Ref{Cdouble}(0)
    _status(s, ccall((sym, libgpf), Cint, (Ptr{Cvoid}, Ref{Cdouble}), s.handle, out))
    return out[]
end
function _vector(s, sym)
    out = Vector{Float64}(undef, s.n_particles)
    _status(s, ccall((sym, libgpf), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Int64), s.handle, out, length(out)))
    return out
end
effective_sample_size(s::DeviceParticleFilterState) = _scalar(s, :gpf_effective_sample_size)
get_ess(s::DeviceParticleFilterState) = effective_sample_size(s)
log_ml_estimate(s::DeviceParticleFilterState) = _scalar(s, :gpf_log_ml_estimate)
get_lml_est(s::DeviceParticleFilterState) = log_ml_estimate(s)
get_log_weights(s::DeviceParticleFilterState) = _vector(s, :gpf_get_log_weights)
get_log_norm_weights(s::DeviceParticleFilterState) = _vector(s, :gpf_get_log_norm_weights)
get_norm_weights(s::DeviceParticleFilterState) = _vector(s, :gpf_get_norm_weights)

# state.parents (test/resample.jl:11)
function Base.getproperty(s::DeviceParticleFilterState, name::Symbol)
    if name === :parents
        out = Vector{Int64}(undef, getfield(s, :n_particles))
        _status(s, ccall((:gpf_get_parents, libgpf), Cint, (Ptr{Cvoid}, Ptr{Int64}, Int64), getfield(s, :handle), out, length(out)))
        return out
    elseif name === :log_weights
        return get_log_weights(s)
    end
    return getfield(s, name)
end

# src/statistics.jl:13-14, 48-50 -- addr = column of the current-step latent (0-based)
function mean(s::DeviceParticleFilterState, addr::Integer)
    out = Ref{Cdouble}(0); _status(s, ccall((:gpf_mean, libgpf), Cint, (Ptr{Cvoid}, Cint, Ref{Cdouble}), s.handle, addr, out)); out[]
end
function var(s::DeviceParticleFilterState, addr::Integer)
    out = Ref{Cdouble}(0); _status(s, ccall((:gpf_var, libgpf), Cint, (Ptr{Cvoid}, Cint, Ref{Cdouble}), s.handle, addr, out)); out[]
end

# past choices along the surviving ancestry, README.md:97-104: mean(state, 5 => 0) == mean(state, 5 => :moving)
# (needs the trajectory store: gpf_history_enable before pf_initialize)
function mean(s::DeviceParticleFilterState, addr::Pair{<:Integer,<:Integer})
    out = Ref{Cdouble}(0); _status(s, ccall((:gpf_history_mean, libgpf), Cint, (Ptr{Cvoid}, Cint, Cint, Ref{Cdouble}), s.handle, addr.first, addr.second, out)); out[]
end
function var(s::DeviceParticleFilterState, addr::Pair{<:Integer,<:Integer})
    out = Ref{Cdouble}(0); _status(s, ccall((:gpf_history_var, libgpf), Cint, (Ptr{Cvoid}, Cint, Cint, Ref{Cdouble}), s.handle, addr.first, addr.second, out)); out[]
end

# stratified initialisation / update (src/initialize.jl:92-109, src/update.jl:193-210): strata = values of the model's discrete latent
function pf_initialize(model::NativeModel, args::Tuple, obs::Vector{Float64}, strata::Vector{Float64}, n::Int; layout::Symbol=:contiguous, kwargs...)
    s = DeviceParticleFilterState(model, n; kwargs...)
    _status(s, ccall((:gpf_initialize_strata, libgpf), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Cint, Cint),
                   s.handle, obs, length(obs), strata, length(strata), layout != :contiguous)); s
end
# ... with a native proposal for the model's other choice (src/initialize.jl:111-129; line_model + LineFixed: test/initialize.jl:66-90)
function pf_initialize(model::NativeModel, args::Tuple, obs::Vector{Float64}, strata::Vector{Float64}, proposal_id::Int, proposal_args::Tuple, n::Int;
                       layout::Symbol=:contiguous, kwargs...)
    s = DeviceParticleFilterState(model, n; kwargs...)
    _status(s, ccall((:gpf_initialize_strata_proposal, libgpf), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Cint, Cint, Cint),
                   s.handle, obs, length(obs), strata, length(strata), layout != :contiguous, proposal_id)); s
end
function pf_update!(s::DeviceParticleFilterState, new_args::Tuple, argdiffs::Tuple, obs::Vector{Float64}, strata::Vector{Float64}; layout::Symbol=:interleaved)
    _status(s, ccall((:gpf_update_strata, libgpf), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Cint, Cint),
                   s.handle, obs, length(obs), strata, length(strata), layout != :contiguous)); s
end

# Gen.sample_unweighted_traces(state, n) (src/utils.jl:189-194): rows of the drawn particles (n x row_width) and their indices
function sample_unweighted_traces(s::DeviceParticleFilterState, n::Int)
    dim = Ref{Cint}(0); w = Ref{Cint}(0)
    _status(s, ccall((:gpf_state_dim, libgpf), Cint, (Ptr{Cvoid}, Ref{Cint}, Ref{Cint}), s.handle, dim, w))
    rows = Matrix{Float64}(undef, w[], n); idx = Vector{Int64}(undef, n)
    _status(s, ccall((:gpf_sample_unweighted, libgpf), Cint, (Ptr{Cvoid}, Int64, Ptr{Cdouble}, Ptr{Int64}), s.handle, n, rows, idx))
    permutedims(rows)[:, 1:dim[]], idx
end

# ---------------------------------------------------------------- multi-GPU: one process per GPU, the exchange inside libgpf
# A shard of a filter of `n_global` particles: rank r of `world` holds the contiguous global range that starts at gid0 (the
# first n_global % world ranks hold one particle more).  The communicator is libgpf's own (RCCL); the host only carries the
# 128-byte id from rank 0 to the others, e.g.  id = MPI.bcast(rank == 0 ? comm_unique_id() : nothing, 0, comm).
mutable struct ShardedDeviceParticleFilterState
    handle::Ptr{Cvoid}
    model::NativeModel
    n_particles::Int          # this shard
    n_global::Int
    rank::Int
    world::Int
end
function comm_unique_id()
    id = Vector{UInt8}(undef, 128)
    ccall((:gpf_comm_unique_id, libgpf), Cint, (Ptr{UInt8},), id) == 0 || error("gpf_comm_unique_id failed")
    return id
end
function pf_initialize(model::NativeModel, model_args::Tuple, observations::Vector{Float64}, n_global::Int, id::Vector{UInt8},
                       rank::Int, world::Int; seed::Integer=1, keep_prev::Bool=false, device::Integer=0)
    base, extra = divrem(n_global, world)
    n = base + (rank < extra ? 1 : 0); gid0 = rank * base + min(rank, extra)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve model begin
        cfg = Ref(GpfConfig(GPF_ABI_VERSION, model.id, length(model.params), keep_prev, pointer(model.params),
                            n, n_global, gid0, UInt64(seed), device, 0, C_NULL))
        ccall((:gpf_create, libgpf), Cint, (Ref{GpfConfig}, Ref{Ptr{Cvoid}}), cfg, h) == 0 ||
            error(unsafe_string(ccall((:gpf_last_error, libgpf), Cstring, (Ptr{Cvoid},), C_NULL)))
    end
    s = ShardedDeviceParticleFilterState(h[], model, n, n_global, rank, world)
    finalizer(x -> ccall((:gpf_destroy, libgpf), Cint, (Ptr{Cvoid},), x.handle), s)
    _status(s, ccall((:gpf_comm_create, libgpf), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint, Cint), s.handle, id, rank, world))
    _status(s, ccall((:gpf_initialize, libgpf), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint), s.handle, observations, length(observations)))
    return s
end
function pf_update!(s::ShardedDeviceParticleFilterState, new_args::Tuple, argdiffs::Tuple, observations::Vector{Float64})
    _status(s, ccall((:gpf_update, libgpf), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint), s.handle, observations, length(observations))); s
end
function pf_rejuvenate!(s::ShardedDeviceParticleFilterState, kern=nothing, kern_args::Tuple=(), n_iters::Int=1; method::Symbol=:move)
    m = method == :move ? 0 : method == :reweight ? 1 : error("Method $method not recognized.")
    _status(s, ccall((:gpf_rejuvenate, libgpf), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{UInt64}), s.handle, m, n_iters, C_NULL)); s
end
# src/resample.jl:19-30 over all shards: ONE call, every collective issued by the library.
# local_only = true: the communication-free "island" resample, pf_resample!(state[shard range], method) on every shard
# (sub-state semantics, src/resample.jl:185-187,205-218), also one call (gpf_resample_local).
function pf_resample!(s::ShardedDeviceParticleFilterState, method::Symbol=:multinomial; check=:warn, sort_particles::Bool=false,
                      local_only::Bool=false, priority_alpha::Union{Nothing,Float64}=nothing)   # priority_fn = w -> priority_alpha * w
    m = method == :multinomial ? 0 : method == :residual ? 1 : method == :stratified ? 2 :
        method == :multinomial_sorted ? 4 : error("Resampling method $method not recognized.")    # (:multinomial_sorted: the opt-in extension, boundary slabs instead of (G-1)/G of all rows)
    chk = check === true ? 2 : (check === :warn ? 1 : 0)
    invalid = Ref{Cint}(0)
    # check = false: NULL for `invalid` keeps the call fully asynchronous (a non-NULL pointer makes the library poll the weight flags)
    inv_ptr = check === false ? Ptr{Cint}(C_NULL) : Base.unsafe_convert(Ptr{Cint}, invalid)
    st = GC.@preserve invalid (local_only ?
        ccall((:gpf_resample_local, libgpf), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Cint}), s.handle, m, sort_particles ? 1 : 0, chk, inv_ptr) :
        (method == :stratified && sort_particles ?   # the reference's default order of the strata (src/resample.jl:145,156-157): the replicated plan
            (priority_alpha === nothing ? ccall((:gpf_shard_resample_sorted, libgpf), Cint, (Ptr{Cvoid}, Cint, Ptr{Cint}), s.handle, chk, inv_ptr) :
                error("sort_particles = true with a priority_fn is not available across shards")) :
         priority_alpha === nothing ?
            ccall((:gpf_shard_resample, libgpf), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cint}), s.handle, m, chk, inv_ptr) :
            ccall((:gpf_shard_resample_tempered, libgpf), Cint, (Ptr{Cvoid}, Cint, Cdouble, Cint, Ptr{Cint}), s.handle, m, priority_alpha, chk, inv_ptr)))
    _status(s, st)
    check === :warn && invalid[] != 0 && @warn("Invalid weights: resampled with uniform weights.")
    return s
end
"one iteration of the README loop on the sharded filter, every rank calls it (gpf.h gpf_shard_step_ess; the unsharded pf_step_ess! with the GLOBAL effective sample size)"
function pf_step_ess!(s::ShardedDeviceParticleFilterState, new_args::Tuple, argdiffs::Tuple, observations::Vector{Float64}; ess_threshold::Real=0.5,
                      method::Symbol=:multinomial, rejuvenate::Union{Nothing,Symbol}=nothing, n_iters::Int=1, check=:warn)
    mid = method == :multinomial ? 0 : method == :residual ? 1 : method == :stratified ? 2 : method == :multinomial_sorted ? 4 : error("Resampling method $method not recognized.")
    rid = rejuvenate === nothing ? -1 : rejuvenate == :move ? 0 : rejuvenate == :reweight ? 1 : error("Method $rejuvenate not recognized.")
    chk = check === true ? 2 : (check === :warn ? 1 : 0)
    resampled = Ref{Cint}(0); invalid = Ref{Cint}(0)
    GC.@preserve resampled invalid begin
        st = ccall((:gpf_shard_step_ess, libgpf), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cdouble, Cint, Cint, Cint, Cint, Ptr{Cint}, Ptr{Cint}),
                   s.handle, observations, length(observations), Float64(ess_threshold), mid, chk, rid, n_iters,
                   Base.unsafe_convert(Ptr{Cint}, resampled), chk == 0 ? Ptr{Cint}(C_NULL) : Base.unsafe_convert(Ptr{Cint}, invalid))
    end
    _status(s, st)
    check === :warn && invalid[] != 0 && @warn("Invalid weights: resampled with uniform weights.")
    return resampled[] != 0
end
"""exchange plan of the i.i.d. resamplers across shards: :push (default) or :pull (gpf.h gpf_comm_set_plan); the same on every rank"""
function shard_plan!(s::ShardedDeviceParticleFilterState, plan::Symbol)
    plan in (:push, :pull) || error("exchange plan :$plan: :push or :pull")
    _status(s, ccall((:gpf_comm_set_plan, libgpf), Cint, (Ptr{Cvoid}, Cint), s.handle, plan == :pull ? 1 : 0)); s
end
"(calls, entries sent to other ranks, entries received from other ranks, bytes per entry) of this rank's sharded resamples so far (gpf.h gpf_comm_traffic)"
function shard_traffic(s::ShardedDeviceParticleFilterState; reset::Bool=false)
    out = zeros(Int64, 4)
    _status(s, ccall((:gpf_comm_traffic, libgpf), Cint, (Ptr{Cvoid}, Ptr{Int64}, Cint), s.handle, out, reset ? 1 : 0))
    (out[1], out[2], out[3], out[4])
end
function shard_plan(s::ShardedDeviceParticleFilterState)
    p = Ref{Cint}(0)
    _status(s, ccall((:gpf_comm_plan, libgpf), Cint, (Ptr{Cvoid}, Ptr{Cint}), s.handle, p))
    p[] == 1 ? :pull : :push
end
"""how the rows of the resamplers with ascending targets (:stratified with sort_particles = false, the opt-in sorted multinomial) cross shards: :p2p = peer
stores into the destination ranks' slot-addressed receive windows (no host wait, no ncclGroup; the default where the mailboxes are up), :rccl = packed
entries through grouped ncclSend / ncclRecv (gpf.h gpf_comm_set_exchange); the same on every rank"""
function shard_exchange!(s::ShardedDeviceParticleFilterState, mode::Symbol)
    mode in (:p2p, :rccl, :p2p_all) || error("exchange mode :$mode: :p2p, :p2p_all or :rccl")
    _status(s, ccall((:gpf_comm_set_exchange, libgpf), Cint, (Ptr{Cvoid}, Cint), s.handle, mode == :rccl ? 0 : (mode == :p2p ? 1 : 2))); s
end
function shard_exchange(s::ShardedDeviceParticleFilterState)
    m = Ref{Cint}(0)
    _status(s, ccall((:gpf_comm_exchange, libgpf), Cint, (Ptr{Cvoid}, Ptr{Cint}), s.handle, m))
    m[] == 0 ? :rccl : (m[] == 1 ? :p2p : :p2p_all)
end
"(us per grouped exchange of `entries` packed entries with every peer, GB/s per link, us per mailbox round, us of an empty launch) on this machine (gpf.h gpf_comm_calibrate; collective)"
function shard_calibrate(s::ShardedDeviceParticleFilterState, entries::Integer; reps::Integer=20)
    out = zeros(Float64, 4)
    _status(s, ccall((:gpf_comm_calibrate, libgpf), Cint, (Ptr{Cvoid}, Int64, Cint, Ptr{Cdouble}), s.handle, entries, reps, out))
    (out[1], out[2], out[3], out[4])
end
"per-phase microseconds of the sharded resamples since `shard_phase_timing!(s, true)` (gpf.h gpf_phase_times): summaries, plan, pack, host wait, exchange, commit + propagate"
shard_phase_timing!(s::ShardedDeviceParticleFilterState, on::Bool) = (_status(s, ccall((:gpf_phase_timing, libgpf), Cint, (Ptr{Cvoid}, Cint), s.handle, on ? 1 : 0)); s)
function shard_phase_times(s::ShardedDeviceParticleFilterState)
    us = zeros(Float64, 6); n = Ref{Int64}(0)
    _status(s, ccall((:gpf_phase_times, libgpf), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ref{Int64}), s.handle, us, n))
    (us, n[])
end
effective_sample_size(s::ShardedDeviceParticleFilterState) = _scalar(s, :gpf_shard_effective_sample_size)
get_ess(s::ShardedDeviceParticleFilterState) = effective_sample_size(s)
log_ml_estimate(s::ShardedDeviceParticleFilterState) = _scalar(s, :gpf_shard_log_ml_estimate)
get_lml_est(s::ShardedDeviceParticleFilterState) = log_ml_estimate(s)

end # module
