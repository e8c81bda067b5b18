# SPDX-License-Identifier: MIT
#
# MarkovModelsAMD.jl -- the binding a MarkovModels.jl maintainer would add so that
# `pdfposteriors`, `αrecursion` and `βrecursion` run on an AMD MI355X through
# libmarkovmodels_amd.so (C ABI: include/markovmodels_amd.h).
#
# NOT TESTED IN THIS REPOSITORY: the build image has no Julia.  The tested binding
# of the same ABI is the Python/ctypes one in markovmodels.jl_amd/_lib.py; this file
# mirrors it call for call.  Device memory comes from AMDGPU.jl (ROCArray).
#
# Seam (see INTEGRATION.md): the reference dispatches on CuArray storage
# (src/fsm.jl:42-48, src/inference.jl:14-26, src/linalg.jl:163,240,335).  Here the
# dispatch is on a device handle type, one level up: one ccall per inference call.

module MarkovModelsAMD

using MarkovModels
using SparseArrays
using Semirings
using AMDGPU
using Libdl

# The package's own generic functions are EXTENDED with methods for the device types below (`using MarkovModels`
# already exports compile, batch, pdfposteriors, ...: src/MarkovModels.jl:14-45 -- defining functions of the same
# names in this module would shadow them instead).
import MarkovModels: compile, batch, pdfposteriors, αrecursion, βrecursion, totalsum, totalcumsum

# what this module adds to the package's API
export ROCCompiledFSM, ROCBatch, to_device, bestpath, maxstateposteriors, pdfposteriors_generic, last_redo_count,
       set_deterministic!, set_posterior_floor!, set_rccl, allreduce_logz, allgather_ttl

const LIB = get(ENV, "MARKOVMODELS_AMD_LIB", "libmarkovmodels_amd.so")

const MM_LOG, MM_TROPICAL, MM_PROB = Cint(0), Cint(1), Cint(2)
const MM_CSC = Cint(0)

struct MMError <: Exception
    code::Cint
    msg::String
end

function check(rc::Cint)
    rc == 0 && return
    msg = unsafe_string(ccall((:mm_last_error, LIB), Cstring, ()))
    rc == -2 && throw(DimensionMismatch(msg))      # src/linalg.jl:166-167
    throw(MMError(rc, msg))
end

semiring_id(::Type{<:LogSemiring}) = MM_LOG
semiring_id(::Type{<:TropicalSemiring}) = MM_TROPICAL
semiring_id(::Type{<:ProbSemiring}) = MM_PROB      # the generic entry only (pdfposteriors_generic)

"Device-resident compiled FSM: replaces adapt(CuArray, ::CompiledFSM) (src/inference.jl:14-26)."
mutable struct ROCCompiledFSM{K}
    handle::Ptr{Cvoid}
    S1::Int
    P1::Int
end

"""
    compile(ROCCompiledFSM, fsm, Ĉ)

`compile(fsm, Ĉ)` (src/inference.jl:11-12) + device adapt.  `Ĉ` is the state map
of examples/prepare-lfmmi-graphs.jl:15-23 (exactly one stored entry per row).
"""
compile(::Type{ROCCompiledFSM}, fsm::FSM{K}, Ĉ::AbstractSparseMatrix) where K =
    _create(K, SparseMatrixCSC(fsm.T̂), SparseVector(fsm.α̂), Ĉ)

"""
    to_device(cfsm::CompiledFSM) -> ROCCompiledFSM

What `adapt(CuArray, cfsm)` / `cfsm |> gpu` is on the CUDA path (src/inference.jl:14-26): the compiled FSM of the
reference, as `compile(fsm, Ĉ)` made it on the host, handed to the engine.  `batch(to_device(c)...)` then replaces
`batch(c...) |> gpu`.
"""
to_device(c::CompiledFSM{K}) where K = _create(K, SparseMatrixCSC(c.T̂), SparseVector(c.α̂), c.Ĉ)

function _create(::Type{K}, T̂::SparseMatrixCSC, α̂::SparseVector, Ĉ::AbstractSparseMatrix) where K
    Ct = SparseMatrixCSC(copy(Ĉ'))                      # column s of Ĉ' = row s of Ĉ
    all(diff(Ct.colptr) .== 1) || throw(ArgumentError("Ĉ must have exactly one entry per row"))
    state2pdf = Vector{Int32}(Ct.rowval)                # 1-based pdf of every state
    vals = val.(nonzeros(T̂))
    avals = val.(nonzeros(α̂))
    T = eltype(vals)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve T̂ α̂ state2pdf vals avals begin
        check(ccall((:mm_fsm_create, LIB), Cint,
            (Cint, Int64, Int64, Cint, Cint, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64,
             Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Int32}, Int32, Ref{Ptr{Cvoid}}),
            semiring_id(K), size(T̂, 1), nnz(T̂), MM_CSC, 8, 1, sizeof(T),
            pointer(T̂.colptr), pointer(T̂.rowval), pointer(vals), nnz(α̂),
            pointer(SparseArrays.nonzeroinds(α̂)), pointer(avals), pointer(state2pdf),
            Int32(size(Ĉ, 2)), h))
    end
    obj = ROCCompiledFSM{K}(h[], size(T̂, 1), size(Ĉ, 2))
    finalizer(o -> ccall((:mm_fsm_destroy, LIB), Cint, (Ptr{Cvoid},), o.handle), obj)
    obj
end

"batch(cfsm...) (src/inference.jl:28-36); repeating one handle shares its storage."
mutable struct ROCBatch{K}
    handle::Ptr{Cvoid}
    fsms::Vector{ROCCompiledFSM{K}}      # keeps the FSM handles alive
    P::Int
end

function batch(f1::ROCCompiledFSM{K}, fs::ROCCompiledFSM{K}...) where K
    all_ = ROCCompiledFSM{K}[f1, fs...]
    hs = Ptr{Cvoid}[f.handle for f in all_]
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:mm_batch_create, LIB), Cint, (Ptr{Ptr{Cvoid}}, Int64, Ref{Ptr{Cvoid}}), hs, length(hs), h))
    obj = ROCBatch{K}(h[], all_, f1.P1 - 1)
    finalizer(o -> ccall((:mm_batch_destroy, LIB), Cint, (Ptr{Cvoid},), o.handle), obj)
    obj
end

"""
    pdfposteriors(b::ROCBatch, V::ROCArray{Float32,3}, lens)

Drop-in for pdfposteriors(fsm, V̂s, Ĉs) (src/inference.jl:145-161) /
pdfposteriors2 (:164-180).  `V` is P × N × B (Julia column-major: pdf fastest, the
layout of `vcat(V̂s...)` without the phony row/frame: expand() happens inside);
`lens` a ROCVector{Int32} or nothing.  Returns (γ::ROCArray{Float32,3} of size
B × P × N like the reference, ttl::ROCVector{Float32}).
"""
function pdfposteriors(b::ROCBatch, V::ROCArray{Float32,3}, lens = nothing)
    P, N, B = size(V)
    P == b.P || throw(DimensionMismatch("V has $P pdfs, the FSMs $(b.P)"))
    γ = ROCArray{Float32}(undef, B, P, N)
    ttl = ROCArray{Float32}(undef, B)
    lp = lens === nothing ? Ptr{Int32}(C_NULL) : Ptr{Int32}(pointer(lens))
    # strides in elements: V (b, n, p) -> p + P*n + P*N*b ; γ (b, n, p) -> b + B*p + B*P*n
    check(ccall((:mm_pdfposteriors_f32, LIB), Cint,
        (Ptr{Cvoid}, Ptr{Float32}, Int64, Int64, Ptr{Int32}, Int64, Ptr{Float32}, Int64, Int64, Int64,
         Ptr{Float32}, Ptr{Cvoid}),
        b.handle, pointer(V), P * N, P, lp, N, pointer(γ), 1, B * P, B, pointer(ttl),
        AMDGPU.stream().stream))
    γ, ttl
end

function _recursion(sym::Symbol, b::ROCBatch, V::ROCArray{Float32,3}, lens)
    P, N, B = size(V)
    total = ccall((:mm_batch_total_states, LIB), Int64, (Ptr{Cvoid},), b.handle)
    out = ROCArray{Float32}(undef, total, N + 1)          # (ΣS1) × (N+1) like state_A / state_B
    lp = lens === nothing ? Ptr{Int32}(C_NULL) : Ptr{Int32}(pointer(lens))
    check(ccall((sym, LIB), Cint,
        (Ptr{Cvoid}, Ptr{Float32}, Int64, Int64, Ptr{Int32}, Int64, Ptr{Float32}, Int64, Ptr{Cvoid}),
        b.handle, pointer(V), P * N, P, lp, N, pointer(out), total, AMDGPU.stream().stream))
    out
end

"αrecursion(α̂, T̂ᵀ, ĈV̂) (src/inference.jl:62-74) as pdfposteriors calls it (:150-152)."
αrecursion(b::ROCBatch, V::ROCArray{Float32,3}, lens = nothing) =
    _recursion(:mm_alpharecursion_f32, b, V, lens)
"βrecursion(T̂, ĈV̂) (src/inference.jl:99-110)."
βrecursion(b::ROCBatch, V::ROCArray{Float32,3}, lens = nothing) =
    _recursion(:mm_betarecursion_f32, b, V, lens)

"""
    bestpath(b::ROCBatch{<:TropicalSemiring}, V, lens) -> (paths, scores)

docs/src/inference.md:6 (absent from src/ at v0.10.0).  States are returned 1-based.
"""
function bestpath(b::ROCBatch{K}, V::ROCArray{Float32,3}, lens = nothing) where K <: TropicalSemiring
    P, N, B = size(V)
    path = ROCArray{Int32}(undef, N, B)
    score = ROCArray{Float32}(undef, B)
    lp = lens === nothing ? Ptr{Int32}(C_NULL) : Ptr{Int32}(pointer(lens))
    check(ccall((:mm_viterbi_f32, LIB), Cint,
        (Ptr{Cvoid}, Ptr{Float32}, Int64, Int64, Ptr{Int32}, Int64, Ptr{Int32}, Int64, Ptr{Float32},
         Ptr{Int32}, Int64, Ptr{Cvoid}),
        b.handle, pointer(V), P * N, P, lp, N, pointer(path), N, pointer(score), C_NULL, 0,
        AMDGPU.stream().stream))
    Array(path) .+ Int32(1), Array(score)
end

"""
    maxstateposteriors(b::ROCBatch{<:TropicalSemiring}, V, lens) -> Matrix{Float32}

docs/src/inference.md:5 (absent from src/ at v0.10.0): the max-marginals alpha (*) beta (/) best of the tropical
semiring, (sum of S+1) x (N+1), 0 along a best path.
"""
maxstateposteriors(b::ROCBatch{K}, V::ROCArray{Float32,3}, lens = nothing) where K <: TropicalSemiring =
    _recursion(:mm_maxstateposteriors_f32, b, V, lens)   # one device call: tropical alpha, beta and the combination

"""
    totalsum(b::ROCBatch, n) / totalcumsum(b::ROCBatch, n) -> Vector{Float32}

src/algorithms.jl:8-29 for every FSM of the batch (natural-log values of the batch's semiring).
"""
function _totalsum(b::ROCBatch, n::Integer, cumulative::Bool)
    out = ROCArray{Float32}(undef, length(b.fsms))
    check(ccall((:mm_totalsum_f32, LIB), Cint, (Ptr{Cvoid}, Int64, Cint, Ptr{Float32}, Ptr{Cvoid}),
        b.handle, n, cumulative, pointer(out), AMDGPU.stream().stream))
    Array(out)
end
totalsum(b::ROCBatch, n::Integer) = _totalsum(b, n, false)
totalcumsum(b::ROCBatch, n::Integer) = _totalsum(b, n, true)

"""
    last_redo_count(b::ROCBatch) -> Int

How many utterances of the last `pdfposteriors` call the fast kernels handed to the exact ones (computed twice: only
the time differs).  Synchronises the current stream.
"""
function last_redo_count(b::ROCBatch)
    n = Ref{Int64}(0)
    check(ccall((:mm_batch_last_redo_count, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Int64}), b.handle, AMDGPU.stream().stream, n))
    Int(n[])
end

"""
    pdfposteriors_generic(b::ROCBatch{K}, V̂::ROCArray{T,3}, Ĉs = nothing) -> (γ, ttl)

The reference's `pdfposteriors(fsm, V̂s, Ĉs)` (src/inference.jl:145-161) over its whole argument space: any semiring the
FSMs were created with (Log, Tropical, Prob), `T` Float32 or Float64, `V̂` the (P+1) × (N+1) × B stack of matrices as
`expand` or anything else made them (semiring VALUES, `val.(...)`), `Ĉs` `nothing` (every FSM's own one-hot map) or
one sparse matrix per utterance with any number of weighted entries per row.  Returns γ of size B × P × N and ttl.
A plain kernel (alpha and beta materialised like the reference does): correctness first; the fast kernels are behind
`pdfposteriors(b, V, lens)`.
"""
function pdfposteriors_generic(b::ROCBatch{K}, V̂::ROCArray{T,3}, Ĉs = nothing) where {K, T <: Union{Float32, Float64}}
    P1, N1, B = size(V̂)
    γ = ROCArray{T}(undef, B, P1 - 1, N1 - 1)
    ttl = ROCArray{T}(undef, B)
    maps = Ptr{Cvoid}[]
    if Ĉs !== nothing
        for Ĉ in Ĉs                                   # CSR of Ĉ = CSC of Ĉ'
            Ct = SparseMatrixCSC(copy(Ĉ'))
            vals = Vector{Float64}(val.(nonzeros(Ct)))
            h = Ref{Ptr{Cvoid}}(C_NULL)
            GC.@preserve Ct vals check(ccall((:mm_statemap_create, LIB), Cint,
                (Cint, Int64, Int32, Int64, Cint, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}),
                semiring_id(K), size(Ĉ, 1), Int32(size(Ĉ, 2)), nnz(Ct), 8, 1, 8,
                pointer(Ct.colptr), pointer(Ct.rowval), pointer(vals), h))
            push!(maps, h[])
        end
    end
    try
        # strides in elements: V̂ (b, n, p) -> p + P1*n + P1*N1*b ; γ (b, n, p) -> b + B*p + B*P*n
        check(ccall((:mm_pdfposteriors_ex, LIB), Cint,
            (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Cint, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Int64, Int64, Int64,
             Ptr{Cvoid}, Ptr{Cvoid}),
            b.handle, isempty(maps) ? C_NULL : pointer(maps), sizeof(T), pointer(V̂), P1 * N1, P1, N1,
            pointer(γ), 1, B * (P1 - 1), B, pointer(ttl), AMDGPU.stream().stream))
    finally
        foreach(h -> ccall((:mm_statemap_destroy, LIB), Cint, (Ptr{Cvoid},), h), maps)
    end
    γ, ttl
end

"""
    set_rccl(lib)

Tell the engine which RCCL the communicators passed to `allreduce_logz` / `allgather_ttl` belong to: the handle
`Libdl.dlopen` returned for it (Julia opens libraries with local visibility, so the engine cannot find RCCL among the
process's global symbols; it never opens an RCCL of its own).
"""
set_rccl(lib::Ptr{Cvoid}) = check(ccall((:mm_set_rccl, LIB), Cint, (Ptr{Cvoid},), lib))
set_rccl(path::AbstractString) = set_rccl(Libdl.dlopen(path))

"No float atomics in the item kernel (the wave kernel, the default numerator path, is deterministic anyway): bit-identical γ on every run."
set_deterministic!(b::ROCBatch, on::Bool = true) =
    (check(ccall((:mm_batch_set_deterministic, LIB), Cint, (Ptr{Cvoid}, Cint), b.handle, on ? 1 : 0)); b)

"""
    set_posterior_floor!(b::ROCBatch, floor = 1f-30)

Posteriors below `floor` may come out as 0 from the fast (linear-domain) kernels; the default keeps every posterior
above 1e-30 and sends utterances with sharp emissions to the exact kernels (3-6x the time).  `1f-12` keeps them on the
fast path (LF-MMI gradients do not see the difference).
"""
set_posterior_floor!(b::ROCBatch, floor::Real = 1f-30) =
    (check(ccall((:mm_batch_set_posterior_floor, LIB), Cint, (Ptr{Cvoid}, Cfloat), b.handle, Float32(floor))); b)

"""
    allreduce_logz(comm, ttl::ROCVector{Float32}) -> Float64
    allgather_ttl(comm, ttl::ROCVector{Float32}, Bmax, world) -> Matrix{Float32}   (Bmax × world)

The only exchange of a sharded batch (utterances are independent: src/fsmops.jl:28-36): the total log-likelihood
the LF-MMI loss consumes (examples/test_cuda.jl:140-152).  `comm` is an RCCL communicator handle (ncclComm_t) of this
process, one process per GPU; call `set_rccl` once with the library it was made with.  `world` = ranks of `comm`.
"""
function allreduce_logz(comm::Ptr{Cvoid}, ttl::ROCVector{Float32})
    s = ROCArray{Float64}(undef, 1)
    check(ccall((:mm_allreduce_logz, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64, Ptr{Float64}, Ptr{Cvoid}),
        comm, pointer(ttl), length(ttl), pointer(s), AMDGPU.stream().stream))
    Array(s)[1]
end
function allgather_ttl(comm::Ptr{Cvoid}, ttl::ROCVector{Float32}, Bmax::Integer, world::Integer)
    pad = ROCArray(fill(-Inf32, Bmax))
    copyto!(pad, 1, ttl, 1, length(ttl))
    out = ROCArray{Float32}(undef, world * Bmax)
    check(ccall((:mm_allgather_ttl, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64, Ptr{Float32}, Ptr{Cvoid}),
        comm, pointer(pad), Bmax, pointer(out), AMDGPU.stream().stream))
    reshape(Array(out), Bmax, world)
end

end # module
