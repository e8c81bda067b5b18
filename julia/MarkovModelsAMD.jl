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
# `using MarkovModels` loads CUDA.jl (a hard dependency of the package: src/MarkovModels.jl:7-8); on an AMD box CUDA.jl
# loads without a device (CUDA.functional() == false) and nothing here calls into it.
#
# Seam (see INTEGRATION.md): the reference dispatches on CuArray storage
# (src/fsm.jl:42-48, src/inference.jl:14-26, src/linalg.jl:163,240,335).  Here the
# dispatch is on a device handle type, one level up: one ccall per inference call.

module MarkovModelsAMD

using MarkovModels
using SparseArrays
using Semirings
using AMDGPU
using Adapt
using Libdl
using LinearAlgebra

# The package's own generic functions are EXTENDED with methods for the device types below (`using MarkovModels`
# already exports compile, batch, pdfposteriors, ...: src/MarkovModels.jl:14-45 -- defining functions of the same
# names in this module would shadow them instead).
import MarkovModels: compile, batch, pdfposteriors, αrecursion, βrecursion, totalsum, totalcumsum

# what this module adds to the package's API
export ROCCompiledFSM, ROCBatch, to_device, compile_many, bestpath, maxstateposteriors, pdfposteriors_generic, last_redo_count,
       last_fallback_count, last_exact_first, team_xcd_stats, reserve_ex!, set_deterministic!, set_posterior_floor!, set_exact_policy!,
       set_mark_policy!, set_gamma_mode!, set_rccl, allreduce_logz, allgather_ttl, ROCSparseCSR, ROCSparseVec, elmul!, eldiv!,
       compiled_cache_clear!, compiled_cache_limits!

const LIB = get(ENV, "MARKOVMODELS_AMD_LIB", "libmarkovmodels_amd.so")

const MM_LOG, MM_TROPICAL, MM_PROB = Cint(0), Cint(1), Cint(2)
const MM_CSC = Cint(0)
const MM_ABI_VERSION = 4      # include/markovmodels_amd.h: the version every ccall tuple of this file was written against

# A library of another ABI version would be called with the wrong argument lists: refuse it when the module is loaded.
function __init__()
    v = ccall((:mm_abi_version, LIB), Cint, ())
    v == MM_ABI_VERSION || error("$(LIB) has ABI version $(v), MarkovModelsAMD.jl was written against $(MM_ABI_VERSION)")
    nothing
end

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

"""
    Adapt.adapt_structure(ROCArray, cfsm::CompiledFSM) -> ROCCompiledFSM

The reference's own seam (src/inference.jl:14-26: `adapt_structure(::Type{<:CuArray}, cfsm)`, reached through
`cfsm |> gpu` / `adapt(CuArray, cfsm)`): `adapt(ROCArray, cfsm)` hands the compiled FSM to the engine.  The FSM type
itself (src/fsm.jl:42-48) has no device form here: the engine takes compiled FSMs (`compile(fsm, Ĉ)` first).
"""
Adapt.adapt_structure(::Type{<:ROCArray}, c::CompiledFSM) = to_device(c)

"""
    compile_many(fsms::Vector{FSM{K}}, Ĉs) -> Vector{ROCCompiledFSM{K}}

`compile.(fsms, Ĉs)` + device adapt for a mini-batch of NEW graphs in one call (mm_fsm_create_many: host threads, one device
allocation, one copy) -- what examples/test_cuda.jl:74-78 does every training step for the numerator graphs.
"""
function compile_many(fsms::Vector{<:FSM{K}}, Ĉs::Vector{<:AbstractSparseMatrix}; threads::Integer = 0) where K
    n = length(fsms)
    n == length(Ĉs) || throw(DimensionMismatch("one Ĉ per FSM"))
    Ts = [SparseMatrixCSC(f.T̂) for f in fsms]
    αs = [SparseVector(f.α̂) for f in fsms]
    vals = [val.(nonzeros(T)) for T in Ts]
    avals = [val.(nonzeros(a)) for a in αs]
    ainds = [SparseArrays.nonzeroinds(a) for a in αs]
    s2ps = map(Ĉs) do Ĉ
        Ct = SparseMatrixCSC(copy(Ĉ'))
        all(diff(Ct.colptr) .== 1) || throw(ArgumentError("Ĉ must have exactly one entry per row"))
        Vector{Int32}(Ct.rowval)
    end
    F = eltype(vals[1])
    S1 = Int64[size(T, 1) for T in Ts]
    NZ = Int64[nnz(T) for T in Ts]
    NI = Int64[nnz(a) for a in αs]
    P1 = Int32[size(Ĉ, 2) for Ĉ in Ĉs]
    out = fill(Ptr{Cvoid}(C_NULL), n)
    GC.@preserve Ts αs vals avals ainds s2ps begin
        p_ptr = Ptr{Cvoid}[pointer(T.colptr) for T in Ts]
        p_idx = Ptr{Cvoid}[pointer(T.rowval) for T in Ts]
        p_val = Ptr{Cvoid}[pointer(v) for v in vals]
        p_ai = Ptr{Cvoid}[pointer(a) for a in ainds]
        p_av = Ptr{Cvoid}[pointer(a) for a in avals]
        p_s2p = Ptr{Int32}[pointer(s) for s in s2ps]
        check(ccall((:mm_fsm_create_many, LIB), Cint,
            (Int64, Cint, Cint, Cint, Cint, Cint, Ptr{Int64}, Ptr{Int64}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}},
             Ptr{Int64}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Int32}}, Ptr{Int32}, Cint, Ptr{Ptr{Cvoid}}),
            n, semiring_id(K), MM_CSC, 8, 1, sizeof(F), S1, NZ, p_ptr, p_idx, p_val, NI, p_ai, p_av, p_s2p, P1, threads, out))
    end
    map(1:n) do i
        obj = ROCCompiledFSM{K}(out[i], S1[i], P1[i])
        finalizer(o -> ccall((:mm_fsm_destroy, LIB), Cint, (Ptr{Cvoid},), o.handle), obj)
        obj
    end
end

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
    keep::Any                            # device arrays an asynchronous call on this batch still reads (replaced by the next call)
end

function batch(f1::ROCCompiledFSM{K}, fs::ROCCompiledFSM{K}...) where K
    all_ = ROCCompiledFSM{K}[f1, fs...]
    hs = Ptr{Cvoid}[f.handle for f in all_]
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:mm_batch_create, LIB), Cint, (Ptr{Ptr{Cvoid}}, Int64, Ref{Ptr{Cvoid}}), hs, length(hs), h))
    obj = ROCBatch{K}(h[], all_, f1.P1 - 1, nothing)
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
function pdfposteriors(b::ROCBatch, V::ROCArray{Float32,3}, lens = nothing; out::Union{Nothing, ROCArray{Float32,3}} = nothing)
    P, N, B = size(V)
    P == b.P || throw(DimensionMismatch("V has $P pdfs, the FSMs $(b.P)"))
    out === nothing || size(out) == (B, P, N) || throw(DimensionMismatch("out must be B × P × N"))
    γ = out === nothing ? ROCArray{Float32}(undef, B, P, N) : out      # (`out`: the array set_gamma_mode!(accumulate = true) adds into)
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

"""
    pdfposteriors(fsm::FSM{K}, V̂s::Vector{<:ROCMatrix}, Ĉs; expanded = true, seqlengths = nothing) -> (γ, ttl)

The reference's own signature (src/inference.jl:145): `fsm` the `rawunion` of the batch's FSMs (src/fsmops.jl:28-36), `V̂s`
one (P+1) × (N+1) matrix per utterance as `expand` made them, `Ĉs` the state maps -- examples/test_cuda.jl:128 runs
unchanged but for the array type.  The blocks of the union are cut apart again by the rows of the `Ĉs` (block b has
size(Ĉs[b], 1) states) and looked up in a PROCESS-WIDE cache of compiled graphs keyed by their content (`hash` of the
block's T̂, α̂ and Ĉ): a graph is packed and uploaded the first time it is seen -- the misses of one call together, by
`compile_many` -- and found again on every later call (the denominator graph; an utterance's numerator across epochs).
The whole batch is remembered per `(fsm, Ĉs)` object pair as well, so a loop that passes the same objects pays one
dictionary look-up.  `V̂s` are stacked on the device like `vcat(V̂s...)` (:146).
With Float32 log-semiring (or, since round 6, ProbSemiring) FSMs, one-hot `Ĉs` and `expanded = true` the fast kernels run on
V̂[1:P, 1:N, b] IN PLACE (strides).
The lengths: `seqlengths` (what `expand` was given: a Vector or ROCVector of integers) keeps the call asynchronous; without it
they are read off the phony row on the host, and the form `expand` gives a matrix (phony row zero(K) up to the length and
one(K) after, real rows zero(K) beyond it) is CHECKED -- a matrix of another form goes to the generic entry like
`expanded = false`.  Anything else (Float64; other semirings; general sparse `Ĉs`) goes to `pdfposteriors_generic`.
Nothing synchronises the device: the batch lives in the cache, and the stacked V̂ and the lengths stay referenced by the batch
(`b.keep`) until the next call on it replaces them -- stream order then guarantees the earlier call has read them.
"""
function pdfposteriors(fsm::FSM{K}, V̂s::Vector{<:ROCMatrix{T}}, Ĉs::Vector{<:AbstractSparseMatrix}; expanded::Bool = true,
                       seqlengths = nothing) where {K, T}
    B = length(V̂s)
    B == length(Ĉs) || throw(DimensionMismatch("one Ĉ per utterance"))
    P1, N1 = size(V̂s[1])
    all(size(v) == (P1, N1) for v in V̂s) || throw(DimensionMismatch("all V̂ must share one (P+1) × (N+1) shape"))
    all(size(Ĉ, 2) == P1 for Ĉ in Ĉs) || throw(DimensionMismatch("V̂ has $P1 rows, a Ĉ has another number of pdfs"))
    b, onehot = _cached_batch(fsm, Ĉs)
    V̂ = cat(V̂s...; dims = 3)                                    # (P+1) × (N+1) × B on the device
    # (ProbSemiring{Float32} too: the library keeps a log-semiring twin of such FSMs and runs the fast kernels on log V̂ -- V̂ then
    # holds likelihoods, ttl comes back as a probability: mm_pdfposteriors_f32 in the header)
    fast = expanded && onehot && (K <: LogSemiring || K <: ProbSemiring) && T === Float32 && eltype(val(one(K))) === Float32
    z0, o1 = K <: ProbSemiring ? (0f0, 1f0) : (-Inf32, 0f0)      # val(zero(K)), val(one(K))
    P, N = P1 - 1, N1 - 1
    lens = nothing
    if fast
        if seqlengths !== nothing
            lens = seqlengths isa ROCArray ? ROCArray{Int32}(seqlengths) : ROCArray(Int32.(collect(seqlengths)))
        else
            # expand (src/inference.jl:54-60): the phony row is zero(K) up to seqlength and one(K) after, the real rows are
            # zero(K) beyond it -- read and verified on the host (one round trip; pass `seqlengths` to avoid it)
            ph = Array(V̂[P1, :, :])                             # (N+1) × B
            L = vec(sum(ph .== z0, dims = 1))
            step_ok = all(all(ph[1:L[j], j] .== z0) && all(ph[L[j]+1:end, j] .== o1) && L[j] <= N for j in 1:B)
            tail_ok = step_ok && all(mapreduce(x -> x == z0, &, V̂[1:P, L[j]+1:N1, j]; init = true) for j in 1:B if L[j] < N1)   # (a GPU reduction returns a host scalar)
            fast = step_ok && tail_ok
            lens = ROCArray(Int32.(L))
        end
    end
    fast || return pdfposteriors_generic(b, V̂, onehot ? nothing : Ĉs)
    γ = ROCArray{Float32}(undef, B, P, N)
    ttl = ROCArray{Float32}(undef, B)
    # V̂ (b, n, p) -> p + P1*n + P1*N1*b: the kernels read the real pdfs and frames in place
    check(ccall((:mm_pdfposteriors_f32, LIB), Cint,
        (Ptr{Cvoid}, Ptr{Float32}, Int64, Int64, Ptr{Int32}, Int64, Ptr{Float32}, Int64, Int64, Int64,
         Ptr{Float32}, Ptr{Cvoid}),
        b.handle, pointer(V̂), P1 * N1, P1, pointer(lens), N, pointer(γ), 1, B * P, B, pointer(ttl),
        AMDGPU.stream().stream))
    # (asynchronous: the batch lives in the cache; the stacked V̂ and the lengths are kept until the NEXT call on this batch
    # replaces them -- stream order then guarantees this call has read them)
    b.keep = (V̂, lens)
    γ, ttl
end

# ---- the compiled-graph cache behind the method above: two BOUNDED least-recently-used tables
# _COMPILED: 128-bit content key of (T̂ block, α̂ block, Ĉ) -> ROCCompiledFSM (two independent 64-bit hashes of the same fields: a
#            collision of both is not a practical event; the Python mirror uses a 128-bit digest likewise)
# _BATCHES:  identity of the caller's fsm object -> (Ĉs object, ROCBatch, onehot): a loop that passes the same objects pays one look-up.
# Both hold strong references, so both are bounded: a training loop that builds a new numerator union every step would otherwise
# keep every ROCBatch (its device workspace, its descriptors, its last stacked V̂ through `keep`) until the device is full.  A batch
# that falls out of _BATCHES is destroyed at once (mm_batch_destroy frees with hipFree, which waits for the device: a call still in
# flight on it finishes first); compiled graphs that fall out of _COMPILED are freed by their finalizers when no batch uses them.
mutable struct _LRU{Kt, Vt}
    d::Dict{Kt, Tuple{Vt, Int}}       # key -> (value, stamp of its last use)
    clock::Int
    limit::Int
end
_LRU{Kt, Vt}(limit::Integer) where {Kt, Vt} = _LRU{Kt, Vt}(Dict{Kt, Tuple{Vt, Int}}(), 0, Int(limit))
function _lru_get(c::_LRU, k)
    e = get(c.d, k, nothing)
    e === nothing && return nothing
    c.clock += 1
    c.d[k] = (e[1], c.clock)
    e[1]
end
"Insert; returns the values that fell out (oldest first) so that the caller can release what they own."
function _lru_put!(c::_LRU{Kt, Vt}, k, v) where {Kt, Vt}
    c.clock += 1
    c.d[k] = (v, c.clock)
    out = Vt[]
    while length(c.d) > c.limit
        oldest = first(c.d)
        for kv in c.d
            kv.second[2] < oldest.second[2] && (oldest = kv)
        end
        push!(out, oldest.second[1])
        delete!(c.d, oldest.first)
    end
    out
end
const _COMPILED = _LRU{NTuple{2, UInt64}, Any}(8192)
const _BATCHES = _LRU{UInt, Any}(16)
"Forget every compiled graph and batch the reference-shaped `pdfposteriors` has cached."
function compiled_cache_clear!()
    for e in values(_BATCHES.d)
        finalize(e[1][2])
    end
    empty!(_BATCHES.d); empty!(_COMPILED.d); nothing
end
"Bounds of the two tables: compiled graphs (default 8192) and whole batches (default 16)."
compiled_cache_limits!(; graphs::Integer = 8192, batches::Integer = 16) = (_COMPILED.limit = graphs; _BATCHES.limit = batches; nothing)

_isonehot(Ĉ::AbstractSparseMatrix{K}) where K =
    all(diff(SparseMatrixCSC(copy(Ĉ')).colptr) .== 1) && all(x -> x == one(K), nonzeros(Ĉ))   # (one(ProbSemiring) = 1: not iszero ∘ val)

function _cached_batch(fsm::FSM{K}, Ĉs) where K
    id = objectid(fsm)
    hit = _lru_get(_BATCHES, id)
    # (objectid of an immutable FSM is the identity of its fields: the same T̂ / α̂ / λ arrays; the entry keeps `fsm` itself, so the id
    # cannot be reused by another object while the entry lives)
    hit !== nothing && hit[4] === fsm && hit[1] === Ĉs && return hit[2], hit[3]
    T̂, α̂ = SparseMatrixCSC(fsm.T̂), SparseVector(fsm.α̂)
    sum(size(Ĉ, 1) for Ĉ in Ĉs) == size(T̂, 1) || throw(DimensionMismatch("the Ĉs' rows do not add up to the states of fsm"))
    onehot = all(_isonehot, Ĉs)
    keys = NTuple{2, UInt64}[]
    blocks = Dict{NTuple{2, UInt64}, Any}()
    lo = 0
    for Ĉ in Ĉs
        r = lo+1:lo+size(Ĉ, 1)
        Tb, ab = T̂[r, r], α̂[r]
        # (a general sparse Ĉ rides along as an argument of the generic entry: its FSM handle gets a placeholder map)
        Cb = onehot ? Ĉ : sparse(1:size(Ĉ, 1), [fill(1, size(Ĉ, 1) - 1); size(Ĉ, 2)], fill(one(K), size(Ĉ, 1)), size(Ĉ, 1), size(Ĉ, 2))
        content = (K, Tb.colptr, Tb.rowval, val.(nonzeros(Tb)), SparseArrays.nonzeroinds(ab), val.(nonzeros(ab)), Cb.colptr, Cb.rowval, size(Cb))
        k = (hash(content, UInt(0x243f6a8885a308d3)), hash(content, UInt(0x13198a2e03707344)))
        push!(keys, k)
        haskey(_COMPILED.d, k) || haskey(blocks, k) || (blocks[k] = (Tb, ab, Cb))
        lo = last(r)
    end
    if !isempty(blocks)                      # the misses of this call, compiled together
        ks = collect(Base.keys(blocks))
        if length(ks) == 1
            Tb, ab, Cb = blocks[ks[1]]
            _lru_put!(_COMPILED, ks[1], _create(K, Tb, ab, Cb))
        else
            fs = [FSM(blocks[k][2], blocks[k][1], eltype(fsm.λ)[]) for k in ks]   # (the struct's own constructor, src/fsm.jl:7-17, 44: labels play no part in inference)
            made = compile_many(fs, [blocks[k][3] for k in ks])
            for (k, c) in zip(ks, made)
                _lru_put!(_COMPILED, k, c)
            end
        end
    end
    # (the graphs of THIS call are fetched before anything else is inserted: a batch larger than the table's bound still finds them)
    cfs = ROCCompiledFSM{K}[]
    for k in keys
        c = _lru_get(_COMPILED, k)
        c === nothing && error("compiled-graph cache: the bound ($(_COMPILED.limit)) is smaller than one batch's distinct graphs")
        push!(cfs, c::ROCCompiledFSM{K})
    end
    b = batch(cfs...)
    for old in _lru_put!(_BATCHES, id, (Ĉs, b, onehot, fsm))
        finalize(old[2])                      # mm_batch_destroy now: workspace, descriptors, and the V̂ it kept
    end
    b, onehot
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

"... and how many of those the float64 exact kernels handed on to the log-domain kernels (normally 0)."
function last_fallback_count(b::ROCBatch)
    n = Ref{Int64}(0)
    check(ccall((:mm_batch_last_fallback_count, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Int64}), b.handle, AMDGPU.stream().stream, n))
    Int(n[])
end
"true if the last `pdfposteriors` call skipped the float32 kernels (the inputs of the call before were hard)."
last_exact_first(b::ROCBatch) = ccall((:mm_batch_last_exact_first, LIB), Cint, (Ptr{Cvoid},), b.handle) != 0
"(workgroups of the team kernels' launches since the last call whose whole team sat on one XCD, all such workgroups): a measurement aid; synchronises the device."
function team_xcd_stats(b::ROCBatch)
    out = zeros(Cint, 2)
    check(ccall((:mm_batch_team_xcd_stats, LIB), Cint, (Ptr{Cvoid}, Ptr{Cint}), b.handle, out))
    Int(out[1]), Int(out[2])
end
"Size the generic entry's workspace (before capturing `pdfposteriors_generic` in a hipGraph)."
reserve_ex!(b::ROCBatch, ::Type{T}, N1::Integer) where T =
    (check(ccall((:mm_batch_reserve_ex, LIB), Cint, (Ptr{Cvoid}, Cint, Int64), b.handle, sizeof(T), N1)); b)

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
    B == length(b.fsms) || throw(DimensionMismatch("V̂ holds $B matrices, the batch $(length(b.fsms)) FSMs"))
    # rows of V̂ against the pdfs of the map in force (src/inference.jl:146-150); the C entry checks the same (MM_ERR_DIM)
    for (i, f) in enumerate(b.fsms)
        want = Ĉs === nothing ? f.P1 : size(Ĉs[i], 2)
        P1 == want || throw(DimensionMismatch("V̂ has $P1 rows, the state map of utterance $i has $want pdfs (was expand() applied?)"))
    end
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
            (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Cint, Int32, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Int64, Int64, Int64,
             Ptr{Cvoid}, Ptr{Cvoid}),
            b.handle, isempty(maps) ? C_NULL : pointer(maps), sizeof(T), Int32(P1), pointer(V̂), P1 * N1, P1, N1,
            pointer(γ), 1, B * (P1 - 1), B, pointer(ttl), AMDGPU.stream().stream))
        isempty(maps) || AMDGPU.synchronize()           # (the call is asynchronous: the maps must outlive it)
    finally
        foreach(h -> ccall((:mm_statemap_destroy, LIB), Cint, (Ptr{Cvoid},), h), maps)
    end
    γ, ttl
end

"""
    set_exact_policy!(b::ROCBatch, policy::Symbol = :auto)

Which linear-domain kernels a shared-graph batch starts with (mm_batch_set_exact_policy): `:auto` (float32 first; the
wide-exponent kernels first while the last FINISHED call left utterances marked -- read without synchronising, so a pipelined
caller's kernel choice depends on host / device timing), `:f32_first`, `:f64_first` (both: the launches of a call are a
function of the call alone, identical call sequences give identical bits).  Pins the path of `αrecursion` / `βrecursion` too
(`:auto`: the log-domain kernel first while the last export handed it more than half of the utterances; `:f64_first`: that kernel alone).
"""
set_exact_policy!(b::ROCBatch, policy::Symbol = :auto) =
    (check(ccall((:mm_batch_set_exact_policy, LIB), Cint, (Ptr{Cvoid}, Cint), b.handle,
                 policy === :auto ? 0 : policy === :f32_first ? 1 : policy === :f64_first ? 2 : throw(ArgumentError("policy")))); b)

"""
    set_mark_policy!(b::ROCBatch, policy::Symbol = :decide)

What a range mark of the float32 linear-domain kernels means (mm_batch_set_mark_policy): `:decide` -- the finish kernel clears it
when its two criteria say nothing that matters was lost (log γ within 1e-4 relative above ~1e-24, an absolute error below ~1e-27
for smaller posteriors) --, `:keep` -- it always stays and the exact kernels compute the utterance: the relative bar down to 1e-30,
at the price of a second pass on inputs that raise marks.
"""
set_mark_policy!(b::ROCBatch, policy::Symbol = :decide) =
    (check(ccall((:mm_batch_set_mark_policy, LIB), Cint, (Ptr{Cvoid}, Cint), b.handle,
                 policy === :decide ? 0 : policy === :keep ? 1 : throw(ArgumentError("policy")))); b)

"""
    set_gamma_mode!(b::ROCBatch; accumulate = false, scale = 1f0)

What `pdfposteriors(b, V, lens)` does with the posteriors (mm_batch_set_gamma_mode): `γ = scale * γ`, or with `accumulate = true`
`γ_out += scale * γ` into the array passed as `out` -- the numerator call with `scale = -1` on the array the denominator call has just
written leaves the LF-MMI gradient `γ_den - γ_num` (examples/test_cuda.jl:140-152) there without a third pass.  Batches of the wave
kernel only (numerator graphs); others throw `MMError(-4, …)`.
"""
set_gamma_mode!(b::ROCBatch; accumulate::Bool = false, scale::Real = 1f0) =
    (check(ccall((:mm_batch_set_gamma_mode, LIB), Cint, (Ptr{Cvoid}, Cint, Cfloat), b.handle, accumulate ? 1 : 0, Float32(scale))); b)

# ---- the reference's semiring linear algebra on the device (src/linalg.jl): mul! and the sparse-vector broadcast --------------
"""
    ROCSparseCSR{K}(A::SparseMatrixCSC{K})  /  ROCSparseVec{K}(x::SparseVector{K})

What `CuSparseMatrixCSR(adapt(CuArray, A))` / `adapt(CuArray, x)` are on the CUDA path (src/linalg.jl:80-131, test/test_linalg.jl:96):
rowPtr / colVal (`Cint`, 1-based) / nzVal of A on the device.  `LinearAlgebra.mul!(c, A, b)` and `mul!(C, A, B, α, β)` on them are
single calls of the HIP library (mm_spmv / mm_spmm), generic in `K` ∈ {Log, Tropical, Prob}Semiring{Float32 | Float64} like the
reference's methods (src/linalg.jl:163-184, 240-262); dense operands are `ROCArray{K}` (a one-field immutable wrapper around a
float: bit-identical to an array of floats).
"""
struct ROCSparseCSR{K}
    rowPtr::ROCVector{Cint}
    colVal::ROCVector{Cint}
    nzVal::ROCVector{K}
    dims::NTuple{2, Int}
end
function ROCSparseCSR(A::SparseMatrixCSC{K}) where K
    At = SparseMatrixCSC(copy(A'))                     # CSC of A' = CSR of A
    ROCSparseCSR{K}(ROCArray(Cint.(At.colptr)), ROCArray(Cint.(At.rowval)), ROCArray(nonzeros(At)), size(A))
end
Base.size(A::ROCSparseCSR) = A.dims
Base.size(A::ROCSparseCSR, i::Integer) = A.dims[i]
struct ROCSparseVec{K}
    nzInd::ROCVector{Cint}
    nzVal::ROCVector{K}
    n::Int
end
ROCSparseVec(x::SparseVector{K}) where K = ROCSparseVec{K}(ROCArray(Cint.(SparseArrays.nonzeroinds(x))), ROCArray(nonzeros(x)), length(x))
_floatbytes(::Type{K}) where K = sizeof(K)              # K wraps one float

function LinearAlgebra.mul!(c::ROCVector{K}, A::ROCSparseCSR{K}, b::ROCVector{K}) where K
    check(ccall((:mm_spmv, LIB), Cint,
        (Cint, Cint, Int64, Int64, Int64, Ptr{Cint}, Ptr{Cint}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}),
        semiring_id(K), _floatbytes(K), size(A, 1), size(A, 2), length(A.nzVal), pointer(A.rowPtr), pointer(A.colVal), 1,
        pointer(A.nzVal), pointer(b), length(b), pointer(c), length(c), AMDGPU.stream().stream))
    c
end
function LinearAlgebra.mul!(C::ROCMatrix{K}, A::ROCSparseCSR{K}, B::ROCMatrix{K}, α::Number = true, β::Number = false) where K
    check(ccall((:mm_spmm, LIB), Cint,
        (Cint, Cint, Int64, Int64, Int64, Ptr{Cint}, Ptr{Cint}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Int64, Int64,
         Int64, Cdouble, Ptr{Cvoid}),
        semiring_id(K), _floatbytes(K), size(A, 1), size(A, 2), length(A.nzVal), pointer(A.rowPtr), pointer(A.colVal), 1,
        pointer(A.nzVal), pointer(B), size(B, 1), size(B, 2), stride(B, 2), pointer(C), size(C, 1), size(C, 2), stride(C, 2),
        Float64(β), AMDGPU.stream().stream))
    C
end
"elmul!(out, y, x::ROCSparseVec) / eldiv!(out, x::ROCSparseVec, y) (src/linalg.jl:287-328): out = zero(K), out[i] = x[i] ⊗ y[i] (⊘) at x's stored entries."
function _svdv!(op::Integer, out::ROCVector{K}, x::ROCSparseVec{K}, y::ROCVector{K}) where K
    check(ccall((:mm_svdv, LIB), Cint,
        (Cint, Cint, Cint, Int64, Int64, Ptr{Cint}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}),
        semiring_id(K), _floatbytes(K), op, x.n, length(x.nzVal), pointer(x.nzInd), 1, pointer(x.nzVal), pointer(y), length(y),
        pointer(out), length(out), AMDGPU.stream().stream))
    out
end
elmul!(out::ROCVector{K}, y::ROCVector{K}, x::ROCSparseVec{K}) where K = _svdv!(0, out, x, y)
eldiv!(out::ROCVector{K}, x::ROCSparseVec{K}, y::ROCVector{K}) where K = _svdv!(1, out, x, y)

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
above 1e-30 and sends utterances with sharp emissions to the exact kernels (~1.4x the time since round 5; 3-6x before).
`1f-12` keeps them on the fast path (LF-MMI gradients do not see the difference).
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
