# PeriodicSchurMI355X.jl — reference-side binding of libpsd_mi355x.so (include/psd_mi355x.h).
#
# Loaded next to RalphAS/PeriodicSchurDecompositions.jl v0.1.6, this module re-points the package's hot path
# (pschur!/pschur, phessenberg!, gpschur, both ordschur! families, checkpsd) at the MI355X engine for Float64 and
# ComplexF64 operands; everything else of the package (Krylov driver, eigvecs, generic element types) keeps running the
# Julia code, which now reaches the engine through these methods wherever it calls them.
#
#     using PeriodicSchurDecompositions, LinearAlgebra
#     include("julia/PeriodicSchurMI355X.jl"); using .PeriodicSchurMI355X
#     F = pschur!(A, :R)                      # A::Vector{Matrix{Float64}} -> PeriodicSchur, computed on the GPU
#
# There is no Julia toolchain in the build image of this repository, so this file is reviewed, not executed, here
# (INTEGRATION.md); the tested mirror of the same interface is periodicschurdecompositions.jl_amd/__init__.py.
# Citations are file:line of the reference sources (/root/reference/src, `PSD.jl` = PeriodicSchurDecompositions.jl).
module PeriodicSchurMI355X

using LinearAlgebra
using LinearAlgebra: checksquare
import PeriodicSchurDecompositions
import PeriodicSchurDecompositions: pschur!, pschur, phessenberg!, gpschur, PeriodicSchur, GeneralizedPeriodicSchur,
                                    checkpsd
const PSD = PeriodicSchurDecompositions

export set_train!, engine_version

const libpsd = get(ENV, "LIBPSD_MI355X", joinpath(@__DIR__, "..", "periodicschurdecompositions.jl_amd", "libpsd_mi355x.so"))

const INFO_NOCONV = 1_000_000
const INFO_NOTIMPL = 2_000_000
const INFO_RUNTIME = 3_000_000

const BlasElt = Union{Float64, ComplexF64}

# ---------------------------------------------------------------------------------------------------------------------
# context: one per task (the C ABI allows one call at a time per context)
mutable struct Ctx
    ptr::Ptr{Cvoid}
    function Ctx(device::Integer = 0)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:psd_create, libpsd), Cint, (Ref{Ptr{Cvoid}}, Cint), r, device)
        rc == 0 || error("psd_create failed (info=$rc): no usable MI355X; there is no CPU fallback in the library")
        c = new(r[])
        finalizer(x -> (ccall((:psd_destroy, libpsd), Cint, (Ptr{Cvoid},), x.ptr); nothing), c)
        c
    end
end
const _ctx = Ref{Union{Nothing, Ctx}}(nothing)
ctx() = something(_ctx[], (_ctx[] = Ctx()))

engine_version() = unsafe_string(ccall((:psd_version, libpsd), Cstring, ()))
"bulges per multishift train (0 or 1: the reference's one-shift-one-sweep iteration, sweep for sweep)"
set_train!(m::Integer) = ccall((:psd_set_train, libpsd), Cint, (Ptr{Cvoid}, Cint), ctx().ptr, m)
"period shard of the Schur vectors (one context per GPU, one process each): rank `r` of `w` forms and updates only the Z_j of its slice"
set_shard!(r::Integer, w::Integer) = ccall((:psd_set_shard, libpsd), Cint, (Ptr{Cvoid}, Cint, Cint), ctx().ptr, r, w)
"factor-sliced sweep windows: G workgroups per window, each with the blocks of its slice of the period (clamped to what fits one GPU)"
set_slices!(G::Integer) = ccall((:psd_set_slices, libpsd), Cint, (Ptr{Cvoid}, Cint), ctx().ptr, G)
"1 if this context's Hessenberg reductions take the pipe form (fixed when the context was created)"
hess_pipe() = ccall((:psd_get_hess_pipe, libpsd), Cint, (Ptr{Cvoid},), ctx().ptr)

# info -> the exception the reference throws at the same place
function _throw(info::Integer)
    info == 0 && return nothing
    info < 0 && throw(ArgumentError("libpsd_mi355x: argument $(-info) invalid"))
    info >= INFO_RUNTIME && error("libpsd_mi355x: HIP runtime failure (code $(info - INFO_RUNTIME))")
    info >= INFO_NOTIMPL && throw(PSD.NotImplemented("not implemented in libpsd_mi355x"))          # PSD.jl:30
    info >= INFO_NOCONV && error("convergence failed at level $(info - INFO_NOCONV)")               # PSD.jl:892
    error("libpsd_mi355x: info=$info")
end
# ordschur! codes: 2000 + j  IllConditionedException(j) (ordschur.jl:61), 3000 SingularException (utils.jl:128)
function _throw_ord(info::Integer)
    2000 <= info < 3000 && throw(PSD.IllConditionedException(info - 2000))
    info == 3000 && throw(LinearAlgebra.SingularException(0))
    _throw(info)
end

_ptrs(M::Vector{<:StridedMatrix{T}}) where {T} = Ptr{Float64}[Ptr{Float64}(pointer(m)) for m in M]
function _dense(M::AbstractVector{<:AbstractMatrix{T}}) where {T <: BlasElt}
    # the C ABI takes column-major n x n blocks with ld = n: Matrix{T} as they are, anything else is copied
    Matrix{T}[(m isa Matrix{T}) ? m : Matrix{T}(m) for m in M]
end
function _check(A)
    isempty(A) && throw(DimensionMismatch("empty sequence"))
    n = checksquare(A[1])
    for a in A                                                                                    # PSD.jl:214-222
        checksquare(a) == n || throw(DimensionMismatch("matrices must have equal order"))
    end
    n
end
_values(α, β, sc) = α ./ β .* 2.0 .^ sc                                                           # generalized.jl:74-76

# ---------------------------------------------------------------------------------------------------------------------
# phessenberg!(A) — PSD.jl:213-259: same packed layout (Hessenberg / QR objects over the caller's matrices) and tau
function phessenberg!(A::Vector{Matrix{T}}) where {T <: BlasElt}
    p = length(A); n = _check(A)
    tau = Matrix{T}(undef, n, p); info = Ref{Cint}(0)
    Ap = _ptrs(A)
    f = T <: Real ? :psd_d_phessenberg : :psd_z_phessenberg
    GC.@preserve A tau begin
        if T <: Real
            ccall((:psd_d_phessenberg, libpsd), Cint,
                  (Ptr{Cvoid}, Cint, Cint, Ptr{Ptr{Float64}}, Ptr{Float64}, Ptr{Cvoid}, Ref{Cint}),
                  ctx().ptr, n, p, Ap, tau, C_NULL, info)
        else
            ccall((:psd_z_phessenberg, libpsd), Cint,
                  (Ptr{Cvoid}, Cint, Cint, Ptr{Ptr{Float64}}, Ptr{Float64}, Ptr{Cvoid}, Ref{Cint}),
                  ctx().ptr, n, p, Ap, Ptr{Float64}(pointer(tau)), C_NULL, info)
        end
    end
    _throw(info[])
    H1 = Hessenberg(A[1], tau[1:(n - 1), 1])                                                      # PSD.jl:249-251
    pH = [LinearAlgebra.QR(A[j], tau[:, j]) for j in 2:p]                                         # PSD.jl:252-256
    return H1, pH
end

# ---------------------------------------------------------------------------------------------------------------------
# pschur!(A, lr; wantZ, wantT, maxitfac), Float64 — PSD.jl:120-152
function pschur!(A::Vector{Matrix{Float64}}, lr::Symbol = :R; wantZ::Bool = true, wantT::Bool = true, maxitfac = 30)
    orient = PSD.char_lr(lr)                                                                      # PSD.jl:155-177
    p = length(A); n = _check(A)
    Z = wantZ ? [Matrix{Float64}(undef, n, n) for _ in 1:p] : Matrix{Float64}[]
    wr = Vector{Float64}(undef, n); wi = similar(wr)
    si = Ref{Cint}(0); info = Ref{Cint}(0)
    Ap = _ptrs(A); Zp = _ptrs(Z)
    GC.@preserve A Z wr wi begin
        ccall((:psd_d_pschur, libpsd), Cint,
              (Ptr{Cvoid}, Cint, Cint, Ptr{Ptr{Float64}}, Ptr{UInt8}, Cchar, Cint, Cint, Cint,
               Ptr{Ptr{Float64}}, Ptr{Float64}, Ptr{Float64}, Ref{Cint}, Ptr{Cvoid}, Ptr{Int32}, Int64, Ref{Cint}),
              ctx().ptr, n, p, Ap, C_NULL, orient, wantT, wantZ, maxitfac,
              wantZ ? Zp : C_NULL, wr, wi, si, C_NULL, C_NULL, 0, info)
    end
    _throw(info[])
    js = Int(si[])                                       # 1 for :R, p for :L (PSD.jl:1092-1094)
    T1 = A[js]; T = [A[j] for j in 1:p if j != js]
    wantZ || (Z = [similar(T1, 0, 0)])                   # PSD.jl:1074-1076
    PeriodicSchur(T1, T, Z, complex.(wr, wi), orient, js)
end

# pschur!(A, lr; ...), ComplexF64 — PSD.jl:1106-1111 (the all-true signature of generalized.jl:108-137)
function pschur!(A::Vector{Matrix{ComplexF64}}, lr::Symbol = :R; wantZ::Bool = true, wantT::Bool = true, maxitfac = 30)
    gps = pschur!(A, trues(length(A)), lr; wantZ = wantZ, wantT = wantT, maxitfac = maxitfac)
    PeriodicSchur(gps.T1, gps.T, gps.Z, gps.values, gps.orientation, gps.schurindex)              # PSD.jl:1110
end

# pschur!(A, S, lr; wantZ, wantT) — generalized.jl:138-146 (ComplexF64), rgeneralized.jl:3-45 (Float64)
function pschur!(A::Vector{Matrix{T}}, S::AbstractVector{Bool}, lr::Symbol = :R;
                 wantZ::Bool = true, wantT::Bool = true, maxitfac = (T <: Real ? 120 : 30),
                 aggressive::Bool = false) where {T <: BlasElt}
    orient = PSD.char_lr(lr)
    p = length(A); n = _check(A)
    length(S) == p || throw(DimensionMismatch("one sign per factor"))
    (orient == 'L' ? S[p] : S[1]) || throw(ArgumentError("The leftmost entry in S must be true"))  # generalized.jl:140
    Z = wantZ ? [Matrix{T}(undef, n, n) for _ in 1:p] : Matrix{T}[]
    α = zeros(ComplexF64, n); β = zeros(Float64, n); sc = zeros(Int32, n)
    si = Ref{Cint}(0); info = Ref{Cint}(0)
    Ap = _ptrs(A); Zp = _ptrs(Z); Sb = UInt8.(S)
    GC.@preserve A Z α β sc Sb begin
        if T <: Real
            ccall((:psd_d_gpschur, libpsd), Cint,
                  (Ptr{Cvoid}, Cint, Cint, Ptr{Ptr{Float64}}, Ptr{UInt8}, Cchar, Cint, Cint, Cint, Ptr{Ptr{Float64}},
                   Ptr{ComplexF64}, Ptr{Float64}, Ptr{Int32}, Ref{Cint}, Ptr{Cvoid}, Ref{Cint}),
                  ctx().ptr, n, p, Ap, Sb, orient, wantT, wantZ, maxitfac, wantZ ? Zp : C_NULL, α, β, sc, si, C_NULL, info)
        else
            ccall((:psd_z_pschur, libpsd), Cint,
                  (Ptr{Cvoid}, Cint, Cint, Ptr{Ptr{Float64}}, Ptr{UInt8}, Cchar, Cint, Cint, Cint, Ptr{Ptr{Float64}},
                   Ptr{ComplexF64}, Ptr{Float64}, Ptr{Int32}, Ref{Cint}, Ptr{Cvoid}, Ptr{Int32}, Int64, Ref{Cint}),
                  ctx().ptr, n, p, Ap, Sb, orient, wantT, wantZ, maxitfac, wantZ ? Zp : C_NULL, α, β, sc, si, C_NULL,
                  C_NULL, 0, info)
        end
    end
    _throw(info[])
    js = Int(si[]); T1 = A[js]; Tv = [A[j] for j in 1:p if j != js]
    wantZ || (Z = [similar(T1, 0, 0)])
    βT = T <: Real ? β : complex.(β)
    GeneralizedPeriodicSchur(collect(Bool, S), js, T1, Tv, Z, α, βT, Int.(sc), orient)            # generalized.jl:31-48
end

# pschur (copying) — PSD.jl:108-113, generalized.jl:87-91: the package's generic methods copy and call pschur!,
# which dispatches to the methods above for Matrix{Float64} / Matrix{ComplexF64}; other matrix types are densified
function pschur(A::AbstractVector{<:AbstractMatrix{T}}, lr::Symbol = :R; kwargs...) where {T <: BlasElt}
    pschur!([Matrix{T}(a) for a in A], lr; kwargs...)
end
function pschur(A::AbstractVector{<:AbstractMatrix{T}}, S::AbstractVector{Bool}, lr::Symbol = :R;
                kwargs...) where {T <: BlasElt}
    pschur!([Matrix{T}(a) for a in A], S, lr; kwargs...)
end

# gpschur(As, Bs) — generalized.jl:1191-1211: the pairs interleaved with alternating signature, complexified
function gpschur(As::AbstractVector{<:AbstractMatrix{T}}, Bs::AbstractVector{<:AbstractMatrix{T}};
                 kwargs...) where {T <: BlasElt}
    Cs, Ss = PSD._mkpsargs(collect(As), Bs)
    pschur!(Matrix{ComplexF64}[Matrix{ComplexF64}(c) for c in Cs], Ss; kwargs...)
end

# ---------------------------------------------------------------------------------------------------------------------
# pschur!(H1, Hs; wantT, wantZ, Q, maxitfac, rev), Float64 — PSD.jl:322-330 (callers: :150, krylov.jl:583,591)
function pschur!(H1::StridedMatrix{Float64}, Hs::AbstractVector{<:StridedMatrix{Float64}};
                 wantZ::Bool = true, wantT::Bool = true, Q::Union{Nothing, Vector{<:StridedMatrix{Float64}}} = nothing,
                 maxitfac = 30, rev::Bool = false)
    p = length(Hs) + 1; n = checksquare(H1)
    H = _dense(vcat([H1], collect(Hs)))
    Zs = wantZ ? (Q === nothing ? [Matrix{Float64}(I, n, n) for _ in 1:p] : _dense(Q)) : Matrix{Float64}[]
    wr = Vector{Float64}(undef, n); wi = similar(wr); info = Ref{Cint}(0)
    Hp = _ptrs(H); Qp = _ptrs(Zs)
    GC.@preserve H Zs wr wi begin
        ccall((:psd_d_pschur_hess, libpsd), Cint,
              (Ptr{Cvoid}, Cint, Cint, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Cint, Cint, Cint, Ptr{Float64}, Ptr{Float64},
               Ptr{Cvoid}, Ptr{Int32}, Int64, Ref{Cint}),
              ctx().ptr, n, p, Hp, wantZ ? Qp : C_NULL, wantT, wantZ, maxitfac, wr, wi, C_NULL, C_NULL, 0, info)
    end
    _throw(info[])
    # results back into the caller's storage (in-place contract), then the orientation bookkeeping of PSD.jl:1078-1092
    H[1] === H1 || copyto!(H1, H[1])
    for l in 1:(p - 1); H[l + 1] === Hs[l] || copyto!(Hs[l], H[l + 1]); end
    wantZ || (Zs = [similar(H1, 0, 0)])
    λ = complex.(wr, wi)
    if rev
        Zr = wantZ ? vcat([Zs[1]], [Zs[p + 2 - l] for l in 2:p]) : Zs
        return PeriodicSchur(H1, [Hs[p - l] for l in 1:(p - 1)], Zr, λ, 'L', p)
    end
    PeriodicSchur(H1, collect(Hs), Zs, λ, 'R', 1)
end

# pschur!(H1, Hs, S; wantT, wantZ, Q, maxitfac, rev) — generalized.jl:166-175 (ComplexF64), rgeneralized.jl:49-59 (Float64)
function pschur!(H1::StridedMatrix{T}, Hs::AbstractVector{<:StridedMatrix{T}}, S::AbstractVector{Bool};
                 wantZ::Bool = true, wantT::Bool = true, Q::Union{Nothing, Vector{<:StridedMatrix{T}}} = nothing,
                 maxitfac = (T <: Real ? 120 : 30), rev::Bool = false, aggressive::Bool = false) where {T <: BlasElt}
    p = length(Hs) + 1; n = checksquare(H1)
    S[1] || throw(ArgumentError("The leftmost entry in S must be true"))
    H = _dense(vcat([H1], collect(Hs)))
    Zs = wantZ ? (Q === nothing ? [Matrix{T}(I, n, n) for _ in 1:p] : _dense(Q)) : Matrix{T}[]
    α = zeros(ComplexF64, n); β = zeros(Float64, n); sc = zeros(Int32, n); info = Ref{Cint}(0)
    Hp = _ptrs(H); Qp = _ptrs(Zs); Sb = UInt8.(S)
    f = T <: Real ? :psd_d_gpschur_hess : :psd_z_pschur_hess
    GC.@preserve H Zs α β sc Sb begin
        if T <: Real
            ccall((:psd_d_gpschur_hess, libpsd), Cint,
                  (Ptr{Cvoid}, Cint, Cint, Ptr{Ptr{Float64}}, Ptr{UInt8}, Ptr{Ptr{Float64}}, Cint, Cint, Cint,
                   Ptr{ComplexF64}, Ptr{Float64}, Ptr{Int32}, Ptr{Cvoid}, Ptr{Int32}, Int64, Ref{Cint}),
                  ctx().ptr, n, p, Hp, Sb, wantZ ? Qp : C_NULL, wantT, wantZ, maxitfac, α, β, sc, C_NULL, C_NULL, 0, info)
        else
            ccall((:psd_z_pschur_hess, libpsd), Cint,
                  (Ptr{Cvoid}, Cint, Cint, Ptr{Ptr{Float64}}, Ptr{UInt8}, Ptr{Ptr{Float64}}, Cint, Cint, Cint,
                   Ptr{ComplexF64}, Ptr{Float64}, Ptr{Int32}, Ptr{Cvoid}, Ptr{Int32}, Int64, Ref{Cint}),
                  ctx().ptr, n, p, Hp, Sb, wantZ ? Qp : C_NULL, wantT, wantZ, maxitfac, α, β, sc, C_NULL, C_NULL, 0, info)
        end
    end
    _throw(info[])
    H[1] === H1 || copyto!(H1, H[1])
    for l in 1:(p - 1); H[l + 1] === Hs[l] || copyto!(Hs[l], H[l + 1]); end
    wantZ || (Zs = [similar(H1, 0, 0)])
    βT = T <: Real ? β : complex.(β)
    if rev                                                                                        # generalized.jl:905-925
        Zr = wantZ ? vcat([Zs[1]], [Zs[p + 2 - l] for l in 2:p]) : Zs
        return GeneralizedPeriodicSchur(reverse(collect(Bool, S)), p, H1, [Hs[p - l] for l in 1:(p - 1)], Zr, α, βT,
                                        Int.(sc), 'L')
    end
    GeneralizedPeriodicSchur(collect(Bool, S), 1, H1, collect(Hs), Zs, α, βT, Int.(sc), 'R')
end

# _phessenberg!(A, S; wantQ) — generalized.jl:988-1082 (signed Hessenberg-triangular reduction)
function PSD._phessenberg!(A::Vector{Matrix{T}}, S::AbstractVector{Bool}; wantQ::Bool = true) where {T <: BlasElt}
    p = length(A); n = _check(A)
    Q = wantQ ? [Matrix{T}(undef, n, n) for _ in 1:p] : Matrix{T}[]
    info = Ref{Cint}(0); Ap = _ptrs(A); Qp = _ptrs(Q); Sb = UInt8.(S)
    GC.@preserve A Q Sb begin
        if T <: Real
            ccall((:psd_d_gphessenberg, libpsd), Cint,
                  (Ptr{Cvoid}, Cint, Cint, Ptr{Ptr{Float64}}, Ptr{UInt8}, Ptr{Ptr{Float64}}, Ptr{Cvoid}, Ref{Cint}),
                  ctx().ptr, n, p, Ap, Sb, wantQ ? Qp : C_NULL, C_NULL, info)
        else
            ccall((:psd_z_gphessenberg, libpsd), Cint,
                  (Ptr{Cvoid}, Cint, Cint, Ptr{Ptr{Float64}}, Ptr{UInt8}, Ptr{Ptr{Float64}}, Ptr{Cvoid}, Ref{Cint}),
                  ctx().ptr, n, p, Ap, Sb, wantQ ? Qp : C_NULL, C_NULL, info)
        end
    end
    _throw(info[])
    return A, Q
end

# _rphessenberg!(Ap, A, Q) — rhessx.jl:55-109 (the Krylov driver's projected problem, krylov.jl:809)
function PSD._rphessenberg!(Ap::Matrix{T}, A::Vector{Matrix{T}}, Q::Union{Nothing, Vector{Matrix{T}}} = nothing) where {T <: BlasElt}
    m, n = size(Ap); p = length(A) + 1
    (m == n || m == n + 1) || throw(ArgumentError("Ap must be n x n or (n+1) x n"))               # rhessx.jl:62
    info = Ref{Cint}(0)
    Apt = _ptrs(A); Qp = Q === nothing ? Ptr{Float64}[] : _ptrs(Q)
    nq, nqc = Q === nothing ? (0, 0) : size(Q[1])
    f = T <: Real ? :psd_d_rphessenberg : :psd_z_rphessenberg
    GC.@preserve Ap A Q begin
        if T <: Real
            ccall((:psd_d_rphessenberg, libpsd), Cint,
                  (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Float64}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Cint, Cint, Ref{Cint}),
                  ctx().ptr, m, n, p, Ap, Apt, Q === nothing ? C_NULL : Qp, nq, nqc, info)
        else
            ccall((:psd_z_rphessenberg, libpsd), Cint,
                  (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Float64}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Cint, Cint, Ref{Cint}),
                  ctx().ptr, m, n, p, Ptr{Float64}(pointer(Ap)), Apt, Q === nothing ? C_NULL : Qp, nq, nqc, info)
        end
    end
    _throw(info[])
    return Ap, A, Q
end

# ---------------------------------------------------------------------------------------------------------------------
# ordschur!(P, select; wantZ, Z) — ordschur.jl:11-73 (ComplexF64), rordschur.jl:3-132 (Float64)
_userT(P) = begin                                          # full user-order list of the factors (T1 at schurindex)
    p = P.period; Ts = Vector{typeof(P.T1)}(undef, p); il = 0
    for l in 1:p; Ts[l] = (l == P.schurindex) ? P.T1 : P.T[il += 1]; end
    Ts
end
function LinearAlgebra.ordschur!(P::PeriodicSchur{T}, select::AbstractVector{Bool};
                                 wantZ::Bool = true, Z = nothing) where {T <: BlasElt}
    p = P.period; n = size(P.T1, 1); js = P.schurindex
    length(select) == n || throw(DimensionMismatch("select must have one entry per eigenvalue"))
    js in (1, p) || throw(ArgumentError("only implemented for schurindex in (1,p)"))               # ordschur.jl:32
    if Z !== nothing && P.orientation == 'R'
        throw(PSD.NotImplemented("no logic for reversing supplementary Z"))                        # ordschur.jl:36-38
    end
    Ts = _userT(P); Zs = Z === nothing ? P.Z : Z
    wantZ = wantZ && !isempty(Zs) && size(Zs[1], 1) == n
    info = Ref{Cint}(0); sel = UInt8.(select)
    Tp = _ptrs(Ts); Zp = wantZ ? _ptrs(Zs) : Ptr{Float64}[]
    if T <: Real
        wr = zeros(n); wi = zeros(n)
        GC.@preserve Ts Zs sel wr wi ccall((:psd_d_ordschur, libpsd), Cint,
            (Ptr{Cvoid}, Cint, Cint, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Cchar, Cint, Ptr{UInt8}, Cint,
             Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}, Ref{Cint}),
            ctx().ptr, n, p, Tp, wantZ ? Zp : C_NULL, P.orientation, js, sel, wantZ, wr, wi, C_NULL, info)
        _throw_ord(info[])
        P.values .= complex.(wr, wi)                                                              # rordschur.jl:124-130
    else
        α = zeros(ComplexF64, n); β = zeros(n); sc = zeros(Int32, n)
        GC.@preserve Ts Zs sel α β sc ccall((:psd_z_ordschur, libpsd), Cint,
            (Ptr{Cvoid}, Cint, Cint, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Cchar, Cint, Ptr{UInt8}, Cint,
             Ptr{ComplexF64}, Ptr{Float64}, Ptr{Int32}, Ptr{Cvoid}, Ref{Cint}),
            ctx().ptr, n, p, Tp, wantZ ? Zp : C_NULL, P.orientation, js, sel, wantZ, α, β, sc, C_NULL, info)
        _throw_ord(info[])
        P.values .= _values(α, β, sc)                                                             # ordschur.jl:118
    end
    return P
end

# ordschur!(P::GeneralizedPeriodicSchur, select; wantZ) — ordschur.jl:11-96,206-328 with the signed swaps of
# sylswap.jl:197-538 (2x2 blocks, Float64) and :638-764 (1x1)
function LinearAlgebra.ordschur!(P::GeneralizedPeriodicSchur{T}, select::AbstractVector{Bool};
                                 wantZ::Bool = true) where {T <: BlasElt}
    p = P.period; n = size(P.T1, 1); js = P.schurindex
    length(select) == n || throw(DimensionMismatch("select must have one entry per eigenvalue"))
    js in (1, p) || throw(ArgumentError("only implemented for schurindex in (1,p)"))
    Ts = _userT(P); Zs = P.Z
    wantZ = wantZ && !isempty(Zs) && size(Zs[1], 1) == n
    α = zeros(ComplexF64, n); β = zeros(n); sc = zeros(Int32, n); info = Ref{Cint}(0)
    sel = UInt8.(select); Sb = UInt8.(P.S)
    Tp = _ptrs(Ts); Zp = wantZ ? _ptrs(Zs) : Ptr{Float64}[]
    GC.@preserve Ts Zs sel Sb α β sc begin
        if T <: Real
            ccall((:psd_d_gordschur, libpsd), Cint,
                  (Ptr{Cvoid}, Cint, Cint, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ptr{UInt8}, Cchar, Cint, Ptr{UInt8}, Cint,
                   Ptr{ComplexF64}, Ptr{Float64}, Ptr{Int32}, Ptr{Cvoid}, Ref{Cint}),
                  ctx().ptr, n, p, Tp, wantZ ? Zp : C_NULL, Sb, P.orientation, js, sel, wantZ, α, β, sc, C_NULL, info)
        else
            ccall((:psd_z_gordschur, libpsd), Cint,
                  (Ptr{Cvoid}, Cint, Cint, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ptr{UInt8}, Cchar, Cint, Ptr{UInt8}, Cint,
                   Ptr{ComplexF64}, Ptr{Float64}, Ptr{Int32}, Ptr{Cvoid}, Ref{Cint}),
                  ctx().ptr, n, p, Tp, wantZ ? Zp : C_NULL, Sb, P.orientation, js, sel, wantZ, α, β, sc, C_NULL, info)
        end
    end
    _throw_ord(info[])
    P.α .= α; P.β .= (T <: Real ? β : complex.(β)); P.αscale .= Int.(sc)                          # ordschur.jl:75-96
    return P
end

# ---------------------------------------------------------------------------------------------------------------------
# checkpsd(P, As; quiet, thresh, strict) — diagnostics.jl:190-263, evaluated on the device (matrix cores)
function checkpsd(P::PSD.AbstractPeriodicSchur{T}, Hs::AbstractVector{<:AbstractMatrix{T}};
                  quiet = false, thresh = 100, strict = true) where {T <: BlasElt}
    p = length(Hs); n = size(P.T1, 1)
    P.period == p || throw(DimensionMismatch("length of Hs vector must match period of P"))        # diagnostics.jl:194
    for l in 1:p
        checksquare(Hs[l]) == n || throw(DimensionMismatch("size of Hs matrices must match P"))   # :197-202
    end
    Ts = _dense(_userT(P)); Zs = _dense(P.Z); As = _dense(Hs)
    S = P isa GeneralizedPeriodicSchur ? UInt8.(P.S) : fill(0x01, p)
    err = zeros(p); orth = zeros(p); tri = zeros(p); ok = Ref{Cint}(0); info = Ref{Cint}(0)
    Tp = _ptrs(Ts); Zp = _ptrs(Zs); Ap = _ptrs(As)
    GC.@preserve Ts Zs As S err orth tri begin
        if T <: Real
            wi = Float64.(imag.(P.values))
            GC.@preserve wi ccall((:psd_d_checkpsd, libpsd), Cint,
                  (Ptr{Cvoid}, Cint, Cint, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ptr{UInt8}, Cchar, Cint,
                   Ptr{Float64}, Cdouble, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Cint}, Ref{Cint}),
                  ctx().ptr, n, p, Tp, Zp, Ap, S, P.orientation, P.schurindex, wi, thresh, strict, err, orth, tri, ok, info)
        else
            ccall((:psd_z_checkpsd, libpsd), Cint,
                  (Ptr{Cvoid}, Cint, Cint, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ptr{UInt8}, Cchar, Cint,
                   Cdouble, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Cint}, Ref{Cint}),
                  ctx().ptr, n, p, Tp, Zp, Ap, S, P.orientation, P.schurindex, thresh, strict, err, orth, tri, ok, info)
        end
    end
    _throw(info[])
    if !quiet                                                                                     # diagnostics.jl:235-259
        cmp = strict ? 0.0 : 10 * eps(Float64) * n
        for l in 1:p
            tri[l] > cmp && @warn "triangularity fails for l=$l"
            orth[l] > 10 * eps(Float64) * n && @warn "orthogonality fails for l=$l"
            err[l] > thresh && @warn "large factorization error ($(err[l]) ϵ‖Aₗ‖₁) for l=$l"
        end
    end
    return ok[] != 0, err
end

end # module
