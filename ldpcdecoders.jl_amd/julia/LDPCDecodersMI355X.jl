# LDPCDecodersMI355X.jl -- Julia shim over libldpc_mi355x.so (include/ldpc_mi355x.h).
#
# NOT EXECUTED in this repository's pipeline: no Julia runtime exists in the build image
# or on the GPU box.  The same C ABI is exercised from Python ctypes
# (ldpcdecoders.jl_amd/_capi.py, decoder.py) and by the test-suite; this file is the
# binding a maintainer of QuantumSavory/LDPCDecoders.jl would add.  It mirrors
#
#   BeliefPropagationDecoder(H, per, max_iters)   src/decoders/belief_propagation.jl:61-67
#   reset!(decoder)                               :83-91
#   decode!(decoder, syndrome)                    :121-188
#   batchdecode!(decoder, syndromes, errors[, success])   :220-231, abstract_decoder.jl:44-48
#
# so that `MI355XBeliefPropagationDecoder <: LDPCDecoders.AbstractDecoder` is a drop-in
# wherever an AbstractDecoder is accepted (generic batchdecode!, QuantumClifford's extension).
# BP+OSD gets its own mirror type below (the reference's OSD decoder stores a concretely typed
# BeliefPropagationDecoder and reads `scratch.log_probabs`, belief_propagation_osd.jl:19,52).
module LDPCDecodersMI355X

using SparseArrays
import LDPCDecoders
import LDPCDecoders: AbstractDecoder, decode!, batchdecode!, reset!

export MI355XBeliefPropagationDecoder, MI355XBeliefPropagationOSDDecoder, MI355XBPOTSDecoder

const libldpc = get(ENV, "LDPC_MI355X_LIB", "libldpc_mi355x.so")

const LDPC_OK = Cint(0)

struct LDPCMI355XError <: Exception
    status::Cint
    msg::String
end

function check(status::Cint)
    status == LDPC_OK && return nothing
    msg = unsafe_string(ccall((:ldpc_last_error, libldpc), Cstring, ()))
    # shape errors are assertion failures in the reference (belief_propagation.jl:221-222)
    status == 1 && throw(ArgumentError(msg))
    throw(LDPCMI355XError(status, msg))
end

"Host mirrors of the two scratch fields other code reads (belief_propagation.jl:3-18)."
struct MI355XScratch
    log_probabs::Vector{Float64}
    channel_probs::Vector{Float64}
    err::Vector{Float64}
end

mutable struct MI355XBeliefPropagationDecoder <: AbstractDecoder
    per::Float64
    max_iters::Int
    s::Int
    n::Int
    sparse_H::SparseMatrixCSC{Bool,Int}
    sparse_HT::SparseMatrixCSC{Bool,Int}
    scratch::MI355XScratch
    handle::Ptr{Cvoid}       # ldpc_bp_decoder* (with `devices`: the root's, owned by `multi`)
    multi::Ptr{Cvoid}        # ldpc_bp_multi* when the decoder partitions its batches over several GPUs, else C_NULL
    # reusable staging (column-major Julia matrices already have the ABI's [B][s] image)
    syn_u8::Vector{UInt8}
    err_u8::Vector{UInt8}
    conv_u8::Vector{UInt8}
end

"""
    MI355XBeliefPropagationDecoder(H, per, max_iters; device=-1, devices=nothing, exchange=0, llr_exact=false)

`llr_exact = true`: `scratch.log_probabs` from the full posterior odds (`ldpc_bp_options.llr_exact`; the default cuts the odds
to their upper 32 bits: LLRs within 5e-7 of the reference's).  The BP+OSD type below asks for it.

`devices = 0:7` makes `batchdecode!` one call that partitions the columns of its `syndromes` matrix over those GPUs
(`ldpc_bp_create_multi`: contiguous shards, one handle and stream per GPU; the host arrays go pinned host -> each
shard's own GPU and back).  `exchange` only matters for device-resident batches (`ldpc_bp_decode_batch_multi_device`:
0 auto = RCCL send/recv from `devices[1]`, 1 hipMemcpyPeer, 2 RCCL).
"""
function MI355XBeliefPropagationDecoder(H, per::Float64, max_iters::Int; device::Integer=-1,
                                        devices::Union{Nothing,AbstractVector{<:Integer}}=nothing, exchange::Integer=0,
                                        llr_exact::Bool=false)
    s, n = size(H)
    sparse_H = SparseMatrixCSC{Bool,Int}(sparse(H))          # :63
    sparse_HT = SparseMatrixCSC{Bool,Int}(sparse(H'))        # :64
    colptr = Int64.(sparse_H.colptr .- 1)                    # zero-based for the ABI
    rowval = Int64.(rowvals(sparse_H) .- 1)
    opts = zeros(Int32, 16); opts[1] = Int32(device)         # ldpc_bp_options: device, waves_per_tile, resident_tiles,
    opts[6] = Int32(llr_exact)                               # kernel_variant, defer_threshold, llr_exact, reserved[10]
    h = Ref{Ptr{Cvoid}}(C_NULL)
    m = C_NULL
    if devices === nothing
        check(ccall((:ldpc_bp_create, libldpc), Cint,
                    (Int64, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Float64, Int64, Ptr{Int32}, Ptr{Ptr{Cvoid}}),
                    s, n, length(rowval), colptr, rowval, per, max_iters, opts, h))
    else
        devs = Int32.(collect(devices))
        mr = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:ldpc_bp_create_multi, libldpc), Cint,
                    (Int32, Ptr{Int32}, Int32, Int64, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Float64, Int64, Ptr{Int32}, Ptr{Ptr{Cvoid}}),
                    length(devs), devs, exchange, s, n, length(rowval), colptr, rowval, per, max_iters, opts, mr))
        m = mr[]
        h[] = ccall((:ldpc_bp_multi_handle, libldpc), Ptr{Cvoid}, (Ptr{Cvoid}, Int32), m, 0)
    end
    d = MI355XBeliefPropagationDecoder(per, max_iters, s, n, sparse_H, sparse_HT,
            MI355XScratch(zeros(n), fill(per, n), zeros(n)), h[], m, UInt8[], UInt8[], UInt8[])
    finalizer(d) do x
        if x.multi != C_NULL
            ccall((:ldpc_bp_destroy_multi, libldpc), Cint, (Ptr{Cvoid},), x.multi)   # (owns every per-GPU handle)
        elseif x.handle != C_NULL
            ccall((:ldpc_bp_destroy, libldpc), Cint, (Ptr{Cvoid},), x.handle)
        end
        x.handle = C_NULL; x.multi = C_NULL
    end
    return d
end

# the host-buffer entry: one GPU, or the batch partitioned over `devices` (same argument list, belief_propagation.jl:220-231)
host_decode(d::MI355XBeliefPropagationDecoder, B, llr) = d.multi != C_NULL ?
    ccall((:ldpc_bp_decode_batch_multi, libldpc), Cint,
          (Ptr{Cvoid}, Int64, Ptr{UInt8}, Ptr{UInt8}, Ptr{UInt8}, Ptr{Float64}, Ptr{Int32}),
          d.multi, B, d.syn_u8, d.err_u8, d.conv_u8, llr, C_NULL) :
    ccall((:ldpc_bp_decode_batch, libldpc), Cint,
          (Ptr{Cvoid}, Int64, Ptr{UInt8}, Ptr{UInt8}, Ptr{UInt8}, Ptr{Float64}, Ptr{Int32}),
          d.handle, B, d.syn_u8, d.err_u8, d.conv_u8, llr, C_NULL)

"""
    last_status(d)

Waits for everything enqueued on the handle and throws if a team of workgroups lost a member in one of
those calls (`ldpc_bp_last_status`, include/ldpc_mi355x.h).  Only callers of the asynchronous device entry
need it; `decode!` / `batchdecode!` above use the synchronous host entry, which repairs such a call itself.
"""
last_status(d::MI355XBeliefPropagationDecoder) = d.multi != C_NULL ?
    check(ccall((:ldpc_bp_multi_last_status, libldpc), Cint, (Ptr{Cvoid},), d.multi)) :
    check(ccall((:ldpc_bp_last_status, libldpc), Cint, (Ptr{Cvoid},), d.handle))

"`(-1)^x` only needs the parity; anything but 0/1 must never match the convergence `==` (:136,:181)."
@inline function syndrome_byte(x)::UInt8
    v = Int(x)                      # InexactError for non-integral floats, like (-1)^2.5 -> DomainError
    (v == 0 || v == 1) ? UInt8(v) : UInt8(2 + (v & 1))
end

function reset!(d::MI355XBeliefPropagationDecoder)            # :83-91 (device scratch is reset per call)
    d.scratch.log_probabs .= 0.0
    d.scratch.channel_probs .= d.per
    d.scratch.err .= 0.0
    d
end

function decode!(d::MI355XBeliefPropagationDecoder, syndrome::AbstractVector)   # :121-188
    length(syndrome) == d.s || throw(BoundsError(syndrome, d.s))
    reset!(d)
    resize!(d.syn_u8, d.s); resize!(d.err_u8, d.n); resize!(d.conv_u8, 1)
    @inbounds for i in 1:d.s
        d.syn_u8[i] = syndrome_byte(syndrome[i])
    end
    check(host_decode(d, 1, d.scratch.log_probabs))
    @inbounds for j in 1:d.n
        d.scratch.err[j] = d.err_u8[j]
    end
    return d.scratch.err, d.conv_u8[1] != 0                   # alias of the scratch, like :187
end

function batchdecode!(d::MI355XBeliefPropagationDecoder, syndromes::AbstractMatrix,
                      errors::AbstractMatrix, success::AbstractVector{Bool})   # :220-231
    @assert size(syndromes, 2) == size(errors, 2)             # :221
    @assert size(syndromes, 2) == length(success)             # :222
    B = size(syndromes, 2)
    size(syndromes, 1) == d.s || throw(DimensionMismatch("syndromes has $(size(syndromes,1)) rows, decoder has $(d.s) checks"))
    size(errors, 1) == d.n || throw(DimensionMismatch("errors has $(size(errors,1)) rows, decoder has $(d.n) bits"))
    B == 0 && return errors, success
    resize!(d.syn_u8, d.s * B); resize!(d.err_u8, d.n * B); resize!(d.conv_u8, B)
    # BitMatrix / Matrix{Int} / views are normalised to the ABI's byte image; column i of the
    # s x B matrix is the i-th contiguous run of s bytes
    @inbounds for i in 1:B, r in 1:d.s
        d.syn_u8[(i - 1) * d.s + r] = syndrome_byte(syndromes[r, i])
    end
    check(host_decode(d, B, Ptr{Float64}(C_NULL)))            # one call, one matrix -- on one GPU or partitioned over `devices`
    @inbounds for i in 1:B
        success[i] = d.conv_u8[i] != 0                        # :226
        for j in 1:d.n
            errors[j, i] = d.err_u8[(i - 1) * d.n + j]        # :227 (0/1 -> eltype(errors))
        end
    end
    # the reference's per-column loop (:224-228) leaves the scratch with the LAST column's decision and LLRs: the
    # decision is in hand, the LLRs come from a second call on that one column (deterministic per syndrome)
    @inbounds for j in 1:d.n
        d.scratch.err[j] = d.err_u8[(B - 1) * d.n + j]
    end
    GC.@preserve d begin
        last_err = Vector{UInt8}(undef, d.n); last_conv = Vector{UInt8}(undef, 1)
        syn_last = pointer(d.syn_u8, (B - 1) * d.s + 1)
        check(d.multi != C_NULL ?
              ccall((:ldpc_bp_decode_batch_multi, libldpc), Cint,
                    (Ptr{Cvoid}, Int64, Ptr{UInt8}, Ptr{UInt8}, Ptr{UInt8}, Ptr{Float64}, Ptr{Int32}),
                    d.multi, 1, syn_last, last_err, last_conv, d.scratch.log_probabs, C_NULL) :
              ccall((:ldpc_bp_decode_batch, libldpc), Cint,
                    (Ptr{Cvoid}, Int64, Ptr{UInt8}, Ptr{UInt8}, Ptr{UInt8}, Ptr{Float64}, Ptr{Int32}),
                    d.handle, 1, syn_last, last_err, last_conv, d.scratch.log_probabs, C_NULL))
    end
    return errors, success                                    # :230
end

# 3-argument form: the generic method at abstract_decoder.jl:44-48 allocates `success`
# and re-dispatches to the 4-argument method above; nothing to add.

# ---------------------------------------------------------------------------------------------
# BP+OSD (src/decoders/belief_propagation_osd.jl).  The reference's BeliefPropagationOSDDecoder
# holds a concretely typed `bp_decoder::BeliefPropagationDecoder` (:19), so the MI355X decoder
# cannot be slotted into it; this type mirrors it: BP on the GPU, the ordered-statistics step in
# the library's host code (`ldpc_osd_postprocess_batch`, bit-packed, threaded over the batch).
# ---------------------------------------------------------------------------------------------
mutable struct MI355XBeliefPropagationOSDDecoder <: AbstractDecoder
    bp_decoder::MI355XBeliefPropagationDecoder
    H::BitMatrix
    osd_order::Int
    osd_handle::Ptr{Cvoid}
end

function MI355XBeliefPropagationOSDDecoder(H::BitMatrix, per::Float64, max_iters::Int;
                                           osd_order::Int=0, device::Integer=-1)     # :26-29
    bp = MI355XBeliefPropagationDecoder(H, per, max_iters; device=device, llr_exact=true)   # (OSD orders bits by reliability, :53-55)
    colptr = Int64.(bp.sparse_H.colptr .- 1); rowval = Int64.(rowvals(bp.sparse_H) .- 1)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:ldpc_osd_create, libldpc), Cint,
                (Int64, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Int64, Ptr{Ptr{Cvoid}}),
                bp.s, bp.n, length(rowval), colptr, rowval, osd_order, h))
    d = MI355XBeliefPropagationOSDDecoder(bp, H, osd_order, h[])
    finalizer(d) do x
        x.osd_handle != C_NULL && ccall((:ldpc_osd_destroy, libldpc), Cint, (Ptr{Cvoid},), x.osd_handle)
        x.osd_handle = C_NULL
    end
    return d
end

function decode!(d::MI355XBeliefPropagationOSDDecoder, syndrome::AbstractVector)      # :49-61
    bp = d.bp_decoder
    bp_err, converged = decode!(bp, syndrome)              # fills bp.syn_u8, bp.err_u8, scratch.log_probabs
    out = Vector{UInt8}(undef, bp.n)
    check(ccall((:ldpc_osd_postprocess_batch, libldpc), Cint,
                (Ptr{Cvoid}, Int64, Ptr{UInt8}, Ptr{UInt8}, Ptr{Float64}, Ptr{UInt8}, Int32),
                d.osd_handle, 1, bp.syn_u8, bp.err_u8, bp.scratch.log_probabs, out, 1))
    return Bool.(out), converged                           # a Bool vector, like :60
end

function batchdecode!(d::MI355XBeliefPropagationOSDDecoder, syndromes::AbstractMatrix,
                      errors::AbstractMatrix, success::AbstractVector{Bool})
    # one BP launch + one threaded OSD pass give the same columns as the reference's generic
    # per-column loop (abstract_decoder.jl:31-42, test_bposd_decoder.jl:49-57)
    @assert size(syndromes, 2) == size(errors, 2)
    @assert size(syndromes, 2) == length(success)
    bp = d.bp_decoder
    B = size(syndromes, 2)
    B == 0 && return errors, success
    resize!(bp.syn_u8, bp.s * B); resize!(bp.err_u8, bp.n * B); resize!(bp.conv_u8, B)
    @inbounds for i in 1:B, r in 1:bp.s
        bp.syn_u8[(i - 1) * bp.s + r] = syndrome_byte(syndromes[r, i])
    end
    llr = Vector{Float64}(undef, bp.n * B)
    check(host_decode(bp, B, llr))
    out = Vector{UInt8}(undef, bp.n * B)
    check(ccall((:ldpc_osd_postprocess_batch, libldpc), Cint,
                (Ptr{Cvoid}, Int64, Ptr{UInt8}, Ptr{UInt8}, Ptr{Float64}, Ptr{UInt8}, Int32),
                d.osd_handle, B, bp.syn_u8, bp.err_u8, llr, out, 0))
    @inbounds for i in 1:B
        success[i] = bp.conv_u8[i] != 0
        for j in 1:bp.n
            errors[j, i] = out[(i - 1) * bp.n + j]
        end
    end
    return errors, success
end

# ---------------------------------------------------------------------------------------------
# BP-OTS (src/decoders/bpots_decoder.jl:39-115, 225-340) over ldpc_bpots_* (LDS-resident kernel for small graphs,
# node-parallel kernel with the messages in global memory up to n ~ 30,000; beyond that LDPCMI355XError(5, ...)).
# ---------------------------------------------------------------------------------------------
mutable struct MI355XBPOTSDecoder <: AbstractDecoder
    per::Float64; max_iters::Int; s::Int; n::Int; T::Int; C::Float64
    handle::Ptr{Cvoid}
end

function MI355XBPOTSDecoder(H::Union{SparseMatrixCSC{Bool,Int},BitMatrix}, per::Float64, max_iters::Int;
                            T::Int=9, C::Float64=2.0, device::Integer=-1)             # :90
    s, n = size(H)
    sp = SparseMatrixCSC{Bool,Int}(sparse(H))
    colptr = Int64.(sp.colptr .- 1); rowval = Int64.(rowvals(sp) .- 1)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:ldpc_bpots_create, libldpc), Cint,
                (Int64, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Float64, Int64, Int64, Float64, Int32, Ptr{Ptr{Cvoid}}),
                s, n, length(rowval), colptr, rowval, per, max_iters, T, C, device, h))
    d = MI355XBPOTSDecoder(per, max_iters, s, n, T, C, h[])
    finalizer(d) do x
        x.handle != C_NULL && ccall((:ldpc_bpots_destroy, libldpc), Cint, (Ptr{Cvoid},), x.handle)
        x.handle = C_NULL
    end
    return d
end

reset!(d::MI355XBPOTSDecoder) = d       # :142-154: the device state is reset inside every decode call

function decode!(d::MI355XBPOTSDecoder, syndrome::AbstractVector)                      # :225-340
    length(syndrome) == d.s || throw(BoundsError(syndrome, d.s))
    syn = UInt8[syndrome_byte(x) for x in syndrome]
    err = Vector{UInt8}(undef, d.n); conv = Vector{UInt8}(undef, 1)
    check(ccall((:ldpc_bpots_decode_batch, libldpc), Cint,
                (Ptr{Cvoid}, Int64, Ptr{UInt8}, Ptr{UInt8}, Ptr{UInt8}, Ptr{Int32}),
                d.handle, 1, syn, err, conv, C_NULL))
    return Int.(err), conv[1] != 0        # best_decisions::Vector{Int}, converged
end
# batchdecode! on it is the reference's generic per-column method (abstract_decoder.jl:31-48);
# a batched override would call ldpc_bpots_decode_batch with B columns exactly like the BP type above.

end # module
