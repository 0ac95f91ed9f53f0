# LDPCDecodersMI355XDropIn.jl -- the SAME-NAME drop-in: after
#
#     using LDPCDecoders
#     include(".../ldpcdecoders.jl_amd/julia/LDPCDecodersMI355XDropIn.jl")
#
# every `LDPCDecoders.BeliefPropagationDecoder` -- the reference's own type, constructed by the reference's own
# constructor (src/decoders/belief_propagation.jl:61-67) -- decodes on the MI355X: this file OVERWRITES the two
# methods that are the hot path,
#
#     decode!(::BeliefPropagationDecoder, syndrome)                          belief_propagation.jl:121-188
#     batchdecode!(::BeliefPropagationDecoder, syndromes, errors, success)   belief_propagation.jl:220-231
#
# and nothing else.  Because the TYPE is untouched, everything that holds one keeps working without an edit:
# `BeliefPropagationOSDDecoder` (its field is concretely typed, belief_propagation_osd.jl:19, and it reads
# `bp_decoder.scratch.log_probabs`, :52 -- filled here), the generic 3-argument `batchdecode!`
# (abstract_decoder.jl:44-48), QuantumClifford's extension.  `reset!` stays the reference's (:83-91) for whoever calls
# it, but the methods below do NOT: it fills the two dense s x n Float64 matrices of the scratch (2 GiB at n = 16384),
# which nobody reads here -- the messages live in HBM, and the device state is reset inside every call; the library
# overwrites `scratch.log_probabs` and this file `scratch.err` completely on every decode! and, with the last column's
# values, on every batchdecode! (as the reference's per-column loop leaves them, :224-228).  What stays of the
# reference's cost is its CONSTRUCTOR, which still allocates those 2 * s * n * 8 bytes on the host once
# (belief_propagation.jl:20-22); LDPCDecodersMI355X.jl next to this file has types of its own and avoids that too.
#
#     LDPCDecodersMI355XDropIn.set_devices!(0:7)      # optional: batchdecode! partitions its columns over these GPUs
#                                                     # (ldpc_bp_create_multi; decoders created afterwards)
#
# NOT EXECUTED in this repository's pipeline (no Julia runtime in the image); LDPCDecodersMI355X.jl next to it is the
# conservative alternative with types of its own.  Overwriting another module's methods is deliberate here; Julia
# forbids it during precompilation, hence `include` at run time rather than a package (or `__precompile__(false)`).
module LDPCDecodersMI355XDropIn

using SparseArrays
import LDPCDecoders
import LDPCDecoders: BeliefPropagationDecoder

const libldpc = get(ENV, "LDPC_MI355X_LIB", "libldpc_mi355x.so")

function check(status::Cint)
    status == 0 && return nothing
    msg = unsafe_string(ccall((:ldpc_last_error, libldpc), Cstring, ()))
    status == 1 && throw(ArgumentError(msg))
    error("libldpc_mi355x: status $status: $msg")
end

mutable struct Handle
    ptr::Ptr{Cvoid}          # ldpc_bp_decoder* (one GPU) or ldpc_bp_multi* (several)
    multi::Bool
    syn_u8::Vector{UInt8}
    err_u8::Vector{UInt8}
    conv_u8::Vector{UInt8}
end

# GPUs that decoders created from now on partition their batches over (empty: the current device)
const DEVICES = Int32[]
set_devices!(devs) = (empty!(DEVICES); append!(DEVICES, Int32.(collect(devs))); DEVICES)

# One library handle per reference decoder object, created at its first decode.  The reference's decoder is an
# IMMUTABLE struct (belief_propagation.jl:38): it has no identity of its own and cannot carry a finalizer (a
# WeakKeyDict keyed on it throws on first use), so the table is keyed on a mutable object every decoder owns exactly
# one of -- its `scratch.log_probabs` vector -- and the handle lives as long as that vector does.
const HANDLES = WeakKeyDict{Vector{Float64},Handle}()
const HANDLES_LOCK = ReentrantLock()

function handle_of(d::BeliefPropagationDecoder)
    lock(HANDLES_LOCK) do
        get!(HANDLES, d.scratch.log_probabs) do
            colptr = Int64.(d.sparse_H.colptr .- 1)          # zero-based CSC pattern of sparse(H) (:63)
            rowval = Int64.(rowvals(d.sparse_H) .- 1)
            h = Ref{Ptr{Cvoid}}(C_NULL)
            multi = !isempty(DEVICES)
            # ldpc_bp_options with llr_exact = 1: a decoder of the reference's type may sit inside the reference's own
            # BeliefPropagationOSDDecoder, which orders the bits by scratch.log_probabs (belief_propagation_osd.jl:52-55)
            opts = zeros(Int32, 16); opts[1] = Int32(-1); opts[6] = Int32(1)
            if multi
                check(ccall((:ldpc_bp_create_multi, libldpc), Cint,
                            (Int32, Ptr{Int32}, Int32, Int64, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Float64, Int64, Ptr{Int32}, Ptr{Ptr{Cvoid}}),
                            length(DEVICES), DEVICES, 0, d.s, d.n, length(rowval), colptr, rowval, d.per, d.max_iters, opts, h))
            else
                check(ccall((:ldpc_bp_create, libldpc), Cint,
                            (Int64, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Float64, Int64, Ptr{Int32}, Ptr{Ptr{Cvoid}}),
                            d.s, d.n, length(rowval), colptr, rowval, d.per, d.max_iters, opts, h))
            end
            hd = Handle(h[], multi, UInt8[], UInt8[], UInt8[])
            finalizer(hd) do x
                if x.ptr != C_NULL
                    x.multi ? ccall((:ldpc_bp_destroy_multi, libldpc), Cint, (Ptr{Cvoid},), x.ptr) :
                              ccall((:ldpc_bp_destroy, libldpc), Cint, (Ptr{Cvoid},), x.ptr)
                end
                x.ptr = C_NULL
            end
            hd
        end
    end
end

# the host-buffer entry of a handle: one GPU or the batch partitioned over several (same argument list)
decode_batch(h::Handle, B, llr) = h.multi ?
    ccall((:ldpc_bp_decode_batch_multi, libldpc), Cint,
          (Ptr{Cvoid}, Int64, Ptr{UInt8}, Ptr{UInt8}, Ptr{UInt8}, Ptr{Float64}, Ptr{Int32}),
          h.ptr, B, h.syn_u8, h.err_u8, h.conv_u8, llr, C_NULL) :
    ccall((:ldpc_bp_decode_batch, libldpc), Cint,
          (Ptr{Cvoid}, Int64, Ptr{UInt8}, Ptr{UInt8}, Ptr{UInt8}, Ptr{Float64}, Ptr{Int32}),
          h.ptr, B, h.syn_u8, h.err_u8, h.conv_u8, llr, C_NULL)

@inline function syndrome_byte(x)::UInt8                      # (-1)^x needs the parity only (:136); 0/1 alone can match (:181)
    v = Int(x)
    (v == 0 || v == 1) ? UInt8(v) : UInt8(2 + (v & 1))
end

function LDPCDecoders.decode!(d::BeliefPropagationDecoder, syndrome::AbstractVector)       # overwrites :121-188
    length(syndrome) == d.s || throw(BoundsError(syndrome, d.s))
    # (:122 calls reset! here; its dense fills are not needed -- see the header -- and both result vectors are
    # overwritten completely below)
    h = handle_of(d)
    resize!(h.syn_u8, d.s); resize!(h.err_u8, d.n); resize!(h.conv_u8, 1)
    @inbounds for i in 1:d.s
        h.syn_u8[i] = syndrome_byte(syndrome[i])
    end
    check(decode_batch(h, 1, d.scratch.log_probabs))
    @inbounds for j in 1:d.n
        d.scratch.err[j] = h.err_u8[j]
    end
    return d.scratch.err, h.conv_u8[1] != 0                                                # the alias of :187
end

function LDPCDecoders.batchdecode!(d::BeliefPropagationDecoder, syndromes::AbstractMatrix,
                                   errors::AbstractMatrix, success::AbstractVector{Bool})   # overwrites :220-231
    @assert size(syndromes, 2) == size(errors, 2)                                          # :221
    @assert size(syndromes, 2) == length(success)                                          # :222
    B = size(syndromes, 2)
    B == 0 && return errors, success
    h = handle_of(d)
    resize!(h.syn_u8, d.s * B); resize!(h.err_u8, d.n * B); resize!(h.conv_u8, B)
    @inbounds for i in 1:B, r in 1:d.s
        h.syn_u8[(i - 1) * d.s + r] = syndrome_byte(syndromes[r, i])
    end
    check(decode_batch(h, B, Ptr{Float64}(C_NULL)))
    @inbounds for i in 1:B
        success[i] = h.conv_u8[i] != 0                                                     # :226
        for j in 1:d.n
            errors[j, i] = h.err_u8[(i - 1) * d.n + j]                                     # :227
        end
    end
    # The reference leaves the scratch holding the LAST column's state (:224-228 runs decode! per column, and decode!
    # fills scratch.err and scratch.log_probabs, :163-168): `scratch.err` gets that column's decision, and
    # `scratch.log_probabs` its LLRs -- asked for from the library for that one column alone (a second call with
    # batch = 1 on the bytes already marshalled: the decoder is deterministic per syndrome, so these are the LLRs of the
    # decode above; shipping 8 n bytes back for EVERY column would double the batch call's I/O for nothing)
    @inbounds for j in 1:d.n
        d.scratch.err[j] = h.err_u8[(B - 1) * d.n + j]
    end
    GC.@preserve h begin
        last_err = Vector{UInt8}(undef, d.n); last_conv = Vector{UInt8}(undef, 1)
        syn_last = pointer(h.syn_u8, (B - 1) * d.s + 1)
        check(h.multi ?
              ccall((:ldpc_bp_decode_batch_multi, libldpc), Cint,
                    (Ptr{Cvoid}, Int64, Ptr{UInt8}, Ptr{UInt8}, Ptr{UInt8}, Ptr{Float64}, Ptr{Int32}),
                    h.ptr, 1, syn_last, last_err, last_conv, d.scratch.log_probabs, C_NULL) :
              ccall((:ldpc_bp_decode_batch, libldpc), Cint,
                    (Ptr{Cvoid}, Int64, Ptr{UInt8}, Ptr{UInt8}, Ptr{UInt8}, Ptr{Float64}, Ptr{Int32}),
                    h.ptr, 1, syn_last, last_err, last_conv, d.scratch.log_probabs, C_NULL))
    end
    return errors, success                                                                 # :230
end

end # module
