# replay.jl -- pins the fixtures in this directory against the REAL reference.
#
# NOT EXECUTED in this repository's pipeline (no Julia runtime in the build image or on the GPU box); written for
# a maintainer who has Julia >= 1.10 and QuantumSavory/LDPCDecoders.jl:
#
#     julia --project=/path/to/LDPCDecoders.jl tests/golden/replay.jl [tests/golden/raw]
#
# For every case under raw/ (written by export_raw.py from the .npz fixtures: raw little-endian arrays + a
# manifest.toml, standard library only) it builds `LDPCDecoders.BeliefPropagationDecoder(H, per, max_iters)` from the
# stored CSC pattern, runs `decode!` on every stored syndrome and compares with what the C oracle (and through it
# the HIP kernels, tests/test_golden.py) produced:
#     hard decisions  `guess`                 == errors[:, b]      bit for bit
#     flag            `converged`             == converged[b]
#     LLRs            `scratch.log_probabs`   ~  llr[:, b]         <= 1e-5, +-Inf and NaN positions exactly
# (iteration counts are not observable in the reference and are not compared).
# Exit status 0 = every case agrees: the oracle -- today "parity unpinned" -- is then pinned by the reference
# itself.  Set LDPC_REPLAY_IMPL=mi355x to replay through the ccall shim (ldpcdecoders.jl_amd/julia) instead,
# i.e. through libldpc_mi355x.so on a machine with an MI355X; LDPC_REPLAY_IMPL=dropin replays through the SAME-NAME
# drop-in (LDPCDecodersMI355XDropIn.jl: the reference's own decoder type with decode! / batchdecode! overwritten) --
# both equally unexecuted here.
using SparseArrays
using TOML
import LDPCDecoders

const IMPL = get(ENV, "LDPC_REPLAY_IMPL", "reference")
if IMPL == "mi355x"
    include(joinpath(@__DIR__, "..", "..", "ldpcdecoders.jl_amd", "julia", "LDPCDecodersMI355X.jl"))
elseif IMPL == "dropin"
    # from here on LDPCDecoders.decode!(::BeliefPropagationDecoder, ...) runs on the GPU: make_decoder below keeps
    # building the reference's own type
    include(joinpath(@__DIR__, "..", "..", "ldpcdecoders.jl_amd", "julia", "LDPCDecodersMI355XDropIn.jl"))
end

const ELTYPES = Dict("UInt8" => UInt8, "Int32" => Int32, "Int64" => Int64, "Float64" => Float64)

function load_array(dir, spec)
    T = ELTYPES[spec["eltype"]]
    a = Array{T}(undef, Int.(spec["dims"])...)
    open(joinpath(dir, spec["file"]), "r") do io
        read!(io, a)
    end
    return ltoh.(a)
end

function make_decoder(H, per, max_iters)
    IMPL == "mi355x" && return LDPCDecodersMI355X.MI355XBeliefPropagationDecoder(H, per, max_iters)
    return LDPCDecoders.BeliefPropagationDecoder(H, per, max_iters)
end

function replay_case(dir)
    m = TOML.parsefile(joinpath(dir, "manifest.toml"))
    s, n, per, max_iters, B = m["s"], m["n"], Float64(m["per"]), m["max_iters"], m["batch"]
    A = Dict(k => load_array(dir, v) for (k, v) in m["arrays"])
    colptr = Int.(A["colptr"]) .+ 1                      # the fixtures are zero-based
    rowval = Int.(A["rowval"]) .+ 1
    H = SparseMatrixCSC{Bool,Int}(s, n, colptr, rowval, fill(true, length(rowval)))
    syn = Int.(A["syndromes"])                           # s x B; entries 2 / 3 stand for "an Int that is not 0/1"
    decoder = make_decoder(H, per, max_iters)
    bad = 0
    worst = 0.0
    for b in 1:B
        guess, conv = LDPCDecoders.decode!(decoder, syn[:, b])
        lp = decoder.scratch.log_probabs
        want_e = A["errors"][:, b]
        want_l = A["llr"][:, b]
        ok = (conv == (A["converged"][b] != 0)) && all(guess .== want_e)
        for j in 1:n
            if isfinite(want_l[j]) && isfinite(lp[j])
                d = abs(lp[j] - want_l[j])
                worst = max(worst, d)
                ok &= d <= 1e-5
            else
                ok &= isequal(lp[j], want_l[j])          # Inf == Inf, -Inf == -Inf, NaN with NaN
            end
        end
        if !ok
            bad += 1
            bad <= 3 && println("  MISMATCH in ", m["case"], " syndrome ", b, ": converged ", conv, " vs ", A["converged"][b],
                                ", differing bits ", count(guess .!= want_e))
        end
    end
    println(rpad(m["case"], 28), " ", B - bad, "/", B, " syndromes agree, max |dLLR| ", worst)
    return bad
end

function main()
    root = length(ARGS) >= 1 ? ARGS[1] : joinpath(@__DIR__, "raw")
    cases = sort(filter(d -> isfile(joinpath(root, d, "manifest.toml")), readdir(root)))
    isempty(cases) && error("no cases under $root (run tests/golden/export_raw.py)")
    bad = sum(replay_case(joinpath(root, c)) for c in cases)
    println(bad == 0 ? "ALL CASES AGREE with $(IMPL)" : "$bad syndromes DISAGREE with $(IMPL)")
    exit(bad == 0 ? 0 : 1)
end

main()
