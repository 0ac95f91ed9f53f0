"""BP (MI355X) + OSD (host) end to end through the mirror of `BeliefPropagationOSDDecoder`,
following test/test_bposd_decoder.jl:1-69, and against the oracle chain
(oracle BP -> oracle OSD) on the same syndromes."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import BPOracle, osd_oracle_postprocess

pytestmark = pytest.mark.gpu


def _syn(H, err):
    return (H.astype(np.int64) @ err.astype(np.int64)) % 2


def test_bposd_decoder(ldpc, gpu):
    """test_bposd_decoder.jl:6-16."""
    rng = np.random.default_rng(1)
    H = ldpc.parity_check_matrix(1000, 10, 9)
    err = rng.random(1000) < 0.01
    bposd = ldpc.BeliefPropagationOSDDecoder(H, 0.01, 100)
    guess, success = bposd.decode_(_syn(H, err))
    assert guess.dtype == np.bool_ and np.array_equal(guess, err) and success is True


def test_bposd_decoder_high_order(ldpc, gpu):
    """test_bposd_decoder.jl:19-34: orders 2..5."""
    rng = np.random.default_rng(2)
    H = ldpc.parity_check_matrix(1000, 10, 9)
    err = rng.random(1000) < 0.01
    syn = _syn(H, err)
    for order in range(2, 6):
        bposd = ldpc.BeliefPropagationOSDDecoder(H, 0.01, 100, osd_order=order)
        guess, _ = bposd.decode_(syn)
        assert np.array_equal(guess, err)


def test_bposd_decoder_large_error_rate(ldpc, gpu):
    """test_bposd_decoder.jl:37-47: per = 0.2, the guess still satisfies the syndrome."""
    rng = np.random.default_rng(3)
    H = ldpc.parity_check_matrix(1000, 10, 9)
    err = rng.random(1000) < 0.2
    syn = _syn(H, err)
    bposd = ldpc.BeliefPropagationOSDDecoder(H, 0.2, 100)
    guess, success = bposd.decode_(syn)
    assert np.array_equal(_syn(H, guess), syn)


def test_bposd_decoder_batch(ldpc, gpu):
    """test_bposd_decoder.jl:50-63 (generic batchdecode!): every column syndrome-consistent."""
    rng = np.random.default_rng(4)
    H = ldpc.parity_check_matrix(1000, 10, 9)
    errors = rng.random((1000, 10)) < 0.01
    syndromes = _syn(H, errors)
    bposd = ldpc.BeliefPropagationOSDDecoder(H, 0.01, 100)
    guesses, successes = ldpc.batchdecode_(bposd, syndromes, np.zeros_like(errors))
    for i in range(10):
        assert np.array_equal(_syn(H, guesses[:, i]), syndromes[:, i])
    assert successes.dtype == np.bool_ and len(successes) == 10


@pytest.mark.parametrize("order", [0, 3])
def test_bposd_equals_oracle_chain_on_bb72(ldpc, gpu, order):
    """BASELINE configs[4] shape: BB [[72,12,6]] H_X, BP then OSD; whole chain vs the oracle chain.
    per is raised so that a visible share of syndromes needs OSD."""
    HX, _ = ldpc.codes.bivariate_bicycle_72_12_6()
    E = ldpc.codes.random_errors(72, 4000, 0.04, seed=order)
    syn = ldpc.codes.syndromes_of(HX, E)
    dec = ldpc.BeliefPropagationOSDDecoder(HX, 0.005, 50, osd_order=order)
    out = np.zeros((72, 4000), dtype=np.uint8)
    guesses, conv = dec.batchdecode_(syn.T, out)
    M = sp.csc_matrix(HX)
    oc = BPOracle(csc=(M.indptr, M.indices), shape=M.shape, per=0.005, max_iters=50)
    oerr, oconv, ollr, _ = oc.batchdecode(syn)
    assert np.array_equal(conv.astype(np.uint8), oconv)
    assert 0.01 < 1 - oconv.mean() < 0.9        # some, not all, needed OSD
    Hd = HX.astype(np.uint8)
    mism = 0
    for b in range(4000):
        ref = osd_oracle_postprocess(Hd, syn[b], oerr[b], ollr[b], order)
        mism += not np.array_equal(guesses[:, b], ref)
    # LLRs agree to ~1 ulp (device log vs libm); a reliability tie broken differently could in
    # principle reorder columns -- none is expected on this workload
    assert mism == 0
    assert np.array_equal(ldpc.codes.syndromes_of(HX, guesses.T), syn)


def test_device_resident_bposd_pipeline(ldpc, gpu):
    """BASELINE configs[4] pipeline: HBM-resident syndromes -> BP on the GPU -> only what still needs
    OSD goes to the host; must equal the plain host-buffer path column for column."""
    import torch

    HX, _ = ldpc.codes.bivariate_bicycle_72_12_6()
    E = ldpc.codes.random_errors(72, 20000, 0.03, seed=5)
    syn = ldpc.codes.syndromes_of(HX, E)
    for order in (0, 2):
        dec = ldpc.BeliefPropagationOSDDecoder(HX, 0.005, 50, osd_order=order)
        ref = np.zeros((72, 20000), dtype=np.uint8)
        ref, rconv = dec.batchdecode_(syn.T, ref)
        err, conv, sent = dec.batchdecode_device(torch.from_numpy(syn).to("cuda:0"))
        assert np.array_equal(err.cpu().numpy(), ref.T)
        assert np.array_equal(conv.cpu().numpy().astype(bool), rconv)
        assert sent == (20000 if order else int((~rconv).sum()))


def test_config5_full_size_bb72_batch_2_pow_20(ldpc, gpu):
    """BASELINE configs[4] at its stated size: BB [[72,12,6]] H_X, per 0.005, batch 2**20, BP on the GPU + OSD-0 on
    the host through the device-resident pipeline.  Size-independent properties on the WHOLE batch -- every output
    reproduces its syndrome (test/test_bposd_decoder.jl:37-47,59-61: OSD always does), `sent` is exactly the number
    of syndromes BP left unconverged -- and the oracle chain (BP oracle + dense OSD oracle) on a 20,000 subset."""
    import torch

    B = 1 << 20
    HX, _ = ldpc.codes.bivariate_bicycle_72_12_6()
    E = ldpc.codes.random_errors(72, B, 0.005, seed=2026)
    syn = ldpc.codes.syndromes_of(HX, E)
    d_syn = torch.from_numpy(syn).to("cuda:0")
    dec = ldpc.BeliefPropagationOSDDecoder(HX, 0.005, 50, osd_order=0)
    err, conv, sent = dec.batchdecode_device(d_syn)
    torch.cuda.synchronize()
    err = err.cpu().numpy()
    conv = conv.cpu().numpy()
    assert sent == int((conv == 0).sum()) and 0 < sent < B // 1000
    assert np.array_equal(ldpc.codes.syndromes_of(HX, err), syn), "an output does not reproduce its syndrome"
    # oracle chain on a subset that contains EVERY syndrome BP left unconverged plus random converged ones
    rng = np.random.default_rng(5)
    idx = np.unique(np.concatenate([np.nonzero(conv == 0)[0], rng.choice(B, 20000, replace=False)]))
    M = sp.csc_matrix(HX)
    oc = BPOracle(csc=(M.indptr, M.indices), shape=M.shape, per=0.005, max_iters=50)
    oerr, oconv, ollr, _ = oc.batchdecode(syn[idx])
    assert np.array_equal(conv[idx], oconv)
    Hd = HX.astype(np.uint8)
    for k, b in enumerate(idx):
        ref = oerr[k] if oconv[k] else osd_oracle_postprocess(Hd, syn[b], oerr[k], ollr[k], 0)
        assert np.array_equal(err[b], ref), f"syndrome {b} differs from the oracle chain"
