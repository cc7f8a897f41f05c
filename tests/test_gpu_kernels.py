"""-m gpu: every HIP kernel against the CPU oracle, called through the C ABI (ctypes).

Tolerances (SURVEY.md 8c / BASELINE.json north_star): integer outputs (TDOA indexes, arg-max
masks away from exact ties) bit-exact; f32 GEMM / FFT results within f32 round-off of the
float64 oracle; W,H rel-Frobenius <= 1e-4; waveforms RMS <= 1e-4.
"""
import ctypes

import numpy as np
import pytest

from conftest import golden
from oracle import gccnmf_oracle as O

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')
FUSED_K12_DEFAULT = 1       # library defaults of tuning keys 16 and 17 (1 = when the cost model says so; 2 = whenever the shape allows)
FUSED_K34_DEFAULT = 1       # (1 = when the launch's whole rounds of 512 workgroups pay; 2 = whenever the shape allows)


@pytest.fixture(scope='module')
def hip():
    from gcc_nmf_amd import _hip
    assert torch.cuda.is_available(), 'the gpu tests need a ROCm device'
    return _hip


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def stream():
    return torch.cuda.current_stream().cuda_stream


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300)


# ------------------------------------------------------------------------------------------------
# the MFMA GEMM template, every operand layout / wave grid / tail / batch mapping
# ------------------------------------------------------------------------------------------------
GEMM_CASES = [
    # (layout bits, M, N, Kd, batch)   bits: 1 A_KC, 2 B_KC, 4 tail row, 8 wide <1,4> grid
    (0, 300, 200, 100, 1), (0 | 8, 100, 300, 37, 2), (0, 1024, 130, 513, 9),
    (1, 200, 150, 64, 1), (1 | 4, 513, 200, 100, 3), (1 | 4, 513, 1244, 128, 9), (1 | 8, 128, 300, 48, 2),
    (1 | 4 | 8, 129, 260, 40, 1),
    (3, 200, 64, 150, 1), (3 | 4, 513, 128, 300, 2), (3 | 4, 513, 200, 1244, 8), (3 | 8, 100, 256, 75, 1),
    (3 | 4 | 8, 129, 300, 50, 2),
    (16, 1024, 200, 513, 2), (16 | 8, 128, 300, 33, 9), (16, 300, 100, 17, 1),      # rank-1 reduction tail in the epilogue
    (32, 1024, 1244, 512, 9), (32, 700, 130, 75, 2), (32, 100, 64, 16, 1),          # the LDS-free throughput tile (direct.hip, experiment)
]


@pytest.mark.parametrize('layout,M,N,Kd,batch', GEMM_CASES)
def test_debug_gemm(hip, layout, M, N, Kd, batch):
    lib = hip.lib()
    rng = np.random.RandomState(layout * 1000 + M + N + Kd)
    a_kc, b_kc = bool(layout & 1), bool(layout & 2)
    Kd16 = -(-Kd // 16) * 16
    A = rng.standard_normal((batch, M, Kd)).astype(np.float32)          # asymmetric, signed
    B = (rng.standard_normal((batch, Kd, N)) + 0.5).astype(np.float32)
    # padded device images (reduction padding must be zero)
    if a_kc:
        lda, a_rows = Kd16, M
        Ad = np.zeros((batch, a_rows, lda), np.float32)
        Ad[:, :, :Kd] = A
        a_clamp = a_rows - 1
    else:
        lda, a_rows = -(-M // 4) * 4, Kd16
        Ad = np.zeros((batch, a_rows, lda), np.float32)
        Ad[:, :Kd, :M] = A.transpose(0, 2, 1)
        a_clamp = lda - 4
    if b_kc:
        ldb, b_rows = Kd16, N
        Bd = np.zeros((batch, b_rows, ldb), np.float32)
        Bd[:, :, :Kd] = B.transpose(0, 2, 1)
        b_clamp = b_rows - 1
    else:
        ldb, b_rows = -(-N // 4) * 4, Kd16
        Bd = np.zeros((batch, b_rows, ldb), np.float32)
        Bd[:, :Kd, :N] = B
        b_clamp = ldb - 4
    ldc = N + 3
    dA, dB = dev(Ad), dev(Bd)
    dC = torch.full((batch, M, ldc), -7.0, dtype=torch.float32, device='cuda')
    bscale = None
    dscale = None
    if not b_kc and not (layout & 16) and not (layout & 32):
        bscale = (rng.rand(Kd16) + 0.5).astype(np.float32)
        dscale = dev(bscale)
    drow = torch.zeros((batch, N), dtype=torch.float32, device='cuda') if b_kc else None
    rc = lib.gccnmf_debug_gemm(dA.data_ptr(), dB.data_ptr(), dC.data_ptr(), M, N, Kd, lda, ldb, ldc, a_clamp, b_clamp, layout,
                               batch, Ad[0].size, Bd[0].size, M * ldc, 0 if dscale is None else dscale.data_ptr(),
                               0 if drow is None else drow.data_ptr(), stream())
    if (layout & 32) and rc == 3:
        pytest.skip('the LDS-free stream tile (layout bit 32) exists in experiment builds only: make EXPERIMENTS=1')
    assert rc == 0
    torch.cuda.synchronize()
    C = dC.cpu().numpy()
    Bs = B.astype(np.float64)
    if bscale is not None:
        Bs = Bs * bscale[None, :Kd, None]
    ref = np.einsum('bmk,bkn->bmn', A.astype(np.float64), Bs)
    scale = np.abs(A).astype(np.float64).max() * np.abs(Bs).max() * Kd
    err = np.abs(C[:, :, :N] - ref).max()
    assert err < 2e-6 * scale, (err, scale)
    assert np.all(C[:, :, N:] == -7.0), 'wrote outside the valid columns'
    if drow is not None:
        rs = drow.cpu().numpy()
        assert np.abs(rs - B.sum(axis=1)).max() < 1e-4 * Kd


# ------------------------------------------------------------------------------------------------
# the latency-path GEMM (csrc/direct.hip): every epilogue x every tile, ragged shapes, batches, tail row, lazy scale
# ------------------------------------------------------------------------------------------------
DIRECT_CASES = [
    # (epilogue, tile, M, N, Kd, batch, tail, scale)      tile 0 = the library's choice, 1..8 = fixed
    (0, 0, 100, 70, 50, 1, False, False), (0, 5, 513, 1024, 1244, 1, True, False), (0, 1, 33, 20, 200, 2, True, False),
    (0, 2, 48, 40, 37, 3, False, False), (0, 4, 70, 33, 64, 1, True, False), (0, 7, 130, 100, 96, 2, False, False),
    (1, 0, 513, 1244, 128, 1, True, True), (1, 6, 513, 300, 1024, 1, True, True), (1, 3, 40, 90, 33, 2, False, True),
    (1, 8, 200, 170, 80, 1, False, False), (1, 6, 65, 81, 16, 3, True, False),
    (2, 0, 513, 1244, 64, 1, True, False), (2, 6, 513, 200, 100, 2, True, False), (2, 8, 100, 163, 48, 1, False, False),
    (2, 3, 30, 50, 20, 1, False, False),
    (3, 0, 1024, 1244, 513, 1, True, False), (3, 8, 128, 300, 513, 2, True, False), (3, 3, 64, 100, 40, 1, False, False),
    (3, 6, 100, 77, 129, 2, True, False), (3, 7, 70, 64, 32, 1, False, False),
]


@pytest.mark.parametrize('epi,tile,M,N,Kd,batch,tail,scale', DIRECT_CASES)
def test_gemm_direct(hip, epi, tile, M, N, Kd, batch, tail, scale):
    """C = A^T . B with both operands reduction-major, against float64; for UPDH `tail` = the rank-1 epilogue term."""
    lib = hip.lib()
    rng = np.random.RandomState(epi * 7919 + tile * 131 + M + N + Kd)
    Mm = M - 1 if (tail and epi != 3) else M                  # rows on the matrix cores; row M-1 is the VALU tail row
    Kdm = Kd - 1 if (tail and epi == 3) else Kd               # UPDH: the last reduction index is the rank-1 term
    Kd16 = -(-Kdm // 16) * 16
    lda, ldb = -(-M // 4) * 4 + 4, -(-N // 4) * 4 + 8
    A = (rng.rand(batch, Kd, M) + 0.25).astype(np.float32)
    B = (rng.rand(batch, Kd, N) + 0.25).astype(np.float32)
    Ad = np.zeros((batch, Kd16 + 16, lda), np.float32)
    Bd = np.zeros((batch, Kd16 + 16, ldb), np.float32)
    Ad[:, :Kdm, :Mm] = A[:, :Kdm, :Mm]
    Bd[:, :Kdm, :N] = B[:, :Kdm]
    dA, dB = dev(Ad), dev(Bd)
    ldc, ldct = ldb, lda
    C0 = (rng.rand(batch, M, N) + 0.5).astype(np.float32)          # UPDH: the old H
    Cd = np.full((batch, M, ldc), -7.0, np.float32)
    if epi == 3:
        Cd[:, :, :N] = C0
    dC = dev(Cd)
    dCt = torch.full((batch, N, ldct), -7.0, dtype=torch.float32, device='cuda')
    E0 = (rng.rand(batch, M, N) + 0.5).astype(np.float32)
    E0d = np.zeros((batch, M, ldc), np.float32)
    E0d[:, :, :N] = E0
    dE0 = dev(E0d)
    bs = (rng.rand(batch, Kd16) + 0.5).astype(np.float32)
    dbs = dev(bs)
    tailA = np.zeros((batch, Kd16), np.float32)
    tailA[:, :Kdm] = A[:, :Kdm, M - 1]
    dtail = dev(tailA)
    drow = torch.full((batch, N), -7.0, dtype=torch.float32, device='cuda')
    E1 = (rng.rand(batch, M) + 0.5).astype(np.float32)
    E2 = (rng.rand(batch, M) + 0.5).astype(np.float32)
    dE1, dE2 = dev(E1), dev(E2)
    kA = np.ascontiguousarray(A[:, Kd - 1, :])                      # [batch][M]
    kB = np.zeros((batch, ldb), np.float32)
    kB[:, :N] = B[:, Kd - 1, :]
    dkA, dkB = dev(kA), dev(kB)
    d = hip.DirectGemm()
    d.A, d.B, d.sA, d.sB, d.lda, d.ldb = dA.data_ptr(), dB.data_ptr(), Ad[0].size, Bd[0].size, lda, ldb
    d.M, d.N, d.Kd, d.batch = Mm, N, Kdm, batch
    if scale:
        d.bscale, d.s_bscale = dbs.data_ptr(), Kd16
    if tail and epi != 3:
        d.tailA, d.s_tailA, d.tail_row = dtail.data_ptr(), Kd16, M - 1
    if epi == 0:
        d.rowsumB, d.s_rowsumB = drow.data_ptr(), N
    d.C, d.sC, d.ldc = dC.data_ptr(), M * ldc, ldc
    d.Ct, d.sCt, d.ldct = dCt.data_ptr(), N * ldct, ldct
    d.E0, d.sE0, d.lde0 = dE0.data_ptr(), M * ldc, ldc
    d.E1, d.sE1, d.E2, d.sE2 = dE1.data_ptr(), M, dE2.data_ptr(), M
    if tail and epi == 3:
        d.ktailA, d.ktailB, d.s_ktailA, d.s_ktailB = dkA.data_ptr(), dkB.data_ptr(), M, ldb
    d.alpha, d.eps = 0.25, 1e-16
    assert lib.gccnmf_gemm_direct(ctypes.byref(d), epi, tile, stream()) == 0
    torch.cuda.synchronize()
    C, Ct = dC.cpu().numpy(), dCt.cpu().numpy()
    Bs = B.astype(np.float64) * (bs[:, :Kd, None] if scale else 1.0)
    acc = np.einsum('bkm,bkn->bmn', A.astype(np.float64), Bs)              # all M rows, all Kd reduction indexes
    if epi == 0:
        ref = acc
    elif epi in (1, 2):
        ref = E0 / acc
    else:
        ref = (C0.astype(np.float64) * E1[:, :, None]) * (acc / (E2[:, :, None].astype(np.float64) + 0.25 + 1e-16))
    rows = M if (tail or epi == 3) else Mm
    tol = 3e-6 * np.abs(ref).max() * max(1.0, np.sqrt(Kd / 64.0))
    if epi != 2:
        assert np.abs(C[:, :rows, :N] - ref[:, :rows]).max() < tol
        assert np.all(C[:, :, N:] == -7.0) or np.all((C[:, :, N:] == -7.0) | (C[:, :, N:] == 0.0)), 'wrote garbage outside the valid columns'
    if epi == 2:
        assert np.abs(Ct[:, :, :M].transpose(0, 2, 1)[:, :rows] - ref[:, :rows]).max() < tol
        if tail:
            assert np.abs(C[:, M - 1, :N] - ref[:, M - 1]).max() < tol          # the tail row also lands in the row-major buffer
    if epi == 3:
        assert np.abs(Ct[:, :, :M].transpose(0, 2, 1) - ref).max() < tol
    if epi in (2, 3):
        pad = Ct[:, :, -(-M // 4) * 4:]
        assert np.all(pad == -7.0), 'transposed store wrote beyond the 4-float group that holds row M-1'
    if epi == 0:
        assert np.abs(drow.cpu().numpy() - Bs.sum(axis=1)).max() < 1e-5 * Kd


# ------------------------------------------------------------------------------------------------
# STFT / iSTFT
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('n_fft,hop,n', [(1024, 256, 20000), (1024, 128, 6000), (512, 64, 5000), (256, 100, 3000), (2048, 512, 9000),
                                         (4096, 1024, 30000)])      # four frames per workgroup instead of eight
def test_stft_stereo(hip, n_fft, hop, n):
    from gcc_nmf_amd.gccNMFFunctions import computeComplexMixtureSpectrogram
    rng = np.random.RandomState(n_fft + hop)
    x = (rng.standard_normal((2, n)) * 0.1).astype(np.float32)
    X = computeComplexMixtureSpectrogram(x, n_fft, hop, np.hanning)
    ref = O.computeComplexMixtureSpectrogram(x, n_fft, hop, np.hanning)
    assert X.shape == ref.shape and X.dtype == np.complex64
    assert np.abs(X - ref).max() < 1e-5 * np.abs(ref).max()


@pytest.mark.parametrize('n_fft,hop,n', [(64, 16, 2000), (128, 32, 3000), (256, 100, 3000), (512, 128, 9000), (1024, 256, 40000), (2048, 512, 30000),
                                         (4096, 1024, 40000)])
def test_register_pass_fft_is_bitwise_the_stage_per_round_trip_fft(hip, n_fft, hop, n):
    """Round 4 runs up to four radix-2 stages per LDS round trip (fft_core.h: fft_pass); every butterfly computes the same expression on
    the same operands, so the STFT (X, V, coherence) and both inverse forms give the SAME BITS as the one-stage-per-round-trip routine
    (tuning key 15 = 0), at every supported size."""
    from gcc_nmf_amd.engine import GCCNMFEngine
    from gcc_nmf_amd.synthetic import synthetic_batch
    lib = hip.lib()
    xs = synthetic_batch(3, 2, numSamples=n)
    outs = []
    for r16 in (0, 1):
        if lib.gccnmf_set_tuning(15, r16) != 0:
            pytest.skip('key 15 (one FFT stage per LDS round trip) exists in experiment builds only: make EXPERIMENTS=1')
        e = GCCNMFEngine(n, windowSize=n_fft, hopSize=hop, dictionarySize=16, numIterations=2, batch=2, numTargets=2)
        e.upload(xs)
        e.stft()
        X, V, C = e.X.clone(), e.V.clone(), e.CC.clone()
        # the inverse transform of the mixture spectrogram itself (spec = X for both 'targets'), fused and two-kernel forms
        e.spec.copy_(torch.stack([e.X[:, 0], e.X[:, 1], e.X[:, 1], e.X[:, 0]], dim=1))
        ys = []
        for fused in ([True, False] if e.fused_istft or n_fft + 3 * hop <= 2048 else [False]):
            e.fused_istft = fused
            e.istft()
            ys.append(e.y.clone())
        outs.append((X, V, C, ys))
    lib.gccnmf_set_tuning(15, 1)
    for a, b in zip(outs[0][:3], outs[1][:3]):
        assert torch.equal(a, b)
    assert len(outs[0][3]) == len(outs[1][3])
    for a, b in zip(outs[0][3], outs[1][3]):
        assert torch.isfinite(a).all() and torch.equal(a, b)
    if len(outs[1][3]) == 2:
        assert torch.equal(outs[1][3][0], outs[1][3][1])          # fused == two-kernel form, as before


def test_istft_n_fft_4096(hip):
    from gcc_nmf_amd.librosaSTFT import stft, istft
    rng = np.random.RandomState(7)
    sig = (rng.standard_normal(40000) * 0.1).astype(np.float32)
    S = O.stft(sig.copy(), 4096, 1024, 4096, np.hanning, center=True)
    y, yr = istft(S, 1024, 4096, np.hanning), O.istft(S, 1024, 4096, np.hanning)
    assert y.shape == yr.shape and np.abs(y - yr).max() < 1e-5 * np.abs(yr).max()
    assert np.abs(stft(sig.copy(), 4096, 1024, 4096, np.hanning, center=True) - S).max() < 1e-5 * np.abs(S).max()


@pytest.mark.parametrize('key', ['400_100', '1000_250', '1536_384', '1000_300'])
def test_any_n_fft_against_the_reference(hip, key):
    """Non-power-of-two n_fft (the reference takes any size, librosaSTFT.py:162-179): the DFT-as-GEMM path against outputs of the
    UNMODIFIED reference stft / istft (tests/golden/kat_nfft.npz, oracle/make_golden_wh.py)."""
    from gcc_nmf_amd.librosaSTFT import stft, istft
    kat = golden('kat_nfft')
    n_fft, hop = [int(v) for v in key.split('_')]
    y, S, yi, yc = kat['y_' + key], kat['S_' + key], kat['yi_' + key], kat['yc_' + key]
    X = stft(y.copy(), n_fft, hop, n_fft, np.hanning, center=False)
    assert X.shape == S.shape and X.dtype == np.complex64 and X.flags['F_CONTIGUOUS']
    assert np.abs(X - S).max() < 1e-5 * np.abs(S).max()
    got = istft(S, hop, n_fft, np.hanning)
    assert got.shape == yi.shape and got.dtype == np.float32 and np.abs(got - yi).max() < 1e-5 * np.abs(yi).max()
    got = istft(S, hop, n_fft, np.hanning, center=False)
    assert got.shape == yc.shape and np.abs(got - yc).max() < 1e-5 * np.abs(yc).max()


def test_odd_and_tiny_n_fft_against_the_oracle(hip):
    from gcc_nmf_amd.librosaSTFT import stft
    from gcc_nmf_amd import gccNMFFunctions as G
    rng = np.random.RandomState(12)
    y = (rng.standard_normal(5000) * 0.1).astype(np.float32)
    for n_fft, hop in [(375, 125), (30, 7), (2049, 512), (48, 48)]:
        X = stft(y, n_fft, hop, n_fft, np.hanning, center=False)
        ref = O.stft(y, n_fft, hop, n_fft, np.hanning, center=False)
        assert X.shape == ref.shape and np.abs(X - ref).max() < 1e-5 * np.abs(ref).max(), n_fft
    x2 = (rng.standard_normal((2, 9000)) * 0.1).astype(np.float32)
    X = G.computeComplexMixtureSpectrogram(x2, 1000, 250, np.hanning)              # both channels through one GEMM launch
    ref = O.computeComplexMixtureSpectrogram(x2, 1000, 250, np.hanning)
    assert X.shape == ref.shape == (2, 501, 33) and np.abs(X - ref).max() < 1e-5 * np.abs(ref).max()


def test_stft_mono_center_and_errors(hip):
    from gcc_nmf_amd.librosaSTFT import stft, istft, ParameterError
    rng = np.random.RandomState(3)
    y = (rng.standard_normal(7000) * 0.1).astype(np.float32)
    for center in (False, True):
        X = stft(y, 1024, 256, 1024, np.hanning, center=center)
        ref = O.stft(y, 1024, 256, 1024, np.hanning, center=center)
        assert X.shape == ref.shape and np.abs(X - ref).max() < 1e-5 * np.abs(ref).max()
    with pytest.raises(ParameterError):
        stft(np.zeros(100, np.float32), 1024, 256, 1024, np.hanning, center=False)
    with pytest.raises(ParameterError):
        stft(y, 1024, 0, 1024, np.hanning, center=False)
    bad = y.copy()
    bad[5] = np.nan
    with pytest.raises(ParameterError):
        stft(bad, 1024, 256, 1024, np.hanning, center=False)
    with pytest.raises(ParameterError):
        stft(np.zeros(9000, np.float32)[::2], 1024, 256, 1024, np.hanning, center=False)
    with pytest.raises(ParameterError):
        stft(y, 1024, 256, 1024, np.ones(1000), center=False)
    spec = (rng.standard_normal((513, 9)) + 1j * rng.standard_normal((513, 9))).astype(np.complex64)
    for center in (True, False):
        yy = istft(spec, 256, 1024, np.hanning, center=center)
        ref = O.istft(spec, 256, 1024, np.hanning, center=center)
        assert yy.shape == ref.shape and yy.dtype == np.float32
        assert np.abs(yy - ref).max() < 2e-6 * np.abs(ref).max()


def test_stft_istft_golden_kat(hip):
    from gcc_nmf_amd.librosaSTFT import stft, istft
    kat = golden('kat_primitives')
    for n_fft, hop in [(1024, 256), (1024, 128), (512, 64), (256, 100)]:
        ref = kat['stft_%d_%d' % (n_fft, hop)]
        X = stft(kat['stft_sig'].copy(), n_fft, hop, n_fft, np.hanning, center=False)
        assert np.abs(X - ref).max() < 1e-5 * np.abs(ref).max()
    for key, spec, hop, ws in [('istft_1024_256', 'istft_spec', 256, 1024), ('istft_1024_128', 'istft_spec', 128, 1024),
                               ('istft_256_64', 'istft_spec2', 64, 256)]:
        y = istft(kat[spec], hop, ws, np.hanning)
        assert y.shape == kat[key].shape
        assert np.abs(y - kat[key]).max() < 2e-6 * np.abs(kat[key]).max()


def test_signal_estimates(hip):
    from gcc_nmf_amd.gccNMFFunctions import getTargetSignalEstimates
    rng = np.random.RandomState(11)
    S = (rng.standard_normal((3, 2, 513, 40)) + 1j * rng.standard_normal((3, 2, 513, 40))).astype(np.complex64)
    y = getTargetSignalEstimates(S, 1024, 256, np.hanning)
    ref = O.getTargetSignalEstimates(S, 1024, 256, np.hanning)
    assert y.shape == ref.shape == (3, 2, 256 * 39) and y.dtype == np.float32
    assert np.abs(y - ref).max() < 2e-6 * np.abs(ref).max()


# ------------------------------------------------------------------------------------------------
# KL-NMF
# ------------------------------------------------------------------------------------------------
def test_klnmf_golden_kat(hip):
    from gcc_nmf_amd.gccNMFFunctions import performKLNMF
    kat = golden('kat_primitives')
    for name in 'abc':
        K, it, alpha = kat['nmf_%s_params' % name]
        W, H = performKLNMF(kat['nmf_V'], int(K), int(it), alpha if alpha else 0)
        assert W.dtype == np.float32 and W.shape == kat['nmf_%s_W' % name].shape
        assert rel(W, kat['nmf_%s_W' % name]) < 1e-5 and rel(H, kat['nmf_%s_H' % name]) < 1e-5
    W, H = performKLNMF(kat['nmf_V'], 4, 2, 0, seedValue=7)
    assert rel(W, kat['nmf_seed7_W']) < 1e-5 and rel(H, kat['nmf_seed7_H']) < 1e-5


@pytest.fixture(params=[0, 1, 2], ids=['tile-auto', 'tile-throughput', 'tile-small'])
def tile_policy(hip, request):
    """Both GEMM tile shapes (512x64 throughput tile, 128x64 small-batch tile) at every test size."""
    lib = hip.lib()
    assert lib.gccnmf_set_tuning(2, request.param) == 0
    yield request.param
    lib.gccnmf_set_tuning(2, 0)


@pytest.mark.parametrize('F,N,K,iters,alpha', [(513, 90, 128, 12, 0), (513, 201, 192, 6, 0.3), (257, 77, 40, 10, 0), (200, 333, 300, 5, 0),
                                               (129, 64, 64, 8, 0), (1025, 50, 64, 4, 0),
                                               (513, 900, 640, 5, 0)])      # one file: unequal split-K parts (40 k-tiles in 3, 57 in 4)
def test_klnmf_vs_oracle(hip, F, N, K, iters, alpha, tile_policy):
    from gcc_nmf_amd.gccNMFFunctions import performKLNMF
    rng = np.random.RandomState(F + N + K)
    V = (np.abs(rng.standard_normal((F, N))) + 0.01).astype(np.float32)
    state = np.random.get_state()
    W, H = performKLNMF(V, K, iters, alpha)
    Wr, Hr = O.performKLNMF(V, K, iters, alpha)
    assert W.shape == (F, K) and H.shape == (K, N)
    assert rel(W, Wr) < 1e-4 and rel(H, Hr) < 1e-4, (rel(W, Wr), rel(H, Hr))
    # the reference leaves the global RNG seeded-and-advanced; so do we
    np.random.seed(0)
    np.random.random((F, K))
    np.random.random((K, N))
    expected_next = np.random.random()
    performKLNMF(V, K, 0, 0)
    assert np.random.random() == expected_next
    np.random.set_state(state)


@pytest.mark.parametrize('N,K,iters', [(20000, 64, 4), (80000, 48, 3), (4100, 128, 5)])
def test_klnmf_one_big_matrix_takes_the_column_block_path(hip, N, K, iters):
    """performKLNMF on ONE big matrix (the dictionary pre-training call, gccNMFPretraining.py:79-80): beyond 4096 columns the
    reference-named function hands the columns to the batched throughput kernels as in-place column blocks."""
    from gcc_nmf_amd import gccNMFFunctions as G
    assert N > G.LARGE_N_COLUMNS
    rng = np.random.RandomState(N + K)
    V = (np.abs(rng.standard_normal((513, N))) + 0.01).astype(np.float32)
    W, H = G.performKLNMF(V, K, iters, 0)
    Wr, Hr = O.performKLNMF(V, K, iters, 0)
    assert W.shape == Wr.shape and H.shape == Hr.shape
    assert rel(W, Wr) < 1e-4 and rel(H, Hr) < 1e-4, (rel(W, Wr), rel(H, Hr))
    assert np.allclose(np.linalg.norm(W, axis=0), 1.0, atol=1e-5)


def test_klnmf_direct_path_against_the_split_k_path_and_small_batches(hip):
    """The direct-to-register latency path (csrc/direct.hip) against what key 10 = 0 runs for one mixture -- the round-3 split-K launches
    in the lab build (key 5 exists there), the LDS-DMA ring kernel in the product library, where the split-K path is compiled out -- and for
    a handful of files (tuning key 12 lets small batches take the direct path): the same factors to round-off (different summation grouping)."""
    lib = hip.lib()
    lab = lib.gccnmf_set_tuning(5, 3) == 0              # key 5 (parts of the single-file split-K W.H) is an experiment-build key
    other = 'split' if lab else 'ring'
    F, T, K, B = 513, 311, 256, 3
    g = hip_geometry(F, T, K)
    rng = np.random.RandomState(3)
    V = (np.abs(rng.standard_normal((B, F, 2 * T))) + 0.01).astype(np.float32)
    from gcc_nmf_amd.engine import klnmf_initial_factors, padded
    W0, H0 = klnmf_initial_factors(F, 2 * T, K)
    res = {}
    for name, direct, maxb, batch in [(other, 0, 1, 1), ('direct', 1, 1, 1), ('direct-batch', 1, 4, B), ('ring-batch', 0, 1, B)]:
        assert lib.gccnmf_set_tuning(10, direct) == 0 and lib.gccnmf_set_tuning(12, maxb) == 0
        Vd = padded(V[:batch], (batch, g.Fp, g.Np), 'cuda')
        Wd = padded(np.repeat(W0[None], batch, 0), (batch, g.Fp, g.Kp), 'cuda')
        Hd = padded(np.repeat(H0[None], batch, 0), (batch, g.Kp, g.Np), 'cuda')
        ws = torch.zeros(lib.gccnmf_klnmf_workspace_floats(F, 2 * T, K, batch), dtype=torch.float32, device='cuda')
        assert lib.gccnmf_klnmf(Vd.data_ptr(), Wd.data_ptr(), Hd.data_ptr(), ws.data_ptr(), F, 2 * T, K, batch, 7, 0.0, 1e-16, 0, stream()) == 0
        torch.cuda.synchronize()
        res[name] = (Wd.cpu().numpy(), Hd.cpu().numpy())
    lib.gccnmf_set_tuning(10, 1)
    lib.gccnmf_set_tuning(12, 4)
    for b in range(B):
        Wr, Hr = O.performKLNMF(V[b], K, 7, 0)
        for name in ('direct-batch', 'ring-batch') + ((other, 'direct') if b == 0 else ()):
            W, H = res[name]
            assert rel(W[b, :F, :K], Wr) < 1e-4 and rel(H[b, :K, :2 * T], Hr) < 1e-4, (name, b)
            # the padding stays exactly zero (it is a reduction operand)
            assert not W[b, F:].any() and not W[b, :, K:].any() and not H[b, K:].any() and not H[b, :, 2 * T:].any(), (name, b)
    assert rel(res['direct'][0], res[other][0]) < 2e-5


@pytest.mark.parametrize('F,T,K,B,alpha', [(513, 75, 128, 6, 0.0), (513, 40, 200, 9, 0.2), (257, 33, 100, 5, 0.0), (385, 20, 64, 12, 0.0),
                                           (513, 330, 128, 13, 0.0), (513, 50, 96, 5, 0.1), (513, 37, 20, 8, 0.0), (129, 64, 32, 5, 0.0)])
def test_klnmf_short_dictionary_fused_launches(hip, F, T, K, B, alpha):
    """Short dictionaries (the reference driver's K = 128: runGCCNMF.py:41): K1 + K2 of an iteration as ONE launch with R kept in the
    H resident in the registers of a column tile's workgroup (tuning key 16) and K3 + K4a as ONE launch of 64-bin slabs with their W
    rows in registers (key 17; both K <= 128, csrc/direct.hip) against the four-launch form and the oracle; the padding of W and H stays exactly zero; a file's bits do not
    depend on the batch."""
    lib = hip.lib()
    from gcc_nmf_amd.engine import Geometry, padded, klnmf_initial_factors
    N = 2 * T
    g = Geometry(F, T, K)
    rng = np.random.RandomState(F + K + B)
    V = (np.abs(rng.standard_normal((B, F, N))) + 0.01).astype(np.float32)
    W0, H0 = klnmf_initial_factors(F, N, K)
    res = {}
    try:
        for name, k16, k17, files in [('four-launch', 0, 0, list(range(B))), ('fused-12', 2, 0, list(range(B))), ('fused-34', 0, 2, list(range(B))),
                                      ('fused', 2, 2, list(range(B))), ('fused-some', 2, 2, [B - 1, 0, 2, 1, 3])]:
            assert lib.gccnmf_set_tuning(16, k16) == 0 and lib.gccnmf_set_tuning(17, k17) == 0
            b = len(files)
            if F == 513:                                       # the library reports the launches it will use
                assert lib.gccnmf_klnmf_plan(F, N, K, b, 0) & 7 == ((2 if k16 else 0) | (4 if k17 else 0) if K <= 128 else 0), name      # (bit 3: chained, by rule or forced)
            Vd = padded(V[files], (b, g.Fp, g.Np), 'cuda')
            Wd = padded(np.repeat(W0[None], b, 0), (b, g.Fp, g.Kp), 'cuda')
            Hd = padded(np.repeat(H0[None], b, 0), (b, g.Kp, g.Np), 'cuda')
            ws = torch.zeros(lib.gccnmf_klnmf_workspace_floats(F, N, K, b), dtype=torch.float32, device='cuda')
            assert lib.gccnmf_klnmf(Vd.data_ptr(), Wd.data_ptr(), Hd.data_ptr(), ws.data_ptr(), F, N, K, b, 6, alpha, 1e-16, 0, stream()) == 0
            torch.cuda.synchronize()
            res[name] = (Wd.cpu().numpy(), Hd.cpu().numpy())
    finally:
        lib.gccnmf_set_tuning(16, FUSED_K12_DEFAULT)
        lib.gccnmf_set_tuning(17, FUSED_K34_DEFAULT)
    for name in ('four-launch', 'fused-12', 'fused-34', 'fused'):
        W, H = res[name]
        assert np.isfinite(W).all() and np.isfinite(H).all(), name
        assert not W[:, F:].any() and not W[:, :, K:].any() and not H[:, K:].any() and not H[:, :, N:].any(), name
        for b in (0, B - 1):
            Wr, Hr = O.performKLNMF(V[b], K, 6, alpha)
            assert rel(W[b, :F, :K], Wr) < 1e-4 and rel(H[b, :K, :N], Hr) < 1e-4, (name, b, rel(W[b, :F, :K], Wr), rel(H[b, :K, :N], Hr))
        assert rel(W, res['four-launch'][0]) < 2e-5 and rel(H, res['four-launch'][1]) < 2e-5, name
    assert np.array_equal(res['fused-some'][0][0], res['fused'][0][B - 1]) and np.array_equal(res['fused-some'][1][2], res['fused'][1][2])


def test_klnmf_slab_launch_for_whole_rounds_and_the_rest_on_the_two_launches(hip):
    """70 files at K = 32: the cost model of tuning key 17 puts one whole round of slab workgroups (64 files) on the fused K3 + K4a launch
    and the remaining 6 files on the two launches behind it; every file against the oracle, and against the four-launch form."""
    lib = hip.lib()
    from gcc_nmf_amd.engine import Geometry, padded, klnmf_initial_factors
    F, T, K, B = 513, 20, 32, 70
    N = 2 * T
    g = Geometry(F, T, K)
    rng = np.random.RandomState(70)
    V = (np.abs(rng.standard_normal((B, F, N))) + 0.01).astype(np.float32)
    W0, H0 = klnmf_initial_factors(F, N, K)
    res = {}
    try:
        for name, k16, k17 in [('four-launch', 0, 0), ('auto', 0, 1)]:
            assert lib.gccnmf_set_tuning(16, k16) == 0 and lib.gccnmf_set_tuning(17, k17) == 0
            assert lib.gccnmf_klnmf_plan(F, N, K, B, 0) == (4 if k17 else 0)
            Vd = padded(V, (B, g.Fp, g.Np), 'cuda')
            Wd = padded(np.repeat(W0[None], B, 0), (B, g.Fp, g.Kp), 'cuda')
            Hd = padded(np.repeat(H0[None], B, 0), (B, g.Kp, g.Np), 'cuda')
            ws = torch.zeros(lib.gccnmf_klnmf_workspace_floats(F, N, K, B), dtype=torch.float32, device='cuda')
            assert lib.gccnmf_klnmf(Vd.data_ptr(), Wd.data_ptr(), Hd.data_ptr(), ws.data_ptr(), F, N, K, B, 5, 0.0, 1e-16, 0, stream()) == 0
            torch.cuda.synchronize()
            res[name] = (Wd.cpu().numpy(), Hd.cpu().numpy())
    finally:
        lib.gccnmf_set_tuning(16, FUSED_K12_DEFAULT)
        lib.gccnmf_set_tuning(17, FUSED_K34_DEFAULT)
    W, H = res['auto']
    assert np.isfinite(W).all() and np.isfinite(H).all()
    assert not W[:, F:].any() and not W[:, :, K:].any() and not H[:, K:].any() and not H[:, :, N:].any()
    for b in (0, 63, 64, 69):
        Wr, Hr = O.performKLNMF(V[b], K, 5, 0.0)
        assert rel(W[b, :F, :K], Wr) < 1e-4 and rel(H[b, :K, :N], Hr) < 1e-4, (b, rel(W[b, :F, :K], Wr), rel(H[b, :K, :N], Hr))
    assert rel(W, res['four-launch'][0]) < 2e-5 and rel(H, res['four-launch'][1]) < 2e-5
    assert np.array_equal(W[64:], res['four-launch'][0][64:]) and np.array_equal(H[64:], res['four-launch'][1][64:])      # the rest: the same launches


def hip_geometry(F, T, K):
    from gcc_nmf_amd.engine import Geometry
    return Geometry(F, T, K)


def test_klnmf_batch_is_file_independent(hip):
    """A file's result does not depend on the batch it rides in (XCD-affine map for batch >= 8 included): bit for bit between
    batches of the batched kernels; a file processed ALONE or with a few others takes the direct latency path (csrc/direct.hip: the
    reduction split over the waves of a workgroup, added in a fixed order) and agrees to summation-order accuracy."""
    lib = hip.lib()
    F, T, K, B = 513, 30, 128, 9
    N = 2 * T
    rng = np.random.RandomState(5)
    from gcc_nmf_amd.engine import Geometry, padded, klnmf_initial_factors
    g = Geometry(F, T, K)
    V = (np.abs(rng.standard_normal((B, F, N))) + 0.01).astype(np.float32)
    W0, H0 = klnmf_initial_factors(F, N, K)
    outs = []
    for files in ([0, 3, 1, 2, 4], [5, 8, 7, 6, 2], list(range(B)), [8], [5, 8]):
        b = len(files)
        dV = padded(V[files], (b, g.Fp, g.Np), 'cuda')
        dW = padded(np.repeat(W0[None], b, 0), (b, g.Fp, g.Kp), 'cuda')
        dH = padded(np.repeat(H0[None], b, 0), (b, g.Kp, g.Np), 'cuda')
        ws = torch.zeros(lib.gccnmf_klnmf_workspace_floats(F, N, K, b), dtype=torch.float32, device='cuda')
        assert lib.gccnmf_klnmf(dV.data_ptr(), dW.data_ptr(), dH.data_ptr(), ws.data_ptr(), F, N, K, b, 5, 0.0, 1e-16, 0, stream()) == 0
        outs.append((dW.cpu().numpy(), dH.cpu().numpy()))
    assert np.array_equal(outs[0][0][0], outs[2][0][0]) and np.array_equal(outs[0][1][0], outs[2][1][0])
    assert np.array_equal(outs[0][0][1], outs[2][0][3]) and np.array_equal(outs[0][1][1], outs[2][1][3])
    assert np.array_equal(outs[1][0][1], outs[2][0][8]) and np.array_equal(outs[1][1][1], outs[2][1][8])
    assert rel(outs[3][0][0], outs[2][0][8]) < 2e-6 and rel(outs[3][1][0], outs[2][1][8]) < 2e-6        # alone: the direct path
    assert rel(outs[4][0][1], outs[2][0][8]) < 2e-6 and rel(outs[4][1][0], outs[2][1][5]) < 2e-6        # a pair: the direct path
    for Wq, Hq in (outs[3], outs[4]):
        assert not Wq[:, F:, :].any() and not Wq[:, :, K:].any() and not Hq[:, K:, :].any() and not Hq[:, :, N:].any()
    # padding stayed zero
    Wp, Hp = outs[2]
    assert not Wp[:, F:, :].any() and not Wp[:, :, K:].any() and not Hp[:, K:, :].any() and not Hp[:, :, N:].any()
    # and it matches the oracle
    Wr, Hr = O.performKLNMF(V[8], K, 5, 0)
    assert rel(Wp[8, :F, :K], Wr) < 1e-4 and rel(Hp[8, :K, :N], Hr) < 1e-4


@pytest.mark.parametrize('F,T,K,B,flags', [(513, 622, 64, 26, 0), (513, 330, 512, 50, 0), (513, 200, 1024, 70, 2), (513, 622, 96, 64, 0), (513, 330, 256, 40, 0)])
def test_narrow_items_and_resident_workgroups_are_bitwise(hip, F, T, K, B, flags):
    """The throughput-tile launch in every form it can take (csrc/gemm_dma.h): full 512 x 64 tiles only (key 9 = 0: the reference form) |
    the launcher's own choice -- half-height tiles for the files of a partial last round and for outputs of at most 256 rows, a narrow
    item for a file's ragged last column tile (key 9 = 1) | every tile as two narrow halves (key 9 = 2) | in an experiment build, 512
    resident workgroups pulling items by ticket (key 18), with and without the next item's first k-tile requested under the epilogue (19).
    Every output element sees the same k order in each of them: W and H are bit for bit those of the reference form.
    (K1/K3 at 26 x 20 = 520 items; the H update at K = 512, 50 x 11 = 550; the unfused R.H^T at K = 1024, 70 x 16 = 1120; 64 files.)"""
    lib = hip.lib()
    from gcc_nmf_amd.engine import Geometry, padded, klnmf_initial_factors
    N = 2 * T
    g = Geometry(F, T, K)
    rng = np.random.RandomState(B)
    V = (np.abs(rng.standard_normal((B, F, N))) + 0.01).astype(np.float32)
    W0, H0 = klnmf_initial_factors(F, N, K)
    dV = padded(V, (B, g.Fp, g.Np), 'cuda')
    outs = []
    forms = [(0, 0, 0), (1, 0, 0), (2, 0, 0), (3, 0, 0)]          # (3: half-height tiles everywhere)
    experiments = lib.gccnmf_set_tuning(18, 0) == 0    # an experiment build (make EXPERIMENTS=1) also carries the resident-workgroup grid
    if experiments:
        forms += [(0, 1, 0), (1, 1, 1), (2, 1, 1), (1, 1, 0)]
    try:
        for narrow, resident, prefetch in forms:
            assert lib.gccnmf_set_tuning(9, narrow) == 0
            if experiments:
                assert lib.gccnmf_set_tuning(18, resident) == 0 and lib.gccnmf_set_tuning(19, prefetch) == 0
            assert lib.gccnmf_set_tuning(16, 0) == 0 and lib.gccnmf_set_tuning(17, 0) == 0          # K <= 128: the four-launch form, on the throughput tile
            dW = padded(np.repeat(W0[None], B, 0), (B, g.Fp, g.Kp), 'cuda')
            dH = padded(np.repeat(H0[None], B, 0), (B, g.Kp, g.Np), 'cuda')
            ws = torch.zeros(lib.gccnmf_klnmf_workspace_floats(F, N, K, B), dtype=torch.float32, device='cuda')
            assert lib.gccnmf_klnmf(dV.data_ptr(), dW.data_ptr(), dH.data_ptr(), ws.data_ptr(), F, N, K, B, 3, 0.1, 1e-16, flags, stream()) == 0
            outs.append((dW.cpu().numpy(), dH.cpu().numpy()))
            # once more on the same stream: a resident-workgroup launch finds the ticket counters as the previous one left them
            assert lib.gccnmf_klnmf(dV.data_ptr(), dW.data_ptr(), dH.data_ptr(), ws.data_ptr(), F, N, K, B, 1, 0.1, 1e-16, flags, stream()) == 0
    finally:
        for key, value in ((9, 1), (18, 0), (19, 1), (16, 1), (17, 1)):
            lib.gccnmf_set_tuning(key, value)
    for form, (Wq, Hq) in zip(forms[1:], outs[1:]):
        assert np.array_equal(outs[0][0], Wq) and np.array_equal(outs[0][1], Hq), form
    Wp, Hp = outs[1]
    assert not Wp[:, F:, :].any() and not Wp[:, :, K:].any() and not Hp[:, K:, :].any() and not Hp[:, :, N:].any()      # padding stayed zero
    Wr, Hr = O.performKLNMF(V[B - 1], K, 3, 0.1)                      # a file of the last round against the oracle
    assert rel(outs[1][0][B - 1, :F, :K], Wr) < 1e-4 and rel(outs[1][1][B - 1, :K, :N], Hr) < 1e-4


# ------------------------------------------------------------------------------------------------
# localisation, scores, masks, reconstruction
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope='module')
def scene():
    x = O.synthetic_mixture(1, numSamples=40000)
    return O.runGCCNMF(x, 16000, 1024, 256, 128, 1.0, 3, dictionarySize=64, numIterations=15, return_intermediates=True), x


def test_angular_spectrogram_and_peaks(hip, scene):
    from gcc_nmf_amd import gccNMFFunctions as G
    r, x = scene
    freqs = np.linspace(0, 8000.0, 513)
    A = G.getAngularSpectrogram(r['C'], freqs, 1.0, 128)
    assert A.shape == r['A'].shape and A.dtype == np.float64
    assert np.abs(A - r['A']).max() < 1e-3
    idx = G.estimateTargetTDOAIndexesFromAngularSpectrum(np.mean(A, axis=-1), 1.0, 128, 3)
    assert idx == r['idx'] and isinstance(idx, list)
    idx2, meanA = G.getTargetTDOAEstimates(r['X'], 16000, 1.0, 128, 3)
    assert idx2 == r['idx'] and np.abs(meanA - r['meanA']).max() < 1e-3
    kat = golden('kat_primitives')
    s = kat['peaks_spectrum']
    assert G.estimateTargetTDOAIndexesFromAngularSpectrum(s, 1.0, len(s), 2) == list(kat['peaks_top2'])
    assert G.estimateTargetTDOAIndexesFromAngularSpectrum(s, 1.0, len(s), 3) == list(kat['peaks_top3'])
    with pytest.raises(ValueError):
        G.estimateTargetTDOAIndexesFromAngularSpectrum(s, 1.0, len(s), 9)
    with pytest.raises(ValueError):
        G.estimateTargetTDOAIndexesFromAngularSpectrum(s, 1.0, len(s), None)


def test_scores_masks_reconstruction(hip, scene):
    from gcc_nmf_amd import gccNMFFunctions as G
    r, x = scene
    freqs = np.linspace(0, 8000.0, 513)
    stereoH = np.array(np.hsplit(r['H'], 2))
    Gs = G.getTargetTDOAGCCNMFs(r['C'], 1.0, 128, freqs, r['idx'], r['W'], stereoH)
    assert Gs.shape == r['G'].shape and Gs.dtype == np.float32
    assert np.abs(Gs - r['G']).max() < 2e-5 * np.abs(r['G']).max()
    M = G.getTargetCoefficientMasks(r['G'], 3)
    assert np.array_equal(M, r['M'])                                   # same scores in -> identical masks out
    kat = golden('kat_primitives')
    assert np.array_equal(G.getTargetCoefficientMasks(kat['masks_in'], 3), kat['masks_out'])   # tie + NaN cases
    allnan = kat['masks_in'].copy()
    allnan[:, 0, 0] = np.nan
    with pytest.raises(ValueError):
        G.getTargetCoefficientMasks(allnan, 3)
    Sp = G.getTargetSpectrogramEstimates(r['M'], r['X'], r['W'], stereoH)
    assert Sp.shape == r['S'].shape and Sp.dtype == np.complex64
    assert np.abs(Sp - r['S']).max() < 1e-5 * np.abs(r['S']).max()
    # soft masks go through the same entry point
    soft = np.random.RandomState(2).rand(*r['M'].shape).astype(np.float32)
    Sp2 = G.getTargetSpectrogramEstimates(soft, r['X'], r['W'], stereoH)
    ref2 = O.getTargetSpectrogramEstimates(soft, r['X'], r['W'], stereoH)
    assert np.abs(Sp2 - ref2).max() < 1e-5 * np.abs(ref2).max()


def test_reference_call_sequence_dropin(hip, scene):
    """gccNMF/runGCCNMF.py:36-52, line by line, on the replacement module's names."""
    from gcc_nmf_amd.gccNMFFunctions import (computeComplexMixtureSpectrogram, performKLNMF, getAngularSpectrogram,
                                            estimateTargetTDOAIndexesFromAngularSpectrum, getTargetTDOAGCCNMFs,
                                            getTargetCoefficientMasks, getTargetSpectrogramEstimates, getTargetSignalEstimates,
                                            hanning, linspace, concatenate, array, hsplit, mean)
    r, stereoSamples = scene
    sampleRate, windowSize, hopSize, numTDOAs, d, numTargets = 16000, 1024, 256, 128, 1.0, 3
    complexMixtureSpectrogram = computeComplexMixtureSpectrogram(stereoSamples, windowSize, hopSize, hanning)
    numChannels, numFrequencies, numTime = complexMixtureSpectrogram.shape
    frequenciesInHz = linspace(0, sampleRate / 2.0, numFrequencies)
    V = concatenate(abs(complexMixtureSpectrogram), axis=-1)
    W, H = performKLNMF(V, dictionarySize=64, numIterations=15, sparsityAlpha=0)
    stereoH = array(hsplit(H, numChannels))
    spectralCoherenceV = complexMixtureSpectrogram[0] * complexMixtureSpectrogram[1].conj() / abs(complexMixtureSpectrogram[0]) / abs(complexMixtureSpectrogram[1])
    angularSpectrogram = getAngularSpectrogram(spectralCoherenceV, frequenciesInHz, d, numTDOAs)
    meanAngularSpectrum = mean(angularSpectrogram, axis=-1)
    targetTDOAIndexes = estimateTargetTDOAIndexesFromAngularSpectrum(meanAngularSpectrum, d, numTDOAs, numTargets)
    targetTDOAGCCNMFs = getTargetTDOAGCCNMFs(spectralCoherenceV, d, numTDOAs, frequenciesInHz, targetTDOAIndexes, W, stereoH)
    targetCoefficientMasks = getTargetCoefficientMasks(targetTDOAGCCNMFs, numTargets)
    targetSpectrogramEstimates = getTargetSpectrogramEstimates(targetCoefficientMasks, complexMixtureSpectrogram, W, stereoH)
    targetSignalEstimates = getTargetSignalEstimates(targetSpectrogramEstimates, windowSize, hopSize, hanning)
    assert targetTDOAIndexes == r['idx']
    assert rel(W, r['W']) < 1e-4 and rel(H, r['H']) < 1e-4
    # the masks are exact except at genuine near-ties of the oracle's own two best target scores
    flipped = np.argmax(targetCoefficientMasks, 0) != np.argmax(r['M'], 0)
    srt = np.sort(np.asarray(r['G'], np.float64), axis=0)
    gap = (srt[-1] - srt[-2]) / np.maximum(np.abs(srt[-1]), 1e-300)
    assert (gap[flipped] < 1e-4).all(), (int(flipped.sum()), float(gap[flipped].max()))
    assert targetSignalEstimates.shape == r['y'].shape and targetSignalEstimates.dtype == np.float32
    rms = np.sqrt(np.mean((targetSignalEstimates.astype(np.float64) - r['y']) ** 2))
    assert rms < 1e-4, rms


@pytest.mark.parametrize('F,T,K', [(513, 40, 128), (257, 33, 64), (200, 50, 100)])
def test_klnmf_fused_and_unfused_w_update_agree(hip, F, T, K):
    """The W update + normalisation fused into the R.H^T epilogue (default when 128 < F-1 <= 512) against the two-launch
    form (GCCNMF_FLAG_UNFUSED_W_UPDATE) and the oracle."""
    lib = hip.lib()
    from gcc_nmf_amd.engine import Geometry, padded, klnmf_initial_factors
    N, B = 2 * T, 3
    g = Geometry(F, T, K)
    rng = np.random.RandomState(F + K)
    V = (np.abs(rng.standard_normal((B, F, N))) + 0.01).astype(np.float32)
    W0, H0 = klnmf_initial_factors(F, N, K)
    res = []
    for flags in (0, 2):
        dV = padded(V, (B, g.Fp, g.Np), 'cuda')
        dW = padded(np.repeat(W0[None], B, 0), (B, g.Fp, g.Kp), 'cuda')
        dH = padded(np.repeat(H0[None], B, 0), (B, g.Kp, g.Np), 'cuda')
        ws = torch.zeros(lib.gccnmf_klnmf_workspace_floats(F, N, K, B), dtype=torch.float32, device='cuda')
        assert lib.gccnmf_klnmf(dV.data_ptr(), dW.data_ptr(), dH.data_ptr(), ws.data_ptr(), F, N, K, B, 6, 0.0, 1e-16, flags, stream()) == 0
        res.append((dW.cpu().numpy(), dH.cpu().numpy()))
    (Wf, Hf), (Wu, Hu) = res
    assert rel(Wf, Wu) < 1e-5 and rel(Hf, Hu) < 1e-5
    assert not Wf[:, F:, :].any() and not Wf[:, :, K:].any()
    for b in range(B):
        Wr, Hr = O.performKLNMF(V[b], K, 6, 0)
        assert rel(Wf[b, :F, :K], Wr) < 1e-4 and rel(Hf[b, :K, :N], Hr) < 1e-4


@pytest.fixture(params=[1, 0], ids=['lds-dma', 'register-staged'])
def dma_throughput_tile(hip, request):
    """Both operand-staging paths of the throughput tile (csrc/gemm_dma.h LDS-DMA, the default, and csrc/gemm_mfma.h
    register staging), forced at any problem size."""
    lib = hip.lib()
    assert lib.gccnmf_set_tuning(2, 1) == 0 and lib.gccnmf_set_tuning(3, request.param) == 0
    yield
    lib.gccnmf_set_tuning(2, 0)
    lib.gccnmf_set_tuning(3, 1)


@pytest.mark.parametrize('F,N,K,iters,alpha', [(513, 90, 128, 12, 0), (513, 1244, 192, 5, 0.3), (257, 77, 40, 10, 0), (200, 333, 300, 5, 0),
                                               (1025, 50, 64, 4, 0)])
def test_klnmf_dma_staging_vs_oracle(hip, dma_throughput_tile, F, N, K, iters, alpha):
    from gcc_nmf_amd.gccNMFFunctions import performKLNMF
    rng = np.random.RandomState(F + N + K)
    V = (np.abs(rng.standard_normal((F, N))) + 0.01).astype(np.float32)
    W, H = performKLNMF(V, K, iters, alpha)
    Wr, Hr = O.performKLNMF(V, K, iters, alpha)
    assert rel(W, Wr) < 1e-4 and rel(H, Hr) < 1e-4, (rel(W, Wr), rel(H, Hr))


# ------------------------------------------------------------------------------------------------
# chained launches of the iteration (tuning key 21; csrc/gemm_dma.h: GemmSync)
# ------------------------------------------------------------------------------------------------
def _klnmf_run(lib, V, W0, H0, F, N, K, B, iters, flags=0):
    from gcc_nmf_amd.engine import Geometry, padded
    g = Geometry(F, N // 2, K)
    dV = padded(V, (B, g.Fp, g.Np), 'cuda')
    dW = padded(np.repeat(W0[None], B, 0), (B, g.Fp, g.Kp), 'cuda')
    dH = padded(np.repeat(H0[None], B, 0), (B, g.Kp, g.Np), 'cuda')
    ws = torch.zeros(lib.gccnmf_klnmf_workspace_floats(F, N, K, B), dtype=torch.float32, device='cuda')
    assert lib.gccnmf_klnmf(dV.data_ptr(), dW.data_ptr(), dH.data_ptr(), ws.data_ptr(), F, N, K, B, iters, 0.0, 1e-16, flags, stream()) == 0
    torch.cuda.synchronize()
    import ctypes
    st = ctypes.c_int(-1)
    assert lib.gccnmf_klnmf_chain_status(ws.data_ptr(), F, N, K, B, ctypes.byref(st)) == 0 and st.value == 0      # every hand-over clean, one XCC per list
    return dW, dH


@pytest.mark.parametrize('B,T,K,iters', [(16, 622, 1024, 6), (64, 622, 1024, 3), (24, 330, 512, 5), (8, 1300, 384, 4), (26, 622, 1024, 4), (13, 622, 512, 3),
                                         (77, 622, 384, 2), (64, 622, 256, 4), (27, 311, 256, 3)])
def test_chained_iteration_is_bitwise_the_four_launches(hip, B, T, K, iters):
    """Tuning key 21: K1 | K2 (2), the whole iteration K1 | K2 | K3 | K4 (4), or EVERY iteration of the call (8) as ONE launch whose
    consumers wait on per-tile / per-file ready counters in the XCD's L2 instead of on kernel boundaries.  Same tile programs, same k order per element: W and H after several
    iterations are bit for bit those of the four-launch form -- and finite (a consumer that gave up waiting poisons them with NaN)."""
    from gcc_nmf_amd.engine import klnmf_initial_factors
    lib = hip.lib()
    F, N = 513, 2 * T
    rng = np.random.RandomState(B + K)
    V = (np.abs(rng.standard_normal((B, F, N))) + 0.01).astype(np.float32)
    W0, H0 = klnmf_initial_factors(F, N, K)
    res = {}
    try:
        for chain in (0, 2, 4, 8):
            assert lib.gccnmf_set_tuning(21, chain) == 0
            res[chain] = _klnmf_run(lib, V, W0, H0, F, N, K, B, iters)
            if chain:
                res[(chain, 'again')] = _klnmf_run(lib, V, W0, H0, F, N, K, B, iters)
        # a file's tiles spread over the XCD lists (file-major equal eighths), handed over at agent scope (key 23 = 2)
        assert lib.gccnmf_set_tuning(23, 2) == 0 and lib.gccnmf_set_tuning(21, 8) == 0
        res[(8, 'spread lists')] = _klnmf_run(lib, V, W0, H0, F, N, K, B, iters)
        assert lib.gccnmf_set_tuning(23, 3) == 0                  # whole files per XCD at any batch size (by rule only where they balance)
        res[(8, 'whole files')] = _klnmf_run(lib, V, W0, H0, F, N, K, B, iters)
        assert lib.gccnmf_set_tuning(23, 1) == 0
        # batch a multiple of 8: also on the plain launch's lists (key 23 = 0)
        if B % 8 == 0:
            assert lib.gccnmf_set_tuning(23, 0) == 0 and lib.gccnmf_set_tuning(21, 8) == 0
            res[(8, 'plain lists')] = _klnmf_run(lib, V, W0, H0, F, N, K, B, iters)
    finally:
        lib.gccnmf_set_tuning(21, 1)
        lib.gccnmf_set_tuning(23, 1)
    W, H = res[0]
    assert torch.isfinite(W).all() and torch.isfinite(H).all()
    for key, (Wc, Hc) in res.items():
        assert torch.equal(Wc, W) and torch.equal(Hc, H), key


def test_chained_iteration_in_two_file_groups_on_two_streams(hip):
    """The engine's two file groups (two streams, GCCNMF_FLAG_GROUPS) with chained launches in each: bitwise the unchained result."""
    from gcc_nmf_amd.engine import GCCNMFEngine
    from gcc_nmf_amd.synthetic import synthetic_batch
    lib = hip.lib()
    xs = synthetic_batch(700, 32)
    outs = []
    try:
        for chain in (0, 4, 8):
            assert lib.gccnmf_set_tuning(21, chain) == 0
            e = GCCNMFEngine(160000, dictionarySize=1024, numIterations=6, batch=32, nmf_groups=2)
            assert e.nmf_groups == 2
            e.upload(xs)
            e.stft()
            e.klnmf()
            torch.cuda.synchronize()
            outs.append((e.W.clone(), e.H.clone()))
    finally:
        lib.gccnmf_set_tuning(21, 1)
    assert torch.isfinite(outs[0][0]).all()
    for W, H in outs[1:]:
        assert torch.equal(outs[0][0], W) and torch.equal(outs[0][1], H)


def test_chained_launch_completes_with_one_workgroup_per_cu(hip):
    """Freedom from deadlock rests on the in-order dispatcher: a consumer is only ever dispatched after every producer it waits for.  The
    sharpest case is the least residency: key 22 (experiment build) makes a chained launch reserve so much LDS that ONE workgroup fits a
    CU -- 256 resident workgroups, each possibly spinning.  The run completes, is finite (no consumer timed out) and bitwise the same."""
    lib = hip.lib()
    if lib.gccnmf_set_tuning(22, 1) != 0:
        pytest.skip('key 22 (chained launches with one workgroup per CU) exists in experiment builds only: make EXPERIMENTS=1')
    from gcc_nmf_amd.engine import klnmf_initial_factors
    B, T, K, iters = 32, 622, 1024, 4
    F, N = 513, 2 * T
    rng = np.random.RandomState(5)
    V = (np.abs(rng.standard_normal((B, F, N))) + 0.01).astype(np.float32)
    W0, H0 = klnmf_initial_factors(F, N, K)
    try:
        assert lib.gccnmf_set_tuning(21, 4) == 0
        Ws, Hs = _klnmf_run(lib, V, W0, H0, F, N, K, B, iters)
        assert lib.gccnmf_set_tuning(21, 8) == 0
        Wc, Hc = _klnmf_run(lib, V, W0, H0, F, N, K, B, iters)       # the whole call as one launch, still one workgroup per CU
        assert lib.gccnmf_set_tuning(22, 0) == 0
        assert lib.gccnmf_set_tuning(21, 0) == 0
        W, H = _klnmf_run(lib, V, W0, H0, F, N, K, B, iters)
    finally:
        lib.gccnmf_set_tuning(22, 0)
        lib.gccnmf_set_tuning(21, 1)
    assert torch.isfinite(W).all() and torch.isfinite(Ws).all()
    assert torch.equal(Ws, W) and torch.equal(Hs, H) and torch.equal(Wc, W) and torch.equal(Hc, H)


def test_wide_one_pass_w_update_at_batch_scale(hip):
    """K <= 128 with at least 256 workgroups of 32 atoms (64 files at K = 128): the W update runs as nmf_update_w_onepass_kernel<32, 1> -- two
    18-row register sets per thread (190 VGPRs, 2 waves per SIMD, no spill: -Rpass-analysis=kernel-resource-usage), norms and column sums summed in
    another order than the 16-atom form of smaller batches.  Against the oracle on the first and the last file, and -- lab build, key 20 --
    against the 16-atom form to round-off."""
    lib = hip.lib()
    from gcc_nmf_amd.engine import klnmf_initial_factors
    B, F, T, K, iters = 64, 513, 40, 128, 6
    N = 2 * T
    rng = np.random.RandomState(11)
    V = (np.abs(rng.standard_normal((B, F, N))) + 0.01).astype(np.float32)
    W0, H0 = klnmf_initial_factors(F, N, K)
    W, H = _klnmf_run(lib, V, W0, H0, F, N, K, B, iters)
    for b in (0, B - 1):
        Wr, Hr = O.performKLNMF(V[b], K, iters, 0)
        assert rel(W[b, :F, :K].cpu().numpy(), Wr) < 1e-4 and rel(H[b, :K, :N].cpu().numpy(), Hr) < 1e-4
    assert torch.allclose(torch.linalg.norm(W[:, :F, :K], dim=1), torch.ones_like(W[:, 0, :K]), atol=1e-5)
    if lib.gccnmf_set_tuning(20, 0) == 0:                      # experiment build: the 16-atom form at the same batch
        try:
            W16, H16 = _klnmf_run(lib, V, W0, H0, F, N, K, B, iters)
        finally:
            lib.gccnmf_set_tuning(20, 1)
        assert rel(W.cpu().numpy(), W16.cpu().numpy()) < 1e-5 and rel(H.cpu().numpy(), H16.cpu().numpy()) < 1e-5


@pytest.mark.parametrize('B,T,K,iters,alpha', [(64, 622, 128, 5, 0.0), (40, 311, 100, 6, 0.2), (24, 330, 64, 6, 0.0), (27, 100, 32, 4, 0.0)])
def test_short_dictionary_chained_call(hip, B, T, K, iters, alpha):
    """K <= 128 (the reference driver's K = 128: runGCCNMF.py:41): the three launches of an iteration -- K1 + K2 on column tiles, K3 + K4a on bin
    slabs, the one-pass W update -- of EVERY iteration as ONE chained launch whose stages hand over per file (csrc/direct.hip:
    gccnmf_short_chain_kernel; tuning key 21).  At 64 files the plain call runs the same three item programs: bit for bit the same factors.
    At other batch sizes the plain call's launch forms follow the batch size: equal to summation order, and to the oracle."""
    lib = hip.lib()
    from gcc_nmf_amd.engine import klnmf_initial_factors
    F, N = 513, 2 * T
    rng = np.random.RandomState(B + K)
    V = (np.abs(rng.standard_normal((B, F, N))) + 0.01).astype(np.float32)
    W0, H0 = klnmf_initial_factors(F, N, K)

    def run(chain):
        from gcc_nmf_amd.engine import Geometry, padded
        g = Geometry(F, T, K)
        dV = padded(V, (B, g.Fp, g.Np), 'cuda')
        dW = padded(np.repeat(W0[None], B, 0), (B, g.Fp, g.Kp), 'cuda')
        dH = padded(np.repeat(H0[None], B, 0), (B, g.Kp, g.Np), 'cuda')
        ws = torch.zeros(lib.gccnmf_klnmf_workspace_floats(F, N, K, B), dtype=torch.float32, device='cuda')
        assert lib.gccnmf_set_tuning(21, chain) == 0
        assert bool(lib.gccnmf_klnmf_plan(F, N, K, B, 0) & 8) == bool(chain)
        assert lib.gccnmf_klnmf(dV.data_ptr(), dW.data_ptr(), dH.data_ptr(), ws.data_ptr(), F, N, K, B, iters, alpha, 1e-16, 0, stream()) == 0
        torch.cuda.synchronize()
        return dW, dH
    try:
        W, H = run(0)
        Wc, Hc = run(8)
        Wc2, Hc2 = run(8)
    finally:
        lib.gccnmf_set_tuning(21, 1)
    assert torch.isfinite(Wc).all() and torch.isfinite(Hc).all()
    assert torch.equal(Wc, Wc2) and torch.equal(Hc, Hc2)
    assert not Wc[:, F:].any() and not Wc[:, :, K:].any() and not Hc[:, K:].any() and not Hc[:, :, N:].any()      # the padding stays exactly zero
    if B == 64:
        assert lib.gccnmf_klnmf_plan(F, N, K, B, 0) & 6 == 6
        assert torch.equal(Wc, W) and torch.equal(Hc, H)
    else:
        assert rel(Wc.cpu().numpy(), W.cpu().numpy()) < 1e-5 and rel(Hc.cpu().numpy(), H.cpu().numpy()) < 1e-5
    for b in (0, B - 1):
        Wr, Hr = O.performKLNMF(V[b], K, iters, alpha)
        assert rel(Wc[b, :F, :K].cpu().numpy(), Wr) < 1e-4 and rel(Hc[b, :K, :N].cpu().numpy(), Hr) < 1e-4


def test_short_dictionary_chained_call_with_one_workgroup_per_cu(hip):
    lib = hip.lib()
    if lib.gccnmf_set_tuning(22, 1) != 0:
        pytest.skip('key 22 (chained launches with one workgroup per CU) exists in experiment builds only: make EXPERIMENTS=1')
    from gcc_nmf_amd.engine import klnmf_initial_factors
    B, T, K, iters = 32, 311, 128, 4
    F, N = 513, 2 * T
    rng = np.random.RandomState(9)
    V = (np.abs(rng.standard_normal((B, F, N))) + 0.01).astype(np.float32)
    W0, H0 = klnmf_initial_factors(F, N, K)
    try:
        assert lib.gccnmf_set_tuning(21, 8) == 0
        Ws, Hs = _klnmf_run(lib, V, W0, H0, F, N, K, B, iters)
        assert lib.gccnmf_set_tuning(22, 0) == 0
        Wc, Hc = _klnmf_run(lib, V, W0, H0, F, N, K, B, iters)
    finally:
        lib.gccnmf_set_tuning(22, 0)
        lib.gccnmf_set_tuning(21, 1)
    assert torch.isfinite(Ws).all() and torch.equal(Ws, Wc) and torch.equal(Hs, Hc)


def test_a_long_call_is_several_chained_launches(hip):
    """A call of more iterations than one chained launch takes (tuning key 24, default 2048) continues in further launches on the same counters:
    with 3 iterations per launch an 8-iteration call is 3 + 3 + 2 -- bit for bit the one-launch call and the plain one, K > 128 and K <= 128."""
    lib = hip.lib()
    from gcc_nmf_amd.engine import klnmf_initial_factors
    for B, T, K in [(16, 622, 1024), (24, 100, 64)]:
        F, N, iters = 513, 2 * T, 8
        rng = np.random.RandomState(K)
        V = (np.abs(rng.standard_normal((B, F, N))) + 0.01).astype(np.float32)
        W0, H0 = klnmf_initial_factors(F, N, K)
        try:
            assert lib.gccnmf_set_tuning(21, 8) == 0
            assert lib.gccnmf_klnmf_plan(F, N, K, B, 0) & 8
            W1, H1 = _klnmf_run(lib, V, W0, H0, F, N, K, B, iters)
            assert lib.gccnmf_set_tuning(24, 3) == 0
            W3, H3 = _klnmf_run(lib, V, W0, H0, F, N, K, B, iters)
            assert lib.gccnmf_set_tuning(24, 1) == 0
            Wq, Hq = _klnmf_run(lib, V, W0, H0, F, N, K, B, iters)
        finally:
            lib.gccnmf_set_tuning(21, 1)
            lib.gccnmf_set_tuning(24, 2048)
        assert torch.isfinite(W1).all()
        assert torch.equal(W3, W1) and torch.equal(H3, H1) and torch.equal(Wq, W1) and torch.equal(Hq, H1)
