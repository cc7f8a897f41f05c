"""CPU checks of the real-time oracle (oracle/rt_oracle.py) against an independent brute-force evaluation of the
reference's formulas (gccNMF/realtime/gccNMFProcessor.py:253-269) and of the overlap-add bookkeeping
(gccNMF/realtime/utils.py:99-116)."""
import numpy as np

from oracle import rt_oracle as R


def test_history_ring():
    h = R.CircularHistory((2, 5))
    for i in range(7):
        h.set(np.array([[i], [10 + i]], float))
    u = h.getUnraveledArray()
    assert u.shape == (2, 5) and u[0].tolist() == [2, 3, 4, 5, 6] and u[1].tolist() == [12, 13, 14, 15, 16]
    h.set(np.array([[7, 8, 9], [17, 18, 19]], float))        # a multi-column append that wraps
    assert h.getUnraveledArray()[0].tolist() == [5, 6, 7, 8, 9]


def test_process_frames_against_brute_force():
    rng = np.random.RandomState(0)
    ws, K, D, Tc = 64, 7, 9, 3
    F = ws // 2 + 1
    W = rng.rand(F, K).astype(np.float32) + 0.05
    p = R.GCCNMFProcessorOracle(16000, ws, Tc, W, 0.1, D, localizationEnabled=True, localizationWindowSize=2, numTDOAHistory=4)
    p.setTargetTDOARange(3.0, 2.0, 2.0, 0.1)
    frames = rng.standard_normal((2, ws, Tc)).astype(np.float32)
    out, im = p.processFrames(frames, return_intermediates=True)
    win = np.sqrt(np.hamming(ws))
    f = np.linspace(0, 8000, F)
    tau = np.linspace(-0.1 / 340.29, 0.1 / 340.29, D)
    for t in range(Tc):
        X = np.array([np.fft.rfft(frames[c, :, t] * win) for c in range(2)])
        C = X[0] * np.conj(X[1]) / np.abs(X[0]) / np.abs(X[1])
        G = np.real(C[:, None] * np.exp(-2j * np.pi * np.outer(f, tau)))           # (F, D)
        S = G.T @ W                                                                 # (D, K)
        am = S.argmax(axis=0)
        assert np.array_equal(am, im['argmaxTDOA'][:, t])
        hm = np.exp(-(np.abs(am - 3.0) / 2.0) ** 2.0) / 1.1 + 0.1
        assert np.allclose(hm, im['HMask'][:, t], atol=1e-6)
        tf = (W @ hm) / W.sum(1)
        assert np.allclose(tf, im['tfMask'][:, t], atol=1e-5)
        y = np.array([np.fft.irfft(tf * X[c]) * win for c in range(2)])
        assert np.allclose(y, out[:, :, t], atol=1e-5)
        assert np.allclose(G.mean(0), im['gccPHAT'][:, t], atol=1e-5)
    # localisation: argmax of the mean of the last 2 gccPHAT columns becomes the NEXT target
    assert im['targetTDOAIndex'] == float(np.argmax(im['gccPHAT'][:, -2:].mean(1)))
    # boxcar mode
    p.targetMode = R.TARGET_MODE_BOXCAR
    p.setTargetTDOARange(4.0, 2.0, 1.0, 0.0)
    _, im2 = p.processFrames(frames, return_intermediates=True)
    assert np.array_equal(im2['HMask'], (np.abs(im2['argmaxTDOA'] - 4.0) < 2.0).astype(float))


def test_overlap_add_bookkeeping():
    ws, hop, B = 64, 32, 32
    ola = R.OverlapAddOracle(2, ws, hop, B, B // hop)
    rng = np.random.RandomState(1)
    x = rng.standard_normal((2, 20 * B)).astype(np.float32)
    out = np.concatenate([ola.processFrames(x[:, b * B:(b + 1) * B], lambda w: w.copy()) for b in range(20)], axis=1)
    # identity processing, 50 % overlap: every sample is covered by two windows and comes out two blocks late
    assert np.allclose(out[:, 4 * B:], 2 * x[:, 2 * B:-2 * B], atol=1e-6)
    # several windows per block
    ola = R.OverlapAddOracle(2, ws, 16, B, 2)
    out = np.concatenate([ola.processFrames(x[:, b * B:(b + 1) * B], lambda w: w.copy()) for b in range(20)], axis=1)
    assert np.allclose(out[:, 4 * B:], 4 * x[:, 2 * B:-2 * B], atol=1e-5)


def test_silence_does_not_produce_nan_audio():
    ws, K, D = 64, 5, 8
    W = np.random.RandomState(2).rand(ws // 2 + 1, K).astype(np.float32) + 0.1
    p = R.GCCNMFProcessorOracle(16000, ws, 1, W, 0.1, D)
    with np.errstate(all='ignore'):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            out, im = p.processFrames(np.zeros((2, ws, 1), np.float32), return_intermediates=True)
    assert np.isfinite(out).all() and not out.any()
    assert np.isnan(im['gccPHAT']).all() and (im['argmaxTDOA'] == 0).all()


def test_asymmetric_windows_reconstruct_at_low_latency():
    """(parity unpinned extension) analysis * synthesis = periodic Hann on the last 2M samples, zero before; overlap-add at hop M is the
    identity, complete ONE block after the newest sample; the product's helper builds the same pair."""
    K, S = 512, 128
    a, sy = R.asymmetric_windows(K, S)
    M = S // 2
    prod = a.astype(np.float64) * sy
    assert not prod[:K - S].any()
    assert np.allclose(prod[K - S:], 0.5 * (1 - np.cos(2 * np.pi * np.arange(S) / S)), atol=1e-6)
    assert np.allclose(prod[K - S:K - M] + prod[K - M:], 1.0, atol=1e-6)                  # COLA at hop M
    assert a.max() <= 1.0 + 1e-6 and a[K - M - 1] > 0.99 and np.isfinite(sy).all()
    from gcc_nmf_amd.realtime import asymmetricWindows
    pa, ps = asymmetricWindows(K, S)
    assert np.array_equal(pa, a) and np.array_equal(ps, sy)
    ola = R.OverlapAddOracle(2, K, M, M, 1, outputDelayBlocks=1)
    rng = np.random.RandomState(3)
    x = rng.standard_normal((2, 40 * M)).astype(np.float32)
    win = (a * sy)[None, :, None]
    y = np.concatenate([ola.processFrames(x[:, b * M:(b + 1) * M], lambda w: w * win) for b in range(40)], axis=1)
    assert np.allclose(y[:, 9 * M:], x[:, 8 * M:-M], atol=1e-5)                          # input, one block late


def test_coefficient_inference_generalises_the_reference_mask():
    """(parity unpinned extension) numHUpdates = 0 is the reference's mask; every H update lowers the KL divergence D(|X| || W h)."""
    rng = np.random.RandomState(5)
    ws, K, D, Tc = 128, 24, 16, 3
    W = R.make_rt_dictionary(9, ws // 2 + 1, K)
    frames = rng.standard_normal((2, ws, Tc)).astype(np.float32)
    outs = []
    for n in (0, 1, 4):
        p = R.GCCNMFProcessorOracle(16000, ws, Tc, W, 0.1, D, localizationEnabled=False, numHUpdates=n)
        p.setTargetTDOARange(5.0, 3.0, 2.0, 0.05)
        y, im = p.processFrames(frames, return_intermediates=True)
        outs.append((y, im))
    assert outs[0][1]['tfMask'].shape == (ws // 2 + 1, Tc) and outs[1][1]['tfMask'].shape == (2, ws // 2 + 1, Tc)
    assert np.all(outs[2][1]['tfMask'] >= 0) and np.all(outs[2][1]['tfMask'] <= outs[2][1]['HMask'].max() + 1e-9)    # a convex mix of HMask
    X = outs[0][1]['X']
    W64 = W.astype(np.float64)

    def kl(h, v):
        wh = W64 @ h
        return float(np.sum(v * np.log(v / wh) - v + wh))
    v = np.abs(X[0]).astype(np.float64)
    h = np.ones((K, Tc))
    prev = kl(h, v)
    for _ in range(4):
        h *= (W64.T @ (v / (W64 @ h))) / W64.sum(0)[:, None]
        cur = kl(h, v)
        assert cur < prev
        prev = cur
