"""Generate tests/golden/* by running the UNMODIFIED reference (seanwood/gcc-nmf).

Runs only inside the build container, where the reference checkout is mounted
read-only at /root/reference (it does not exist on the GPU box).  The reference
has no golden vectors of its own, so these fixtures --
outputs of the reference's own functions on the reference's own ``data/*.wav``
and on seeded synthetic inputs -- are what pins ``oracle/gccnmf_oracle.py``.

    python oracle/make_golden.py            # rewrites tests/golden/

Versions used are recorded in tests/golden/MANIFEST.json.
"""
import hashlib
import json
import os
import shutil
import sys
import time

import numpy as np

REF_ROOT = os.environ.get('GCCNMF_REFERENCE_ROOT', '/root/reference')
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, 'tests', 'golden')

sys.path.insert(0, REF_ROOT)
sys.path.insert(0, REPO)
import gccNMF.gccNMFFunctions as R            # noqa: E402  (the reference, unmodified)
import gccNMF.librosaSTFT as RS               # noqa: E402
from oracle import gccnmf_oracle as O         # noqa: E402  (only for the synthetic-signal recipe)

WAVS = ['dev1_female3_liverec_130ms_1m', 'dev_A_1_2_3_4', 'dev_B_1_8_9_16',
        'dev_C_2_7_10_15', 'dev_D_13_14_15_16', 'dev_Sq1_Co_A']


def reference_pipeline(stereoSamples, sampleRate, ws, hop, D, d, S, K, iters, alpha=0):
    """gccNMF/runGCCNMF.py:36-52, called function by function on the reference module."""
    X = R.computeComplexMixtureSpectrogram(stereoSamples, ws, hop, np.hanning)
    numChannels, F, T = X.shape
    freqs = np.linspace(0, sampleRate / 2.0, F)
    V = np.concatenate(np.abs(X), axis=-1)
    W, H = R.performKLNMF(V, dictionarySize=K, numIterations=iters, sparsityAlpha=alpha)
    stereoH = np.array(np.hsplit(H, numChannels))
    C = X[0] * X[1].conj() / np.abs(X[0]) / np.abs(X[1])
    A = R.getAngularSpectrogram(C, freqs, d, D)
    meanA = np.mean(A, axis=-1)
    idx = R.estimateTargetTDOAIndexesFromAngularSpectrum(meanA, d, D, S)
    G = R.getTargetTDOAGCCNMFs(C, d, D, freqs, idx, W, stereoH)
    M = R.getTargetCoefficientMasks(G, S)
    Sp = R.getTargetSpectrogramEstimates(M, X, W, stereoH)
    y = R.getTargetSignalEstimates(Sp, ws, hop, np.hanning)
    return dict(X=X, V=V, W=W, H=H, C=C, A=A, meanA=meanA, idx=np.array([int(i) for i in idx]), G=G, M=M, S=Sp, y=y)


def near_ties(G, below=1e-2):
    """Sparse record of the reference's near-tied coefficient assignments: flat (k, t) positions whose two best target scores
    differ by less than `below` relative to the best, and that relative gap.  A position that is not listed is a
    well-separated arg-max which every implementation must reproduce exactly."""
    srt = np.sort(np.asarray(G, np.float64), axis=0)
    gap = (srt[-1] - srt[-2]) / np.maximum(np.abs(srt[-1]), 1e-300)
    pos = np.flatnonzero(gap.ravel() < below)
    return dict(tie_pos=pos.astype(np.int32), tie_gap=gap.ravel()[pos].astype(np.float32), tie_below=np.float64(below))


def save(name, **arrays):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrays)
    print('wrote %s (%.1f KB)' % (path, os.path.getsize(path) / 1024.0))


def main():
    os.makedirs(os.path.join(OUT, 'data'), exist_ok=True)
    manifest = {'numpy': np.__version__, 'scipy': __import__('scipy').__version__,
                'reference_root': REF_ROOT, 'generated_unix': int(time.time()), 'wav_sha1': {}}

    # ---- fixtures: the reference's own stereo mixtures ------------------------------------
    for w in WAVS:
        src = os.path.join(REF_ROOT, 'data', w + '_mix.wav')
        dst = os.path.join(OUT, 'data', w + '_mix.wav')
        shutil.copyfile(src, dst)
        manifest['wav_sha1'][w] = hashlib.sha1(open(dst, 'rb').read()).hexdigest()

    ws, D, d, S = 1024, 128, 1.0, 3

    # ---- config 1 / 2: dev1, hop 256, K = 128 and K = 1024, full stage dump ---------------
    x, sr = R.loadMixtureSignal(os.path.join(REF_ROOT, 'data', WAVS[0] + '_mix.wav'))
    for K in (128, 1024):
        t0 = time.time()
        r = reference_pipeline(x, sr, ws, 256, D, d, S, K, 100)
        print('dev1 K=%d: %.1f s, idx=%s' % (K, time.time() - t0, r['idx']))
        sub = 1 if K == 128 else 8
        save('dev1_hop256_K%d' % K,
             idx=r['idx'], meanA=r['meanA'], A_sub=r['A'][:, ::16],
             X_sub=r['X'][:, ::4, ::7], V_sum=np.float64(r['V'].astype(np.float64).sum()),
             C_sub=r['C'][::4, ::7],
             W_sub=r['W'][:, ::sub], H_sub=r['H'][::sub, :], sub=np.int64(sub),
             G_sub=r['G'][:, ::sub, ::3],
             argmax=np.argmax(r['M'], axis=0).astype(np.uint8),
             mask_counts=r['M'].sum(axis=(1, 2)).astype(np.int64),
             S_sub=r['S'][:, :, ::8, ::5],
             y=r['y'].astype(np.float32), **near_ties(r['G']))

    # ---- the other five mixtures + dev1 at the reference's default hop 128: summaries -----
    # (SURVEY section 4 asks for K in {128, 1024} on all six)
    for K in (128, 1024):
        for w, hop in [(WAVS[0], 128)] + [(w, 256) for w in WAVS[1:]]:
            x, sr = R.loadMixtureSignal(os.path.join(REF_ROOT, 'data', w + '_mix.wav'))
            t0 = time.time()
            r = reference_pipeline(x, sr, ws, hop, D, d, S, K, 100)
            print('%s hop %d K=%d: idx=%s (%.1f s)' % (w, hop, K, r['idx'], time.time() - t0))
            save('%s_hop%d_K%d' % (w, hop, K), idx=r['idx'], meanA=r['meanA'],
                 argmax=np.argmax(r['M'], axis=0).astype(np.uint8),
                 mask_counts=r['M'].sum(axis=(1, 2)).astype(np.int64),
                 y_sub=r['y'][:, :, ::8].astype(np.float32),
                 y_rms=np.float64(np.sqrt(np.mean(r['y'].astype(np.float64) ** 2))), **near_ties(r['G']))

    # ---- synthetic mixture i = 0 (SURVEY 8d recipe), K = 128 ------------------------------
    xs = O.synthetic_mixture(0)
    r = reference_pipeline(xs, 16000, ws, 256, D, d, S, 128, 100)
    print('synthetic 0: idx=%s' % r['idx'])
    save('synthetic0_hop256_K128', x_head=xs[:, :4096], x_sum=np.float64(xs.astype(np.float64).sum()),
         idx=r['idx'], meanA=r['meanA'], argmax=np.argmax(r['M'], axis=0).astype(np.uint8),
         y_sub=r['y'][:, :, ::8].astype(np.float32), **near_ties(r['G']))

    # ---- small known-answer cases for each primitive --------------------------------------
    rng = np.random.RandomState(1234)
    sig = (rng.standard_normal(5000) * 0.1).astype(np.float32)
    kat = {'stft_sig': sig}
    for n_fft, hop in [(1024, 256), (1024, 128), (512, 64), (256, 100)]:
        kat['stft_%d_%d' % (n_fft, hop)] = RS.stft(sig.copy(), n_fft, hop, n_fft, np.hanning, center=False)
    spec = (rng.standard_normal((513, 9)) + 1j * rng.standard_normal((513, 9))).astype(np.complex64)
    kat['istft_spec'] = spec
    kat['istft_1024_256'] = RS.istft(spec, 256, 1024, np.hanning)
    kat['istft_1024_128'] = RS.istft(spec, 128, 1024, np.hanning)
    spec2 = (rng.standard_normal((129, 20)) + 1j * rng.standard_normal((129, 20))).astype(np.complex64)
    kat['istft_spec2'] = spec2
    kat['istft_256_64'] = RS.istft(spec2, 64, 256, np.hanning)
    # KL-NMF on small shapes, with and without sparsity
    Vs = np.abs(rng.standard_normal((33, 50))).astype(np.float32) + 0.01
    kat['nmf_V'] = Vs
    for name, K, it, alpha in [('a', 8, 10, 0), ('b', 5, 25, 0.5), ('c', 16, 3, 0)]:
        W, H = R.performKLNMF(Vs, K, it, alpha)
        kat['nmf_%s_W' % name] = W
        kat['nmf_%s_H' % name] = H
        kat['nmf_%s_params' % name] = np.array([K, it, alpha], dtype=np.float64)
    W, H = R.performKLNMF(Vs, 4, 2, 0, seedValue=7)
    kat['nmf_seed7_W'] = W
    kat['nmf_seed7_H'] = H
    # angular spectrum peak picking on a hand-made spectrum
    spectrum = np.array([0, 1, 0, 3, 3, 2, 5, 1, 4, 0, 2, 9], dtype=np.float64)
    kat['peaks_spectrum'] = spectrum
    kat['peaks_top2'] = np.array(R.estimateTargetTDOAIndexesFromAngularSpectrum(spectrum, 1.0, len(spectrum), 2))
    kat['peaks_top3'] = np.array(R.estimateTargetTDOAIndexesFromAngularSpectrum(spectrum, 1.0, len(spectrum), 3))
    # coefficient masks with ties and NaNs
    g = rng.standard_normal((3, 6, 7)).astype(np.float32)
    g[1, 2, 3] = g[0, 2, 3]            # tie -> first index
    g[2, 1, 1] = np.nan                # NaN ignored by nanargmax
    kat['masks_in'] = g
    kat['masks_out'] = R.getTargetCoefficientMasks(g, 3)
    # pcm conversions
    pcm = np.array([-32768, -1, 0, 1, 32767], dtype=np.int16)
    import gccNMF.wavfile as RW
    kat['pcm_in'] = pcm
    kat['pcm2float'] = RW.pcm2float(pcm)
    fl = np.array([-1.5, -1.0, -0.5, 0.0, 0.3, 0.99997, 1.0, 2.0], dtype=np.float32)
    kat['float_in'] = fl
    kat['float2pcm'] = RW.float2pcm(fl)
    kat['tdoas_128'] = R.getTDOAsInSeconds(1.0, 128)
    kat['freqs_513'] = R.getFrequenciesInHz(16000, 513)
    save('kat_primitives', **kat)

    with open(os.path.join(OUT, 'MANIFEST.json'), 'w') as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
