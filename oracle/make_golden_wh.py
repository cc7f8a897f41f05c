"""Additional reference goldens (UNMODIFIED reference, run in the build container only): the KL-NMF factors of all six reference
mixtures at K = 1024 -- sub-sampled W[:, ::16], H[::16, ::2] -- so that the device's W / H are pinned on every wav, not on dev1 only
(tests/test_gpu_pipeline.py), plus two known-answer cases for non-power-of-two n_fft (stft / istft of the reference at 400, 1000, 1536).

    python oracle/make_golden_wh.py         # writes tests/golden/wh_sub_K1024.npz, tests/golden/kat_nfft.npz
"""
import os
import sys

import numpy as np

REF_ROOT = os.environ.get('GCCNMF_REFERENCE_ROOT', '/root/reference')
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, 'tests', 'golden')
sys.path.insert(0, REF_ROOT)
sys.path.insert(0, REPO)
import gccNMF.gccNMFFunctions as R            # noqa: E402  (the reference, unmodified)
import gccNMF.librosaSTFT as RS               # noqa: E402

WAVS = ['dev1_female3_liverec_130ms_1m', 'dev_A_1_2_3_4', 'dev_B_1_8_9_16', 'dev_C_2_7_10_15', 'dev_D_13_14_15_16', 'dev_Sq1_Co_A']


def main():
    out = {}
    for w in WAVS:
        x, sr = R.loadMixtureSignal(os.path.join(REF_ROOT, 'data', w + '_mix.wav'))
        X = R.computeComplexMixtureSpectrogram(x, 1024, 256, np.hanning)
        V = np.concatenate(np.abs(X), axis=-1)
        W, H = R.performKLNMF(V, dictionarySize=1024, numIterations=100, sparsityAlpha=0)
        out[w + '_W'] = W[:, ::16].copy()
        out[w + '_H'] = H[::16, ::2].copy()
        out[w + '_norms'] = np.array([np.linalg.norm(W), np.linalg.norm(H)])
        print(w, W.shape, H.shape, out[w + '_norms'], flush=True)
    np.savez_compressed(os.path.join(OUT, 'wh_sub_K1024.npz'), **out)

    kat = {}
    rng = np.random.RandomState(77)
    for n_fft, hop in [(400, 100), (1000, 250), (1536, 384), (1000, 300)]:
        y = (rng.standard_normal(n_fft * 6 + 123) * 0.1).astype(np.float32)
        S = RS.stft(y.copy(), n_fft=n_fft, hop_length=hop, win_length=n_fft, window=np.hanning, center=False)
        yi = RS.istft(S, hop_length=hop, win_length=n_fft, window=np.hanning)
        yc = RS.istft(S, hop_length=hop, win_length=n_fft, window=np.hanning, center=False)
        key = '%d_%d' % (n_fft, hop)
        kat.update({'y_' + key: y, 'S_' + key: np.asarray(S), 'yi_' + key: np.asarray(yi), 'yc_' + key: np.asarray(yc)})
        print('kat', key, S.shape, S.dtype, yi.shape, yi.dtype, flush=True)
    np.savez_compressed(os.path.join(OUT, 'kat_nfft.npz'), **kat)


if __name__ == '__main__':
    main()
