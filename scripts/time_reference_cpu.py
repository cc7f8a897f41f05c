#!/usr/bin/env python
"""The UNMODIFIED reference's CPU path timed on this box's host cores (SURVEY 8d "CPU baseline timing"): the functions of
gccNMF/gccNMFFunctions.py in the order of gccNMF/runGCCNMF.py:36-52, K = 1024, 100 iterations, hop 256 -- the parameters of the
GPU bench -- on bench file 0 (synthetic, SURVEY 8d recipe), per stage and end to end, best of `--reps` after one warm-up.

The reference checkout does not travel with the repository; it is staged into git-ignored scratch for ONE gpurun call and removed
again (scripts/stage_reference.sh stage|unstage), exactly as for the drop-in driver test.  Nothing in tests/, bench.py or the
product reads it.  Output: one JSON object (kept under profiles/).

    bash scripts/stage_reference.sh stage
    gpurun -- 'python scripts/time_reference_cpu.py > gpurun_out/reference_cpu.json'
    bash scripts/stage_reference.sh unstage
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--reps', type=int, default=2)
    p.add_argument('--dictionary-size', type=int, default=1024)
    p.add_argument('--iterations', type=int, default=100)
    p.add_argument('--hop', type=int, default=256)
    a = p.parse_args()
    root = os.environ.get('GCCNMF_REFERENCE_ROOT')
    for cand in (root, os.path.join(REPO, 'oracle', '_ref', 'reference_checkout'), '/root/reference'):
        if cand and os.path.isdir(os.path.join(cand, 'gccNMF')):
            root = cand
            break
    else:
        raise SystemExit('no reference checkout (bash scripts/stage_reference.sh stage)')
    sys.path.insert(0, root)
    import gccNMF.gccNMFFunctions as R                       # the reference, as it is
    from gcc_nmf_amd.synthetic import synthetic_batch
    try:
        from threadpoolctl import threadpool_info
        pools = [{k: i.get(k) for k in ('internal_api', 'num_threads', 'version')} for i in threadpool_info()]
    except Exception:
        pools = None

    sr, ws, D, d, S = 16000, 1024, 128, 1.0, 3
    x = synthetic_batch(0, 1)[0]

    def once():
        t = {}
        c = [time.perf_counter()]

        def lap(name):
            c.append(time.perf_counter())
            t[name] = c[-1] - c[-2]
        X = R.computeComplexMixtureSpectrogram(x, ws, a.hop, np.hanning)
        lap('computeComplexMixtureSpectrogram')
        nch, F, T = X.shape
        f = np.linspace(0, sr / 2.0, F)
        V = np.concatenate(np.abs(X), axis=-1)
        lap('V')
        W, H = R.performKLNMF(V, dictionarySize=a.dictionary_size, numIterations=a.iterations, sparsityAlpha=0)
        lap('performKLNMF')
        stereoH = np.array(np.hsplit(H, nch))
        C = X[0] * X[1].conj() / np.abs(X[0]) / np.abs(X[1])
        lap('spectralCoherenceV')
        A = R.getAngularSpectrogram(C, f, d, D)
        lap('getAngularSpectrogram')
        idx = R.estimateTargetTDOAIndexesFromAngularSpectrum(np.mean(A, axis=-1), d, D, S)
        lap('estimateTargetTDOAIndexesFromAngularSpectrum')
        G = R.getTargetTDOAGCCNMFs(C, d, D, f, idx, W, stereoH)
        lap('getTargetTDOAGCCNMFs')
        M = R.getTargetCoefficientMasks(G, S)
        lap('getTargetCoefficientMasks')
        Sp = R.getTargetSpectrogramEstimates(M, X, W, stereoH)
        lap('getTargetSpectrogramEstimates')
        y = R.getTargetSignalEstimates(Sp, ws, a.hop, np.hanning)
        lap('getTargetSignalEstimates')
        t['end_to_end'] = c[-1] - c[0]
        return t, T, [int(i) for i in idx], y

    once()
    runs = [once() for _ in range(a.reps)]
    best = min(runs, key=lambda r: r[0]['end_to_end'])
    t, T, idx, y = best
    # round 3: OpenBLAS scales NEGATIVELY on these matrix sizes with a 64- or 128-thread pool (the oracle port: 133 frames/s on 128
    # threads, 753 on 16), so the reference is also timed at a few smaller BLAS thread counts; the best is what a user would tune to
    sweep = {}
    try:
        from threadpoolctl import threadpool_limits
        for th in (1, 4, 8, 16, 32):
            with threadpool_limits(limits=th):
                r = min((once() for _ in range(a.reps)), key=lambda v: v[0]['end_to_end'])
            sweep[str(th)] = {'frames_per_s': T / r[0]['end_to_end'], 'nmf_only_frames_per_s': T / r[0]['performKLNMF'], 'seconds': r[0]}
    except ImportError:
        pass
    best_threads = max(sweep, key=lambda k: sweep[k]['frames_per_s']) if sweep else None
    print(json.dumps({
        'blas_thread_sweep': sweep, 'best_blas_threads': best_threads,
        'best_frames_per_s': max([T / t['end_to_end']] + [v['frames_per_s'] for v in sweep.values()]),
        'what': 'unmodified seanwood/gcc-nmf functions (gccNMF/gccNMFFunctions.py) in runGCCNMF.py order on bench file 0, host CPU only',
        'parameters': {'n_fft': ws, 'hop': a.hop, 'dictionary_size': a.dictionary_size, 'nmf_iterations': a.iterations, 'tdoas': D, 'targets': S},
        'frames': T, 'frames_per_s': T / t['end_to_end'], 'nmf_only_frames_per_s': T / t['performKLNMF'],
        'seconds': t, 'reps': a.reps, 'tdoa_indexes': idx, 'waveform_shape': list(y.shape),
        'host_cpus': os.cpu_count(), 'thread_pools': pools, 'numpy': np.__version__}))


if __name__ == '__main__':
    main()
