"""CPU-only checks of the host side: the C-ABI library loads and exports exactly what include/gccnmf_hip.h
declares, geometry helpers, wav conventions, the drop-in module's exported names, and that the product
fails loudly (never falls back to the CPU) when no device is present."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import REPO, golden

HEADER = os.path.join(REPO, 'include', 'gccnmf_hip.h')


def declared_functions(experiments=False):
    """Entry points the header declares: for the product build, or (experiments=True) with the `#ifdef GCCNMF_EXPERIMENTS` blocks."""
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    if not experiments:
        src = re.sub(r'#ifdef GCCNMF_EXPERIMENTS.*?#endif', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(?:int|long|gccnmf_allreduce_fn)\s+(gccnmf_\w+)\s*\(', src)))


@pytest.mark.parametrize('flavor', ['product', 'experiments'])
def test_library_exports_every_declared_symbol(flavor):
    """Header, ctypes table and shared object agree entry point by entry point -- for the product library, and for the lab build
    (make EXPERIMENTS=1) with the header's GCCNMF_EXPERIMENTS blocks when that library has been built."""
    from gcc_nmf_amd import _hip
    path = _hip.LIB_PATH if flavor == 'product' else os.path.join(os.path.dirname(_hip.LIB_PATH), 'libgccnmf_hip_exp.so')
    if flavor == 'experiments' and not os.path.exists(path):
        pytest.skip('libgccnmf_hip_exp.so has not been built (make -C gcc_nmf_amd/csrc EXPERIMENTS=1)')
    names = declared_functions(experiments=(flavor == 'experiments'))
    product = declared_functions()
    assert 40 <= len(product) <= 46, 'the product header is meant to shrink, not grow (round 6 added four: gccnmf_magnitude, gccnmf_klnmf_ragged + its workspace size, gccnmf_klnmf_chain_status)'
    handle = ctypes.CDLL(path)
    for n in names:
        assert hasattr(handle, n), n
    assert sorted(_hip.SIGNATURES) == product, 'ctypes table and header disagree'
    assert sorted(set(_hip.SIGNATURES) | set(_hip.EXPERIMENT_SIGNATURES)) == declared_functions(experiments=True)
    if flavor == 'product':
        for n in _hip.EXPERIMENT_SIGNATURES:
            assert not hasattr(handle, n), 'the product library must not carry %s' % n
        assert handle.gccnmf_set_tuning(18, 0) != 0 and handle.gccnmf_set_tuning(1, 0) != 0          # experiment keys are rejected
    else:
        assert handle.gccnmf_set_tuning(18, 0) == 0
    assert _hip.lib().gccnmf_version() >= 100


def test_pitches_and_workspace_sizes():
    from gcc_nmf_amd import _hip
    lib = _hip.lib()
    v = [ctypes.c_int() for _ in range(4)]
    assert lib.gccnmf_pitches(513, 622, 1024, *[ctypes.byref(x) for x in v]) == 0
    assert [x.value for x in v] == [528, 1024, 1280, 640]
    assert lib.gccnmf_pitches(513, 622, 128, *[ctypes.byref(x) for x in v]) == 0
    assert [x.value for x in v] == [528, 128, 1280, 640]
    assert lib.gccnmf_pitches(1, 622, 128, *[ctypes.byref(x) for x in v]) == 1           # GCCNMF_ERR_ARG
    base = 528 * 1280 + 528 * 1024 + 3 * 1024                      # R, U, three K-vectors per file
    direct = 1024 * 528 + 1280 * 1024 + 1280 * 528                 # Wt, Ht, Rt: the transposed copies of the direct path, per file
    chain = lambda batch: batch * (2 * 20 + 2) + 32                # ready counters of the chained launches (K1 -> K2, K2 -> K3 per column tile; K3 -> K4, K4 -> K1 per file) + error flag
    assert lib.gccnmf_klnmf_workspace_floats(513, 1244, 1024, 2) == 2 * (base + direct) + chain(2)
    assert lib.gccnmf_klnmf_workspace_floats(513, 1244, 1024, 64) == 64 * base + chain(64)               # (a handful of files at most)
    assert lib.gccnmf_klnmf_workspace_floats(513, 1244, 1024, 1) == base + 4 * (528 * 1280 + 1024) + direct + chain(1)      # + the split-K partials of one file alone
    assert lib.gccnmf_klnmf_workspace_floats(513, 0, 1024, 1) == -1
    assert lib.gccnmf_klnmf_workspace_floats(513, 1244, 128, 64) == 64 * (528 * 1280 + 528 * 128 + 3 * 128) + chain(64)     # (the fused short-dictionary launches need no scratch)
    # argument checking happens before any HIP call, so it is testable without a GPU
    assert lib.gccnmf_klnmf(0, 0, 0, 0, 513, 1244, 1024, 1, 1, 0.0, 1e-16, 0, 0) == 1
    assert lib.gccnmf_stft_stereo(0, 0, 0, 1000, 256, 1, 1, 0, 0, 0, 0, 0, 0) == 1
    assert lib.gccnmf_istft_ola(0, 3, 1024, 256, 4, 1, 0, 0, 1.0, 1, 0, 0, 0) == 1


def test_shared_run_argument_checks_and_workspace_sizes():
    """gccnmf_klnmf_shared_run / the shard descriptor: sizes and argument checking happen before any HIP call."""
    from gcc_nmf_amd import _hip
    lib = _hip.lib()
    Fp, Kp = 528, 1024
    # column blocks of one [Fp][ld] matrix: R has the matrix's layout, Upart / rowsum_part are per block
    assert lib.gccnmf_klnmf_shared_shard_workspace_floats(513, 640, 1024, 31, 20032) == Fp * 20032 + 31 * (Fp * Kp + Kp)
    # whole padded files back to back
    assert lib.gccnmf_klnmf_shared_shard_workspace_floats(513, 1244, 1024, 64, 0) == 64 * (Fp * 1280 + Fp * Kp + Kp)
    # ONE file in the plain layout also carries the scratch of the latency path: the split-K partials (4 x max(Fp*Np, Fp*Kp) + 4 x Kp)
    # and the transposed copies Wt | Ht | Rt of the direct kernels
    one = Fp * 1280 + Fp * Kp + Kp
    lat = 4 * (Fp * 1280 + Kp) + Kp * Fp + 1280 * Kp + 1280 * Fp
    assert lib.gccnmf_klnmf_shared_shard_workspace_floats(513, 1244, 1024, 1, 0) == one + lat
    assert lib.gccnmf_klnmf_shared_shard_workspace_floats(513, 1244, 1024, 1, 1280) == one + lat
    assert lib.gccnmf_klnmf_shared_workspace_floats(513, 1244, 1024, 1) == one + lat + 2 * Kp
    assert lib.gccnmf_klnmf_shared_partial_floats(513, 1024) == Fp * Kp + Kp
    for bad in [(513, 100, 1024, 2, 640), (513, 640, 1024, 3, 1280), (513, 640, 1024, 1, 642), (513, 0, 1024, 1, 0)]:
        assert lib.gccnmf_klnmf_shared_shard_workspace_floats(*bad) == -1      # ragged block in a batch / blocks beyond ld / pitch % 4 / N = 0
    arr = (_hip.SharedShard * 1)()
    assert lib.gccnmf_klnmf_shared_run(arr, 1, 8, 8, 8, 513, 1024, 1, 0.0, 1e-16, None, None, None) == 1     # null shard pointers
    assert lib.gccnmf_klnmf_shared_run(None, 9, 8, 8, 8, 513, 1024, 1, 0.0, 1e-16, None, None, None) == 1    # > GCCNMF_MAX_SHARDS
    assert lib.gccnmf_klnmf_shared_run(None, 0, 0, 8, 8, 513, 1024, 1, 0.0, 1e-16, None, None, None) == 1    # no W
    assert lib.gccnmf_set_tuning(8, 5) == 1 and lib.gccnmf_set_tuning(9, 4) == 1 and lib.gccnmf_set_tuning(7, 3) == 1
    assert lib.gccnmf_set_tuning(18, 2) == 1 and lib.gccnmf_set_tuning(19, 2) == 1
    assert lib.gccnmf_set_tuning(10, 2) == 1 and lib.gccnmf_set_tuning(11, 9) == 1 and lib.gccnmf_set_tuning(12, 17) == 1
    assert lib.gccnmf_set_tuning(16, 3) == 1 and lib.gccnmf_set_tuning(17, 3) == 1 and lib.gccnmf_set_tuning(17, 1) == 0 and lib.gccnmf_set_tuning(16, 1) == 0
    d = _hip.DirectGemm()
    assert lib.gccnmf_gemm_direct(None, 0, 0, None) == 1 and lib.gccnmf_gemm_direct(ctypes.byref(d), 0, 0, None) == 1        # null operands
    assert lib.gccnmf_rccl_comm_init(None, 2, 0, None) == 1 and lib.gccnmf_rccl_allreduce(None, None, 4, None) == 1
    assert lib.gccnmf_stft_dft(0, 0, 0, 1000, 250, 1, 1, 0, 0, 0, 0) == 1 and lib.gccnmf_dft_workspace_floats(1000, 0, 2) == -1


def _gemm_plan(lib, M, N, batch, xcd=1, concurrent=0, narrow=1):
    pl = (ctypes.c_int * 8)()
    n = lib.gccnmf_debug_gemm_plan(M, N, batch, xcd, concurrent, narrow, pl, None, 0)
    items = (ctypes.c_int * (6 * max(n, 1)))()
    assert lib.gccnmf_debug_gemm_plan(M, N, batch, xcd, concurrent, narrow, pl, items, n) == n
    return dict(zip(('lists', 'cw', 'cr', 'split', 'rag', 'tiles_m', 'tiles_n', 'grid'), pl)), np.array(items[:6 * n]).reshape(n, 6)


def test_throughput_tile_work_lists_cover_every_tile_exactly_once():
    """The ordered item lists of a throughput-tile launch (csrc/gemm_dma.h; gccnmf_debug_gemm_plan runs the kernel's own tile decode on
    the host): in every form -- wide tiles only, narrow ragged tiles, everything split -- each 32-column block of every (file, row tile)
    is computed by exactly one item, items of a list are ordered longest first, and a ragged last column tile (N = 1244 = 19 x 64 + 28)
    costs one narrow item instead of a padded wide one where that does not cost the launch another round of workgroup slots."""
    from gcc_nmf_amd import _hip
    lib = _hip.lib()
    try:
        for policy in (0, 1, 2):
            assert lib.gccnmf_set_tuning(9, policy) == 0
            for M, N, B in [(512, 1244, 64), (512, 1244, 26), (1024, 1244, 64), (512, 1244, 5), (300, 100, 9), (512, 1280, 32), (512, 1244, 72),
                            (512, 1244, 16), (512, 1244, 40), (512, 33, 8), (512, 32, 8), (1536, 1244, 13), (512, 660, 50), (512, 1244, 51)]:
                for concurrent in (0, 1):
                    pl, items = _gemm_plan(lib, M, N, B, 1, concurrent)
                    cover = np.zeros((B, pl['tiles_m'], 2 * pl['tiles_n']), int)
                    for lst, t, f, tm, c0, nw in items:
                        assert c0 % 32 == 0 and nw in (1, 2) and 0 <= lst < pl['lists']
                        cover[f, tm, c0 // 32: c0 // 32 + nw] += 1
                    need = np.ones_like(cover)
                    if pl['rag']:
                        need[:, :, -1] = 0                       # columns that do not exist: nobody computes them
                        assert policy != 0 and N - 64 * (pl['tiles_n'] - 1) <= 32
                    assert np.array_equal(cover, need), (policy, M, N, B)
                    for lst in range(pl['lists']):               # longest first: wide, then narrow; tickets are dense
                        mine = items[items[:, 0] == lst]
                        assert np.array_equal(mine[:, 1], np.arange(len(mine))) and np.all(np.diff(mine[:, 5]) <= 0)
                    if policy == 0:
                        assert pl['rag'] == 0 and pl['split'] == 0 and np.all(items[:, 5] == 2)
                    if policy == 2:
                        assert np.all(items[:, 5] == 1)
                    if policy == 1:
                        assert pl['split'] == 0
        assert lib.gccnmf_set_tuning(9, 1) == 0
        # the headline shape: 19 wide + 1 narrow item per file instead of 20 wide ones (152 + 8 = the 160 slots per XCD the padded tiles take)
        pl, items = _gemm_plan(lib, 512, 1244, 64)
        assert (pl['cw'], pl['cr'], pl['split'], pl['rag']) == (152, 8, 0, 1) and (items[:, 5] == 1).sum() == 64
        # 51 files: 128 padded tiles per XCD are exactly two rounds of 64 slots, 122 + 7 items would start a third -> wide tiles, unless the
        # launch shares the chip with another file group's (its early finishers are used at once)
        assert _gemm_plan(lib, 512, 1244, 51)[0]['rag'] == 0 and _gemm_plan(lib, 512, 1244, 51, concurrent=1)[0]['rag'] == 1
        # a kernel instantiation without the narrow loop never gets narrow items
        pl, items = _gemm_plan(lib, 512, 1244, 26, narrow=0)
        assert pl['rag'] == 0 and pl['split'] == 0 and np.all(items[:, 5] == 2) and len(items) == 520
    finally:
        lib.gccnmf_set_tuning(9, 1)
    assert lib.gccnmf_debug_gemm_plan(0, 5, 1, 1, 0, 1, None, None, 0) == -1


def test_chained_launch_lists_hold_whole_files():
    """The lists of a chained launch (gccnmf_debug_gemm_plan, narrow_capable bit 1): every item of a file sits in ONE list -- list
    file % 8, so that every producer and consumer of a file share one XCD's L2 in every GEMM of the iteration -- the files of a list in
    ascending order, a file's items together (wide tiles, then its ragged ones), every 32-column block computed exactly once, at any
    batch size; the longest list has ceil(batch / 8) files."""
    from gcc_nmf_amd import _hip
    lib = _hip.lib()
    for M, N, B, narrow in [(512, 1244, 64, 1), (1024, 1244, 64, 1), (512, 1024, 64, 0), (512, 1244, 26, 1), (1024, 1244, 77, 1), (512, 1280, 13, 1),
                            (512, 1244, 8, 1), (1536, 660, 50, 1)]:
        pl, items = _gemm_plan(lib, M, N, B, narrow=narrow | 2)
        assert pl['lists'] == 8 and pl['split'] == 0 and pl['cr'] == 0
        assert pl['cw'] == -(-B // 8) * pl['tiles_m'] * pl['tiles_n'] and pl['grid'] == 8 * pl['cw']
        cover = np.zeros((B, pl['tiles_m'], 2 * pl['tiles_n']), int)
        for lst, t, f, tm, c0, nw in items:
            assert f % 8 == lst and nw in (1, 2)
            cover[f, tm, c0 // 32: c0 // 32 + nw] += 1
        need = np.ones_like(cover)
        if pl['rag']:
            need[:, :, -1] = 0
            assert narrow and N - 64 * (pl['tiles_n'] - 1) <= 32
        assert np.array_equal(cover, need), (M, N, B)
        per = pl['tiles_m'] * pl['tiles_n']
        for lst in range(8):
            mine = items[items[:, 0] == lst]
            assert np.array_equal(mine[:, 1], np.arange(len(mine))) and len(mine) == per * len(range(lst, B, 8))
            assert np.array_equal(mine[:, 2], np.repeat(np.arange(lst, B, 8), per))           # whole files, ascending
            for q in range(len(mine) // per):
                assert np.all(np.diff(mine[q * per:(q + 1) * per, 5]) <= 0)                    # a file's ragged items behind its wide ones
    try:
        assert lib.gccnmf_set_tuning(9, 2) == 0
        assert lib.gccnmf_debug_gemm_plan(512, 1244, 64, 1, 0, 3, (ctypes.c_int * 8)(), None, 0) == -1       # no chained form of the all-halves test layout
    finally:
        lib.gccnmf_set_tuning(9, 1)
    assert lib.gccnmf_debug_gemm_plan(512, 1244, 5, 1, 0, 3, (ctypes.c_int * 8)(), None, 0) == -1           # fewer than 8 files: one list, no chain


def test_klnmf_plan_is_a_pure_function_of_shape_batch_and_tuning():
    """gccnmf_klnmf_plan: which launches gccnmf_klnmf will use (no device needed).  Short dictionaries at batch scale take the fused
    launches when whole rounds of 512 workgroups pay; file groups that run side by side (GCCNMF_FLAG_GROUPS) are planned together."""
    from gcc_nmf_amd import _hip
    lib = _hip.lib()
    plan = lib.gccnmf_klnmf_plan
    assert plan(513, 1244, 1024, 1, 0) == 1 and plan(513, 1244, 1024, 4, 0) == 1 and plan(513, 1244, 1024, 64, 0) == 8     # direct path: up to 4 files
    # bit 3: the whole call as one chained launch -- K >= 256 a multiple of 128, from 20 files on (whole-file lists where they balance: from 24 files;
    # else the tile list in equal eighths with agent-scope hand-over), no other file group beside it (round 6, profiles/r06y_files_sweep_*.txt)
    assert [plan(513, 1244, 1024, b, 0) for b in (16, 19, 20, 24, 25, 26, 32, 51, 52, 77, 104)] == [0, 0, 8, 8, 8, 8, 8, 8, 8, 8, 8]
    assert plan(513, 1244, 1024, 32, 4 | (2 << 8)) == 0 and plan(513, 1244, 256, 64, 0) == 8 and plan(513, 1244, 320, 64, 0) == 0 and plan(513, 1244, 384, 64, 0) == 8
    assert plan(513, 1244, 1024, 64, 1) == 0 and plan(513, 1244, 1024, 64, 2) == 0              # no XCD-affine lists / unfused W update: plain launches
    try:
        assert lib.gccnmf_set_tuning(21, 0) == 0 and plan(513, 1244, 1024, 64, 0) == 0
        assert lib.gccnmf_set_tuning(21, 4) == 0 and plan(513, 1244, 1024, 16, 0) == 8 and plan(513, 1244, 1024, 25, 4 | (2 << 8)) == 8     # forced form
        assert lib.gccnmf_set_tuning(21, 3) == 1
    finally:
        lib.gccnmf_set_tuning(21, 1)
    # (bit 3 beside them, round 6: where the plain call runs both fused launches for EVERY file -- and the batch gives every XCD three whole files
    # in balanced lists -- the three launches of every iteration run as ONE chained launch, bit for bit the plain call)
    assert plan(513, 1244, 128, 64, 0) == 6 | 8 and plan(513, 1244, 128, 25, 0) == 2 and plan(513, 1244, 128, 26, 0) == 0
    assert plan(513, 1244, 128, 96, 0) == 6 and plan(513, 1244, 128, 128, 0) == 6                # 96: a round of 64 files on the slabs, 32 behind it: no chain
    assert plan(513, 1244, 128, 56, 0) == 14 and plan(513, 1244, 128, 40, 0) == 2 and plan(513, 1244, 64, 64, 0) == 14 and plan(513, 2486, 128, 64, 0) == 14
    assert plan(513, 1244, 129, 64, 0) == 0 and plan(500, 1244, 128, 64, 0) == 0                # K > 128 / F not 64 n + 1: the batched tiles
    groups = lambda n: 4 | (n << 8)
    assert plan(513, 1244, 128, 32, groups(2)) == 6 and plan(513, 1244, 128, 16, groups(4)) == 6 and plan(513, 1244, 128, 32, 0) == 2
    assert plan(513, 0, 128, 64, 0) == -1
    try:
        assert lib.gccnmf_set_tuning(16, 0) == 0 and lib.gccnmf_set_tuning(17, 2) == 0
        assert plan(513, 1244, 128, 26, 0) == 4 and plan(513, 1244, 64, 5, 0) == 4
        assert plan(257, 500, 128, 26, 0) == 0                                                  # 4 slabs cannot hold the tail sums of 128 atoms
    finally:
        lib.gccnmf_set_tuning(16, 1)
        lib.gccnmf_set_tuning(17, 1)


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a device is present')
    from gcc_nmf_amd import HipLibraryError
    from gcc_nmf_amd import gccNMFFunctions as G
    from gcc_nmf_amd.engine import GCCNMFEngine
    V = np.ones((5, 6), np.float32)
    with pytest.raises(HipLibraryError):
        G.performKLNMF(V, 2, 1, 0)
    with pytest.raises(HipLibraryError):
        G.computeComplexMixtureSpectrogram(np.zeros((2, 4096), np.float32), 1024, 256, np.hanning)
    with pytest.raises(HipLibraryError):
        GCCNMFEngine(16000)
    import gcc_nmf_amd
    src = ''.join(open(os.path.join(os.path.dirname(gcc_nmf_amd.__file__), f)).read()
                  for f in os.listdir(os.path.dirname(gcc_nmf_amd.__file__)) if f.endswith('.py'))
    assert 'import oracle' not in src and 'from oracle' not in src, 'the product must never import the oracle'


def test_dropin_module_names():
    """Everything the reference driver pulls out of `from gccNMFFunctions import *` (runGCCNMF.py:27-54)."""
    from gcc_nmf_amd import gccNMFFunctions as G
    for name in ['getMixtureFileName', 'getSourceEstimateFileName', 'loadMixtureSignal', 'getMaxTDOA', 'getTDOAsInSeconds',
                 'getFrequenciesInHz', 'computeComplexMixtureSpectrogram', 'performKLNMF', 'getAngularSpectrogram',
                 'estimateTargetTDOAIndexesFromAngularSpectrum', 'getTargetTDOAGCCNMFs', 'getTargetCoefficientMasks',
                 'getTargetSpectrogramEstimates', 'getTargetSignalEstimates', 'saveTargetSignalEstimates',
                 'getTargetTDOAEstimates', 'SPEED_OF_SOUND_IN_METRES_PER_SECOND', 'stft', 'istft', 'wavread', 'wavwrite',
                 'hanning', 'linspace', 'float32', 'concatenate', 'array', 'hsplit', 'mean']:
        assert hasattr(G, name), name
    assert G.SPEED_OF_SOUND_IN_METRES_PER_SECOND == 340.29
    assert G.getMixtureFileName('a/b') == 'a/b_mix.wav' and G.getSourceEstimateFileName('a/b', 1) == 'a/b_sim_2.wav'
    kat = golden('kat_primitives')
    assert np.array_equal(G.getTDOAsInSeconds(1.0, 128), kat['tdoas_128'])
    assert np.array_equal(G.getFrequenciesInHz(16000, 513), kat['freqs_513'])
    import inspect
    sig = inspect.signature(G.performKLNMF)
    assert list(sig.parameters) == ['V', 'dictionarySize', 'numIterations', 'sparsityAlpha', 'epsilon', 'seedValue']
    assert sig.parameters['epsilon'].default == 1e-16 and sig.parameters['seedValue'].default == 0
    assert list(inspect.signature(G.computeComplexMixtureSpectrogram).parameters) == \
        ['stereoSamples', 'windowSize', 'hopSize', 'windowFunction', 'fftSize']
    assert list(inspect.signature(G.getTargetTDOAGCCNMFs).parameters) == \
        ['coherenceV', 'microphoneSeparationInMetres', 'numTDOAs', 'frequenciesInHz', 'targetTDOAIndexes', 'W', 'stereoH']


def test_wav_conventions(tmp_path):
    from gcc_nmf_amd import wavfile as Wf
    kat = golden('kat_primitives')
    assert np.array_equal(Wf.pcm2float(kat['pcm_in']), kat['pcm2float'])
    assert np.array_equal(Wf.float2pcm(kat['float_in']), kat['float2pcm'])
    with pytest.raises(TypeError):
        Wf.pcm2float(np.zeros(3, np.float32))
    with pytest.raises(TypeError):
        Wf.float2pcm(np.zeros(3, np.int16))
    x = (np.random.RandomState(0).rand(2, 1000).astype(np.float32) - 0.5) * 0.5
    p = str(tmp_path / 'a.wav')
    Wf.wavwrite(x, p, 16000)
    y, sr = Wf.wavread(p)
    assert sr == 16000 and y.shape == (2, 1000) and y.dtype == np.float32
    assert np.abs(y - x).max() <= 1.0 / 32768
    loud = x * 10
    Wf.wavwrite(loud, p, 16000)                       # clip protection: rescaled to 0.99 peak
    y, _ = Wf.wavread(p)
    assert abs(np.abs(y).max() - 0.99) < 1e-3
    with pytest.raises(ValueError):
        Wf.wavwrite(loud, p, 16000, clipProtection=False)
    ref, sr = Wf.wavread(os.path.join(REPO, 'tests', 'golden', 'data', 'dev1_female3_liverec_130ms_1m_mix.wav'))
    assert ref.shape == (2, 160000) and sr == 16000 and ref.dtype == np.float32


def test_constant_tables():
    from gcc_nmf_amd.engine import fft_twiddles, steering_tables, klnmf_initial_factors, num_frames
    from oracle import gccnmf_oracle as O
    tw = fft_twiddles(1024).view(np.complex64)
    assert tw.shape == (512,) and abs(tw[256] - (-1j)) < 1e-7 and tw[0] == 1
    f, tau = np.linspace(0, 8000, 513), O.getTDOAsInSeconds(1.0, 128)
    trig = steering_tables(f, tau, 528, 128)
    E = np.exp(np.outer(f, -(2j * np.pi) * tau))
    assert trig.shape == (2, 528, 128) and np.allclose(trig[0, :513], E.real, atol=1e-7) and np.allclose(trig[1, :513], -E.imag, atol=1e-7)
    assert not trig[:, 513:].any()
    W, H = klnmf_initial_factors(33, 50, 8)
    Wr, Hr = O.initKLNMF(33, 50, 8)
    assert np.array_equal(W, Wr) and np.array_equal(H, Hr) and W.dtype == np.float32
    assert num_frames(160000, 1024, 256) == 622 and num_frames(160000, 1024, 128) == 1243


def test_ordered_dictionary_matches_reference_rule():
    from gcc_nmf_amd.pretraining import getOrderedDictionary
    rng = np.random.RandomState(0)
    W = rng.rand(40, 9).astype(np.float32)
    Wo = getOrderedDictionary(W)
    cent = (np.arange(40)[:, None] * Wo).sum(0) / Wo.sum(0)
    assert Wo.shape == W.shape and np.all(np.diff(cent) >= 0)
    assert sorted(map(tuple, Wo.T.round(6))) == sorted(map(tuple, W.T.round(6)))       # a permutation of the atoms


def test_bench_contract_defaults_and_loud_failure_without_gpu(monkeypatch):
    """bench.py: no flags = 1 GPU, a handful of steps, the BASELINE workload (64 files, K = 1024, 100 iterations); without
    a device it fails loudly (HipLibraryError) instead of timing a fallback."""
    import importlib
    import sys
    import pytest
    import torch
    monkeypatch.setattr(sys, 'argv', ['bench.py'])
    bench = importlib.import_module('bench')
    a = bench.parse()
    assert (a.gpus, a.files, a.dictionary_size, a.iterations, a.hop, a.seconds, a.mode) == (1, 64, 1024, 100, 256, 10.0, 'separate')
    assert 1 <= a.steps <= 10 and a.warmup >= 1 and a.nmf_groups is None
    assert bench.F32_MFMA_PEAK_TFLOPS == 157.3
    if not torch.cuda.is_available():
        from gcc_nmf_amd._hip import HipLibraryError
        from gcc_nmf_amd.engine import GCCNMFEngine
        with pytest.raises(HipLibraryError):
            GCCNMFEngine(160000, dictionarySize=1024, batch=1)


def test_fft_size_limits_are_named():
    """Like the reference, any n_fft is taken (powers of two: radix-2 kernels; other sizes: DFT as a GEMM); the limits are 2..8192."""
    from gcc_nmf_amd import librosaSTFT as L
    for n_fft in (16384, 1, 8193):
        with pytest.raises(L.ParameterError, match='not supported'):
            L.stft(np.zeros(40000, np.float32), n_fft=n_fft, hop_length=256, window=np.hanning, center=False)
    with pytest.raises(L.ParameterError, match='not supported'):
        L.istft(np.zeros((8194, 4), np.complex64), hop_length=256, window=np.hanning)


def test_frame_is_the_reference_strided_view():
    """gccNMF/librosaSTFT.py:370-435: y_frames[i, j] == y[j * hop + i], a view, tail samples dropped, same errors."""
    from gcc_nmf_amd import librosaSTFT as L
    y = np.arange(1000, dtype=np.float32)
    f = L.frame(y, frame_length=256, hop_length=100)
    assert f.shape == (256, 1 + (1000 - 256) // 100) and f[7, 3] == y[3 * 100 + 7] and np.shares_memory(f, y)
    assert np.array_equal(f[:, -1], y[700:956])
    with pytest.raises(L.ParameterError, match='Invalid hop_length'):
        L.frame(y, 256, 0)
    with pytest.raises(L.ParameterError, match='contiguous'):
        L.frame(y[::2], 256, 100)
    with pytest.raises(L.ParameterError, match='too short'):
        L.frame(y[:100], 256, 100)
    # the DFT-as-GEMM tables of the any-n_fft path: shapes, and that they ARE the transform (float64 reference on the host)
    w, n_fft, Fp = np.hanning(10), 10, 16
    b, ib = L.dft_basis(w, n_fft, Fp), L.idft_basis(w, n_fft, Fp)
    assert b.shape == (16, 32) and ib.shape == (32, 64)
    x = np.random.RandomState(0).standard_normal(n_fft)
    X = np.conj(np.fft.rfft(w * x))
    assert np.allclose(x @ b[:n_fft, :6].astype(np.float64), X.real, atol=1e-6) and np.allclose(x @ b[:n_fft, Fp:Fp + 6].astype(np.float64), X.imag, atol=1e-6)
    full = np.concatenate([X.conj(), X[-2:0:-1]])
    want = w * np.fft.ifft(full).real
    got = X.real @ ib[:6, :n_fft].astype(np.float64) + X.imag @ ib[Fp:Fp + 6, :n_fft].astype(np.float64)
    assert np.allclose(got, want, atol=1e-6)


def test_lds_dma_statements_set_m0_themselves():
    """M0 (the LDS-DMA destination base) is a compiler-reserved register that cannot be declared as an inline-asm clobber (hipcc only
    warns).  The rule that makes that safe -- every asm statement that issues an LDS-DMA load writes M0 itself, in front of the load, and
    no other statement reads M0 -- is checked on the sources, so a later edit cannot silently break it (fragility)."""
    csrc = os.path.join(REPO, 'gcc_nmf_amd', 'csrc')
    dma = 0
    for name in sorted(os.listdir(csrc)):
        if not name.endswith(('.h', '.hip')):
            continue
        src = open(os.path.join(csrc, name)).read()
        for m in re.finditer(r'asm\s+volatile\s*\((.*?)\)\s*;', src, flags=re.S):
            text = m.group(1)
            strings = ''.join(re.findall(r'"((?:[^"\\]|\\.)*)"', text.split(':')[0]))
            if re.search(r'global_load_lds|buffer_load\w*[^"]*\blds\b', strings):
                dma += 1
                assert 's_mov_b32 m0' in strings and strings.index('s_mov_b32 m0') < re.search(r'global_load_lds|buffer_load', strings).start(), (name, strings)
            elif re.search(r'\bm0\b', strings):
                raise AssertionError('%s: an asm statement touches m0 without being an LDS-DMA load: %s' % (name, strings))
    assert dma >= 2


def test_reciprocal_forms_of_the_update_epilogues_against_the_reference_quotients():
    """only V / (W.H) is an IEEE quotient on the device; the other three divisions of gccNMFFunctions.py:76-80 are
    evaluated by the lean epilogues (csrc/gemm_dma.h) as products with a correctly rounded reciprocal, and K2 associates differently:
        H update (:76)    reference (H s) * (num / den)      device (H * num) * (s * (1 / den))      s = the pending norms of :81
        W update (:77)    reference W * (num / den)          device W * (num * (1 / den))
        normalise (:80)   reference W / norm                 device W * (1 / norm)
    The same float32 expressions evaluated here in NumPy on 4 M random operands of the magnitudes the iteration sees: the distance to the
    reference's expression is at most 1 ulp (normalise), 2 ulp (W update), 4 ulp (H update; 99.8 % within 2) -- 5e-7 relative at worst, against
    the 1e-4 bar on W and H after 100 iterations."""
    rng = np.random.default_rng(0)
    n = 4_000_000
    f = np.float32

    def ulps(a, b):
        return np.abs(a.astype(f).view(np.int32).astype(np.int64) - b.astype(f).view(np.int32).astype(np.int64))
    h = rng.random(n, dtype=f) + f(1e-3)
    num = rng.random(n, dtype=f) * f(50) + f(0.01)
    den = rng.random(n, dtype=f) * f(30) + f(0.5)
    s = rng.random(n, dtype=f) * f(3) + f(0.1)
    one = f(1)
    d_h = ulps((h * num) * (s * (one / den)), (h * s) * (num / den))
    d_w = ulps(h * (num * (one / den)), h * (num / den))
    d_n = ulps(num * (one / den), num / den)
    assert d_n.max() <= 1 and d_w.max() <= 2 and d_h.max() <= 4 and (d_h <= 2).mean() > 0.995
