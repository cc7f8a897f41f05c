"""The reference's own driver, byte-for-byte unchanged, running on top of this package's module names
(SURVEY.md 8b).  Needs the reference checkout, so it runs only in the build container; the hot functions are
stubbed with the oracle here (no GPU in this container) -- what is under test is the interception mechanics:
every hot-path call of runGCCNMF.py must land in gcc_nmf_amd.gccNMFFunctions.  The same driver run with the
real HIP functions is tests/test_gpu_kernels.py::test_reference_call_sequence_dropin's call sequence."""
import os
import sys

import numpy as np
import pytest

from conftest import golden

REF = os.environ.get('GCCNMF_REFERENCE_ROOT', '/root/reference')
HOT = ['computeComplexMixtureSpectrogram', 'performKLNMF', 'getAngularSpectrogram', 'estimateTargetTDOAIndexesFromAngularSpectrum',
       'getTargetTDOAGCCNMFs', 'getTargetCoefficientMasks', 'getTargetSpectrogramEstimates', 'getTargetSignalEstimates']


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, 'gccNMF', 'runGCCNMF.py')), reason='reference checkout not present')
def test_unmodified_reference_driver_lands_in_this_package(tmp_path, monkeypatch):
    pytest.importorskip('matplotlib')
    from scipy.io import wavfile
    from gcc_nmf_amd import dropin
    import gcc_nmf_amd.gccNMFFunctions as G
    from oracle import gccnmf_oracle as O
    calls = {}

    def stub(name):
        fn = getattr(O, name)

        def wrapped(*a, **k):
            calls[name] = calls.get(name, 0) + 1
            return fn(*a, **k)
        return wrapped
    for name in HOT:
        monkeypatch.setattr(G, name, stub(name))
    saved = {k: sys.modules.get(k) for k in list(dropin._ALIASES) + ['gccNMFPlotting', 'gccNMF', 'gccNMF.gccNMFPlotting']}
    try:
        out = dropin.run_reference_driver(REF, str(tmp_path))
    finally:
        dropin.uninstall()
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    assert all(calls.get(n) == 1 for n in HOT), calls
    g = golden('dev1_female3_liverec_130ms_1m_hop128_K128')          # the driver's own defaults: hop 128, K=128, 100 iterations
    for i in range(3):
        sr, pcm = wavfile.read(os.path.join(out, 'dev1_female3_liverec_130ms_1m_sim_%d.wav' % (i + 1)))
        assert sr == 16000 and pcm.shape == (158976, 2) and pcm.dtype == np.int16
        expect = (g['y_sub'][i] * 32768).clip(-32768, 32767).astype(np.int16)       # float2pcm of the reference's own output
        assert np.abs(pcm.T[:, ::8].astype(int) - expect).max() <= 1


def test_install_aliases():
    from gcc_nmf_amd import dropin
    import gcc_nmf_amd.gccNMFFunctions as G
    saved = {k: sys.modules.get(k) for k in dropin._ALIASES}
    try:
        mod = dropin.install()
        assert mod is G and sys.modules['gccNMFFunctions'] is G and sys.modules['gccNMF.gccNMFFunctions'] is G
        ns = {}
        exec('from gccNMFFunctions import *', ns)
        assert ns['performKLNMF'] is G.performKLNMF and 'hanning' in ns and 'hsplit' in ns
    finally:
        dropin.uninstall()
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v
