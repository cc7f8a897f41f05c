"""Where the wall time of the UNMODIFIED reference driver goes on top of this package: gccNMF/runGCCNMF.py run as __main__ through
dropin.run_reference_driver with every function it binds from gccNMFFunctions wrapped in a timer (the wrappers are installed on the
replacement module before the driver's star-import).  Two runs in one process: cold (imports, first HIP call, buffer pools empty) and warm.
    python scripts/driver_split.py <reference_root> [--resident] > gpurun_out/driver_split.json"""
import json
import os
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
root = sys.argv[1]
resident = '--resident' in sys.argv
os.environ.setdefault('MPLBACKEND', 'Agg')

t0 = time.perf_counter()
import numpy, scipy.signal, torch                                  # noqa: E401,E402
t_numpy_torch = time.perf_counter() - t0
t0 = time.perf_counter()
import matplotlib.pyplot                                           # noqa: E402  (gccNMFPlotting.py imports it at module level)
t_matplotlib = time.perf_counter() - t0
t0 = time.perf_counter()
from gcc_nmf_amd import dropin, gccNMFFunctions as G, _hip         # noqa: E402
_hip.lib()
torch.cuda.init()
torch.zeros(1, device='cuda')
torch.cuda.synchronize()
t_package_and_hip_init = time.perf_counter() - t0

TIMED = ['loadMixtureSignal', 'computeComplexMixtureSpectrogram', 'performKLNMF', 'getAngularSpectrogram',
         'estimateTargetTDOAIndexesFromAngularSpectrum', 'getTargetTDOAGCCNMFs', 'getTargetCoefficientMasks',
         'getTargetSpectrogramEstimates', 'getTargetSignalEstimates', 'saveTargetSignalEstimates']
original = dict((n, getattr(G, n)) for n in TIMED)
times = {}


def wrap(name):
    fn = original[name]

    def timed(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            times[name] = times.get(name, 0.0) + time.perf_counter() - t
    return timed


runs = []
try:
    for n in TIMED:
        setattr(G, n, wrap(n))
    for label in ('cold', 'warm', 'warm2'):
        times.clear()
        for m in ('gccNMFPlotting', 'gccNMF.gccNMFPlotting'):
            if label == 'cold':
                sys.modules.pop(m, None)
        t0 = time.perf_counter()
        dropin.run_reference_driver(root, tempfile.mkdtemp(prefix='gccnmf_split_'), resident=resident)
        wall = time.perf_counter() - t0
        hot = sum(v for k, v in times.items() if k not in ('loadMixtureSignal', 'saveTargetSignalEstimates'))
        runs.append({'run': label, 'wall_ms': 1e3 * wall, 'the_eight_named_functions_ms': 1e3 * hot,
                     'wav_read_ms': 1e3 * times.get('loadMixtureSignal', 0), 'wav_writes_ms': 1e3 * times.get('saveTargetSignalEstimates', 0),
                     'driver_numpy_imports_and_rest_ms': 1e3 * (wall - sum(times.values())),
                     'per_function_ms': dict((k, 1e3 * v) for k, v in times.items())})
finally:
    for n in TIMED:
        setattr(G, n, original[n])
    dropin.uninstall()
print(json.dumps({'what': 'unmodified gccNMF/runGCCNMF.py (hop 128, K = 128, 100 iterations, dev1 mixture) on gcc_nmf_amd via dropin.install(resident=%s); '
                          'the driver itself draws no plots (gccNMFPlotting is imported, not called)' % resident,
                  'before_the_driver_ms': {'import numpy scipy torch': 1e3 * t_numpy_torch, 'import matplotlib.pyplot': 1e3 * t_matplotlib,
                                           'import gcc_nmf_amd + dlopen + HIP context': 1e3 * t_package_and_hip_init},
                  'runs': runs}, indent=1))
