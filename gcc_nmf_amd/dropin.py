"""Run the reference's own driver (gccNMF/runGCCNMF.py) UNCHANGED on top of this package.

The reference binds its algorithm functions by name through star-imports:
    runGCCNMF.py:27   from gccNMFFunctions import *        (a TOP-LEVEL module, resolved from the script dir)
    runGCCNMF.py:28   from gccNMFPlotting import *         (which re-exports gccNMF.gccNMFFunctions, gccNMFPlotting.py:27)
so both module names must resolve to ``gcc_nmf_amd.gccNMFFunctions`` *before* the script runs;
putting this package earlier on PYTHONPATH is not enough because sys.path[0] (the script dir) wins
(SURVEY.md section 8b).  ``install()`` seeds ``sys.modules`` accordingly.
"""
import os
import runpy
import shutil
import sys
import tempfile

_ALIASES = {
    'gccNMFFunctions': 'gcc_nmf_amd.gccNMFFunctions',
    'gccNMF.gccNMFFunctions': 'gcc_nmf_amd.gccNMFFunctions',
    'gccNMF.librosaSTFT': 'gcc_nmf_amd.librosaSTFT',
    'gccNMF.wavfile': 'gcc_nmf_amd.wavfile',
}


def install(resident=False):
    """After this call ``from gccNMFFunctions import *`` / ``from gccNMF.gccNMFFunctions import ...`` bind the
    MI355X implementations.  Returns the replacement module.

    ``resident=True`` (opt-in): every array a named function returns is read-only and keeps its device image while it lives; passing the
    same object on to the next function -- what runGCCNMF.py:36-52 does with X, W, the scores, the masks and the spectrogram estimates --
    then skips the re-upload (gcc_nmf_amd/_staging.py).  Default: writable outputs, every argument uploaded, as in earlier rounds."""
    import importlib
    for alias, target in _ALIASES.items():
        sys.modules[alias] = importlib.import_module(target)
    sys.modules['gccNMFFunctions'].set_resident(resident)
    return sys.modules['gccNMFFunctions']


def uninstall():
    mod = sys.modules.get('gccNMFFunctions')
    if mod is not None and hasattr(mod, 'set_resident'):
        mod.set_resident(False)
    for alias in _ALIASES:
        sys.modules.pop(alias, None)


def run_reference_driver(reference_root, workdir=None, resident=False):
    """Execute ``<reference_root>/gccNMF/runGCCNMF.py`` byte-for-byte unchanged (as ``__main__``) against this
    package.  The driver reads ``../data/<prefix>_mix.wav`` and writes ``../data/<prefix>_sim_N.wav`` relative to
    its working directory, so it runs from a scratch copy of ``data/``.  Returns the directory holding the outputs."""
    reference_root = os.path.abspath(reference_root)
    script = os.path.join(reference_root, 'gccNMF', 'runGCCNMF.py')
    if not os.path.exists(script):
        raise FileNotFoundError(script)
    workdir = workdir or tempfile.mkdtemp(prefix='gccnmf_dropin_')
    data = os.path.join(workdir, 'data')
    cwd = os.path.join(workdir, 'gccNMF')
    os.makedirs(cwd, exist_ok=True)
    if not os.path.isdir(data):
        shutil.copytree(os.path.join(reference_root, 'data'), data)
    os.environ.setdefault('MPLBACKEND', 'Agg')              # gccNMFPlotting imports matplotlib
    install(resident=resident)
    old_cwd, old_path = os.getcwd(), list(sys.path)
    try:
        os.chdir(cwd)
        # script dir first (for `from gccNMFPlotting import *`), reference root for the `gccNMF` package
        sys.path[:0] = [os.path.join(reference_root, 'gccNMF'), reference_root]
        runpy.run_path(script, run_name='__main__')
    finally:
        os.chdir(old_cwd)
        sys.path[:] = old_path
    return data


if __name__ == '__main__':
    out = run_reference_driver(sys.argv[1] if len(sys.argv) > 1 else '/root/reference',
                               sys.argv[2] if len(sys.argv) > 2 else None)
    print('outputs in', out)
