"""gcc_nmf_amd -- the per-frame GCC-NMF hot path of seanwood/gcc-nmf, hand-written for MI355X (gfx950).

    gcc_nmf_amd.gccNMFFunctions   drop-in for the reference's gccNMF/gccNMFFunctions.py (NumPy in / out)
    gcc_nmf_amd.librosaSTFT       stft / istft with the reference's signatures
    gcc_nmf_amd.engine            GCCNMFEngine: a whole batch of mixtures resident in HBM
    gcc_nmf_amd.dropin            install(): run the reference's runGCCNMF.py unchanged on top of this package
    gcc_nmf_amd.distributed       file sharding and shared-dictionary training across ranks (RCCL)

All arithmetic lives in libgccnmf_hip.so (csrc/, C ABI in include/gccnmf_hip.h).  Importing the
package does not need a GPU; calling any hot-path function without the library or a device raises
``HipLibraryError`` -- there is no CPU fallback.
"""
import os as _os
import sys as _sys

# Multi-process GPU work on this driver stack needs dmabuf IPC: without HSA_ENABLE_IPC_MODE_LEGACY=0 RCCL (and any CUDA-tensor sharing
# across processes) fails with `hipIpcGetMemHandle: invalid argument`.  The runtime reads it when HIP initialises, so it is set on
# import -- before torch has touched a device in any ordinary program -- unless the user chose a value.  It is a PROCESS-wide switch
# (every HIP user in the process sees it): GCCNMF_NO_IPC_ENV=1 keeps this package's hands off it.  If HIP was already running when the
# package was imported the setting cannot take effect any more; distributed.collective_hook says so in its error messages.
HIP_STARTED_BEFORE_IMPORT = bool('torch' in _sys.modules and getattr(_sys.modules['torch'], 'cuda', None) is not None and
                                 _sys.modules['torch'].cuda.is_initialized())
IPC_ENV_SET_BY_PACKAGE = False
if _os.environ.get('GCCNMF_NO_IPC_ENV', '') in ('', '0') and 'HSA_ENABLE_IPC_MODE_LEGACY' not in _os.environ:
    _os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    IPC_ENV_SET_BY_PACKAGE = True

from ._hip import HipLibraryError, LIB_PATH   # noqa: F401,E402

__version__ = '0.1.0'
