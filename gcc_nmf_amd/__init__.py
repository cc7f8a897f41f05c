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

# Multi-process GPU work on this driver stack needs dmabuf IPC: without HSA_ENABLE_IPC_MODE_LEGACY=0 RCCL (and any CUDA-tensor sharing
# across processes) fails with `hipIpcGetMemHandle: invalid argument`.  The runtime reads it when HIP initialises, so it is set on
# import -- before torch has touched a device in any ordinary program -- unless the user chose a value.  distributed.collective_hook
# re-checks it where it matters.
_os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

from ._hip import HipLibraryError, LIB_PATH   # noqa: F401,E402

__version__ = '0.1.0'
