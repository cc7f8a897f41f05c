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
from ._hip import HipLibraryError, LIB_PATH   # noqa: F401

__version__ = '0.1.0'
