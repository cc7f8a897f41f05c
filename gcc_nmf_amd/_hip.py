"""ctypes binding of libgccnmf_hip.so (include/gccnmf_hip.h).

The library is the product: there is NO CPU fallback.  If the shared object is
missing (or its symbols do not match the header) importing this module's
``lib()`` raises ``HipLibraryError`` -- loudly, on every call path.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('GCCNMF_HIP_LIB') or os.path.join(_HERE, 'libgccnmf_hip.so')     # override: A/B builds only

c_int, c_long, c_float, c_void_p = ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_void_p
P_INT = ctypes.POINTER(ctypes.c_int)


class SharedShard(ctypes.Structure):
    """gccnmf_shared_shard (include/gccnmf_hip.h)"""
    _fields_ = [('V', c_void_p), ('H', c_void_p), ('workspace', c_void_p), ('N', c_int), ('batch', c_int), ('ld', c_int)]


class DirectGemm(ctypes.Structure):
    """gccnmf_direct_gemm (include/gccnmf_hip.h): descriptor of one latency-path GEMM (csrc/direct.hip)"""
    _fields_ = [('A', c_void_p), ('B', c_void_p), ('sA', c_long), ('sB', c_long), ('lda', c_int), ('ldb', c_int),
                ('M', c_int), ('N', c_int), ('Kd', c_int), ('batch', c_int),
                ('bscale', c_void_p), ('s_bscale', c_long), ('tailA', c_void_p), ('s_tailA', c_long), ('tail_row', c_int),
                ('rowsumB', c_void_p), ('s_rowsumB', c_long), ('C', c_void_p), ('sC', c_long), ('ldc', c_int),
                ('Ct', c_void_p), ('sCt', c_long), ('ldct', c_int), ('E0', c_void_p), ('sE0', c_long), ('lde0', c_int),
                ('E1', c_void_p), ('sE1', c_long), ('E2', c_void_p), ('sE2', c_long), ('ktailA', c_void_p), ('ktailB', c_void_p),
                ('s_ktailA', c_long), ('s_ktailB', c_long), ('alpha', c_float), ('eps', c_float),
                ('tiles_m', c_int), ('tiles_n', c_int), ('xc', c_int), ('sm', c_int), ('sn', c_int), ('trace', c_void_p)]


# gccnmf_allreduce_fn: int (*)(void* ctx, float* buf, long count, void* stream)
RCCL_UNIQUE_ID_BYTES = 128
ALLREDUCE_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_long, c_void_p)

STATUS = {0: 'GCCNMF_OK', 1: 'GCCNMF_ERR_ARG (bad argument)', 2: 'GCCNMF_ERR_LAUNCH (HIP launch failed)',
          3: 'GCCNMF_ERR_UNSUPPORTED', 4: 'GCCNMF_ERR_COLLECTIVE (all-reduce hook / RCCL failed)'}

# name -> (restype, argtypes); mirrors include/gccnmf_hip.h declaration by declaration
SIGNATURES = {
    'gccnmf_version': (c_int, []),
    'gccnmf_set_tuning': (c_int, [c_int, c_int]),
    'gccnmf_pitches': (c_int, [c_int, c_int, c_int, P_INT, P_INT, P_INT, P_INT]),
    'gccnmf_stft_stereo': (c_int, [c_void_p, c_long, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p]),
    'gccnmf_dft_workspace_floats': (c_long, [c_int, c_int, c_int]),
    'gccnmf_stft_dft': (c_int, [c_void_p, c_long, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'gccnmf_istft_dft': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_float, c_int, c_void_p, c_void_p, c_void_p]),
    'gccnmf_stft_stereo_pcm16': (c_int, [c_void_p, c_long, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_void_p]),
    'gccnmf_pack_pcm16': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'gccnmf_klnmf_workspace_floats': (c_long, [c_int, c_int, c_int, c_int]),
    'gccnmf_klnmf': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                             c_float, c_int, c_void_p]),
    'gccnmf_klnmf_ragged_workspace_floats': (c_long, [c_int, c_int, c_int, c_int]),
    'gccnmf_klnmf_ragged': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, P_INT, c_int, c_int, c_int, c_int, c_float,
                                    c_float, c_int, c_void_p]),
    'gccnmf_klnmf_chain_status': (c_int, [c_void_p, c_int, c_int, c_int, c_int, P_INT]),
    'gccnmf_klnmf_plan': (c_int, [c_int, c_int, c_int, c_int, c_int]),
    'gccnmf_klnmf_stage': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float,
                                   c_int, c_int, c_void_p]),
    'gccnmf_klnmf_shared_workspace_floats': (c_long, [c_int, c_int, c_int, c_int]),
    'gccnmf_klnmf_shared_partial_floats': (c_long, [c_int, c_int]),
    'gccnmf_klnmf_shared_begin': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'gccnmf_klnmf_shared_step_a': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                           c_int, c_float, c_float, c_void_p]),
    'gccnmf_klnmf_shared_step_b': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'gccnmf_klnmf_shared_finish': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'gccnmf_klnmf_shared_shard_workspace_floats': (c_long, [c_int, c_int, c_int, c_int, c_int]),
    'gccnmf_klnmf_shared_run': (c_int, [ctypes.POINTER(SharedShard), c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float,
                                        c_float, c_void_p, c_void_p, c_void_p]),
    'gccnmf_rccl_available': (c_int, []),
    'gccnmf_rccl_unique_id': (c_int, [ctypes.c_char_p]),
    'gccnmf_rccl_comm_init': (c_int, [ctypes.c_char_p, c_int, c_int, ctypes.POINTER(c_void_p)]),
    'gccnmf_rccl_comm_destroy': (c_int, [c_void_p]),
    'gccnmf_rccl_allreduce': (c_int, [c_void_p, c_void_p, c_long, c_void_p]),
    'gccnmf_rccl_allreduce_hook': (c_void_p, []),
    'gccnmf_angular_spectrogram': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                           c_void_p]),
    'gccnmf_pick_tdoa_peaks': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'gccnmf_scores_workspace_floats': (c_long, [c_int, c_int, c_int, c_int]),
    'gccnmf_target_scores_masks': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                           c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'gccnmf_argmax_targets': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'gccnmf_coherence': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'gccnmf_magnitude': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'gccnmf_reconstruct_workspace_floats': (c_long, [c_int, c_int, c_int, c_int]),
    'gccnmf_reconstruct': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                   c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'gccnmf_istft_ola': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_int,
                                 c_void_p, c_void_p, c_void_p]),
    'gccnmf_ola_frames_halo': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    'gccnmf_rt_process_block': (c_int, [c_void_p] * 19 + [c_int] * 13 + [c_void_p]),
    'gccnmf_rt_process_block_ll': (c_int, [c_void_p] * 23 + [c_int] * 15 + [c_void_p]),
    'gccnmf_gemm_direct': (c_int, [ctypes.POINTER(DirectGemm), c_int, c_int, c_void_p]),
    'gccnmf_debug_gemm_plan': (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, P_INT, P_INT, c_int]),
    'gccnmf_debug_gemm': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                  c_int, c_int, c_long, c_long, c_long, c_void_p, c_void_p, c_void_p]),
}


# entry points of the lab build only (make -C gcc_nmf_amd/csrc EXPERIMENTS=1 -> libgccnmf_hip_exp.so; `#ifdef GCCNMF_EXPERIMENTS` in the header)
EXPERIMENT_SIGNATURES = {
    'gccnmf_debug_set_trace': (c_int, [c_void_p, c_int]),
    'gccnmf_debug_mfma_peak': (c_int, [c_void_p, c_int, c_int, c_void_p]),
}


class HipLibraryError(RuntimeError):
    pass


_lib = None


def lib():
    """The loaded shared library with typed prototypes; raises HipLibraryError if it cannot be used."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            '%s is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` or '
            '`make -C gcc_nmf_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback.' % LIB_PATH)
    try:
        handle = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise HipLibraryError('cannot load %s: %s' % (LIB_PATH, e))
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError:
            if os.environ.get('GCCNMF_HIP_LIB'):      # an A/B build of another revision: entry points it lacks are simply not callable
                continue
            raise HipLibraryError('%s does not export %s (stale build? re-run make -C gcc_nmf_amd/csrc)' % (LIB_PATH, name))
        fn.restype = restype
        fn.argtypes = argtypes
    for name, (restype, argtypes) in EXPERIMENT_SIGNATURES.items():
        fn = getattr(handle, name, None)
        if fn is not None:
            fn.restype = restype
            fn.argtypes = argtypes
    # A/B runs without code changes: GCCNMF_TUNE="9=2,8=1" applies gccnmf_set_tuning(key, value) pairs at load time
    for kv in filter(None, os.environ.get('GCCNMF_TUNE', '').split(',')):
        key, value = [int(v) for v in kv.split('=')]
        if handle.gccnmf_set_tuning(key, value) != 0:
            # a rejected pair would silently measure the defaults (a product build rejects the experiment-only keys): always loud.
            # GCCNMF_TUNE_LENIENT=1 (one tuning string across libraries of different revisions) downgrades it to a warning.
            msg = 'GCCNMF_TUNE: gccnmf_set_tuning(%d, %d) was rejected by %s' % (key, value, LIB_PATH)
            if os.environ.get('GCCNMF_TUNE_LENIENT', '') in ('', '0'):
                raise HipLibraryError(msg)
            import warnings
            warnings.warn(msg + ' -- running with that key at its default')
    _lib = handle
    return _lib


def check(status, what):
    if status != 0:
        raise HipLibraryError('%s failed: %s' % (what, STATUS.get(status, 'status %d' % status)))
