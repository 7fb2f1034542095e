"""ctypes binding of libvf_hip.so (the C ABI declared in include/vf_hip.h).

The library is built in-tree by ``build_library()`` (``hipcc --offload-arch=gfx950``; it
cross-compiles without a GPU).  There is no CPU fallback: ``load_library()`` raises if the
shared object is missing, and every wrapper raises ``VfError`` with ``vf_last_error()`` when
a call fails.
"""
import ctypes
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(_HERE)
LIB_PATH = os.environ.get('VF_LIBRARY') or os.path.join(_HERE, 'libvf_hip.so')     # override: experiments only
SOURCES = [os.path.join(_HERE, 'csrc', f) for f in
           ('vf_engine.hip', 'vf_conv_mfma.h', 'vf_conv_gsplit.h', 'vf_small_kernels.h', 'vf_persistent.h',
            'vf_conv_bf16x6.h', 'vf_fused_top.h', 'vf_fc_tile.h', 'vf_conv_first.h', 'vf_savp3.h', 'vf_engine_savp3.inc')] + \
          [os.path.join(REPO, 'include', 'vf_hip.h')]

# every symbol include/vf_hip.h declares
EXPORTS = ('vf_abi_version', 'vf_last_error', 'vf_weight_count', 'vf_create', 'vf_destroy',
           'vf_load_weights', 'vf_set_context', 'vf_rollout', 'vf_export', 'vf_register',
           'vf_allgather_scores', 'vf_comm_init_all', 'vf_comm_destroy', 'vf_allgather_scores_group',
           'vf_macs_per_sample_step', 'vf_set_profiling', 'vf_get_profile',
           'vf_set_dedup', 'vf_set_persistent', 'vf_set_xcd_queues', 'vf_set_fuse_top', 'vf_device_status',
           'vf_set_phase_stats', 'vf_debug_phase_stats', 'vf_debug_poison_status', 'vf_set_sched_option')
ABI_VERSION = 7


class VfError(RuntimeError):
    pass


class VfConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ('height', 'width', 'adim', 'sdim', 'ndesig', 'n_context', 'sequence_length',
                 'num_masks', 'max_batch', 'device', 'precision', 'ncam', 'n_draws', 'arch', 'zdim', 'layer_spec')]


def _hipcc():
    for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise VfError('hipcc not found; cannot build libvf_hip.so')


def library_is_stale():
    if os.environ.get('VF_LIBRARY'):
        return False
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(s) > built for s in SOURCES)


def build_library(force=False, verbose=False):
    """Compile the HIP engine for gfx950 into visual_foresight_amd/libvf_hip.so."""
    if not force and not library_is_stale():
        return LIB_PATH
    cmd = [_hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-shared', '-fPIC',
           '-o', LIB_PATH + '.tmp', SOURCES[0]]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or proc.returncode:
        print(' '.join(cmd))
        print(proc.stdout)
    if proc.returncode:
        raise VfError('hipcc failed building libvf_hip.so')
    os.replace(LIB_PATH + '.tmp', LIB_PATH)
    return LIB_PATH


_lib = None


def load_library():
    """dlopen libvf_hip.so and declare the prototypes.  Raises VfError when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VfError('%s is missing - run `python -c "import __graft_entry__ as g; g.build()"` '
                      '(there is no CPU fallback for the predictor)' % LIB_PATH)
    # PyTorch (the buffer carrier) ships its own copy of the HIP runtime.  It must be the FIRST one this process loads:
    # when libvf_hip.so pulls in /opt/rocm's libamdhip64 before torch has loaded its bundled one, the process ends up
    # with two runtimes and the second reports "no ROCm-capable device" (seen with build() followed by smoke() in one
    # process).  Importing torch here makes the order independent of what the caller did first.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    P = ctypes.c_void_p
    lib.vf_abi_version.restype = ctypes.c_int
    lib.vf_last_error.restype = ctypes.c_char_p
    lib.vf_weight_count.restype = ctypes.c_size_t
    lib.vf_weight_count.argtypes = [ctypes.POINTER(VfConfig)]
    lib.vf_macs_per_sample_step.restype = ctypes.c_double
    lib.vf_macs_per_sample_step.argtypes = [ctypes.POINTER(VfConfig)]
    lib.vf_create.argtypes = [ctypes.POINTER(VfConfig), ctypes.POINTER(P)]
    lib.vf_destroy.argtypes = [P]
    lib.vf_load_weights.argtypes = [P, P, ctypes.c_size_t]
    lib.vf_set_context.argtypes = [P, P, P, P, P, P]
    lib.vf_rollout.argtypes = [P, P, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.c_float,
                               ctypes.POINTER(ctypes.c_float), P, P, P]
    lib.vf_register.argtypes = [P, P, P, P, P, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, P, P, P, P, P]
    lib.vf_allgather_scores.argtypes = [P, P, P, ctypes.c_int32, P, P]
    lib.vf_comm_init_all.argtypes = [ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(P)]
    lib.vf_comm_destroy.argtypes = [P]
    lib.vf_allgather_scores_group.argtypes = [ctypes.c_int32, ctypes.POINTER(P), ctypes.POINTER(P), ctypes.POINTER(P),
                                              ctypes.c_int32, ctypes.POINTER(P), ctypes.POINTER(P)]
    lib.vf_set_phase_stats.argtypes = [P, ctypes.c_int32]
    lib.vf_debug_phase_stats.argtypes = [P, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32),
                                         ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_uint64)]
    lib.vf_debug_poison_status.argtypes = [P]
    lib.vf_export.argtypes = [P, ctypes.c_int32, ctypes.c_int32, P, P, P, P]
    lib.vf_set_profiling.argtypes = [P, ctypes.c_int32]
    lib.vf_get_profile.argtypes = [P, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64),
                                   ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    lib.vf_set_dedup.argtypes = [P, ctypes.c_int32]
    lib.vf_set_dedup.restype = ctypes.c_int
    lib.vf_set_persistent.argtypes = [P, ctypes.c_int32]
    lib.vf_set_xcd_queues.argtypes = [P, ctypes.c_int32]
    lib.vf_set_fuse_top.argtypes = [P, ctypes.c_int32]
    lib.vf_set_sched_option.argtypes = [P, ctypes.c_int32, ctypes.c_int32]
    lib.vf_set_sched_option.restype = ctypes.c_int
    lib.vf_set_fuse_top.restype = ctypes.c_int
    lib.vf_set_xcd_queues.restype = ctypes.c_int
    lib.vf_device_status.argtypes = [P, ctypes.POINTER(ctypes.c_int32)]
    lib.vf_set_persistent.restype = lib.vf_device_status.restype = ctypes.c_int
    lib.vf_set_profiling.restype = lib.vf_get_profile.restype = ctypes.c_int
    for name in ('vf_create', 'vf_destroy', 'vf_load_weights', 'vf_set_context', 'vf_rollout',
                 'vf_export', 'vf_register', 'vf_allgather_scores', 'vf_comm_init_all', 'vf_comm_destroy',
                 'vf_allgather_scores_group', 'vf_set_phase_stats',
                 'vf_debug_phase_stats', 'vf_debug_poison_status'):
        getattr(lib, name).restype = ctypes.c_int
    if lib.vf_abi_version() != ABI_VERSION:
        raise VfError('libvf_hip.so ABI version %d, expected %d (stale build? run __graft_entry__.build())'
                      % (lib.vf_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise VfError('libvf_hip error %d: %s' % (rc, load_library().vf_last_error().decode()))
