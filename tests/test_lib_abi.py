"""CPU checks of the C-ABI library: it loads, exports what include/vf_hip.h declares, and its
shape bookkeeping agrees with the Python architecture table (no compute calls without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

from visual_foresight_amd import _lib
from visual_foresight_amd.video_prediction import cdna_arch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    _lib.build_library()
    return _lib.load_library()


def test_exports_every_declared_symbol(lib):
    header = open(os.path.join(REPO, 'include', 'vf_hip.h')).read()
    declared = set(re.findall(r'\b(vf_[a-z_]+)\s*\(', header))
    assert declared, 'no prototypes found in the header'
    assert declared == set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.vf_abi_version() == _lib.ABI_VERSION == 7


@pytest.mark.parametrize('H,W,adim,sdim,nd', [(64, 64, 4, 5, 1), (48, 64, 3, 3, 2), (128, 128, 5, 5, 4)])
def test_weight_count_and_macs_agree_with_arch_table(lib, H, W, adim, sdim, nd):
    cfg = cdna_arch.CdnaConfig(height=H, width=W, adim=adim, sdim=sdim, ndesig=nd)
    c = _lib.VfConfig(H, W, adim, sdim, nd, 2, 15, 10, 8, 0)
    n = sum(int(np.prod(s)) for s in cdna_arch.tensor_shapes(cfg).values())
    assert lib.vf_weight_count(ctypes.byref(c)) == n
    macs = sum(cdna_arch.macs_per_sample_step(cfg).values())
    assert lib.vf_macs_per_sample_step(ctypes.byref(c)) == pytest.approx(macs, rel=1e-12)


@pytest.mark.parametrize('H,W,adim,sdim,nd', [(128, 128, 12, 5, 1), (48, 80, 6, 3, 2)])
def test_savp_weight_count_and_macs_agree_with_arch_table(lib, H, W, adim, sdim, nd):
    from visual_foresight_amd.video_prediction import savp_arch
    cfg = savp_arch.SavpConfig(height=H, width=W, adim=adim, sdim=sdim, ndesig=nd)
    c = _lib.VfConfig(H, W, adim, sdim, nd, 2, 15, 10, 8, 0, 0, 1, 1, cfg.arch_id)
    assert cfg.arch_id == 1
    n = sum(int(np.prod(s)) for s in cfg.tensor_shapes().values())
    assert lib.vf_weight_count(ctypes.byref(c)) == n
    assert lib.vf_macs_per_sample_step(ctypes.byref(c)) == pytest.approx(sum(cfg.macs_per_sample_step().values()), rel=1e-12)
    # four scales: a 128x128 step costs about what a 64x64 step of the three-scale network costs
    if H == 128:
        assert sum(cfg.macs_per_sample_step().values()) < 1.1 * sum(cdna_arch.macs_per_sample_step(cdna_arch.CdnaConfig()).values())
    bad = _lib.VfConfig(72, 64, adim, sdim, nd, 2, 15, 10, 8, 0, 0, 1, 1, 1)
    assert lib.vf_weight_count(ctypes.byref(bad)) == 0 and b'multiples of 16' in lib.vf_last_error()


def test_savp_weights_file_roundtrip(tmp_path):
    from visual_foresight_amd.video_prediction import savp_arch
    cfg = savp_arch.SavpConfig(height=32, width=32, adim=6)
    w = cdna_arch.CdnaWeights.random(cfg, seed=5, bias_scale=0.1, ln_jitter=0.1)
    assert w.tensors['lna/g'].std() > 0 and w.tensors['enc0/w'].shape == (5, 5, 16, 32)
    w.save(str(tmp_path))
    r = cdna_arch.CdnaWeights.load(str(tmp_path))
    assert r.cfg.arch == 'savp' and list(r.tensors) == list(w.tensors)
    for k in w.tensors:
        np.testing.assert_array_equal(w.tensors[k], r.tensors[k])
    with pytest.raises(ValueError):         # a CDNA engine must not swallow a SAVP checkpoint
        cdna_arch.CdnaWeights.load(str(tmp_path), cdna_arch.CdnaConfig(height=32, width=32, adim=6))


def test_survey_mac_count():
    macs = sum(cdna_arch.macs_per_sample_step(cdna_arch.CdnaConfig()).values())
    assert macs == pytest.approx(1.63e9, rel=2e-3)      # 1.63 GMAC per sample-step


def test_invalid_config_is_rejected(lib):
    bad = _lib.VfConfig(60, 64, 4, 5, 1, 2, 15, 10, 8, 0)
    assert lib.vf_weight_count(ctypes.byref(bad)) == 0
    assert b'multiples of 8' in lib.vf_last_error()
    with pytest.raises(_lib.VfError):
        _lib.check(-1)


def test_predictor_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation
    with pytest.raises(_lib.VfError):
        HipVPredEvaluation('', {'designated_pixel_count': 1, 'run_batch_size': 2})


def test_weights_file_roundtrip(tmp_path):
    cfg = cdna_arch.CdnaConfig(height=32, width=32)
    w = cdna_arch.CdnaWeights.random(cfg, seed=5, bias_scale=0.1, ln_jitter=0.1)
    w.save(str(tmp_path))
    r = cdna_arch.CdnaWeights.load(str(tmp_path), cdna_arch.CdnaConfig(height=32, width=32, ndesig=2))
    for k in w.tensors:
        np.testing.assert_array_equal(w.tensors[k], r.tensors[k])
    with pytest.raises(ValueError):
        cdna_arch.CdnaWeights.load(str(tmp_path), cdna_arch.CdnaConfig(height=64, width=64))
    again = cdna_arch.CdnaWeights.random(cfg, seed=5, bias_scale=0.1, ln_jitter=0.1)
    for k in w.tensors:
        np.testing.assert_array_equal(w.tensors[k], again.tensors[k])


def test_c_host_builds_from_the_header_alone(tmp_path):
    """tools/c_host/vf_c_host.c - the C driver of the boundary (run on the GPU by tests/test_gpu_c_host.py) - compiles
    with gcc against include/vf_hip.h and links against the shipped library: every entry point it calls is exported
    with a C-callable signature."""
    import shutil
    import subprocess
    if not shutil.which('gcc') or not os.path.exists('/opt/rocm/include/hip/hip_runtime_api.h'):
        pytest.skip('gcc or the HIP headers are not available')
    exe = str(tmp_path / 'vf_c_host')
    cmd = ['gcc', '-O2', '-std=c11', '-Wall', '-Werror', '-D__HIP_PLATFORM_AMD__', '-I', os.path.join(REPO, 'include'),
           '-I', '/opt/rocm/include', os.path.join(REPO, 'tools', 'c_host', 'vf_c_host.c'), _lib.LIB_PATH,
           '-L/opt/rocm/lib', '-lamdhip64', '-Wl,-rpath,/opt/rocm/lib', '-o', exe]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert proc.returncode == 0, proc.stdout
    # no arguments: usage, exit code 1 - and no HIP call has been made (runs without a GPU)
    run = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                         env=dict(os.environ, LD_LIBRARY_PATH=os.path.dirname(_lib.LIB_PATH)))
    assert run.returncode == 1 and 'usage' in run.stdout
