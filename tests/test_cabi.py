"""CPU: the C-ABI library loads and exports every symbol include/tamtr_hip.h declares (no compute without a GPU),
and the product refuses CPU tensors instead of falling back."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, 'include', 'tamtr_hip.h')).read()
    return sorted(set(re.findall(r'^int\s+(tamtr_\w+)\s*\(', src, flags=re.M)))


def test_library_exports_every_declared_symbol():
    import tamtr_amd
    from tamtr_amd import _lib
    if not os.path.exists(tamtr_amd.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    names = _declared()
    assert len(names) >= 13
    h = ctypes.CDLL(tamtr_amd.LIB_PATH)
    for n in names:
        assert hasattr(h, n), f'{n} declared in tamtr_hip.h but not exported'
    assert sorted(_lib.EXPORTS) == names, 'ctypes signature table out of sync with the header'
    assert _lib.lib().tamtr_abi_version() == _lib.ABI_VERSION


def test_bad_arguments_are_rejected_without_a_gpu():
    from tamtr_amd import _lib
    h = _lib.lib()
    z = ctypes.c_void_p(0)
    assert h.tamtr_maxsigmoid_gate_fwd(z, z, z, z, z, z, z, 1, 1, 32, 16, 10, 1.0, 0, z) == -1
    assert h.tamtr_msdeform_attn_fwd(z, z, z, z, z, 1, 1, 1, 64, 1, 1, 4, 0, z) == -1
    assert h.tamtr_contrastive_logits_fwd(z, z, z, z, z, z, z, 1, 1, 1, 64, 0, z) == -1
    # next-3: the gate's 3x3 value convolution and the BatchNorm combine (null operands; channel counts the kernel is not built for)
    one = ctypes.c_void_p(16)   # a non-null, 16-byte aligned address that is never dereferenced: the checks come first
    assert h.tamtr_conv3x3_cl_stats_fwd(z, 64, z, z, z, z, z, z, 1, 8, 16, 64, 64, 1e-3, 0.03, z) == -1
    assert h.tamtr_conv3x3_cl_stats_fwd(one, 48, one, one, z, z, z, one, 1, 8, 16, 48, 64, 1e-3, 0.03, z) == -2     # C1 % 32
    assert h.tamtr_conv3x3_cl_stats_fwd(one, 64, one, one, z, z, z, one, 1, 8, 16, 64, 96, 1e-3, 0.03, z) == -2     # C2 % 64
    assert h.tamtr_conv3x3_cl_stats_fwd(one, 32, one, one, z, z, z, one, 1, 8, 16, 64, 64, 1e-3, 0.03, z) == -1     # pitch < C1
    assert h.tamtr_conv3x3_pack_weight(z, z, 64, 64, 0, z) == -1
    assert h.tamtr_conv3x3_pack_weight(one, one, 40, 64, 0, z) == -2
    assert h.tamtr_bn_finalize(z, z, z, z, 64, 10, 1e-3, 0.03, z) == -1
    assert h.tamtr_conv3x3_tiles(2, 40, 40) == 2 * 5 * 3 and h.tamtr_conv3x3_tiles(1, 8, 16) == 1
    assert h.tamtr_selective_scan_chunk() == 64                                     # checkpoint interval of hstate (ABI 23)


def test_cpu_tensors_are_refused_not_emulated():
    import tamtr_amd.ops as ops
    from tamtr_amd import TamtrHipError
    with pytest.raises(TamtrHipError):
        ops.contrastive_logits(torch.zeros(1, 1, 64), torch.zeros(1, 1, 64), torch.zeros(()), torch.zeros(1))
    with pytest.raises(TamtrHipError):
        ops.ms_deform_attn_core(torch.zeros(1, 4, 1, 8), [(2, 2)], torch.zeros(1, 1, 1, 1, 1, 2), torch.zeros(1, 1, 1, 1, 1))
    with pytest.raises(TamtrHipError):
        ops.maxsigmoid_gate(torch.zeros(1, 32, 2, 2), torch.zeros(1, 3, 32), torch.zeros(1), torch.zeros(1, 32, 2, 2), 1)


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, 'tam-tr_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                s = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', s, flags=re.M), f'{f} imports the oracle'
