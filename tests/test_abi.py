"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol declared in
include/gear_hip.h, the ctypes table matches the header, and host-side argument errors surface without a GPU."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT
from gear_amd import _lib


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "gear_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gear_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 8
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/gear_hip.h but not exported by libgear_hip.so"


def test_ctypes_table_matches_header():
    assert sorted(_lib.SIGNATURES.keys()) == _header_symbols()
    lib = _lib.load()
    assert lib.gear_abi_version() >= 1


def test_argument_errors_are_reported_without_gpu():
    lib = _lib.load()
    # bad bit width -> status < 0 and a message, before any launch
    rc = lib.gear_quant_pack_lastdim(None, 4, 64, 64, 3, 0, None, None, None, None, None)
    assert rc < 0
    assert b"bits" in lib.gear_last_error()
    rc = lib.gear_quant_pack_lastdim(None, 4, 100, 64, 2, 0, None, None, None, None, None)
    assert rc < 0 and b"divisible" in lib.gear_last_error()
    rc = lib.gear_gemv_outer(None, None, None, None, 8, 3, 128, 128, 64, 2, 0, 0, 0, None, None, 0, None)
    assert rc < 0 and b"n_rep" in lib.gear_last_error()
    rc = lib.gear_gemv_outer(None, None, None, None, 8, 1, 128, 128, 64, 8, 0, 0, 0, None, None, 0, None)
    assert rc < 0


def test_operators_refuse_cpu_tensors():
    """No CPU fallback: the Python operators raise on non-GPU tensors instead of computing something else."""
    from gear_amd.quant import new_pack, matmul
    x = torch.randn(1, 2, 16, 128).half()
    with pytest.raises(_lib.GearError):
        new_pack.triton_quantize_and_pack_along_last_dim(x, 64, 2)
    with pytest.raises(_lib.GearError):
        new_pack.quant_and_pack_kcache(x.transpose(2, 3).contiguous(), 16, 2)
    with pytest.raises(AssertionError):          # reference assert: T % group_size == 0 (new_pack.py:222)
        new_pack.triton_quantize_and_pack_along_last_dim(x, 48, 2)
    with pytest.raises(AssertionError):          # reference assert: 4-D (new_pack.py:218)
        new_pack.triton_quantize_and_pack_along_last_dim(x[0], 64, 2)
    with pytest.raises(AssertionError):          # matmul.py:217
        matmul.cuda_bmm_fA_qB_outer(64, x[:, :, :1], torch.zeros(1, 2, 128, 8, dtype=torch.int32),
                                    torch.zeros(1, 2, 128, 2).half(), torch.zeros(1, 2, 128, 2).half(), 3)


def test_pack_unpack_tensor_format(golden):
    """Payload format helpers against the reference's golden packs (pure integer ops, runs on CPU)."""
    import numpy as np
    from gear_amd.quant import new_pack
    f = golden("f1_quant_pack.npz")
    raw = torch.from_numpy(f["raw4"])
    assert np.array_equal(new_pack.pack_tensor(raw, 4, 2).numpy(), f["raw4_pack2"])
    assert np.array_equal(new_pack.pack_tensor(raw, 4, 3).numpy(), f["raw4_pack3"])
    assert np.array_equal(new_pack.unpack_tensor(torch.from_numpy(f["raw4_pack3"]), 4, 3).numpy(), f["raw4_unpack3"])
    assert np.array_equal(new_pack.pack_tensor(raw & 3, 2, 3).numpy(), f["raw2_pack3"])
    assert np.array_equal(new_pack.pack_tensor(raw & 3, 2, 2).numpy(), f["raw2_pack2"])


def test_compression_config_ratio_bookkeeping_matches_the_reference():
    """compress_ratio / calculate_compress_ratio_list / calculate_compress_ratio_total (compress_config.py:87-281; the GenerationBench
    drivers call them right after building the config: evaluation_gsm8k.py:407) against fixture F11, made by executing the
    reference's class: every legacy method name, the dense-K / dense-V corners of "Picache", and today's dispatcher names, for
    which the reference appends nothing and the total divides by zero."""
    import json
    import os
    from gear_amd.simulated import CompressionConfig
    cases = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "f11_compress_ratio.json")))
    assert len(cases) == 75
    for c in cases:
        cfg = CompressionConfig(**c["kwargs"])
        cfg.copy_for_all_attention()
        cfg.calculate_compress_ratio_list(c["seqlen"], c["model_dim"])
        assert cfg.compress_ratio_list == c["list"], c["kwargs"]
        if c["total"] == "zde":
            with pytest.raises(ZeroDivisionError):
                cfg.calculate_compress_ratio_total()
        else:
            assert cfg.calculate_compress_ratio_total() == c["total"]
