"""gear_amd -- MI355X-native KV-cache compress / decompress hot path of opengear-project/GEAR.

Hand-written HIP kernels for gfx950 (gear_amd/csrc -> libgear_hip.so, C ABI in include/gear_hip.h) behind the
reference's own Python operator API:

    gear_amd.quant.new_pack   <->  cuda_supported_gear/quant/new_pack.py
    gear_amd.quant.matmul     <->  cuda_supported_gear/quant/matmul.py
    gear_amd.modeling_llamagear <-> cuda_supported_gear/modeling_llamagear.py (compression glue + attention)
    gear_amd.simulated        <->  GenerationBench/GenerationTest/GEARLM/Simulated/{compress_function,compress_config}.py

There is no CPU fallback: every operator raises if libgear_hip.so is missing or the tensors are not on the GPU.
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401
