"""ctypes binding of libgear_hip.so (C ABI: include/gear_hip.h).  Fails loudly -- no fallback path."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgear_hip.so")
CSRC = os.path.join(_HERE, "csrc")

MODE_FP16_STEPWISE = 0
MODE_FP32 = 1

_lib = None

_vp, _i64, _i, _sz = C.c_void_p, C.c_int64, C.c_int, C.c_size_t

# name -> (restype, argtypes); mirrors include/gear_hip.h one to one
SIGNATURES = {
    "gear_last_error": (C.c_char_p, []),
    "gear_abi_version": (_i, []),
    "gear_set_option": (_i, [C.c_char_p, _i]),
    "gear_compress_key_fused_workspace": (_sz, [_i64, _i, _i, _i]),
    "gear_compress_key_fused": (_i, [_vp, _i64, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i64, _i64, _i, _i, _i, _vp, _vp, _i64, _i64,
                                     _vp, _i, _i, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "gear_compress_value_fused_workspace": (_sz, [_i64, _i, _i, _i]),
    "gear_compress_value_fused": (_i, [_vp, _i64, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i64, _i64,
                                       _vp, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "gear_vsel_candidates": (_i, [_vp, _i64, _i, _i64, _i64, _i, _i, _i64, _i, _i, _vp, _vp]),
    "gear_vsel_thresholds": (_i, [_vp, _i, _i64, _i, _i64, _i, _vp, _vp, _vp]),
    "gear_compress_value_sharded": (_i, [_vp, _i64, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i64, _i64,
                                         _vp, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "gear_attn_decode_cache": (_i, [_vp, _vp, _i, _i, _i, _vp, C.c_float, _vp, _vp, _vp, _sz, _vp]),
    "gear_cache_tiles_build": (_i, [_vp, _i, _i, _i, _i, _i, _vp]),
    "gear_outlier_chunk_index_ex": (_i, [_vp, _i64, _i64, _i64, _i64, _i, _i, _i, _i, _vp, _i, _vp]),
    "gear_quant_rows_whole": (_i, [_vp, _i64, _i, _i64, _i64, _i, _i, _i64, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "gear_quant_rows_ragged": (_i, [_vp, _i64, _i, _i64, _i64, _i, _i, _i64, _i, _i, _i, _i, _vp, _vp, _vp]),
    "gear_quant_pack_lastdim": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "gear_quant_pack_k": (_i, [_vp, _i64, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "gear_unpack_dequant_lastdim": (_i, [_vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp, _vp]),
    "gear_unpack_dequant_k": (_i, [_vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _vp, _vp]),
    "gear_gemv_outer_workspace": (_sz, [_i64, _i, _i, _i]),
    "gear_compress_rows": (_i, [_vp, _i64, _i, _i64, _i64, _i, _i, _i64, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gear_lowrank_workspace": (_sz, [_i64, _i, _i, _i]),
    "gear_lowrank": (_i, [_vp, _i, _i, _i64, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "gear_decompress_rows": (_i, [_vp, _vp, _vp, _i64, _i, _i64, _i64, _i, _i, _i64, _i, _i, _i, _i, _vp, _vp, _i, _i, _i,
                                  _vp, _vp, _i, _vp, _vp]),
    "gear_attn_decode_workspace": (_sz, [_i, _i, _i, _i]),
    "gear_attn_decode_f16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, C.c_float, _vp, _vp, _vp, _sz, _vp]),
    "gear_attn_decode": (_i, [_vp] * 17 + [_i] * 18 + [C.c_float, _vp, _vp, _vp, _sz, _vp]),
    "gear_attn_decode_seg": (_i, [_vp] * 17 + [_i] * 21 + [C.c_float, _vp, _vp, _vp, _sz, _vp]),
    "gear_attn_decode_dyn": (_i, [_vp] * 17 + [_i] * 21 + [_vp, C.c_float, _vp, _vp, _vp, _sz, _vp]),
    "gear_rope_append_dyn": (_i, [_vp, _i, _i, _i, _i, _vp, C.c_float, _vp, _vp, _vp, _i, _vp]),
    "gear_decode_state_advance": (_i, [_vp, _vp]),
    "gear_outlier_chunk_index": (_i, [_vp, _i64, _i, _i, _i, _vp, _vp]),
    "gear_attn_decode_idx": (_i, [_vp] * 19 + [_i] * 18 + [C.c_float, _vp, _vp, _vp, _sz, _vp]),
    "gear_gemv_f16": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "gear_gemv_f16_add": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "gear_gemv_f16_norm": (_i, [_vp, _vp, _vp, C.c_float, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "gear_gemv_qkv_rope": (_i, [_vp, _vp, _vp, C.c_float, _vp, _i, _i, _i, _i, _i, _i, _i, _i, C.c_float, _vp,
                                _vp, _vp, _vp, _vp, _vp]),
    "gear_rope_append": (_i, [_vp, _i, _i, _i, _i, _i, C.c_float, _vp, _vp, _vp, _i, _i, _vp]),
    "gear_add_rmsnorm": (_i, [_vp, _vp, _vp, _i64, _i, C.c_float, _vp, _vp, _vp]),
    "gear_silu_mul": (_i, [_vp, _i64, _i, _vp, _vp]),
    "gear_transpose_f16": (_i, [_vp, _i64, _i, _i, _vp, _vp]),
    "gear_gemv_outer": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _i64, _i64, _vp, _vp, _sz, _vp]),
    "gear_gemv_outer_dim": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "gear_gemv_outer_lrap_workspace": (_sz, [_i64, _i, _i, _i]),
    "gear_gemv_outer_lrap": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp,
                                  _sz, _vp]),
    "gear_compress_block_workspace": (_sz, [_i64, _i]),
    "gear_compress_block": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _sz, _vp]),
    "gear_compress_block_status_ptr": (_vp, [_vp]),
    "gear_kone_timeouts": (_i, []),
    "gear_kone_fallback_heads": (_i, []),
    "gear_xchg_bytes": (_sz, [_i, _sz]),
    "gear_xchg_alloc": (_i, [_sz, C.POINTER(_vp)]),
    "gear_xchg_free": (_i, [_vp]),
    "gear_xchg_export": (_i, [_vp, _vp]),
    "gear_xchg_open": (_i, [_vp, C.POINTER(_vp)]),
    "gear_xchg_close": (_i, [_vp]),
    "gear_xchg_allgather": (_i, [_vp, _i, _sz, _i, _i, _vp, _vp, _vp, _vp]),
}

ABI_VERSION = 5      # what this table was written against (gear_abi_version() of the library must match)


def _source_hash() -> str:
    """sha256 over the kernel sources (names + contents): what the built library is checked against."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(CSRC, n) for n in sorted(os.listdir(CSRC)) if n.endswith((".hip", ".h")) or n == "Makefile"]
    files.append(os.path.join(os.path.dirname(_HERE), "include", "gear_hip.h"))     # the ABI header is a dependency of every object
    for path in files:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile gear_amd/csrc/*.hip for gfx950 into gear_amd/libgear_hip.so (hipcc cross-compiles without a GPU) and record
    the hash of the sources it was built from next to it."""
    import fcntl
    cmd = ["make", "-C", CSRC, "-j8"]
    # ranks started together (bench.py under torch.distributed.run, the spawned workers of the tests) must not run make in the
    # same directory at once: the first takes the lock and builds, the others find the library up to date
    with open(LIB_PATH + ".lock", "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            if force:
                subprocess.check_call(["make", "-C", CSRC, "clean"], stdout=subprocess.DEVNULL)
            subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
            with open(LIB_PATH + ".src", "w") as f:
                f.write(_source_hash())
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)
    return LIB_PATH


def load():
    """Load the HIP library.  A library that is missing or was built from other sources than the ones in the tree is rebuilt
    first (make is incremental); without a compiler that is an error -- gear_amd has no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    stale = not os.path.exists(LIB_PATH)
    if not stale:
        try:
            with open(LIB_PATH + ".src") as f:
                stale = f.read().strip() != _source_hash()
        except OSError:
            stale = True
    if stale:
        try:
            build()
        except (OSError, subprocess.CalledProcessError) as e:
            raise RuntimeError(
                f"{LIB_PATH} is missing or older than gear_amd/csrc and could not be rebuilt ({e}): run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or make -C gear_amd/csrc). gear_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI and the library disagree
        fn.restype = res
        fn.argtypes = args
    got = lib.gear_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} reports ABI version {got}, gear_amd/_lib.py was written against {ABI_VERSION}: rebuild it")
    _lib = lib
    return lib


class GearError(RuntimeError):
    pass


def check(rc: int, what: str):
    if rc != 0:
        msg = load().gear_last_error()
        raise GearError(f"{what}: status {rc}: {msg.decode() if msg else ''}")


class CacheView(C.Structure):
    """ctypes mirror of `gear_cache_view` (include/gear_hip.h): same field order, pointers then ints."""
    _PTRS = ("kcode", "kscale", "kmn", "kP", "kQ", "koidx", "koval", "vcode", "vscale", "vmn", "vP", "vQ", "voidx", "voval",
             "kwin", "vwin", "kochunk", "vochunk", "ktile", "kcnt", "vtile", "vcnt")
    _INTS = ("B", "Hkv", "D", "tcap", "ldk", "lsk", "group", "bits", "mode", "rk", "rv", "kk_cap", "kk0", "kkb", "kv", "seg0",
             "seglen", "wcap", "nbk_pitch", "ktile_cap", "nck", "vtile_cap", "nblk")
    _fields_ = [(n, C.c_void_p) for n in _PTRS] + [(n, C.c_int) for n in _INTS]


def stream_ptr(t=None) -> int:
    """hipStream_t of the current stream of `t`'s device (the current device when t is None).  The library never calls
    hipSetDevice: launches go to the stream that is passed, so the stream must belong to the device that holds the
    tensors -- require_gpu() checks that all tensors of a call share one device."""
    if t is not None:
        return torch.cuda.current_stream(t.device).cuda_stream
    return torch.cuda.current_stream().cuda_stream


def set_option(name: str, value: int):
    """gear_set_option(): select an alternative (equally exact) code path -- used by the tests."""
    check(load().gear_set_option(name.encode(), int(value)), "gear_set_option")


def ptr(t):
    return None if t is None else t.data_ptr()


def require_gpu(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise GearError("gear_amd operators run on the GPU only (tensor is on %s); there is no CPU fallback" % t.device)
        if not t.is_contiguous():
            raise GearError("gear_amd operators need contiguous tensors")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise GearError(f"gear_amd operators need all tensors on one device (got {dev} and {t.device})")
    if dev is not None and dev.index is not None and dev.index != torch.cuda.current_device():
        raise GearError(f"tensors live on {dev} but the current device is cuda:{torch.cuda.current_device()}: wrap the call in "
                        "torch.cuda.device(tensor.device) (kernels are enqueued on the current device's stream)")
