"""Fixture F8-ref: the attention cache STATE MACHINES pinned against the reference's own code (SURVEY.md section 8c, rows a7 / a8
/ f-4).  Runs in the build container only (needs /root/reference); writes tests/golden/f8_ref_*.npz (data only).

What is executed is the reference's source, extracted with `ast` from the files where they lie and exec'd unmodified:
  * key_compression, value_compression, matmul_withlrap     cuda_supported_gear/modeling_llamagear.py:23-111
  * LlamaAttention_GEAR.forward                               cuda_supported_gear/modeling_llamagear.py:177-484
  * LlamaAttention_KIVI.forward                               cuda_supported_gear/modeling_llama_kivi.py:81-289
The modules cannot be imported whole under the installed transformers (5.x; the reference pins 4.38.2) and the CUDA extension
`kivi_gemv` cannot be built (no nvcc), so the names those functions look up are bound as follows:
  * triton_quantize_and_pack_along_last_dim[_witherror], headwise_lrap, unpack_and_dequant_vcache: the reference's
    quant/new_pack.py, imported as is (Triton kernels under TRITON_INTERPRET=1 on CPU tensors);
  * cuda_bmm_fA_qB_outer(group, fA, qB, scales, zeros, bits): fA @ unpack_and_dequant_vcache(qB, ...) with fp32 accumulation and
    one fp16 rounding -- the check the reference applies to its own kernel (quant/gemv.py:70-74, :118-123; gemv_cuda.cu:343-345);
  * the model plumbing around the cache path (`self`): q / k / v projections = slices of the input (the trace is stated on
    post-RoPE q, k, v), o_proj = identity, rotary embedding = identity.  Projections and RoPE are HF code, not the path.
For the low-rank methods two of the reference's defects make its own forward useless as a parity target (SURVEY appendix B:
B1 `_witherror` packs only the first columns, B2 the K error is reshaped instead of transposed).  The build follows the
documented stances (all columns packed, low-rank of the true error); for those cases ("stance" in the case name) the forward
still is the reference's, with key_compression / value_compression replaced by the six lines below that call the SAME reference
leaf functions with the two defects removed.  The "kivi" cases run the reference forward with nothing replaced.

Stored per case: inputs q, k, v (post-RoPE, fp16), every torch.rand basis the run drew (in order), the attention output of the
last prompt position and of every decode step, and the final cache: slot 8, packed K / V codes, scales, zero points, factors.
A second group of entries pins matmul_withlrap alone (key and value; prefill factors only and prefill + stacked block factors).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_f8_ref.py
"""
import ast
import math
import os
import sys
import textwrap
import types
import warnings
from typing import List, Optional, Tuple

os.environ["TRITON_INTERPRET"] = "1"
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

REF = "/root/reference/cuda_supported_gear"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REF, "quant"))
import new_pack  # noqa: E402  (the reference's module)


def extract(path, name, klass=None):
    """Source text of a module-level function, or of a method of `klass`, from a reference file."""
    src = open(path).read()
    tree = ast.parse(src)
    body = tree.body
    if klass is not None:
        body = next(n for n in body if isinstance(n, ast.ClassDef) and n.name == klass).body
    node = next(n for n in body if isinstance(n, ast.FunctionDef) and n.name == name)
    lines = src.split("\n")[node.lineno - 1:node.end_lineno]
    return textwrap.dedent("\n".join(lines))


def cuda_bmm_fA_qB_outer(group_size, fA, qB, scales, zeros, bits, mqa=False):
    """Stand-in for the CUDA extension with the ARITHMETIC of its kernels (quant/csrc/gemv_cuda.cu:331-345, bgemv{2,4}_kernel_outer_dim):
    the weight is dequantized in fp32 -- float(scale) * float(code) + float(zero), never rounded to fp16 --, multiplied by float(input),
    accumulated in fp32, and rounded to fp16 ONCE at the end.  (Rounds 3-4 used the reference's own host-side check of that kernel,
    fA @ unpack_and_dequant_vcache(...), quant/gemv.py:70-74, which rounds the dequantized weight to fp16 first: the fixture was then
    2e-3 away from the kernel it stood in for, and the parity tests had to carry that.)  Codes are unpacked by the reference's
    unpack_tensor; the summation order over the input channels is torch.matmul's, the kernel's is a strided per-thread sum + warp
    tree -- fp32 either way."""
    code = new_pack.unpack_tensor(qB, bits, pack_dim=3).float()                      # [B, nh, K, N]
    shape = code.shape
    code = code.view(shape[:-1] + (shape[-1] // group_size, group_size))
    w = scales.float().unsqueeze(-1) * code + zeros.float().unsqueeze(-1)            # fp32, as the kernel
    return torch.matmul(fA.float(), w.view(shape)).to(fA.dtype)


def apply_rotary_pos_emb(q, k, cos, sin, position_ids=None):
    return q, k


def namespace(stance):
    ns = dict(torch=torch, nn=nn, F=F, math=math, warnings=warnings, Optional=Optional, Tuple=Tuple, List=List,
              triton_quantize_and_pack_along_last_dim=new_pack.triton_quantize_and_pack_along_last_dim,
              triton_quantize_and_pack_along_last_dim_witherror=new_pack.triton_quantize_and_pack_along_last_dim_witherror,
              headwise_lrap=new_pack.headwise_lrap, cuda_bmm_fA_qB_outer=cuda_bmm_fA_qB_outer,
              apply_rotary_pos_emb=apply_rotary_pos_emb)
    gear = os.path.join(REF, "modeling_llamagear.py")
    for fn in ("key_compression", "value_compression", "matmul_withlrap"):
        exec(extract(gear, fn), ns)
    if stance:
        def low(cc):
            return "gearl" in cc["compress_method"] or "gearsl" in cc["compress_method"]

        def key_compression(key_full, cc):          # key_full [B,H,D,T]
            code, scale, mn = new_pack.triton_quantize_and_pack_along_last_dim(key_full, cc["group_size"], cc["quantize_bit"])   # B1
            if not low(cc):
                return code, scale, mn, None, None
            err = new_pack.triton_quantize_and_pack_along_last_dim_witherror(key_full, cc["group_size"], cc["quantize_bit"])[3]
            p, q = new_pack.headwise_lrap(err.reshape(key_full.shape), cc["rank"], cc["loop"])                                    # B2
            return code, scale, mn, p, q

        def value_compression(value_full, cc):      # value_full [B,H,T,D]
            code, scale, mn = new_pack.triton_quantize_and_pack_along_last_dim(value_full, cc["group_size"], cc["quantize_bit"])
            if not low(cc):
                return code, scale, mn, None, None
            err = new_pack.triton_quantize_and_pack_along_last_dim_witherror(value_full, cc["group_size"], cc["quantize_bit"])[3]
            p, q = new_pack.headwise_lrap(err.reshape(value_full.shape), cc["rankv"], cc["loop"])
            return code, scale, mn, p, q
        ns["key_compression"], ns["value_compression"] = key_compression, value_compression
    exec(extract(gear, "forward", "LlamaAttention_GEAR"), ns)
    ns["forward_gear"] = ns.pop("forward")
    exec(extract(os.path.join(REF, "modeling_llama_kivi.py"), "forward", "LlamaAttention_KIVI"), ns)
    ns["forward_kivi"] = ns.pop("forward")
    return ns


def stub_self(H, D, cc, bits):
    HD = H * D
    return types.SimpleNamespace(
        q_proj=lambda h: h[..., :HD], k_proj=lambda h: h[..., HD:2 * HD], v_proj=lambda h: h[..., 2 * HD:],
        o_proj=lambda x: x, rotary_emb=lambda x, seq_len=None: (None, None), num_heads=H, num_key_value_heads=H,
        num_key_value_groups=1, head_dim=D, hidden_size=HD, compress_config=cc, residual_length=cc["residual"],
        group_size=cc["group_size"], k_bits=bits, v_bits=bits, layer_idx=0, config=types.SimpleNamespace(pretraining_tp=1))


CASES = {  # name: (which forward, method, bits, stance, prompt length, decode steps)
    "gear_kivi_b2": ("gear", "KIVI", 2, False, 200, 130),       # reference forward + reference compression glue, untouched
    "gear_kivi_b4_t64": ("gear", "KIVI", 4, False, 64, 70),     # prompt exactly one block: V stays fp16 until the first boundary
    "gear_stance_gearl_b2": ("gear", "gearlKIVI", 2, True, 200, 130),
    "gear_stance_gearl_b4_t30": ("gear", "gearlKIVI", 4, True, 30, 110),   # prompt shorter than the window
    "kivi_b2": ("kivi", "KIVI", 2, False, 200, 130),            # modeling_llama_kivi.py: sliding V window, per-token V quantization
    "kivi_b4_t64": ("kivi", "KIVI", 4, False, 64, 70),
    # round 5: a wider trace -- 8 heads, rank 8, two block boundaries past a 512-token prompt (the shapes above are H = 2, rank 4)
    "gear_stance_gearl_b2_h8r8": ("gear", "gearlKIVI", 2, True, 512, 130),
}
H, D, RANK = 2, 128, 4
SHAPE = {"gear_stance_gearl_b2_h8r8": (8, 8)}      # per-case (heads, rank) where they differ from the defaults


def run_case(name, out):
    which, method, bits, stance, TP, STEPS = CASES[name]
    H, RANK = SHAPE.get(name, (globals()["H"], globals()["RANK"]))
    cc = dict(compress_method=method, group_size=64, residual=64, quantize_bit=bits, rank=RANK, rankv=RANK, loop=3)
    ns = namespace(stance)
    fwd = ns["forward_gear"] if which == "gear" else ns["forward_kivi"]
    me = stub_self(H, D, cc, bits)
    g = torch.Generator().manual_seed(4321 + len(name))
    qkv = (torch.randn(3, 1, H, TP + STEPS, D, generator=g) * 0.5).half()
    draws = []
    real_rand = torch.rand

    def rec_rand(*a, **k):
        t = real_rand(*a, **k)
        draws.append(t.clone())
        return t
    torch.rand = rec_rand
    try:
        torch.manual_seed(99)

        def hidden(t0, t1):      # [1, q_len, 3*H*D]: q | k | v rows of the tokens
            return torch.cat([qkv[i][:, :, t0:t1].transpose(1, 2).reshape(1, t1 - t0, H * D) for i in range(3)], -1)
        mask = torch.triu(torch.full((TP, TP), torch.finfo(torch.float16).min, dtype=torch.float16), 1)[None, None]
        o, _, cache = fwd(me, hidden(0, TP), attention_mask=mask, use_cache=True)
        outs = [o[:, -1:]]
        for i in range(STEPS):
            o, _, cache = fwd(me, hidden(TP + i, TP + i + 1), attention_mask=None, past_key_value=cache, use_cache=True)
            outs.append(o)
    finally:
        torch.rand = real_rand
    out[f"{name}_qkv"] = qkv.numpy()
    out[f"{name}_out"] = torch.cat(outs, 1).numpy()            # [1, 1 + STEPS, H*D]
    out[f"{name}_seq"] = np.array([cache[8 if which == "gear" else -1]])
    slots = {"kcode": 0, "kfull": 1, "kscale": 2, "kmn": 3, "vcode": 4, "vfull": 5, "vscale": 6, "vmn": 7}
    for k, i in slots.items():
        if cache[i] is not None:
            out[f"{name}_{k}"] = cache[i].numpy()
    if which == "gear":
        for k, i in (("kp", 9), ("kq", 10), ("vp", 13), ("vq", 14)):
            lst = cache[i]
            if lst is not None and lst[0] is not None:
                for j, t in enumerate(lst):
                    out[f"{name}_{k}{j}"] = t.numpy()
    # headwise_lrap draws p_base then q_base (new_pack.py:296-297): the even draws are the bases that matter
    for j, t in enumerate(draws[0::2]):
        out[f"{name}_P0_{j}"] = t.numpy()
    print(name, "ok:", out[f"{name}_out"].shape, "seq", int(out[f"{name}_seq"][0]), "draws", len(draws), flush=True)


def run_matmul(out):
    """matmul_withlrap alone on a payload made with the reference's leaf functions: K^T [1,H,D,T] / V [1,H,T,D] with T = Tp + nb*64."""
    ns = namespace(False)
    mm = ns["matmul_withlrap"]
    g = torch.Generator().manual_seed(777)
    for bits in (2, 4):
        Tp, nb, r, gs = 128, 2, RANK, 64
        T = Tp + nb * 64
        kT = (torch.randn(1, H, D, T, generator=g) * 0.5).half()
        v = (torch.randn(1, H, T, D, generator=g) * 0.5).half()
        q = (torch.randn(1, H, 1, D, generator=g) * 0.5).half()
        a = torch.softmax(torch.randn(1, H, 1, T, generator=g), -1).half()
        kc, ks, km = new_pack.triton_quantize_and_pack_along_last_dim(kT, gs, bits)
        vc, vs, vm = new_pack.triton_quantize_and_pack_along_last_dim(v, gs, bits)
        fac = lambda *shape: (torch.randn(*shape, generator=g) * 0.1).half()
        # K: P [1,H,T,r] token side, Q [1,H,D,r]; stacked blocks [nb,1,H,64,r] / [nb,1,H,D,r] (modeling_llamagear.py:71-85)
        kp0, kq0, kp1, kq1 = fac(1, H, Tp, r), fac(1, H, D, r), fac(nb, 1, H, 64, r), fac(nb, 1, H, D, r)
        # V: P [1,H,D,r], Q [1,H,T,r]; stacked [nb,1,H,D,r] / [nb,1,H,64,r] (:87-108)
        vp0, vq0, vp1, vq1 = fac(1, H, D, r), fac(1, H, Tp, r), fac(nb, 1, H, D, r), fac(nb, 1, H, 64, r)
        pre = f"mm_b{bits}_"
        fpi = 32 // bits
        for k_, t_ in dict(q=q, a=a, kc=kc, ks=ks, km=km, vc=vc, vs=vs, vm=vm, kp0=kp0, kq0=kq0, kp1=kp1, kq1=kq1, vp0=vp0, vq0=vq0,
                           vp1=vp1, vq1=vq1).items():
            out[pre + k_] = t_.numpy()
        out[pre + "key_none"] = mm(gs, q, kc, ks, km, bits, [None], [None], type="key").numpy()
        out[pre + "key_prefill"] = mm(gs, q, kc[..., :Tp // fpi].contiguous(), ks[..., :Tp // gs].contiguous(),
                                      km[..., :Tp // gs].contiguous(), bits, [kp0], [kq0], type="key").numpy()
        out[pre + "key_stacked"] = mm(gs, q, kc, ks, km, bits, [kp0, kp1], [kq0, kq1], type="key").numpy()
        out[pre + "value_none"] = mm(gs, a, vc, vs, vm, bits, [None], [None], type="value").numpy()
        ap = torch.softmax(torch.randn(1, H, 1, Tp, generator=g), -1).half()
        out[pre + "a_prefill"] = ap.numpy()
        out[pre + "value_prefill"] = mm(gs, ap, vc[:, :, :Tp].contiguous(), vs[:, :, :Tp].contiguous(), vm[:, :, :Tp].contiguous(),
                                        bits, [vp0], [vq0], type="value").numpy()
        out[pre + "value_stacked"] = mm(gs, a, vc, vs, vm, bits, [vp0, vp1], [vq0, vq1], type="value").numpy()
        print("matmul_withlrap bits", bits, "ok", flush=True)


if __name__ == "__main__":
    assert os.path.isdir(REF), "the reference tree is only present in the build container"
    only = sys.argv[1:]
    with torch.no_grad():
        if not only or "mm" in only:
            d = {}
            run_matmul(d)
            np.savez_compressed(os.path.join(HERE, "f8_ref_matmul.npz"), **d)
        for name in CASES:
            if only and name not in only:
                continue
            d = {}
            run_case(name, d)
            np.savez_compressed(os.path.join(HERE, f"f8_ref_{name}.npz"), **d)
