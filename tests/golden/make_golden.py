#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by EXECUTING the reference implementation.

Runs ONLY in the build container (needs /root/reference, read-only).  The reference never travels to
the GPU box; these fixtures (pure data: inputs + the reference's outputs) do.  Nothing here is product
code and no reference source text is stored -- only arrays.

Usage:  python tests/golden/make_golden.py          (CPU only, ~1-2 minutes)

Fixture map:
  f1_quant_pack.npz   a1/a3  quant_and_pack_{v,k}cache, unpack_and_dequant_*, one Triton-interpreted a1 case
  f2_witherror.npz    a2     triton_quantize_and_pack_along_last_dim_witherror (error / scale / mn)
  f3_lowrank.npz      a4/a10 headwise_lrap and fake_poweriteration_group with the drawn P0 captured
  f4_fakequant.npz    a9     token / channel fake quantizers (fp16 and fp32 arithmetic)
  f5_outlier.npz      a11    gears_channelQ / gears_tokenQ (+ an explicit tie case)
  f6_insert.npz       a12    compress_insert_function for KIVI_V2 / GEARL / GEAR
  f10_ragged.npz      a11/a12 gears_channelQ and method GEAR at sequence lengths that are not a multiple of the group
  f9_kcvt.npz         a12    compress_insert_function for KCVT / GEAR-KCVT / GEARL-KCVT and the token_preserving window
  f7_gemv.npz         a6     inp @ dequant_weight_outer recipe of CSG/quant/gemv.py:93-126 (MHA + MQA)
"""
import importlib.util
import os
import sys

os.environ["TRITON_INTERPRET"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

sys.path.insert(0, os.path.join(REF, "cuda_supported_gear", "quant"))
import new_pack  # noqa: E402  (a1-a4)


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


SIM = os.path.join(REF, "GenerationBench", "GenerationTest", "GEARLM", "Simulated")
cf = _load("ref_compress_function", os.path.join(SIM, "compress_function.py"))
cc = _load("ref_compress_config", os.path.join(SIM, "compress_config.py"))


def randn_half(seed, shape, scale=1.0):
    torch.manual_seed(seed)
    return (torch.randn(shape) * scale).half()


def npy(t):
    return t.detach().cpu().numpy()


class RandCapture:
    """Records every tensor torch.rand returns (the reference draws P0/Q0 on the CPU generator)."""

    def __enter__(self):
        self.orig = torch.rand
        self.drawn = []

        def rec(*a, **k):
            t = self.orig(*a, **k)
            self.drawn.append(t.clone())
            return t

        torch.rand = rec
        return self

    def __exit__(self, *exc):
        torch.rand = self.orig


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrs)} arrays")


# ------------------------------------------------------------------------------------------- F1
def f1():
    d = {}
    xk = randn_half(0, (1, 2, 128, 256))   # "K layout" for a1: [B,H,D,T], groups along T
    xv = randn_half(1, (1, 2, 256, 128))   # "V layout": [B,H,T,D], groups along D
    d["xk"], d["xv"] = npy(xk), npy(xv)
    for tag, x in (("k", xk), ("v", xv)):
        for g in (32, 64, 128):
            for b in (2, 4):
                code, scale, mn = new_pack.quant_and_pack_vcache(x.clone(), g, b)
                deq = new_pack.unpack_and_dequant_vcache(code, scale, mn, g, b)
                key = f"last_{tag}_g{g}_b{b}"
                d[key + "_code"], d[key + "_scale"], d[key + "_mn"] = npy(code), npy(scale), npy(mn)
                d[key + "_deq"] = npy(deq)
    # token-major K tile, packed along T (quant_and_pack_kcache)
    for g in (32, 64, 128):
        for b in (2, 4):
            code, scale, mn = new_pack.quant_and_pack_kcache(xv.clone(), g, b)
            deq = new_pack.unpack_and_dequant_kcache(code, scale, mn, g, b)
            key = f"kc_g{g}_b{b}"
            d[key + "_code"], d[key + "_scale"], d[key + "_mn"] = npy(code), npy(scale), npy(mn)
            d[key + "_deq"] = npy(deq)
    # the Triton function itself (interpreter mode; small because it is slow)
    xt = randn_half(2, (1, 2, 16, 256))
    d["xt"] = npy(xt)
    for b in (2, 4):
        code, scale, mn = new_pack.triton_quantize_and_pack_along_last_dim(xt.clone(), 64, b)
        d[f"tri_b{b}_code"], d[f"tri_b{b}_scale"], d[f"tri_b{b}_mn"] = npy(code), npy(scale), npy(mn)
    # raw pack / unpack
    torch.manual_seed(3)
    raw = torch.randint(0, 16, (1, 2, 32, 64), dtype=torch.int32)
    d["raw4"] = npy(raw)
    d["raw4_pack2"] = npy(new_pack.pack_tensor(raw, 4, 2))
    d["raw4_pack3"] = npy(new_pack.pack_tensor(raw, 4, 3))
    d["raw4_unpack3"] = npy(new_pack.unpack_tensor(new_pack.pack_tensor(raw, 4, 3), 4, 3))
    raw2 = raw & 3
    d["raw2_pack3"] = npy(new_pack.pack_tensor(raw2, 2, 3))
    d["raw2_pack2"] = npy(new_pack.pack_tensor(raw2, 2, 2))
    save("f1_quant_pack.npz", **d)


# ------------------------------------------------------------------------------------------- F2
def f2():
    d = {}
    x = randn_half(4, (1, 2, 16, 256))
    d["x"] = npy(x)
    for b in (2, 4):
        code, scale, mn, err = new_pack.triton_quantize_and_pack_along_last_dim_witherror(x.clone(), 64, b)
        d[f"b{b}_refcode_B1"] = npy(code)   # reference's (mostly zero) code: documents defect B1
        d[f"b{b}_scale"], d[f"b{b}_mn"], d[f"b{b}_err"] = npy(scale), npy(mn), npy(err)
    save("f2_witherror.npz", **d)


# ------------------------------------------------------------------------------------------- F3
def f3():
    d = {}
    torch.manual_seed(5)
    E = (torch.randn(1, 2, 256, 128) * 0.1)
    # give E a decaying spectrum so that the rank-r part is well defined
    U = torch.randn(1, 2, 256, 6)
    V = torch.randn(1, 2, 128, 6)
    E = E + (U * torch.tensor([3.0, 2.0, 1.5, 1.0, 0.7, 0.5])) @ V.transpose(2, 3) * 0.05
    d["E"] = npy(E.float())
    Eh = E.half()
    d["Eh"] = npy(Eh)
    for r in (2, 4, 8, 16):
        for loop in (1, 3):
            torch.manual_seed(100 + r + loop)
            with RandCapture() as rc:
                rec = cf.fake_poweriteration_group(E.clone(), loop, r, "cpu", None, None)
            d[f"pi_r{r}_l{loop}_P0"] = npy(rc.drawn[0])
            d[f"pi_r{r}_l{loop}_rec"] = npy(rec)
            torch.manual_seed(200 + r + loop)
            with RandCapture() as rc:
                P, Q = new_pack.headwise_lrap(Eh.clone(), r, loop)
            d[f"lrap_r{r}_l{loop}_P0"] = npy(rc.drawn[0])
            d[f"lrap_r{r}_l{loop}_P"] = npy(P)
            d[f"lrap_r{r}_l{loop}_Q"] = npy(Q)
    save("f3_lowrank.npz", **d)


# ------------------------------------------------------------------------------------------- F4
def f4():
    d = {}
    x = randn_half(6, (1, 4, 256, 128))
    d["x"] = npy(x)
    for b in (2, 4):
        for g in (32, 64, 128):
            d[f"tok_b{b}_g{g}"] = npy(cf.fake_groupwise_token_asymmetric_quantization(x.clone(), b, g))
            d[f"chan_fp16_b{b}_g{g}"] = npy(cf.fake_groupwise_channel_asymmetric_quantization_new(x.clone(), b, g))
            d[f"chan_fp32_b{b}_g{g}"] = npy(
                cf.fake_groupwise_channel_asymmetric_quantization_new(x.clone().float(), b, g))
        # cluster variants, fp32, incl. an unquantized tail (T=256, g=96 -> 64-token tail)
        d[f"chan_cluster_b{b}_g96"] = npy(
            cf.fake_groupwise_channel_asymmetric_quantization_cluster(x.clone().float(), b ** 2 - 1, 96))
        d[f"tok_cluster_b{b}_g64"] = npy(
            cf.fake_groupwise_token_asymmetric_quantization_cluster(x.clone().float(), b ** 2 - 1, 64))
    save("f4_fakequant.npz", **d)


# ------------------------------------------------------------------------------------------- F5
def _boundary_ties(rows, k):
    """True if any row has a tie across the top-k / bottom-k selection boundary."""
    s = np.sort(rows, axis=1)
    return bool(np.any(s[:, k - 1] == s[:, k]) or np.any(s[:, -k] == s[:, -k - 1]))


def _detie_rows(rows, k):
    """Nudge (by one fp16 ulp, inwards) the first UNSELECTED element of every row whose top-k / bottom-k
    boundary is tied.  rows: fp16 [R, len] (modified in place).  Returns the number of nudges."""
    n = 0
    order = np.argsort(rows.astype(np.float32), axis=1, kind="stable")
    for r in range(rows.shape[0]):
        o = order[r]
        if rows[r, o[k - 1]] == rows[r, o[k]]:
            rows[r, o[k]] = np.nextafter(rows[r, o[k]], np.float16(np.inf))
            n += 1
        if rows[r, o[-k]] == rows[r, o[-k - 1]]:
            rows[r, o[-k - 1]] = np.nextafter(rows[r, o[-k - 1]], np.float16(-np.inf))
            n += 1
    return n


def detie(xn, ks, chan=True, tok=True):
    """Remove ties ACROSS the top-k / bottom-k boundary of every channel row (length T) and/or token row
    (length H*D) of xn [B,H,T,D] fp16, for every k in ks.  torch.topk's tie order is implementation-defined,
    so the strict-parity fixtures carry no such tie (random fp16 data has them in a few % of rows)."""
    xn = xn.copy()
    B, H, T, D = xn.shape
    for _ in range(200):
        n = 0
        for k in ks:
            if chan:
                rows = np.ascontiguousarray(xn.transpose(0, 1, 3, 2)).reshape(B * H * D, T)
                n += _detie_rows(rows, k)
                xn = np.ascontiguousarray(rows.reshape(B, H, D, T).transpose(0, 1, 3, 2))
            if tok:
                rows = np.ascontiguousarray(xn.transpose(0, 2, 1, 3)).reshape(B * T, H * D)
                n += _detie_rows(rows, k)
                xn = np.ascontiguousarray(rows.reshape(B, T, H, D).transpose(0, 2, 1, 3))
        if n == 0:
            break
    assert n == 0, "could not remove boundary ties"
    for k in ks:
        if chan:
            assert not _boundary_ties(xn.transpose(0, 1, 3, 2).reshape(-1, T).astype(np.float32), k)
        if tok:
            assert not _boundary_ties(xn.transpose(0, 2, 1, 3).reshape(T, -1).astype(np.float32), k)
    return xn


def outlier_k(numel, B, T, s):
    return int(int(numel * s) / B / T / 2)


def f5():
    d = {}
    xn = npy(randn_half(7, (1, 4, 256, 128)))
    B, H, T, D = xn.shape
    xn = detie(xn, [outlier_k(xn.size, B, T, s) for s in (0.01, 0.02)])
    x = torch.from_numpy(xn)
    d["x"] = npy(x)
    for s in (0.01, 0.02):
        for b in (2, 4):
            tag = f"s{int(s * 100)}_b{b}"
            d[f"chan_{tag}"] = npy(cf.gears_channelQ(x.clone(), b, 64, s))
            d[f"tok_{tag}"] = npy(cf.gears_tokenQ(x.clone(), b, 64, s))
    # explicit tie case: duplicate the extreme values inside rows
    xt = x.clone()
    xt[0, 0, 10, :] = 0.25
    xt[0, 0, 10, 5] = 3.0
    xt[0, 0, 10, 77] = 3.0          # tie among the largest of token row 10 (heads share the row!)
    xt[0, 1, 10, 9] = 3.0
    xt[0, 0, :, 3] = -0.5
    xt[0, 0, 17, 3] = -4.0
    xt[0, 0, 99, 3] = -4.0          # tie among the smallest of channel (h0,d3)
    xt[0, 0, 200, 3] = -4.0
    d["x_tie"] = npy(xt)
    d["chan_tie_s2_b2"] = npy(cf.gears_channelQ(xt.clone(), 2, 64, 0.02))
    d["tok_tie_s2_b2"] = npy(cf.gears_tokenQ(xt.clone(), 2, 64, 0.02))
    save("f5_outlier.npz", **d)


# ------------------------------------------------------------------------------------------- F6
def _cfg(method, bits, g, rank, loop, left):
    c = cc.CompressionConfig(compress_method=method, attention_number=1, quantize_bit=bits, group_size=g,
                             rank=rank, rankv=rank, prefill_rank=rank, prefill_rankv=rank, loop=loop, left=left)
    c.copy_for_all_attention()
    return c


def f6():
    d = {}
    ks = [outlier_k(4 * 256 * 128, 1, 256, s) for s in (0.01, 0.02)]
    k = torch.from_numpy(detie(npy(randn_half(8, (1, 4, 256, 128))), ks, chan=True, tok=False))
    v = torch.from_numpy(detie(npy(randn_half(9, (1, 4, 256, 128))), ks, chan=False, tok=True))
    d["k"], d["v"] = npy(k), npy(v)
    cases = [("KIVI_V2", 4, 64, 0, 0, 0.0), ("KIVI_V2", 2, 64, 0, 0, 0.0),
             ("GEARL", 2, 64, 4, 3, 0.0), ("GEARL", 4, 64, 8, 3, 0.0),
             ("GEAR", 4, 64, 4, 3, 0.01), ("GEAR", 2, 64, 8, 3, 0.02)]
    for (m, b, g, r, loop, left) in cases:
        tag = f"{m}_b{b}_r{r}"
        torch.manual_seed(300 + b + r)
        with RandCapture() as rc:
            ko, vo = cf.compress_insert_function(k.clone(), v.clone(), _cfg(m, b, g, r, loop, left), 0, prefill=True)
        d[tag + "_k"], d[tag + "_v"] = npy(ko), npy(vo)
        if rc.drawn:
            d[tag + "_P0k"], d[tag + "_P0v"] = npy(rc.drawn[0]), npy(rc.drawn[2])
    save("f6_insert.npz", **d)


# ------------------------------------------------------------------------------------------- F7
def f7():
    """Inputs built the way CSG/quant/gemv.py:93-126 builds them (IC=739 odd, OC=128, GS=32), expected
    output = inp @ dequantized weight.  Weight/scale/zero are stored in the layout handed to the CUDA
    kernel ([BS, OC/pack, IC], [BS, OC/g, IC])."""
    d = {}
    B, nh, IC, OC, GS = 2, 4, 739, 128, 32
    for name, wb in (("mha", B * nh), ("mqa", B)):
        torch.manual_seed(10)
        inp = torch.randn((B * nh, 1, IC)).half()
        w0 = torch.randn((wb, IC, OC)).half()
        d[f"{name}_inp"] = npy(inp)
        for bit in (2, 4):
            maxq = 2 ** bit - 1
            w = w0.view(wb, IC, OC // GS, GS)
            mx = torch.max(w, dim=-1)[0]
            mn = torch.min(w, dim=-1)[0]
            scale = (mx - mn) / maxq
            q = (w - mn.unsqueeze(-1))
            q.div_(scale.unsqueeze(-1))
            q = q.clamp_(0, maxq).round_().to(torch.int32).view(wb, IC, OC)
            qw = new_pack.pack_tensor(q, bit, 2)
            # dequant exactly as gemv.py's dequant_weight_outer: half arithmetic
            deq = (q.half().view(wb, IC, OC // GS, GS) * scale.unsqueeze(-1) + mn.unsqueeze(-1)).view(wb, IC, OC)
            if name == "mha":
                ref = inp.float() @ deq.float()
            else:
                ref = (inp.float().view(B, nh, 1, IC) @ deq.float().view(B, 1, IC, OC)).view(B * nh, 1, OC)
            d[f"{name}_b{bit}_qw"] = npy(qw.transpose(1, 2).contiguous())       # [BS, OC/pack, IC]
            d[f"{name}_b{bit}_scale"] = npy(scale.transpose(1, 2).contiguous())  # [BS, OC/g, IC]
            d[f"{name}_b{bit}_mn"] = npy(mn.transpose(1, 2).contiguous())
            d[f"{name}_b{bit}_ref"] = npy(ref)
    d["dims"] = np.array([B, nh, IC, OC, GS])
    save("f7_gemv.npz", **d)


# ------------------------------------------------------------------------------------------- F9
def f9():
    """The dispatcher's remaining methods: groups spanning the whole sequence (K) / all heads (V), and the token_preserving
    window of the KIVI_V2 branch.  T = 192 (not a power of two) on purpose."""
    d = {}
    B, H, T, D = 1, 4, 192, 128
    ks = [outlier_k(B * H * T * D, B, T, 0.02)]
    k = torch.from_numpy(detie(npy(randn_half(18, (B, H, T, D))), ks, chan=True, tok=False))
    v = torch.from_numpy(detie(npy(randn_half(19, (B, H, T, D))), ks, chan=False, tok=True))
    d["k"], d["v"] = npy(k), npy(v)
    cases = [("KCVT", 4, 0, 0.0), ("KCVT", 2, 0, 0.0), ("GEAR-KCVT", 2, 8, 0.02), ("GEAR-KCVT", 4, 4, 0.0),
             ("GEARL-KCVT", 4, 4, 0.0), ("GEARL-KCVT", 2, 8, 0.0)]
    for (m, b, r, left) in cases:
        tag = f"{m}_b{b}_r{r}"
        torch.manual_seed(900 + b + r)
        with RandCapture() as rc:
            ko, vo = cf.compress_insert_function(k.clone(), v.clone(), _cfg(m, b, 64, r, 3, left), 0, prefill=True)
        d[tag + "_k"], d[tag + "_v"] = npy(ko), npy(vo)
        if rc.drawn:
            d[tag + "_P0k"], d[tag + "_P0v"] = npy(rc.drawn[0]), npy(rc.drawn[2])
    # token_preserving: KIVI_V2 on the window [int(0.25 T) : -int(0.25 T)] (T = 256 -> tokens 64..191) and the empty window
    k2, v2 = randn_half(20, (1, 2, 256, 128)), randn_half(21, (1, 2, 256, 128))
    d["k_tp"], d["v_tp"] = npy(k2), npy(v2)
    for name, ss, ls in (("tp_25_25", 0.25, 0.25), ("tp_50_0", 0.5, 0.0)):
        c = _cfg("KIVI_V2", 4, 64, 0, 0, 0.0)
        c.token_preserving, c.start_saving, c.locality_saving = [True], [ss], [ls]
        ko, vo = cf.compress_insert_function(k2.clone(), v2.clone(), c, 0, prefill=True)
        d[name + "_k"], d[name + "_v"] = npy(ko), npy(vo)
    save("f9_kcvt.npz", **d)


# ------------------------------------------------------------------------------------------- F10
def f10():
    """Sequence lengths that are NOT a multiple of the group (every real prompt of the simulated path): gears_channelQ leaves the
    T mod g tail of a channel unquantized (compress_function.py:107-122) while selecting outliers and taking the fill mean over all
    T tokens; method GEAR through the dispatcher on such a tensor (K and V, P0 captured)."""
    d = {}
    for (T, g) in ((200, 64), (150, 32), (70, 64)):
        B, H, D = 1, 2, 128
        ks = [outlier_k(B * H * T * D, B, T, s) for s in (0.02, 0.05)]
        xn = detie(npy(randn_half(30 + T, (B, H, T, D))), ks, chan=True, tok=False)
        x = torch.from_numpy(xn)
        d[f"x_T{T}"] = xn
        for (s, b) in ((0.0, 2), (0.02, 2), (0.02, 4), (0.05, 4)):
            d[f"chan_T{T}_g{g}_s{int(s * 100)}_b{b}"] = npy(cf.gears_channelQ(x.clone(), b, g, s))
    B, H, T, D = 1, 4, 200, 128
    ks = [outlier_k(B * H * T * D, B, T, 0.02)]
    k = torch.from_numpy(detie(npy(randn_half(41, (B, H, T, D))), ks, chan=True, tok=False))
    v = torch.from_numpy(detie(npy(randn_half(42, (B, H, T, D))), ks, chan=False, tok=True))
    d["k"], d["v"] = npy(k), npy(v)
    for (b, r, left) in ((2, 8, 0.02), (4, 4, 0.02), (2, 4, 0.0)):
        tag = f"GEAR_b{b}_r{r}_s{int(left * 100)}"
        torch.manual_seed(1000 + b + r)
        with RandCapture() as rc:
            ko, vo = cf.compress_insert_function(k.clone(), v.clone(), _cfg("GEAR", b, 64, r, 3, left), 0, prefill=True)
        d[tag + "_k"], d[tag + "_v"] = npy(ko), npy(vo)
        d[tag + "_P0k"], d[tag + "_P0v"] = npy(rc.drawn[0]), npy(rc.drawn[2])
    save("f10_ragged.npz", **d)


if __name__ == "__main__":
    torch.set_num_threads(8)
    only = sys.argv[1:]
    for fn in (f1, f2, f3, f4, f5, f6, f7, f9, f10):
        if not only or fn.__name__ in only:
            fn()
