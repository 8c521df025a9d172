"""GPU tests of the fused decode-attention kernel over compressed payloads (gear_amd/csrc/attention.hip):
result == attention over the explicitly reconstructed cache (float64 on the host), for every payload flavour."""
import math

import numpy as np
import pytest
import torch

from conftest import rel_fro
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def host(t):
    return None if t is None else t.detach().cpu().numpy()


def reconstruct(p):
    """Payload -> float64 [B,H,T,D]: dequant + Q P^T, outlier entries = value + Q P^T (no fp16 rounding)."""
    B, H, T, D = p.shape
    code, scale, mn = host(p.code), host(p.scale).astype(np.float64), host(p.mn).astype(np.float64)
    if p.kind == "v":
        q = orc.unpack_tensor(code, p.bits, 3).astype(np.float64)                       # [B,H,T,D]
        deq = q * np.repeat(scale, p.group, axis=3) + np.repeat(mn, p.group, axis=3)
    else:
        q = orc.unpack_tensor(code, p.bits, 3).astype(np.float64)                       # [B,H,D,T]
        deq = (q * np.repeat(scale, p.group, axis=3) + np.repeat(mn, p.group, axis=3)).transpose(0, 1, 3, 2)
    lr = 0.0
    if p.P is not None:
        lr = host(p.Q).astype(np.float64) @ host(p.P).astype(np.float64).transpose(0, 1, 3, 2)
    rec = deq + lr
    if p.oidx is not None:
        oi, ov = host(p.oidx).astype(np.int64), host(p.oval).astype(np.float64)
        if p.kind == "v":      # rows (b,t) over columns h*D+d
            for b in range(B):
                for t in range(T):
                    h, d = oi[b, t] // D, oi[b, t] % D
                    base = ov[b, t]
                    rec[b, h, t, d] = base + (lr[b, h, t, d] if p.P is not None else 0.0)
        else:                  # rows (b,h,d) over tokens
            for b in range(B):
                for h in range(H):
                    for d in range(D):
                        t = oi[b, h, d]
                        rec[b, h, t, d] = ov[b, h, d] + (lr[b, h, t, d] if p.P is not None else 0.0)
    return rec


def ref_attention(q, khat, vhat, kwin, vwin, n_rep):
    q64 = q.astype(np.float64)[:, :, 0]                       # [B,Hq,D]
    K = np.repeat(khat, n_rep, axis=1)
    V = np.repeat(vhat, n_rep, axis=1)
    if kwin is not None:
        K = np.concatenate([K, np.repeat(kwin.astype(np.float64), n_rep, axis=1)], axis=2)
        V = np.concatenate([V, np.repeat(vwin.astype(np.float64), n_rep, axis=1)], axis=2)
    s = np.einsum("bhd,bhtd->bht", q64, K) / math.sqrt(q.shape[-1])
    s -= s.max(-1, keepdims=True)
    a = np.exp(s)
    a /= a.sum(-1, keepdims=True)
    return np.einsum("bht,bhtd->bhd", a, V)[:, :, None]


CASES = [
    # B, Hq, Hkv, T, W, bits, mode, rank, k_out
    (1, 4, 4, 256, 0, 2, "fp32", 4, 3),
    (2, 4, 2, 512, 17, 4, "fp32", 8, 5),
    (1, 8, 8, 4096, 63, 2, "fp32", 8, 40),
    (1, 2, 2, 192, 8, 2, "fp16", 2, 0),
    (1, 4, 4, 1024, 0, 4, "fp16", 0, 0),
    (1, 2, 2, 2112, 1, 2, "fp32", 16, 10),
    (1, 8, 1, 1024, 5, 2, "fp32", 4, 2),
    (1, 4, 4, 2048, 33, 4, "fp32", 4, 20),      # BASELINE configs[1]: 4 bits, rank 4 (8-byte factor rows on the short-chunk kernel)
]


@pytest.fixture(params=["planned", "generic"])
def attn_kernel(request):
    """planned: contexts <= 8k run the 128-token-chunk kernel (every load issued up front); generic: force the
    variable-chunk kernel that longer contexts use, on the same cases."""
    from gear_amd import _lib as L
    L.set_option("attn_generic", 1 if request.param == "generic" else 0)
    yield request.param
    L.set_option("attn_generic", 0)


@pytest.mark.parametrize("B,Hq,Hkv,T,W,bits,mode,rank,k_out", CASES + [(1, 2, 1, 8320, 3, 2, "fp32", 8, 20),
                                                                   (1, 8, 1, 8192, 5, 2, "fp32", 16, 20),   # 70B-style GQA head, rank 16
                                                                   (1, 2, 2, 16384, 7, 2, "fp32", 8, 40)])  # the longest context the cache takes
def test_fused_decode_attention(attn_kernel, B, Hq, Hkv, T, W, bits, mode, rank, k_out):
    from gear_amd import compress as C
    from gear_amd.attention import decode_attention
    torch.manual_seed(61)
    D = 128
    k = (torch.randn(B, Hkv, T, D) * (1 + 2 * (torch.rand(1, Hkv, 1, D) > 0.95))).half().cuda()
    v = torch.randn(B, Hkv, T, D).half().cuda()
    q = torch.randn(B, Hq, 1, D).half().cuda()
    kw = torch.randn(B, Hkv, W, D).half().cuda() if W else None
    vw = torch.randn(B, Hkv, W, D).half().cuda() if W else None
    pk = C.compress_key(k, bits, 64, k_out=k_out, rank=rank, loop=3, mode=mode)
    pv = C.compress_value(v, bits, 64, k_out=k_out, rank=rank, loop=3, mode=mode)
    out, lse = decode_attention(q, pk, pv, kw, vw, return_lse=True)
    ref = ref_attention(host(q), reconstruct(pk), reconstruct(pv), host(kw), host(vw), Hq // Hkv)
    err = rel_fro(host(out).astype(np.float64), ref)
    assert err < 2e-3, err
    assert torch.isfinite(lse).all()


def test_window_only_and_matches_module_semantics():
    """No compressed tokens yet (short prompt): plain fp16 attention over the window."""
    from gear_amd.attention import decode_attention
    torch.manual_seed(62)
    q = torch.randn(2, 4, 1, 128).half().cuda()
    kw, vw = torch.randn(2, 2, 40, 128).half().cuda(), torch.randn(2, 2, 40, 128).half().cuda()
    out = decode_attention(q, None, None, kw, vw)
    ref = ref_attention(host(q), np.zeros((2, 2, 0, 128)), np.zeros((2, 2, 0, 128)), host(kw), host(vw), 2)
    assert rel_fro(host(out).astype(np.float64), ref) < 2e-3


def test_outlier_chunk_index_matches_searchsorted():
    """gear_outlier_chunk_index (through Payload.chunk_index) == numpy searchsorted on every sorted outlier list."""
    from gear_amd import compress as C
    torch.manual_seed(63)
    B, H, T, D, k = 1, 4, 512, 128, 6
    x = torch.randn(B, H, T, D).half().cuda()
    pk = C.compress_key(x, 2, 64, k_out=k)
    pv = C.compress_value(x, 2, 64, k_out=k)
    for p, nb in ((pk, T // 128 + 1), (pv, H + 1)):
        idx = host(p.oidx).astype(np.int64).reshape(-1, k) & 0xFFFF
        tab = host(p.chunk_index())
        assert tab.shape == (idx.shape[0], nb) and tab.dtype == np.uint8
        bounds = np.arange(nb) * 128
        ref = np.stack([np.searchsorted(row, bounds, side="left") for row in idx])
        assert np.array_equal(tab, ref)
    assert p.chunk_index() is p.chunk_index()          # cached


@pytest.mark.parametrize("Hq,Hkv,T,W,bits,rank,k_out", [(4, 2, 640, 9, 2, 8, 6), (8, 2, 1024, 0, 4, 8, 4), (8, 1, 2048, 33, 2, 16, 12),
                                                    (16, 2, 512, 64, 2, 0, 0)])
def test_gqa_group_in_one_workgroup_equals_one_workgroup_per_query_head(Hq, Hkv, T, W, bits, rank, k_out):
    """Round 5 (option attn_gqa_group): the short-chunk kernel serves the 2 / 4 / 8 query heads of a KV head from ONE workgroup (the
    chunk's payload loaded once).  Same registers, same order of operations per head as one workgroup per query head (the default:
    measured faster): outputs equal up to the order of the float atomics of the outlier terms; without outliers bit for bit."""
    from gear_amd import _lib as L
    from gear_amd import compress as C
    from gear_amd.attention import decode_attention
    torch.manual_seed(64)
    B, D = 2, 128
    k = torch.randn(B, Hkv, T, D).half().cuda()
    v = torch.randn(B, Hkv, T, D).half().cuda()
    q = torch.randn(B, Hq, 1, D).half().cuda()
    kw = torch.randn(B, Hkv, W, D).half().cuda() if W else None
    vw = torch.randn(B, Hkv, W, D).half().cuda() if W else None
    pk = C.compress_key(k, bits, 64, k_out=k_out, rank=rank, loop=3, mode="fp32")
    pv = C.compress_value(v, bits, 64, k_out=k_out, rank=rank, loop=3, mode="fp32")
    out_1, lse_1 = decode_attention(q, pk, pv, kw, vw, return_lse=True)
    L.set_option("attn_gqa_group", 1)
    try:
        out_g, lse_g = decode_attention(q, pk, pv, kw, vw, return_lse=True)
    finally:
        L.set_option("attn_gqa_group", 0)
    if k_out == 0:
        assert torch.equal(out_g, out_1) and torch.equal(lse_g, lse_1)
    else:
        assert rel_fro(host(out_g).astype(np.float64), host(out_1).astype(np.float64)) < 1e-4
    ref = ref_attention(host(q), reconstruct(pk), reconstruct(pv), host(kw), host(vw), Hq // Hkv)
    assert rel_fro(host(out_g).astype(np.float64), ref) < 2e-3


@pytest.mark.parametrize("B,Hq,Hkv,T,tcap", [(1, 4, 4, 4096, 4096), (2, 8, 2, 1000, 1152), (1, 8, 1, 8192, 8192), (3, 2, 2, 77, 128),
                                             (1, 6, 2, 300, 300)])
def test_fp16_cache_attention_baseline(B, Hq, Hkv, T, tcap):
    """gear_attn_decode_f16: the UNCOMPRESSED baseline bench.py times beside the compressed cache (the reference harness's model
    "None", test.py:41-62) == float64 attention over the first T tokens of the fp16 cache; ragged last chunk, capacity > T, grouped
    (2 / 4 / 8) and ungrouped (3) query-head ratios."""
    from gear_amd.attention import decode_attention_f16
    torch.manual_seed(65)
    k = torch.randn(B, Hkv, tcap, 128).half().cuda()
    v = torch.randn(B, Hkv, tcap, 128).half().cuda()
    q = torch.randn(B, Hq, 1, 128).half().cuda()
    out, lse = decode_attention_f16(q, k, v, T, return_lse=True)
    ref = ref_attention(host(q), host(k)[:, :, :T].astype(np.float64), host(v)[:, :, :T].astype(np.float64), None, None, Hq // Hkv)
    assert rel_fro(host(out).astype(np.float64), ref) < 1e-3
    s = np.einsum("bhd,bhtd->bht", host(q).astype(np.float64)[:, :, 0], np.repeat(host(k)[:, :, :T].astype(np.float64), Hq // Hkv, 1)) / math.sqrt(128)
    ref_lse = np.log(np.exp(s - s.max(-1, keepdims=True)).sum(-1)) + s.max(-1)
    assert np.allclose(host(lse), ref_lse, rtol=0, atol=2e-3)


@pytest.mark.parametrize("Hq,Hkv,T,W,rank,k_out", [(4, 4, 512, 1, 8, 4), (4, 2, 1024, 64, 8, 0), (8, 1, 384, 100, 16, 6), (2, 2, 4096, 37, 0, 0)])
def test_window_as_one_more_chunk_equals_window_in_the_reduce_kernel(Hq, Hkv, T, W, rank, k_out):
    """Option attn_win_chunk: the fp16 window as one more chunk of the short-chunk kernel's split (its own workgroup in the same
    launch; the reduce kernel only merges: 1 = always, 0 = when the vector short-chunk kernel runs -- the default since the head
    became the fastest grid dimension) against -1 = window scores and values inside the reduce kernel (rounds 1 - 4).  Same softmax,
    merged in a different order: equal to fp32 rounding."""
    from gear_amd import _lib as L
    from gear_amd import compress as C
    from gear_amd.attention import decode_attention
    torch.manual_seed(66)
    B, D = 2, 128
    k = torch.randn(B, Hkv, T, D).half().cuda()
    v = torch.randn(B, Hkv, T, D).half().cuda()
    q = torch.randn(B, Hq, 1, D).half().cuda()
    kw, vw = torch.randn(B, Hkv, W, D).half().cuda(), torch.randn(B, Hkv, W, D).half().cuda()
    pk = C.compress_key(k, 2, 64, k_out=k_out, rank=rank, loop=3, mode="fp32")
    pv = C.compress_value(v, 2, 64, k_out=k_out, rank=rank, loop=3, mode="fp32")
    L.set_option("attn_win_chunk", -1)
    try:
        out_r, lse_r = decode_attention(q, pk, pv, kw, vw, return_lse=True)
        L.set_option("attn_win_chunk", 1)
        out_c, lse_c = decode_attention(q, pk, pv, kw, vw, return_lse=True)
    finally:
        L.set_option("attn_win_chunk", 0)
    out_d, lse_d = decode_attention(q, pk, pv, kw, vw, return_lse=True)          # the default rule takes one of the two
    assert rel_fro(host(out_d).astype(np.float64), host(out_r).astype(np.float64)) < 5e-4 and torch.allclose(lse_d, lse_r, rtol=0, atol=1e-4)
    assert rel_fro(host(out_c).astype(np.float64), host(out_r).astype(np.float64)) < 5e-4      # (one fp16 rounding of the output)
    assert torch.allclose(lse_c, lse_r, rtol=0, atol=1e-4)
    ref = ref_attention(host(q), reconstruct(pk), reconstruct(pv), host(kw), host(vw), Hq // Hkv)
    assert rel_fro(host(out_c).astype(np.float64), ref) < 2e-3
