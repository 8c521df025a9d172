"""GPU parity tests (run with -m gpu on an MI355X): HIP quantizers / dequantizers / GEMV through the C ABI
versus the CPU oracle and the reference-generated golden fixtures.  Integer / fp16 payloads: bit-exact."""
import numpy as np
import pytest
import torch

from conftest import rel_fro
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

BITS = (2, 4)
GROUPS = (32, 64, 128)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def bits_eq(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.dtype == np.float16:
        a, b = a.view(np.uint16), b.view(np.uint16)
    if a.dtype == np.float32:
        a, b = a.view(np.uint32), b.view(np.uint32)
    bad = np.count_nonzero(a != b)
    assert bad == 0, f"{bad} / {a.size} elements differ"


def randn_half(seed, shape, scale=1.0):
    torch.manual_seed(seed)
    return (torch.randn(shape) * scale).half()


@pytest.fixture(scope="module")
def np_():
    from gear_amd.quant import new_pack
    return new_pack


@pytest.fixture(scope="module")
def mm_():
    from gear_amd.quant import matmul
    return matmul


# ------------------------------------------------------------------------------------------ golden (reference) vectors
@pytest.mark.parametrize("tag,xkey", [("k", "xk"), ("v", "xv")])
@pytest.mark.parametrize("g", GROUPS)
@pytest.mark.parametrize("b", BITS)
def test_f1_lastdim_golden(golden, np_, tag, xkey, g, b):
    f = golden("f1_quant_pack.npz")
    code, scale, mn = np_.triton_quantize_and_pack_along_last_dim(dev(f[xkey]), g, b)
    key = f"last_{tag}_g{g}_b{b}"
    bits_eq(host(code), f[key + "_code"])
    bits_eq(host(scale), f[key + "_scale"].reshape(scale.shape))
    bits_eq(host(mn), f[key + "_mn"].reshape(mn.shape))
    deq = np_.unpack_and_dequant_vcache(code, scale.unsqueeze(-1), mn.unsqueeze(-1), g, b)
    bits_eq(host(deq), f[key + "_deq"])
    # and through the reference's vcache entry point
    c2, s2, m2 = np_.quant_and_pack_vcache(dev(f[xkey]), g, b)
    bits_eq(host(c2), f[key + "_code"])
    assert s2.shape == f[key + "_scale"].shape and m2.shape == f[key + "_mn"].shape


@pytest.mark.parametrize("g", GROUPS)
@pytest.mark.parametrize("b", BITS)
def test_f1_kcache_golden(golden, np_, g, b):
    f = golden("f1_quant_pack.npz")
    code, scale, mn = np_.quant_and_pack_kcache(dev(f["xv"]), g, b)
    key = f"kc_g{g}_b{b}"
    bits_eq(host(code), f[key + "_code"])
    bits_eq(host(scale), f[key + "_scale"])
    bits_eq(host(mn), f[key + "_mn"])
    deq = np_.unpack_and_dequant_kcache(code, scale, mn, g, b)
    bits_eq(host(deq), f[key + "_deq"])


@pytest.mark.parametrize("b", BITS)
def test_f2_witherror_golden(golden, np_, b):
    f = golden("f2_witherror.npz")
    code, scale, mn, err = np_.triton_quantize_and_pack_along_last_dim_witherror(dev(f["x"]), 64, b)
    bits_eq(host(scale), f[f"b{b}_scale"])
    bits_eq(host(mn), f[f"b{b}_mn"])
    assert tuple(err.shape) == f[f"b{b}_err"].shape
    bits_eq(host(err), f[f"b{b}_err"])
    # B1: all columns packed == what a1 produces
    c1, _, _ = np_.triton_quantize_and_pack_along_last_dim(dev(f["x"]), 64, b)
    bits_eq(host(code), host(c1))


@pytest.mark.parametrize("name", ("mha", "mqa"))
@pytest.mark.parametrize("b", BITS)
def test_f7_gemv_golden(golden, mm_, name, b):
    f = golden("f7_gemv.npz")
    B, nh, IC, OC, GS = [int(v) for v in f["dims"]]
    nkv = nh if name == "mha" else 1
    inp = dev(f[f"{name}_inp"].reshape(B, nh, 1, IC))
    qw = dev(np.ascontiguousarray(f[f"{name}_b{b}_qw"].transpose(0, 2, 1)).reshape(B, nkv, IC, -1))
    sc = dev(np.ascontiguousarray(f[f"{name}_b{b}_scale"].transpose(0, 2, 1)).reshape(B, nkv, IC, -1))
    mn = dev(np.ascontiguousarray(f[f"{name}_b{b}_mn"].transpose(0, 2, 1)).reshape(B, nkv, IC, -1))
    out = mm_.cuda_bmm_fA_qB_outer(GS, inp, qw, sc, mn, b, mqa=(name == "mqa"))
    ref = f[f"{name}_b{b}_ref"].reshape(B, nh, 1, OC)
    assert rel_fro(host(out).astype(np.float32), ref) < 2e-3
    orc_out = orc.gemv_outer(host(inp), host(qw), host(sc), host(mn), GS, b)
    assert rel_fro(host(out).astype(np.float32), orc_out.astype(np.float32)) < 1e-3


@pytest.mark.parametrize("name", ("mha", "mqa"))
@pytest.mark.parametrize("b", BITS)
def test_f7_gemv_golden_on_the_extension_layout(golden, mm_, name, b):
    """gear_gemv_outer_dim = kivi_gemv.gemv_forward_cuda_outer_dim on ITS argument layout (gemv_cuda.h:13-21: kernel [BS', OC / pack,
    IC], scaling factors / zeros [BS', OC / group, IC], K innermost): the fixture's arrays exactly as the reference extension takes
    them, no transpose on the way in; against the fixture's output, the oracle and the native-layout entry point."""
    f = golden("f7_gemv.npz")
    B, nh, IC, OC, GS = [int(v) for v in f["dims"]]
    nkv = nh if name == "mha" else 1
    inp = dev(f[f"{name}_inp"].reshape(B * nh, 1, IC))
    qw, sc, mn = dev(f[f"{name}_b{b}_qw"]), dev(f[f"{name}_b{b}_scale"]), dev(f[f"{name}_b{b}_mn"])
    assert qw.shape[0] == B * nkv and qw.shape[2] == IC and sc.shape[2] == IC
    out = mm_.gemv_forward_cuda_outer_dim(inp, qw, sc, mn, b, GS, nh, name == "mqa")
    assert tuple(out.shape) == (B * nh, 1, OC)
    ref = f[f"{name}_b{b}_ref"].reshape(B * nh, 1, OC)
    assert rel_fro(host(out).astype(np.float32), ref) < 2e-3
    native = mm_.cuda_bmm_fA_qB_outer(GS, inp.reshape(B, nh, 1, IC), qw.transpose(1, 2).contiguous().reshape(B, nkv, IC, -1),
                                      sc.transpose(1, 2).contiguous().reshape(B, nkv, IC, -1),
                                      mn.transpose(1, 2).contiguous().reshape(B, nkv, IC, -1), b, mqa=(name == "mqa"))
    assert rel_fro(host(out).astype(np.float32), host(native).reshape(B * nh, 1, OC).astype(np.float32)) < 1e-3


def test_gemv_outer_dim_rejects_bad_arguments(mm_):
    from gear_amd import _lib as L
    a = torch.zeros((2, 3, 64), dtype=torch.float16, device="cuda")
    with pytest.raises(L.GearError):
        mm_.gemv_forward_cuda_outer_dim(a, torch.zeros((2, 4, 64), dtype=torch.int32, device="cuda"),
                                        torch.zeros((2, 1, 64), dtype=torch.float16, device="cuda"),
                                        torch.zeros((2, 1, 64), dtype=torch.float16, device="cuda"), 2, 64, 1, False)


# ------------------------------------------------------------------------------------------ oracle on seeded inputs
@pytest.mark.parametrize("shape", [(1, 4, 128, 1024), (2, 3, 64, 320), (1, 32, 128, 4096)])
@pytest.mark.parametrize("g,b", [(64, 2), (64, 4), (32, 2), (128, 4), (64, 8)])
@pytest.mark.parametrize("mode", ["fp16", "fp32"])
def test_lastdim_vs_oracle(np_, shape, g, b, mode):
    if shape[-1] % g:
        pytest.skip("shape not divisible")
    if shape[-1] == 4096 and (g, b) not in ((64, 2), (64, 4)):
        pytest.skip("big shape only for the headline configs")
    x = randn_half(11, shape)
    m = 0 if mode == "fp16" else 1
    code, scale, mn, err = np_.triton_quantize_and_pack_along_last_dim_witherror(x.cuda(), g, b, mode=mode)
    xin = x.numpy() if m == 0 else x.numpy().astype(np.float32)
    r = orc.quant_pack_lastdim(xin, g, b, mode=m, want_err=True, want_deq=True)
    bits_eq(host(code), r["code"])
    bits_eq(host(scale), r["scale"])
    bits_eq(host(mn), r["mn"])
    if m == 0:
        bits_eq(host(err).reshape(shape), r["err"])
    else:
        e = (x.numpy().astype(np.float32) - r["deq"].astype(np.float16).astype(np.float32)).astype(np.float16)
        bits_eq(host(err).reshape(shape), e)
    deq = np_.unpack_and_dequant_vcache(code, scale, mn, g, b, mode=mode)
    bits_eq(host(deq), orc.unpack_dequant_lastdim(r["code"], r["scale"], r["mn"], g, b, mode=m))


@pytest.mark.parametrize("shape", [(1, 4, 256, 128), (2, 2, 128, 64), (1, 32, 4096, 128)])
@pytest.mark.parametrize("g,b", [(64, 2), (64, 4), (32, 4), (128, 2)])
@pytest.mark.parametrize("mode", ["fp16", "fp32"])
def test_kcache_vs_oracle(np_, shape, g, b, mode):
    if shape[2] == 4096 and g != 64:
        pytest.skip("big shape only for the headline configs")
    x = randn_half(12, shape)
    m = 0 if mode == "fp16" else 1
    code, scale, mn = np_.quant_and_pack_kcache(x.cuda(), g, b, mode=mode)
    xin = x.numpy() if m == 0 else x.numpy().astype(np.float32)
    r = orc.quant_pack_k(xin, g, b, mode=m)
    bits_eq(host(code), r["code"])
    bits_eq(host(scale).reshape(r["scale"].shape), r["scale"])
    bits_eq(host(mn).reshape(r["mn"].shape), r["mn"])
    deq = np_.unpack_and_dequant_kcache(code, scale, mn, g, b, mode=mode)
    bits_eq(host(deq), orc.unpack_dequant_k(r["code"], r["scale"], r["mn"], g, b, mode=m))


def test_zero_range_groups_are_defined(np_):
    """Reference: 0/0 -> NaN codes (defect B6).  Build: code 0, scale 0, dequant == mn, error == 0."""
    x = randn_half(13, (1, 2, 8, 256))
    x[0, 0, 3, 64:128] = 1.5
    x[0, 1, :, :] = -0.25
    code, scale, mn, err = np_.triton_quantize_and_pack_along_last_dim_witherror(x.cuda(), 64, 2)
    r = orc.quant_pack_lastdim(x.numpy(), 64, 2, mode=0, want_err=True)
    bits_eq(host(code), r["code"])
    bits_eq(host(scale), r["scale"])
    assert not torch.isnan(err).any()
    assert host(code)[0, 1].any() == False  # noqa: E712
    deq = np_.unpack_and_dequant_vcache(code, scale, mn, 64, 2)
    assert torch.equal(deq[0, 1].cpu(), x[0, 1])


@pytest.mark.parametrize("b", BITS)
@pytest.mark.parametrize("kind", ["kside", "vside", "vside_gqa", "odd"])
def test_gemv_vs_oracle(np_, mm_, b, kind):
    torch.manual_seed(14)
    g = 64
    if kind == "kside":      # q . K^T : K = head_dim, N = tokens
        B, nh, nkv, K, N = 1, 8, 8, 128, 4096
    elif kind == "vside":    # A . V : K = tokens, N = head_dim
        B, nh, nkv, K, N = 1, 8, 8, 4096, 128
    elif kind == "vside_gqa":
        B, nh, nkv, K, N = 2, 8, 2, 1024, 128
    else:
        B, nh, nkv, K, N = 1, 3, 3, 739, 96
        g = 32
    w = randn_half(15, (B, nkv, K, N))
    code, scale, mn = np_.triton_quantize_and_pack_along_last_dim(w.cuda(), g, b)
    a = randn_half(16, (B, nh, 1, K))
    if kind.startswith("vside"):
        a = torch.softmax(a.float() * 3, dim=-1).half()
    out = mm_.cuda_bmm_fA_qB_outer(g, a.cuda(), code, scale, mn, b)
    ref, ref32 = orc.gemv_outer(a.numpy(), host(code), host(scale), host(mn), g, b, want32=True)
    got = host(out).astype(np.float32)
    assert rel_fro(got, ref32) < 1e-3
    # fp16 output: at most 1 ulp from the correctly rounded double-accumulated result
    ulp = np.maximum(np.abs(ref32), 2.0 ** -14) * 2.0 ** -10
    assert np.all(np.abs(got - ref32) <= 1.01 * ulp + 1e-6)


# ------------------------------------------------------------------------------------------ size-independent properties
@pytest.mark.parametrize("b", BITS)
def test_full_size_properties(np_, b):
    """BASELINE config sizes (Llama-2-7B, T=4096): |x - dequant| <= scale/2 (+fp16 rounding), codes in range,
    re-quantizing the dequantized tensor reproduces the codes (idempotence)."""
    B, H, T, D, g = 1, 32, 4096, 128, 64
    x = randn_half(17, (B, H, T, D)).cuda()
    code, scale, mn = np_.quant_and_pack_vcache(x, g, b)
    deq = np_.unpack_and_dequant_vcache(code, scale, mn, g, b)
    s = scale.float().expand(B, H, T, D // g, g).reshape(B, H, T, D)
    bound = s * 0.5 + (x.abs().float() + s * (2 ** b)) * 2.0 ** -9
    assert bool(((x.float() - deq.float()).abs() <= bound).all())
    c2, s2, m2 = np_.quant_and_pack_vcache(deq, g, b)
    d2 = np_.unpack_and_dequant_vcache(c2, s2, m2, g, b)
    assert bool(((d2.float() - deq.float()).abs() <= deq.abs().float() * 2.0 ** -8 + s * 2.0 ** -7 + 1e-3).all())
    kc, ks, km = np_.quant_and_pack_kcache(x, g, b)
    kd = np_.unpack_and_dequant_kcache(kc, ks, km, g, b)
    sk = ks.float().expand(B, H, T // g, g, D).reshape(B, H, T, D)
    boundk = sk * 0.5 + (x.abs().float() + sk * (2 ** b)) * 2.0 ** -9
    assert bool(((x.float() - kd.float()).abs() <= boundk).all())
    # transposed K through the last-dim quantizer == kcache quantizer (test.py:190-193 of the reference)
    ct, st, mt = np_.triton_quantize_and_pack_along_last_dim(x.transpose(2, 3).contiguous(), g, b)
    assert torch.equal(ct.transpose(2, 3).contiguous(), kc)
    assert torch.equal(st.transpose(2, 3).contiguous().view(-1), ks.view(-1))
