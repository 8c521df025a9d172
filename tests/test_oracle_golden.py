"""Pins the CPU oracle (oracle/) against golden vectors produced by running the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import oracle as orc
from conftest import rel_fro

BITS = (2, 4)
GROUPS = (32, 64, 128)


def test_fp16_conversion_roundtrip():
    lib = orc.lib()
    allh = np.arange(65536, dtype=np.uint16)
    f = allh.view(np.float16).astype(np.float32)
    for h in list(range(0, 65536, 97)) + [0, 1, 0x3FF, 0x400, 0x7BFF, 0x7C00, 0x8000, 0xFBFF]:
        v = lib.orc_h2f(int(h))
        if np.isnan(f[h]):
            assert np.isnan(v)
        else:
            assert v == f[h]
            assert lib.orc_f2h(float(f[h])) == h
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(4000) * s for s in (1e-7, 1e-5, 1e-3, 1.0, 300.0, 7e4)]).astype(np.float32)
    ref = x.astype(np.float16).view(np.uint16)
    got = np.array([lib.orc_f2h(float(v)) for v in x], dtype=np.uint16)
    assert np.array_equal(ref, got)


@pytest.mark.parametrize("tag,xkey", [("k", "xk"), ("v", "xv")])
@pytest.mark.parametrize("g", GROUPS)
@pytest.mark.parametrize("b", BITS)
def test_f1_quant_pack_lastdim_bit_exact(golden, tag, xkey, g, b):
    f = golden("f1_quant_pack.npz")
    r = orc.quant_pack_lastdim(f[xkey], g, b, mode=0)
    key = f"last_{tag}_g{g}_b{b}"
    assert np.array_equal(r["code"], f[key + "_code"])
    assert np.array_equal(r["scale"].view(np.uint16), f[key + "_scale"].reshape(r["scale"].shape).view(np.uint16))
    assert np.array_equal(r["mn"].view(np.uint16), f[key + "_mn"].reshape(r["mn"].shape).view(np.uint16))
    deq = orc.unpack_dequant_lastdim(f[key + "_code"], r["scale"], r["mn"], g, b, mode=0)
    assert np.array_equal(deq.view(np.uint16), f[key + "_deq"].view(np.uint16))


@pytest.mark.parametrize("g", GROUPS)
@pytest.mark.parametrize("b", BITS)
def test_f1_quant_pack_kcache_bit_exact(golden, g, b):
    f = golden("f1_quant_pack.npz")
    r = orc.quant_pack_k(f["xv"], g, b, mode=0)
    key = f"kc_g{g}_b{b}"
    assert np.array_equal(r["code"], f[key + "_code"])
    assert np.array_equal(r["scale"].view(np.uint16), f[key + "_scale"].reshape(r["scale"].shape).view(np.uint16))
    assert np.array_equal(r["mn"].view(np.uint16), f[key + "_mn"].reshape(r["mn"].shape).view(np.uint16))
    deq = orc.unpack_dequant_k(r["code"], r["scale"], r["mn"], g, b, mode=0)
    assert np.array_equal(deq.view(np.uint16), f[key + "_deq"].view(np.uint16))


@pytest.mark.parametrize("b", BITS)
def test_f1_triton_function_bit_exact(golden, b):
    f = golden("f1_quant_pack.npz")
    r = orc.quant_pack_lastdim(f["xt"], 64, b, mode=0)
    assert np.array_equal(r["code"], f[f"tri_b{b}_code"])
    assert np.array_equal(r["scale"].view(np.uint16), f[f"tri_b{b}_scale"].view(np.uint16))
    assert np.array_equal(r["mn"].view(np.uint16), f[f"tri_b{b}_mn"].view(np.uint16))


def test_f1_pack_unpack_tensor(golden):
    f = golden("f1_quant_pack.npz")
    assert np.array_equal(orc.pack_tensor(f["raw4"], 4, 2), f["raw4_pack2"])
    assert np.array_equal(orc.pack_tensor(f["raw4"], 4, 3), f["raw4_pack3"])
    assert np.array_equal(orc.unpack_tensor(f["raw4_pack3"], 4, 3), f["raw4_unpack3"])
    assert np.array_equal(orc.pack_tensor(f["raw4"] & 3, 2, 3), f["raw2_pack3"])
    assert np.array_equal(orc.pack_tensor(f["raw4"] & 3, 2, 2), f["raw2_pack2"])
    assert np.array_equal(orc.unpack_tensor(f["raw2_pack2"], 2, 2), (f["raw4"] & 3).astype(np.int16))


@pytest.mark.parametrize("b", BITS)
def test_f2_witherror_bit_exact(golden, b):
    f = golden("f2_witherror.npz")
    r = orc.quant_pack_lastdim(f["x"], 64, b, mode=0, want_err=True)
    assert np.array_equal(r["scale"].view(np.uint16), f[f"b{b}_scale"].view(np.uint16))
    assert np.array_equal(r["mn"].view(np.uint16), f[f"b{b}_mn"].view(np.uint16))
    assert np.array_equal(r["err"].reshape(-1).view(np.uint16), f[f"b{b}_err"].reshape(-1).view(np.uint16))
    # B1: the reference's code tensor is zero beyond the first floor(num_groups/fpi) packed columns
    refc = f[f"b{b}_refcode_B1"]
    ncol = (256 // 64) // (32 // b)
    assert not refc[..., ncol:].any()
    assert np.array_equal(r["code"][..., :ncol], refc[..., :ncol])
    assert r["code"][..., ncol:].any()


@pytest.mark.parametrize("r", (2, 4, 8, 16))
@pytest.mark.parametrize("loop", (1, 3))
def test_f3_lowrank(golden, r, loop):
    f = golden("f3_lowrank.npz")
    rec = orc.fake_poweriteration_group(f["E"], loop, r, f[f"pi_r{r}_l{loop}_P0"])
    assert rel_fro(rec, f[f"pi_r{r}_l{loop}_rec"]) < 1e-3
    # headwise_lrap (fp16 factors): parity is on the product Q P^T
    P, Q = orc.lowrank(f["Eh"], r, loop, f[f"lrap_r{r}_l{loop}_P0"])
    got = orc.lowrank_reconstruct(P.astype(np.float16), Q.astype(np.float16))
    ref = orc.lowrank_reconstruct(f[f"lrap_r{r}_l{loop}_P"], f[f"lrap_r{r}_l{loop}_Q"])
    assert rel_fro(got, ref) < 2e-3


@pytest.mark.parametrize("g", GROUPS)
@pytest.mark.parametrize("b", BITS)
def test_f4_fake_quant(golden, g, b):
    f = golden("f4_fakequant.npz")
    x = f["x"]
    tok = orc.fake_groupwise_token_asymmetric_quantization(x, b, g)
    assert np.array_equal(tok.view(np.uint16), f[f"tok_b{b}_g{g}"].view(np.uint16))
    ch16 = orc.fake_groupwise_channel_asymmetric_quantization_new(x, b, g)
    assert np.array_equal(ch16.view(np.uint16), f[f"chan_fp16_b{b}_g{g}"].view(np.uint16))
    ch32 = orc.fake_groupwise_channel_asymmetric_quantization_new(x.astype(np.float32), b, g)
    assert np.array_equal(ch32, f[f"chan_fp32_b{b}_g{g}"])


@pytest.mark.parametrize("b", BITS)
def test_f4_cluster_variants(golden, b):
    f = golden("f4_fakequant.npz")
    x32 = f["x"].astype(np.float32)
    assert np.array_equal(orc.fake_groupwise_channel_asymmetric_quantization_cluster(x32, b ** 2 - 1, 96),
                          f[f"chan_cluster_b{b}_g96"])
    assert np.array_equal(orc.fake_groupwise_token_asymmetric_quantization_cluster(x32, b ** 2 - 1, 64),
                          f[f"tok_cluster_b{b}_g64"])


@pytest.mark.parametrize("s", (0.01, 0.02))
@pytest.mark.parametrize("b", BITS)
def test_f5_outliers_bit_exact(golden, s, b):
    f = golden("f5_outlier.npz")
    tag = f"s{int(s * 100)}_b{b}"
    assert np.array_equal(orc.gears_channelQ(f["x"], b, 64, s).view(np.uint16), f[f"chan_{tag}"].view(np.uint16))
    assert np.array_equal(orc.gears_tokenQ(f["x"], b, 64, s).view(np.uint16), f[f"tok_{tag}"].view(np.uint16))


def test_f5_tie_case_restored_tensor(golden):
    """Ties across the selection boundary: index choice is implementation-defined; parity is on the tensor."""
    f = golden("f5_outlier.npz")
    for name, fn in (("chan", orc.gears_channelQ), ("tok", orc.gears_tokenQ)):
        got = fn(f["x_tie"], 2, 64, 0.02).astype(np.float32)
        ref = f[f"{name}_tie_s2_b2"].astype(np.float32)
        frac = np.mean(got != ref)
        assert frac < 2e-3, (name, frac)
        assert rel_fro(got, ref) < 2e-2


@pytest.mark.parametrize("case", ["KIVI_V2_b4_r0", "KIVI_V2_b2_r0", "GEARL_b2_r4", "GEARL_b4_r8", "GEAR_b4_r4",
                                  "GEAR_b2_r8"])
def test_f6_compress_insert_function(golden, case):
    f = golden("f6_insert.npz")
    method, b, r = case.split("_b")[0], int(case.split("_b")[1][0]), int(case.split("_r")[1])
    left = {"GEAR_b4_r4": 0.01, "GEAR_b2_r8": 0.02}.get(case, 0.0)
    P0k = f[case + "_P0k"] if (case + "_P0k") in f.files else None
    P0v = f[case + "_P0v"] if (case + "_P0v") in f.files else None
    k, v = orc.compress_insert_function(f["k"], f["v"], method, b, 64, rank=r, rankv=r, loop=3, left=left,
                                        P0k=P0k, P0v=P0v)
    if method == "KIVI_V2":
        assert np.array_equal(k.view(np.uint16), f[case + "_k"].view(np.uint16))
        assert np.array_equal(v.view(np.uint16), f[case + "_v"].view(np.uint16))
    else:
        # north_star tolerance: 1e-3 relative on the reconstructed K/V
        assert rel_fro(k, f[case + "_k"]) < 1e-3
        assert rel_fro(v, f[case + "_v"]) < 1e-3


@pytest.mark.parametrize("case", ["gearl_b2_mha", "gearl_b4_gqa", "kivi_b2_mha"])
def test_f8_attention_state_machine_trace(golden, case):
    """Fixture F8: the committed 130-step decode trace of the attention cache state machine (oracle/attention_oracle.py,
    restating modeling_llamagear.py:177-484).  The trace was produced by this oracle (the reference's forward is not importable),
    so the test pins the ORACLE: an edit that changes what the restatement computes -- and with it what the GPU tests compare
    the product against -- fails here.  Bit-exact (same numpy arithmetic)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("make_f8", os.path.join(os.path.dirname(__file__), "golden", "make_f8.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    f = golden("f8_trace.npz")
    out, state = mk.run(case, load=f)
    assert out.shape == f[case + "_out"].shape == (1, mk.CASES[case][2], 131, 128)
    assert np.array_equal(out.view(np.uint16), f[case + "_out"].view(np.uint16))
    assert np.array_equal(state, f[case + "_state"])
    # 330 tokens: 320 compressed in 1 prefill segment + 2 decode blocks, 10 in the fp16 window
    fpi = 32 // mk.CASES[case][1]
    assert list(f[case + "_state"][:4]) == [330, 320 // fpi, 10, 320]


F9_CASES = ["KCVT_b4_r0", "KCVT_b2_r0", "GEAR-KCVT_b2_r8", "GEAR-KCVT_b4_r4", "GEARL-KCVT_b4_r4", "GEARL-KCVT_b2_r8"]


@pytest.mark.parametrize("case", F9_CASES)
def test_f9_kcvt_methods(golden, case):
    """The dispatcher's whole-row variants (group = seq_len for K, num_head * head_dim for V), T = 192: the oracle vs the
    reference-executed fixture.  KCVT is a pure quantize / dequantize: bit-exact."""
    f = golden("f9_kcvt.npz")
    method, b, r = case.split("_b")[0], int(case.split("_b")[1][0]), int(case.split("_r")[1])
    left = 0.02 if case == "GEAR-KCVT_b2_r8" else 0.0
    P0k = f[case + "_P0k"] if (case + "_P0k") in f.files else None
    P0v = f[case + "_P0v"] if (case + "_P0v") in f.files else None
    k, v = orc.compress_insert_function(f["k"], f["v"], method, b, 64, rank=r, rankv=r, loop=3, left=left, P0k=P0k, P0v=P0v)
    if method == "KCVT":
        assert np.array_equal(k.view(np.uint16), f[case + "_k"].view(np.uint16))
        assert np.array_equal(v.view(np.uint16), f[case + "_v"].view(np.uint16))
    else:
        assert rel_fro(k, f[case + "_k"]) < 1e-3
        assert rel_fro(v, f[case + "_v"]) < 1e-3


@pytest.mark.parametrize("name,ss,ls", [("tp_25_25", 0.25, 0.25), ("tp_50_0", 0.5, 0.0)])
def test_f9_token_preserving_window(golden, name, ss, ls):
    """token_preserving (compress_function.py:431-464): only tokens [int(ss T), T - int(ls T)) are quantized; locality 0 makes
    the slice [s:-0] empty, i.e. the call is a no-op -- the reference's behaviour, kept."""
    f = golden("f9_kcvt.npz")
    k, v = orc.compress_insert_function(f["k_tp"], f["v_tp"], "KIVI_V2", 4, 64, start_saving=ss, locality_saving=ls)
    assert np.array_equal(k.view(np.uint16), f[name + "_k"].view(np.uint16))
    assert np.array_equal(v.view(np.uint16), f[name + "_v"].view(np.uint16))
    if ls == 0.0:
        assert np.array_equal(k, f["k_tp"]) and np.array_equal(v, f["v_tp"])


@pytest.mark.parametrize("name", ("mha", "mqa"))
@pytest.mark.parametrize("b", BITS)
def test_f7_gemv(golden, name, b):
    f = golden("f7_gemv.npz")
    B, nh, IC, OC, GS = [int(v) for v in f["dims"]]
    inp = f[f"{name}_inp"].reshape(B, nh, 1, IC)
    nkv = nh if name == "mha" else 1
    # stored in the CUDA kernel's layout [BS, OC/pack, IC]; the oracle takes matmul.py's [.., K, N/fpi]
    qw = np.ascontiguousarray(f[f"{name}_b{b}_qw"].transpose(0, 2, 1)).reshape(B, nkv, IC, -1)
    sc = np.ascontiguousarray(f[f"{name}_b{b}_scale"].transpose(0, 2, 1)).reshape(B, nkv, IC, -1)
    mn = np.ascontiguousarray(f[f"{name}_b{b}_mn"].transpose(0, 2, 1)).reshape(B, nkv, IC, -1)
    out, out32 = orc.gemv_outer(inp, qw, sc, mn, GS, b, mode=0, want32=True)
    ref = f[f"{name}_b{b}_ref"].reshape(B, nh, 1, OC)
    # reference recipe rounds the dequantized weight to fp16 (gemv.py:70-74); the kernel does not
    assert rel_fro(out32, ref) < 2e-3
    assert rel_fro(out.astype(np.float32), ref) < 2e-3


# ------------------------------------------------------------------------------------------------------------------
# F8-ref: the state machines against traces made by EXECUTING the reference's own forward / compression glue /
# matmul_withlrap source (tests/golden/make_f8_ref.py).  Tolerances: the reference-side GEMV of the fixture rounds the
# dequantized weight to fp16 before the product (its own check of the kernel, quant/gemv.py:70-74) where the kernel -- and
# orc.gemv_outer -- keep fp32 (2e-3, as fixture F7); low-rank factors agree up to the basis of the subspace (products only).
F8_REF_GEAR = ["gear_kivi_b2", "gear_kivi_b4_t64", "gear_stance_gearl_b2", "gear_stance_gearl_b4_t30", "gear_stance_gearl_b2_h8r8"]
F8_REF_KIVI = ["kivi_b2", "kivi_b4_t64"]


def _f8_ref_inputs(f, case):
    qkv = f[case + "_qkv"]                       # [3, 1, H, T, D] post-RoPE q, k, v
    out = f[case + "_out"]                       # [1, 1 + steps, H*D]
    H, T, D = qkv.shape[2], qkv.shape[3], qkv.shape[4]
    steps = out.shape[1] - 1
    return qkv[0], qkv[1], qkv[2], out.reshape(1, steps + 1, H, D).transpose(0, 2, 1, 3), H, D, T - steps, steps


def _run_trace(o, q, k, v, TP, steps):
    mask = np.triu(np.full((TP, TP), np.finfo(np.float16).min, np.float16), 1)[None, None]
    outs = [o.prefill(q[:, :, :TP], k[:, :, :TP], v[:, :, :TP], mask)[:, :, -1:]]
    for i in range(steps):
        t = TP + i
        outs.append(o.decode(q[:, :, t:t + 1], k[:, :, t:t + 1], v[:, :, t:t + 1]))
    return np.concatenate(outs, 2)


@pytest.mark.parametrize("case", F8_REF_GEAR)
def test_f8_ref_gear_state_machine_vs_reference_forward(golden, case):
    """oracle/attention_oracle.py against LlamaAttention_GEAR.forward of the reference (cuda_supported_gear/modeling_llamagear.py:
    177-484) run step by step: attention output of every step, slot 8, the packed cache bit for bit, the factor products."""
    from oracle.attention_oracle import GearAttentionOracle
    f = golden(f"f8_ref_{case}.npz")
    q, k, v, ref, H, D, TP, steps = _f8_ref_inputs(f, case)
    bits = 4 if "_b4" in case else 2
    method = "gearlKIVI" if "gearl" in case else "KIVI"
    rank = 8 if "r8" in case else 4
    cc = dict(compress_method=method, group_size=64, residual=64, quantize_bit=bits, rank=rank, rankv=rank, loop=3)
    drawn = []

    def draw(B, Hh, S, Dm, r):
        p = f[f"{case}_P0_{len(drawn)}"]
        assert p.shape == (B, Hh, Dm, r)
        drawn.append(p)
        return p
    o = GearAttentionOracle(H, H, D, cc, draw)
    got = _run_trace(o, q, k, v, TP, steps)
    assert got.shape == ref.shape
    # (round 5: the fixture's stand-in for the CUDA GEMV has the kernel's arithmetic -- fp32 dequantized weight, fp32 accumulate,
    # one fp16 rounding -- instead of an fp16-rounded weight; the tolerances came down from 2e-3 / 4e-3 with it)
    assert rel_fro(got, ref) < 5e-4, rel_fro(got, ref)
    worst = max(rel_fro(got[:, :, i], ref[:, :, i]) for i in range(steps + 1))
    assert worst < 1e-3, worst
    c = o.c
    assert c["n"] == int(f[case + "_seq"][0]) == TP + steps
    # (a prompt of exactly `residual` tokens strands V in fp16 for good -- modeling_llamagear.py:416 / :335 -- in both)
    for name, key in (("kcode", "kc"), ("vcode", "vc"), ("kscale", "ks"), ("kmn", "km"), ("vscale", "vs"), ("vmn", "vm"),
                      ("kfull", "kfull"), ("vfull", "vfull")):
        if c[key] is None:
            assert f"{case}_{name}" not in f.files, name
        elif c[key].dtype == np.float16:
            assert np.array_equal(c[key].view(np.uint16), f[f"{case}_{name}"].view(np.uint16)), name
        else:
            assert np.array_equal(c[key], f[f"{case}_{name}"]), name
    if method == "gearlKIVI":
        assert len(drawn) == len([n for n in f.files if n.startswith(case + "_P0_")])
        # factors: reference format [prefill, stacked blocks]; the oracle keeps one entry per segment
        for pn, qn, op, oq in (("kp", "kq", "kp", "kq"), ("vp", "vq", "vp", "vq")):
            refs = [(f[f"{case}_{pn}0"], f[f"{case}_{qn}0"])] if f"{case}_{pn}0" in f.files else []
            if f"{case}_{pn}1" in f.files:
                refs += list(zip(f[f"{case}_{pn}1"], f[f"{case}_{qn}1"]))
            assert len(refs) == len(c[op])
            for (P, Qf), Po, Qo in zip(refs, c[op], c[oq]):
                a = Qf.astype(np.float64) @ P.astype(np.float64).transpose(0, 1, 3, 2)
                b = Qo.astype(np.float64) @ Po.astype(np.float64).transpose(0, 1, 3, 2)
                assert rel_fro(b, a) < 3e-3, (pn, rel_fro(b, a))


@pytest.mark.parametrize("case", F8_REF_KIVI)
def test_f8_ref_kivi_state_machine_vs_reference_forward(golden, case):
    """oracle/kivi_oracle.py against LlamaAttention_KIVI.forward of the reference (modeling_llama_kivi.py:81-289), incl. the
    per-token V quantization past the sliding window (:200-213)."""
    from oracle.kivi_oracle import KiviAttentionOracle
    f = golden(f"f8_ref_{case}.npz")
    q, k, v, ref, H, D, TP, steps = _f8_ref_inputs(f, case)
    bits = 4 if "_b4" in case else 2
    o = KiviAttentionOracle(H, H, D, 64, bits, 64)
    got = _run_trace(o, q, k, v, TP, steps)
    assert rel_fro(got, ref) < 5e-4, rel_fro(got, ref)
    assert max(rel_fro(got[:, :, i], ref[:, :, i]) for i in range(steps + 1)) < 1e-3
    c = o.c
    assert c["n"] == int(f[case + "_seq"][0])
    assert np.array_equal(c["kc"], f[case + "_kcode"]) and np.array_equal(c["vc"], f[case + "_vcode"])
    for name, key in (("kscale", "ks"), ("kmn", "km"), ("vscale", "vs"), ("vmn", "vm"), ("vfull", "vfull")):
        assert np.array_equal(c[key].view(np.uint16), f[f"{case}_{name}"].view(np.uint16)), name
    assert c["vfull"].shape[2] == 64                  # the sliding fp16 window


@pytest.mark.parametrize("bits", [2, 4])
def test_f8_ref_matmul_withlrap(golden, bits):
    """oracle matmul_withlrap against the reference's (modeling_llamagear.py:54-111): no factors, prefill factors, prefill +
    stacked block factors, key and value side."""
    from oracle.attention_oracle import matmul_withlrap
    f = golden("f8_ref_matmul.npz")
    g = lambda n: f[f"mm_b{bits}_{n}"]
    fpi, Tp = 32 // bits, 128
    cases = {
        "key_none": (g("q"), g("kc"), g("ks"), g("km"), [None], [None], "key"),
        "key_prefill": (g("q"), np.ascontiguousarray(g("kc")[..., :Tp // fpi]), np.ascontiguousarray(g("ks")[..., :Tp // 64]),
                        np.ascontiguousarray(g("km")[..., :Tp // 64]), [g("kp0")], [g("kq0")], "key"),
        "key_stacked": (g("q"), g("kc"), g("ks"), g("km"), [g("kp0"), g("kp1")], [g("kq0"), g("kq1")], "key"),
        "value_none": (g("a"), g("vc"), g("vs"), g("vm"), [None], [None], "value"),
        "value_prefill": (g("a_prefill"), np.ascontiguousarray(g("vc")[:, :, :Tp]), np.ascontiguousarray(g("vs")[:, :, :Tp]),
                          np.ascontiguousarray(g("vm")[:, :, :Tp]), [g("vp0")], [g("vq0")], "value"),
        "value_stacked": (g("a"), g("vc"), g("vs"), g("vm"), [g("vp0"), g("vp1")], [g("vq0"), g("vq1")], "value"),
    }
    for name, (a, code, scale, mn, pb, qb, typ) in cases.items():
        got = matmul_withlrap(64, a, code, scale, mn, bits, pb, qb, type=typ)
        ref = g(name).reshape(got.shape)
        assert rel_fro(got, ref) < 2e-4, (name, rel_fro(got, ref))     # (fp32-weight stand-in since round 5: was 2e-3)


@pytest.mark.parametrize("B,H,T,bits,g,s,r", [(1, 4, 256, 2, 64, 0.02, 4), (2, 2, 192, 4, 32, 0.05, 8), (1, 3, 200, 2, 64, 0.01, 2),
                                             (1, 2, 128, 4, 64, 0.0, 4)])
def test_fused_cpu_gear_path_equals_the_glue_path(B, H, T, bits, g, s, r):
    """oracle.gear_tensor (bench.py's cpu_baseline: method GEAR on one tensor inside the C library) is bit-identical to the
    function-by-function restatement that is pinned to the reference's fixtures (compress_insert_function, F6)."""
    rng = np.random.default_rng(3)
    k = rng.standard_normal((B, H, T, 128)).astype(np.float16)
    v = rng.standard_normal((B, H, T, 128)).astype(np.float16)
    k[:, :, 5::17, 3::29] *= 6
    v[:, :, 2::9, 1::31] *= 6
    P0k, P0v = rng.random((B, H, 128, r), dtype=np.float32), rng.random((B, H, 128, r), dtype=np.float32)
    rk, rv = orc.compress_insert_function(k, v, "GEAR", bits, g, r, r, 3, s, P0k, P0v)
    gk = orc.gear_tensor(k, "k", bits, g, s, r, 3, P0k)
    gv = orc.gear_tensor(v, "v", bits, g, s, r, 3, P0v)
    assert np.array_equal(gk.view(np.uint16), rk.view(np.uint16))
    assert np.array_equal(gv.view(np.uint16), rv.view(np.uint16))


F10_CHAN = [(T, g, s, b) for (T, g) in ((200, 64), (150, 32), (70, 64)) for (s, b) in ((0.0, 2), (0.02, 2), (0.02, 4), (0.05, 4))]


@pytest.mark.parametrize("T,g,s,b", F10_CHAN)
def test_f10_ragged_channel_rows_bit_exact(golden, T, g, s, b):
    """Sequence length not a multiple of the group: the T mod g tail of every channel stays as it is, selection and fill mean use all
    T tokens (compress_function.py:107-122, :261-296).  Oracle vs the reference-executed fixture, bit for bit."""
    f = golden("f10_ragged.npz")
    x = f[f"x_T{T}"]
    got = orc.gears_channelQ(x, b, g, s)
    assert np.array_equal(got.view(np.uint16), f[f"chan_T{T}_g{g}_s{int(s * 100)}_b{b}"].view(np.uint16))
    assert np.array_equal(got[:, :, (T // g) * g:], x[:, :, (T // g) * g:])          # (the tail is the input)


@pytest.mark.parametrize("case", ["GEAR_b2_r8_s2", "GEAR_b4_r4_s2", "GEAR_b2_r4_s0"])
def test_f10_ragged_gear_method(golden, case):
    f = golden("f10_ragged.npz")
    b, r, left = int(case.split("_b")[1][0]), int(case.split("_r")[1][0]), int(case.split("_s")[1]) / 100
    k, v = orc.compress_insert_function(f["k"], f["v"], "GEAR", b, 64, rank=r, rankv=r, loop=3, left=left,
                                        P0k=f[case + "_P0k"], P0v=f[case + "_P0v"])
    assert rel_fro(k, f[case + "_k"]) < 1e-3
    assert rel_fro(v, f[case + "_v"]) < 1e-3
