"""GPU parity tests of the fused token-major K path (gear_compress_key_fused, csrc/kfused.hip) against the CPU oracle, the
round-1 row-compressor chain and the golden fixtures.  Variants: packed / element-by-element tile arithmetic, candidate /
exact slow selection.  Run with -m gpu on an MI355X."""
import numpy as np
import pytest
import torch

from conftest import rel_fro
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def host(t):
    return t.detach().cpu().numpy()


def randn_half(seed, shape, scale=1.0):
    torch.manual_seed(seed)
    return (torch.randn(shape) * scale).half()


def bits_eq(a, b, what=""):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if a.dtype == np.float16:
        a, b = a.view(np.uint16), b.view(np.uint16)
    elif a.dtype == np.float32:
        a, b = a.view(np.uint32), b.view(np.uint32)
    bad = np.count_nonzero(a != b)
    assert bad == 0, f"{what}: {bad} / {a.size} elements differ (first at {np.argwhere(a != b)[:4].tolist()})"


@pytest.fixture(scope="module")
def C():
    from gear_amd import compress
    return compress


@pytest.fixture(params=["chain", "one", "main", "eout"], autouse=True)
def kpath(request):
    """Every test of this module runs four times: through the kernel chain (select -> dense -> solve -> Q pass; fp32 arithmetic:
    k_dense_kernel, csrc/kone.hip), through the chain with the register-tile kernel of rounds 2 - 5 in its place (option
    kfused_main: k_main_kernel), through the chain with the error matrix written by k_dense_kernel and read by the MFMA Q pass
    (option kfused_eout) and with the single-read
    kernel of csrc/kone.hip (selection + dense part + Gram in one launch, the slabs of a head exchanging candidates inside the
    launch) wherever its plan takes the shape (fp32 arithmetic; variants 2 / 8 / 32 / 64 and the fp16-stepwise mode keep the chain).
    No exchange wait may have run into its bound, and -- ordinary data -- no head may have needed the exact fall-back chain unless
    the test is about hard rows."""
    from gear_amd import _lib as L
    lib = L.load()
    lib.gear_set_option(b"kfused_one", 1 if request.param == "one" else -1)
    lib.gear_set_option(b"kfused_main", 1 if request.param == "main" else 0)
    lib.gear_set_option(b"kfused_eout", 1 if request.param == "eout" else 0)
    t0 = lib.gear_kone_timeouts()
    yield request.param
    lib.gear_set_option(b"kfused_one", 0)
    lib.gear_set_option(b"kfused_main", 0)
    lib.gear_set_option(b"kfused_eout", 0)
    assert lib.gear_kone_timeouts() == t0, "an exchange wait of the single-read K kernel timed out"


def _oracle_key(xn, k, g, b, mode=1):
    """Oracle on K rows (channels x tokens): selection, fill, quantization.  Returns dict + index sets."""
    B, H, T, D = xn.shape
    rows = np.ascontiguousarray(xn.transpose(0, 1, 3, 2)).reshape(B * H * D, T).astype(np.float32)
    orig = rows.copy()
    isml = ilrg = None
    if k > 0:
        isml, ilrg, mean = orc.outlier_select(rows, k)
        fill = mean.astype(np.float16).astype(np.float32) if mode == 0 else mean
        np.put_along_axis(rows, isml, fill[:, None], 1)
        np.put_along_axis(rows, ilrg, fill[:, None], 1)
    q = orc.quant_pack_lastdim(rows if mode == 1 else rows.astype(np.float16), g, b, mode=mode, want_deq=True)
    return q, isml, ilrg, orig


def zero_canon(a):
    """-0 -> +0: the minimum of a group holding both zeros is a zero of either sign (torch.min leaves it open too)."""
    a = np.array(a, copy=True)
    a[a == 0] = 0
    return a


def _check_payload(p, xn, k, g, b, mode=1):
    B, H, T, D = xn.shape
    q, isml, ilrg, rows32 = _oracle_key(xn, k, g, b, mode)
    bits_eq(host(p.scale).reshape(q["scale"].shape), q["scale"], "scale")
    bits_eq(zero_canon(host(p.mn).reshape(q["mn"].shape)), zero_canon(q["mn"]), "mn")
    cq = orc.unpack_tensor(q["code"], b, 1)
    ch = orc.unpack_tensor(host(p.code).reshape(B * H * D, -1), b, 1)
    mask = np.ones_like(cq, bool)
    if k > 0:
        oi = host(p.oidx).astype(np.int64).reshape(B * H * D, 2 * k)
        assert np.array_equal(oi[:, :k], np.sort(isml, 1)), "small-side outlier set"
        assert np.array_equal(oi[:, k:], np.sort(ilrg, 1)), "large-side outlier set"
        ov = host(p.oval).reshape(B * H * D, 2 * k)
        bits_eq(ov, np.take_along_axis(rows32, oi, 1).astype(np.float16), "outlier values")
        np.put_along_axis(mask, isml, False, 1)
        np.put_along_axis(mask, ilrg, False, 1)
    assert np.array_equal(cq[mask], ch[mask]), "codes"
    # Outlier slots carry quant(fill), fill = the row mean (compress_function.py:276-283: torch.mean of the fp32 row, an fp32
    # reduction whose order torch does not specify).  The oracle takes the correctly rounded mean (fp64 sum), the select kernel an
    # fp32 sum in its streaming order: the two can differ in the last place, and where the fill's quotient sits that close to a
    # rounding boundary the code under the slot differs.  Those codes are never read back (the slot's value is restored from the
    # sparse list).  Checked, not waved through: EVERY differing code must be the code of a mean one ulp away, and there must be
    # few of them.
    if k > 0:
        rr, cc_ = np.nonzero((cq != ch) & ~mask)
        assert rr.size <= max(1, (~mask).sum() // 10000), f"fill codes: {rr.size} differ"
        if rr.size:
            mean = rows32.astype(np.float64).sum(1) / rows32.shape[1]
            L = (1 << b) - 1
            sc = np.asarray(q["scale"], np.float32).reshape(B * H * D, -1)[rr, cc_ // g]
            mn_ = np.asarray(q["mn"], np.float32).reshape(B * H * D, -1)[rr, cc_ // g]
            ok = np.zeros(rr.size, bool)
            for step in (-1, 0, 1):
                m32 = mean[rr].astype(np.float32)
                if step:
                    m32 = np.nextafter(m32, np.float32(np.inf * step), dtype=np.float32)
                if mode == 0:
                    f16 = m32.astype(np.float16)
                    t1 = (f16 - mn_.astype(np.float16)).astype(np.float16)
                    c = (t1.astype(np.float32) / sc).astype(np.float16).astype(np.float32)
                else:
                    c = (m32 - mn_) / sc
                ok |= np.clip(np.rint(c), 0, L).astype(ch.dtype) == ch[rr, cc_]
            assert ok.all(), "a fill code that no mean within one ulp of the exact one explains"
    return q, isml, ilrg, rows32


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
@pytest.mark.parametrize("shape", [(1, 4, 256, 128), (2, 2, 128, 128), (1, 8, 1024, 128), (1, 2, 64, 128), (1, 3, 320, 128)])
@pytest.mark.parametrize("b,g,s", [(2, 64, 0.02), (4, 64, 0.01), (2, 32, 0.05), (4, 32, 0.0)])
def test_fused_key_payload_vs_oracle(C, shape, b, g, s, variant):
    B, H, T, D = shape
    x = randn_half(22, shape)
    k = min(C.outlier_count(B, H, T, D, s), T // 2)
    p = C.compress_key_fused(x.cuda(), b, g, k_out=k, mode="fp32", variant=variant)
    _check_payload(p, x.numpy(), k, g, b)
    if k == C.outlier_count(B, H, T, D, s):
        bits_eq(host(C.decompress(p)), orc.gears_channelQ(x.numpy(), b, g, s), "decompressed K")


@pytest.mark.parametrize("shape,k", [((1, 4, 256, 128), 0), ((1, 2, 128, 128), 3), ((2, 2, 64, 128), 1)])
@pytest.mark.parametrize("b,g", [(2, 64), (4, 64), (2, 32)])
def test_fused_key_mode_fp16_vs_oracle_and_rows_path(C, shape, k, b, g):
    """fp16-stepwise arithmetic (the streaming cache's mode): payload == oracle and == the row-compressor chain."""
    x = randn_half(23, shape)
    p = C.compress_key_fused(x.cuda(), b, g, k_out=k, mode="fp16")
    _check_payload(p, x.numpy(), k, g, b, mode=0)
    pr = C.compress_key(x.cuda(), b, g, k_out=k, mode="fp16", path="rows")
    assert torch.equal(p.code, pr.code) and torch.equal(p.scale, pr.scale) and torch.equal(p.mn, pr.mn)
    if k:
        assert torch.equal(p.oidx.view(pr.oidx.shape), pr.oidx) and torch.equal(p.oval.view(pr.oval.shape), pr.oval)


@pytest.mark.parametrize("variant", [0, 1, 4, 5])
@pytest.mark.parametrize("shape,b,s,r", [((1, 3, 320, 128), 2, 0.02, 8), ((1, 4, 256, 128), 2, 0.02, 8), ((1, 2, 1024, 128), 4, 0.01, 4), ((1, 2, 128, 128), 2, 0.0, 16),
                                         ((2, 3, 64, 128), 2, 0.0, 8)])
def test_fused_key_gear_vs_oracle(C, shape, b, s, r, variant):
    """The whole K side of method GEAR (outliers + quant + rank-r error approximation) vs the oracle's
    gearslkivi_channelQ_new (compress_function.py:213-220) -- the north-star 1e-3 statement -- and vs the rows chain."""
    B, H, T, D = shape
    x = randn_half(24, shape)
    k = C.outlier_count(B, H, T, D, s)
    P0 = torch.rand(B, H, D, r)
    p = C.compress_key_fused(x.cuda(), b, 64, k_out=k, rank=r, loop=3, mode="fp32", P0=P0, variant=variant)
    rec = host(C.decompress(p)).astype(np.float32)
    ref = orc.gearslkivi_channelQ_new(x.numpy(), b, 64, s, r, 3, P0.numpy())
    assert rel_fro(rec, ref) < 1e-3
    # the low-rank term alone: Q P^T vs the oracle's factors of the oracle's error (fp16 factors: 2e-3 on the product)
    out = orc.gears_channelQ(x.numpy().astype(np.float32), b, 64, s).astype(np.float32)
    err = x.numpy().astype(np.float32) - out
    lr_ref = orc.lowrank_reconstruct(*orc.lowrank(err, r, 3, P0.numpy()))
    lr = host(torch.matmul(p.Q.float(), p.P.float().transpose(2, 3)))
    assert rel_fro(lr, lr_ref) < 3e-3
    pr = C.compress_key(x.cuda(), b, 64, k_out=k, rank=r, loop=3, mode="fp32", P0=P0, path="rows")
    assert torch.equal(p.code, pr.code) and torch.equal(p.scale, pr.scale)
    lr_rows = host(torch.matmul(pr.Q.float(), pr.P.float().transpose(2, 3)))
    assert rel_fro(lr, lr_rows) < 2e-3


def test_fused_key_hard_rows(C):
    """Columns that defeat the threshold guess or overflow the candidate lists: constant, few-valued, heavy-tailed, one-sided,
    +-0, exact ties across the selection boundary -- the exact slow path must give the oracle's sets (lower token first)."""
    torch.manual_seed(51)
    B, H, T, D = 1, 2, 512, 128
    x = torch.randn(B, H, T, D)
    x[0, 0, :, 0] = 0.5                                             # constant column: every element ties
    x[0, 0, :, 1] = torch.randint(0, 3, (T,)).float()               # three distinct values
    x[0, 0, :, 2] = 0.0
    x[0, 0, 7, 2] = -0.0
    x[0, 0, :, 3] = torch.where(torch.rand(T) < 0.5, torch.tensor(2.0), torch.tensor(-2.0))
    x[0, 0, :, 4] = torch.rand(T)                                   # uniform: nothing beyond 2 sigma
    x[0, 0, :, 5] = torch.randn(T) ** 3                             # heavy tails
    x[0, 0, :, 6] = torch.randn(T).abs()                            # one-sided
    x[0, 1, :, 0] = torch.where(torch.rand(T) < 0.3, torch.randn(T) * 8, torch.randn(T) * 0.01)
    x[0, 1, :, 1] = 0.25
    x[0, 1, [3, 40, 90, 300], 1] = 3.0                              # 4 equal maxima, k = 2..: lowest tokens win
    x[0, 1, [10, 11, 200, 201], 1] = -2.0
    x = x.half()
    for k in (2, 5, 12):
        for variant in (0, 2):
            p = C.compress_key_fused(x.cuda(), 2, 64, k_out=k, mode="fp32", variant=variant)
            _check_payload(p, x.numpy(), k, 64, 2)
    p = C.compress_key_fused(x.cuda(), 2, 64, k_out=2, mode="fp32")
    oi = host(p.oidx).astype(np.int64).reshape(B, H, D, 4)[0, 1, 1]
    assert list(oi) == [10, 11, 3, 40]


def test_fused_key_large_k_takes_slow_path(C):
    """k beyond what the candidate lists hold: the exact slow selection.  (Up to k = T/4 no group is all outliers; beyond
    that a group's minimum can be the fill value itself, i.e. the fp32 row mean, whose last bits depend on the summation
    order -- the reference's torch.mean is an fp32 reduction too -- so there only the selection is compared.)"""
    x = randn_half(53, (1, 2, 256, 128))
    for k in (50, 60):
        p = C.compress_key_fused(x.cuda(), 2, 64, k_out=k, mode="fp32")
        _check_payload(p, x.numpy(), k, 64, 2)
    for k in (100, 128):
        p = C.compress_key_fused(x.cuda(), 2, 64, k_out=k, mode="fp32")
        rows = np.ascontiguousarray(x.numpy().transpose(0, 1, 3, 2)).reshape(-1, 256).astype(np.float32)
        isml, ilrg, _ = orc.outlier_select(rows, k)
        oi = host(p.oidx).astype(np.int64).reshape(-1, 2 * k)
        assert np.array_equal(oi[:, :k], np.sort(isml, 1)) and np.array_equal(oi[:, k:], np.sort(ilrg, 1))
        assert rel_fro(host(C.decompress(p)).astype(np.float32), orc.gears_channelQ(x.numpy(), 2, 64, 2 * k / 256 + 1e-9).astype(np.float32)) < 1e-6


@pytest.mark.parametrize("T,H,k,r", [(8192, 1, 10, 16), (4096, 2, 25, 8), (4096, 1, 40, 8), (16384, 1, 40, 8), (64, 2, 1, 8)])
def test_fused_key_c4_c5_shapes_vs_oracle(C, T, H, k, r):
    """BASELINE configs 4 / 5 at their per-GPU K geometry: 70B (T = 8192, k = 8*128*0.02/2 = 10 per channel row, rank 16),
    13B (k = 40*128*0.01/2 = 25, rank 8) and 7B (k = 40) -- against the oracle, full context; plus the two ends of what the fused
    path takes: T = 16384 (the longest row) and T = 64 (a single tile)."""
    x = randn_half(61, (1, H, T, 128))
    P0 = torch.rand(1, H, 128, r)
    p = C.compress_key_fused(x.cuda(), 2, 64, k_out=k, rank=r, loop=3, mode="fp32", P0=P0)
    q, isml, ilrg, rows32 = _check_payload(p, x.numpy(), k, 64, 2)
    # reference reconstruction from the oracle's pieces
    deq = q["deq"].astype(np.float16).astype(np.float32)
    np.put_along_axis(deq, isml, np.take_along_axis(rows32, isml, 1), 1)
    np.put_along_axis(deq, ilrg, np.take_along_axis(rows32, ilrg, 1), 1)
    out = np.ascontiguousarray(deq.reshape(1, H, 128, T).transpose(0, 1, 3, 2))
    err = x.numpy().astype(np.float32) - out
    lr_ref = orc.lowrank_reconstruct(*orc.lowrank(err, r, 3, P0.numpy()))
    rec = host(C.decompress(p)).astype(np.float32)
    assert rel_fro(rec, out + lr_ref) < 1e-3
    lr = host(torch.matmul(p.Q.float(), p.P.float().transpose(2, 3)))
    assert rel_fro(lr, lr_ref) < 3e-3


def test_fused_key_ill_conditioned_error(C):
    """Gram formulation on an error matrix with a large dynamic range: heavy-tailed K with 8x outlier channels and NO outlier
    extraction (the error keeps the heavy channels), 4-bit and 2-bit -- product Q P^T vs the oracle's step-by-step iteration."""
    torch.manual_seed(71)
    B, H, T, D = 1, 3, 1024, 128
    x = torch.randn(B, H, T, D) * (1 + 7 * (torch.rand(B, H, 1, D) > 0.95)) * (1 + 20 * (torch.rand(B, H, T, 1) > 0.995))
    x = x.half()
    for b in (2, 4):
        P0 = torch.rand(B, H, D, 8)
        p = C.compress_key_fused(x.cuda(), b, 64, k_out=0, rank=8, loop=3, mode="fp32", P0=P0)
        out = orc.gears_channelQ(x.numpy().astype(np.float32), b, 64, 0.0).astype(np.float32)
        err = x.numpy().astype(np.float32) - out
        lr_ref = orc.lowrank_reconstruct(*orc.lowrank(err, 8, 3, P0.numpy()))
        lr = host(torch.matmul(p.Q.float(), p.P.float().transpose(2, 3)))
        assert rel_fro(lr, lr_ref) < 3e-3, (b, rel_fro(lr, lr_ref))
        assert rel_fro(host(C.decompress(p)).astype(np.float32), out + lr_ref) < 1e-3


def test_fused_key_appends_into_a_pitched_cache(C):
    """t_off / row pitch / list capacity: two 64-token blocks appended behind a 128-token prefix land exactly where one call
    over the concatenation would put the quantized backbone (groups never straddle blocks), factors and lists at their rows."""
    from gear_amd import _lib as L
    B, H, D, g, b, r = 1, 2, 128, 64, 2, 8
    Tcap = 512
    fpi = 32 // b
    dev = torch.device("cuda")
    code = torch.zeros((B, H, D, Tcap // fpi), dtype=torch.int32, device=dev)
    scale = torch.zeros((B, H, D, Tcap // g), dtype=torch.float16, device=dev)
    mn = torch.zeros_like(scale)
    nseg, kcap = 4, 6
    P = torch.zeros((nseg, B, H, D, r), dtype=torch.float16, device=dev)
    Q = torch.zeros((B, H, Tcap, r), dtype=torch.float16, device=dev)
    oidx = torch.zeros((B, H, D, 2, kcap), dtype=torch.int16, device=dev)
    oval = torch.zeros((B, H, D, 2, kcap), dtype=torch.float16, device=dev)
    lib = L.load()
    blocks = [(0, 128, 2, 0), (128, 64, 1, 1), (192, 64, 1, 2)]          # (t_off, T, k, segment)
    xs, refs = [], []
    o_off = 0
    for t_off, T, k, seg in blocks:
        x = randn_half(80 + seg, (B, H, T, D)).cuda()
        P0 = torch.rand(B, H, D, r).cuda()
        ws = torch.empty((lib.gear_compress_key_fused_workspace(B * H, T, k, r),), dtype=torch.uint8, device=dev)
        rc = lib.gear_compress_key_fused(L.ptr(x), B * H, T, g, b, 0, k, L.ptr(code), L.ptr(scale), L.ptr(mn), Tcap // fpi,
                                         Tcap // g, t_off, r, 3, L.ptr(P0), L.ptr(P[seg]), B * H, 0, L.ptr(Q), Tcap, t_off,
                                         L.ptr(oidx), L.ptr(oval), kcap, o_off, 0, L.ptr(ws), ws.numel(), L.stream_ptr(x))
        L.check(rc, "gear_compress_key_fused")
        ref = C.compress_key_fused(x, b, g, k_out=k, rank=r, loop=3, mode="fp16", P0=P0)
        xs.append((t_off, T, k, seg, o_off))
        refs.append(ref)
        o_off += k
    torch.cuda.synchronize()
    for (t_off, T, k, seg, oo), ref in zip(xs, refs):
        assert torch.equal(code[..., t_off // fpi:(t_off + T) // fpi], ref.code)
        assert torch.equal(scale[..., t_off // g:(t_off + T) // g], ref.scale)
        assert torch.equal(mn[..., t_off // g:(t_off + T) // g], ref.mn)
        assert torch.equal(P[seg], ref.P) and torch.equal(Q[:, :, t_off:t_off + T], ref.Q)
        ri = ref.oidx.view(B, H, D, 2, k).to(torch.int32) + t_off
        assert torch.equal(oidx[..., oo:oo + k].to(torch.int32), ri)
        assert torch.equal(oval[..., oo:oo + k], ref.oval.view(B, H, D, 2, k))
    assert int(code[..., 256 // fpi:].abs().max()) == 0 and float(Q[:, :, 256:].abs().max()) == 0.0


def test_single_read_kernel_falls_back_per_head_and_stays_exact(C):
    """csrc/kone.hip: a head whose threshold guess fails (here: columns of two values, so that every candidate list either
    overflows or stays empty) is flagged inside the launch and redone by the exact kernel chain -- that head only; the other heads
    keep the single-read result; the payload is the oracle's either way and no exchange wait runs into its bound."""
    from gear_amd import _lib as L
    lib = L.load()
    torch.manual_seed(77)
    B, H, T, D, k = 1, 4, 2048, 128, 12
    x = torch.randn(B, H, T, D)
    x[0, 2] = torch.where(torch.rand(T, D) < 0.5, torch.tensor(1.0), torch.tensor(-1.0))     # head 2: no tail beyond any threshold
    x = x.half()
    lib.gear_set_option(b"kfused_one", 1)
    try:
        f0, t0 = lib.gear_kone_fallback_heads(), lib.gear_kone_timeouts()
        p = C.compress_key_fused(x.cuda(), 2, 64, k_out=k, mode="fp32")
        assert lib.gear_kone_fallback_heads() - f0 == 1, "exactly the degenerate head takes the exact chain"
        assert lib.gear_kone_timeouts() == t0
        _check_payload(p, x.numpy(), k, 64, 2)
        f1 = lib.gear_kone_fallback_heads()
        p = C.compress_key_fused(randn_half(78, (B, H, T, D)).cuda(), 2, 64, k_out=k, mode="fp32")
        assert lib.gear_kone_fallback_heads() == f1, "ordinary data: no head falls back"
    finally:
        lib.gear_set_option(b"kfused_one", 0)
