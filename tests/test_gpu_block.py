"""GPU tests of the single-launch block compressor (csrc/block_fused.hip, gear_compress_block): the 64-token block boundary of
the streaming cache.  Checked three ways: against the kernel chain it replaces (bit-exact payload), against the CPU oracle
(outlier sets, codes, scale / zero point), and -- for the factors, which come from a different but equivalent iteration --
against a float64 restatement of the reference's power iteration (cuda_supported_gear/quant/new_pack.py:291-311) on the very
error matrix the cache implies."""
import numpy as np
import pytest
import torch

from conftest import rel_fro
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(autouse=True)
def _block_kernel_for_every_head_count():
    """The tests cover every head count the kernel accepts (cache.py keeps the chain below 4 KV heads for speed)."""
    from gear_amd import cache as gc
    old = (gc.USE_BLOCK_KERNEL, gc.BLOCK_KERNEL_MIN_HEADS)
    gc.BLOCK_KERNEL_MIN_HEADS = 1
    yield
    gc.USE_BLOCK_KERNEL, gc.BLOCK_KERNEL_MIN_HEADS = old


def _mk(n_layers, B, H, cc, tmax, use_block, seed=5):
    from gear_amd import cache as gc
    gc.USE_BLOCK_KERNEL = use_block
    pool = gc.GearKVCachePool(n_layers, B, H, tmax, cc, "cuda", seed=seed)
    caches = [gc.GearKVCache(B, H, tmax, cc, "cuda", pool=pool, layer=l) for l in range(n_layers)]
    return pool, caches


def _feed(pool, caches, kpre, vpre, knew, vnew, use_block):
    """prefill (per layer, the chain) + appended tokens; blocks through compress_all with or without the block kernel."""
    from gear_amd import cache as gc
    gc.USE_BLOCK_KERNEL = use_block
    try:
        for l, c in enumerate(caches):
            if kpre is not None:
                c.prefill(kpre[l].cuda(), vpre[l].cuda())
        for i in range(knew.shape[3]):
            for l, c in enumerate(caches):
                c.append(knew[l][:, :, i:i + 1].cuda(), vnew[l][:, :, i:i + 1].cuda())
            if caches[0].n_win == 64:
                pool.compress_all()
    finally:
        gc.USE_BLOCK_KERNEL = True
    torch.cuda.synchronize()


def _tile_sets(tile, cnt):
    """[..., cap] entries + [...] counts -> list of sorted arrays (entry order inside a tile is not part of the format)."""
    tile, cnt = host(tile).astype(np.int64) & 0xFFFFFFFF, host(cnt)
    flat_t, flat_c = tile.reshape(-1, tile.shape[-1]), cnt.reshape(-1)
    return [np.sort(flat_t[i, :max(flat_c[i], 0)]) for i in range(flat_c.size)], flat_c


CASES = [
    # layers, B, H, bits, group, rank, left, T0, steps
    (2, 1, 32, 2, 64, 8, 0.02, 128, 130),      # Llama-2-7B heads: V rows of 4096 elements, kv = 40
    (1, 2, 4, 2, 64, 4, 0.02, 200, 150),
    (2, 1, 10, 2, 32, 8, 0.01, 64, 70),        # 13B shard (10 heads): rows of 1280, group 32
    (3, 1, 1, 2, 64, 16, 0.02, 0, 130),        # 70B shard: 1 KV head, rank 16, no prompt
    (1, 1, 40, 4, 64, 4, 0.05, 192, 70),       # 13B heads, 4 bits, prompt ends in the middle of a 128-token chunk
    (1, 2, 8, 4, 32, 2, 0.1, 64, 70),          # kk_blk = 3, rank below the register block
    (1, 1, 4, 2, 64, 8, 0.0, 128, 70),         # no outliers
    (1, 1, 4, 2, 64, 0, 0.02, 128, 70),        # no low-rank part (KIVI + outliers)
    (1, 1, 64, 2, 64, 4, 0.01, 64, 70),        # 64 heads: the row-duty wave has no lane H for the terminal chunk-index entry
]


@pytest.mark.parametrize("layers,B,H,bits,group,rank,left,T0,steps", CASES)
def test_block_kernel_matches_chain(layers, B, H, bits, group, rank, left, T0, steps):
    from gear_amd import cache as gc
    torch.manual_seed(100 + H + bits)
    method = "gearslKIVI" if rank else "KIVI"
    cc = dict(compress_method=method, group_size=group, residual=64, quantize_bit=bits, rank=rank, rankv=rank, loop=3, left=left)
    tmax = T0 + steps + 10
    kpre = vpre = None
    if T0:
        kpre = [torch.randn(B, H, T0, 128).half() for _ in range(layers)]
        vpre = [torch.randn(B, H, T0, 128).half() for _ in range(layers)]
    knew = torch.randn(layers, B, H, steps, 128).half()
    vnew = torch.randn(layers, B, H, steps, 128).half()
    # heavy tails + ties + signed zeros so that the selection paths see something other than a clean Gaussian
    knew[:, :, :, 5::17, 3::29] *= 6
    vnew[:, :, :, 3::11, 7::31] *= 6
    knew[:, :, :, 40:50, 10:20] = 0.5
    vnew[:, :, :, 36:40, 64:128] = -0.25
    knew[:, :, :, 60, 40:44] = -0.0
    vnew[:, :, :, 61, 0:6] = -0.0
    out = {}
    for use_block in (False, True):
        pool, caches = _mk(layers, B, H, cc, tmax, use_block)
        _feed(pool, caches, kpre, vpre, knew, vnew, use_block)
        out[use_block] = (pool, caches)
    assert gc.block_kernel_status() == 0
    pa, ca = out[False]
    pb, cb = out[True]
    n = ca[0].n_comp
    assert n == cb[0].n_comp and n == (T0 // 64 + (T0 % 64 + steps) // 64) * 64
    exact = ["kcode", "kscale", "kmn", "vcode", "vscale", "vmn", "koidx", "koval", "voidx", "voval", "vochunk", "kcnt", "vcnt"]
    for name in exact:
        if name in pa.buf:
            a, b = host(pa.buf[name]), host(pb.buf[name])
            if a.dtype == np.float16:
                a, b = a.view(np.uint16), b.view(np.uint16)
            assert np.array_equal(a, b), name
    for tname, cname in (("ktile", "kcnt"), ("vtile", "vcnt")):
        if tname in pa.buf:
            sa, cnta = _tile_sets(pa.buf[tname], pa.buf[cname])
            sb, cntb = _tile_sets(pb.buf[tname], pb.buf[cname])
            assert np.array_equal(cnta, cntb)
            for x, y in zip(sa, sb):
                assert np.array_equal(x, y), tname
    if rank:
        # factors: same subspace -> same product Q P^T up to the fp16 rounding of the factors
        for l in range(layers):
            a, b = ca[l], cb[l]
            for seg_t in range(a.seg0, n, 64):
                seg = a._segment_of(seg_t)
                for qn, pn in (("kQtok", "kPseg"), ("vQtok", "vPseg")):
                    La = host(getattr(a, qn)[:, :, seg_t:seg_t + 64]).astype(np.float64) @ host(getattr(a, pn)[seg]).astype(np.float64).transpose(0, 1, 3, 2)
                    Lb = host(getattr(b, qn)[:, :, seg_t:seg_t + 64]).astype(np.float64) @ host(getattr(b, pn)[seg]).astype(np.float64).transpose(0, 1, 3, 2)
                    for bb in range(B):
                        for hh in range(H):
                            assert rel_fro(Lb[bb, hh], La[bb, hh]) < 4e-3, (l, seg_t, qn, bb, hh, rel_fro(Lb[bb, hh], La[bb, hh]))


def _error_matrix(x, code, scale, mn, omask, fill16, group, layout):
    """E of one head's block as the kernels define it (fp16-stepwise): x, code [64,128] token-major; scale / mn per group;
    omask True at outliers (E = 0 there)."""
    x32 = x.astype(np.float32)
    if layout == "k":       # groups along tokens, per channel: scale [128, 64/g]
        sc = np.repeat(scale.astype(np.float32), group, 1).T
        zp = np.repeat(mn.astype(np.float32), group, 1).T
    else:                   # groups along channels, per token: scale [64, 128/g]
        sc = np.repeat(scale.astype(np.float32), group, 1)
        zp = np.repeat(mn.astype(np.float32), group, 1)
    v = np.where(omask, fill16.astype(np.float32), x32)
    dq = ((code.astype(np.float32) * sc).astype(np.float16).astype(np.float32) + zp).astype(np.float16).astype(np.float32)
    e = (v - dq).astype(np.float16).astype(np.float64)
    e[omask] = 0.0
    return e


def _ref_lowrank(E, P0, loop):
    """float64 restatement of headwise_lrap (new_pack.py:291-311) on E [S, Dm] with the basis P0 [Dm, r]: returns Q P^T."""
    P = P0.astype(np.float64)
    for i in range(loop):
        if i == loop - 1:
            P = np.linalg.qr(P)[0]
        Q = E @ P
        if i == loop - 1:
            Q = np.linalg.qr(Q)[0]
        P = E.T @ Q
    return Q @ P.T


@pytest.mark.parametrize("H,bits,group,rank,left", [(4, 2, 64, 8, 0.02), (2, 4, 32, 4, 0.0), (8, 2, 64, 16, 0.05)])
def test_block_factors_match_reference_iteration(H, bits, group, rank, left):
    """The block kernel's factors against the reference's power iteration in float64, started from the same P0, on the error
    matrix rebuilt on the host from the cache contents (codes, scale, mn, outlier lists) and the fp16 block."""
    from gear_amd import cache as gc
    torch.manual_seed(7)
    B, seed = 1, 11
    cc = dict(compress_method="gearslKIVI", group_size=group, residual=64, quantize_bit=bits, rank=rank, rankv=rank, loop=3, left=left)
    c = gc.GearKVCache(B, H, 64, cc, "cuda", seed=seed)
    k = torch.randn(B, H, 64, 128).half()
    v = torch.randn(B, H, 64, 128).half()
    k[:, :, 3::7, 5::13] *= 5
    for i in range(64):
        c.append(k[:, :, i:i + 1].cuda(), v[:, :, i:i + 1].cuda())
    c.maybe_compress()
    torch.cuda.synchronize()
    assert c.n_comp == 64 and gc.block_kernel_status() == 0
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    P0k = host(torch.rand((B, H, 128, rank), device="cuda", generator=g))
    P0v = host(torch.rand((B, H, 128, rank), device="cuda", generator=g))
    fpi = 32 // bits
    kcode = orc.unpack_tensor(host(c.kcode[:, :, :, :64 // fpi]), bits, 3)       # [B,H,D,64]
    vcode = orc.unpack_tensor(host(c.vcode[:, :, :64]), bits, 3)                # [B,H,64,D]
    kn, vn = k.numpy(), v.numpy()
    seg = 1
    for h in range(H):
        # K: outlier mask from the lists, fill = fp16(mean of the channel over the block)
        om = np.zeros((64, 128), bool)
        if c.kk_blk:
            oi = host(c.koidx[0, h, :, :, :c.kk_blk]).astype(np.int64) & 0xFFFF      # [D,2,kk]
            for d in range(128):
                om[oi[d].reshape(-1), d] = True
        fill = kn[0, h].astype(np.float64).mean(0).astype(np.float32).astype(np.float16)[None, :].repeat(64, 0)
        E = _error_matrix(kn[0, h], kcode[0, h].T, host(c.kscale[0, h, :, :64 // group]), host(c.kmn[0, h, :, :64 // group]), om, fill,
                          group, "k")
        got = host(c.kQtok[0, h, :64]).astype(np.float64) @ host(c.kPseg[seg, 0, h]).astype(np.float64).T
        ref = _ref_lowrank(E, P0k[0, h], 3)
        assert rel_fro(got, ref) < 3e-3, ("K", h, rel_fro(got, ref))
        om = np.zeros((64, 128), bool)
        if c.kv:
            oi = host(c.voidx[0, :64]).astype(np.int64) & 0xFFFF                      # [64, 2kv]
            for t in range(64):
                cols = oi[t][(oi[t] // 128) == h] % 128
                om[t, cols] = True
        rows = vn[0].transpose(1, 0, 2).reshape(64, H * 128).astype(np.float64)
        fill = rows.mean(1).astype(np.float32).astype(np.float16)[:, None].repeat(128, 1)
        E = _error_matrix(vn[0, h], vcode[0, h], host(c.vscale[0, h, :64]), host(c.vmn[0, h, :64]), om, fill, group, "v")
        got = host(c.vQtok[0, h, :64]).astype(np.float64) @ host(c.vPseg[seg, 0, h]).astype(np.float64).T
        ref = _ref_lowrank(E, P0v[0, h], 3)
        assert rel_fro(got, ref) < 3e-3, ("V", h, rel_fro(got, ref))


def test_block_kernel_is_one_launch_and_rejects_bad_views():
    """Argument errors come back as status codes through the C ABI (no launch)."""
    import ctypes as C_
    from gear_amd import _lib as L
    from gear_amd import cache as gc
    cc = dict(compress_method="gearslKIVI", group_size=64, residual=64, quantize_bit=2, rank=4, rankv=4, loop=3, left=0.02)
    c = gc.GearKVCache(1, 2, 256, cc, "cuda")
    view = gc._view(c.bufs, c.dims, 1, 2, 128, 0, 0, kwin=True)
    lib = L.load()
    ws = torch.zeros(lib.gear_compress_block_workspace(1, 2), dtype=torch.uint8, device="cuda")
    P0 = torch.rand(1, 2, 128, 4, device="cuda")
    args = lambda t_off: (C_.byref(view), t_off, 0, 3, P0.data_ptr(), P0.data_ptr(), c.kPseg.data_ptr(), c.vPseg.data_ptr(), 2,
                          0, 0, ws.data_ptr(), ws.numel(), L.stream_ptr())
    assert lib.gear_compress_block(*args(32)) < 0 and b"offset" in lib.gear_last_error()
    assert lib.gear_compress_block(*args(256)) < 0
    view.mode = 1
    assert lib.gear_compress_block(*args(0)) < 0 and b"mode" in lib.gear_last_error()
    view.mode = 0
    assert lib.gear_compress_block(*args(0)) == 0
    torch.cuda.synchronize()


def test_block_kernel_handoff_under_uneven_load():
    """The V row hand-off (row-duty waves -> V tiles: write-through mask / mean, drained, flag; relaxed polls) checked word for
    word while other work competes for the chip: a second stream streams 1 GiB through HBM, the block kernel runs 25 times on
    fresh data with a varying number of tiles per launch, and every launch's V payload must equal the kernel chain's (whose
    selection happens in a separate launch).  Uniformly idle chips hide hand-off bugs (MI355X_MICROARCH.md)."""
    from gear_amd import cache as gc
    torch.manual_seed(11)
    cc = dict(compress_method="gearslKIVI", group_size=64, residual=64, quantize_bit=2, rank=4, rankv=4, loop=3, left=0.02)
    noise = torch.empty(1 << 28, dtype=torch.float16, device="cuda")
    side = torch.cuda.Stream()
    for it in range(25):
        layers, H = (1 + it % 5), (4, 8, 12, 32)[it % 4]
        out = {}
        for use_block in (True, False):
            pool, caches = _mk(layers, 1, H, cc, 256, use_block, seed=it)
            torch.manual_seed(1000 + it)
            pool.buf["kwin"].copy_(torch.randn(pool.buf["kwin"].shape, device="cuda").half())
            pool.buf["vwin"].copy_(torch.randn(pool.buf["vwin"].shape, device="cuda").half())
            for c in caches:
                c.n_comp, c.n_win, c.seg0, c.kk0 = 64, 64, 64, 0
            torch.cuda.synchronize()
            if use_block:
                with torch.cuda.stream(side):
                    for _ in range(3):
                        noise.mul_(1.0001)
            gc.USE_BLOCK_KERNEL = use_block
            pool.compress_all()
            torch.cuda.synchronize()
            out[use_block] = pool
        for name in ("vcode", "vscale", "vmn", "voidx", "voval", "vochunk", "vcnt", "kcode", "kscale", "kmn", "koidx", "koval"):
            a, b = host(out[False].buf[name]), host(out[True].buf[name])
            if a.dtype == np.float16:
                a, b = a.view(np.uint16), b.view(np.uint16)
            assert np.array_equal(a, b), (it, name)
    assert gc.block_kernel_status() == 0
