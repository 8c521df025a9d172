"""Multi-process CPU tests (gloo, world_size 2) of the head-sharded path: partition math, the all-gather
assembly, and equivalence of the sharded attention module with the unsharded one on the dense (prefill) branch --
the packed-cache branch needs the GPU and is covered by the -m gpu tests."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gear_amd import parallel


def test_partition_math():
    assert parallel.shard_heads(32, 8, 3) == (12, 16)
    assert [parallel.shard_heads(40, 4, r) for r in range(4)] == [(0, 10), (10, 20), (20, 30), (30, 40)]
    assert parallel.shard_heads(8, 8, 7) == (7, 8)          # 70B: one KV head per GPU
    with pytest.raises(ValueError):
        parallel.shard_heads(32, 3, 0)
    assert parallel.outliers_per_shard(40, 8) == 5 and parallel.outliers_per_shard(40, 1) == 40
    assert parallel.outliers_per_shard(0, 4) == 0 and parallel.outliers_per_shard(3, 8) == 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gear_amd.modeling_llamagear import LlamaAttention_GEAR, LlamaConfigLite
        torch.manual_seed(0)                                   # same weights / inputs on every rank
        cfg = LlamaConfigLite(hidden_size=512, num_attention_heads=4, num_key_value_heads=2, num_hidden_layers=1)
        cc = dict(compress_method="gearlKIVI", group_size=64, residual=64, quantize_bit=2, rank=2, rankv=2, loop=3)
        full = LlamaAttention_GEAR(0, cfg, cc)
        local = parallel.shard_attention_weights(full, LlamaAttention_GEAR(0, cfg, cc, tp_rank=rank, tp_world=world))
        x = torch.randn(2, 24, 512)
        mask = torch.full((24, 24), torch.finfo(torch.float32).min).triu(1)[None, None].expand(2, 1, 24, 24)
        ref, _, _ = full(x, attention_mask=mask, use_cache=False)
        got, _, none = local(x, attention_mask=mask, use_cache=False)
        assert none is None
        assert local.num_heads == 2 and local.num_key_value_heads == 1 and local.q_proj.weight.shape[0] == 256
        err = float((got - ref).abs().max())
        # the gather itself: rank r's block lands at slot r
        g = parallel.all_gather_heads(torch.full((1, 1, 4), float(rank)), world)
        ok_gather = g.flatten().tolist() == [0.0] * 4 + [1.0] * 4
        # bench-style timing reduction: MAX over ranks
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ret[rank] = (err, ok_gather, float(t.item()))
    finally:
        dist.destroy_process_group()


def test_sharded_attention_matches_unsharded_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        err, ok_gather, tmax = ret[r]
        assert err < 1e-4, err
        assert ok_gather
        assert tmax == 2.0


def _sel_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(3)                                   # the same full tensor on every rank
        NB, Ht, T, D, k = 2, 4, 24, 128, 5
        v = torch.randn(NB, Ht, T, D).half()
        v[0, :, 3, :10] = 0.5                                  # ties across the two ranks' heads
        v[1, 1, 2, 7] = -0.0
        v[0, 0, 0, :] = 1.0
        v[1, :, 5, :] = 0.25                                   # a constant row: every element ties
        Hl = Ht // world
        mine = v[:, rank * Hl:(rank + 1) * Hl].contiguous()
        filled, mask, oidx, oval = parallel.exact_v_selection(mine, k, rank, world)
        # brute force on the full rows: k smallest / k largest by (order key, lower column first), fill = fp16(float(exact mean))
        rows = v.permute(0, 2, 1, 3).reshape(NB * T, Ht * D)
        key = parallel._order_key(rows.view(torch.int16).to(torch.int64) & 0xFFFF)
        col = torch.arange(Ht * D)
        want = torch.zeros_like(rows, dtype=torch.bool)
        want_side = [torch.zeros_like(want), torch.zeros_like(want)]           # [small, large]
        for r in range(rows.shape[0]):
            comp_l = key[r] * 65536 + (65535 - col)
            comp_s = (65535 - key[r]) * 65536 + (65535 - col)
            want_side[1][r, torch.topk(comp_l, k).indices] = True
            want_side[0][r, torch.topk(comp_s, k).indices] = True
        want = want_side[0] | want_side[1]
        lo = rank * Hl * D
        got_rows = mask.permute(0, 2, 1, 3).reshape(NB * T, Hl * D)
        ok_mask = bool(torch.equal(got_rows, want[:, lo:lo + Hl * D]))
        fill = (rows.double().sum(1) / rows.shape[1]).float().half()
        f_rows = filled.permute(0, 2, 1, 3).reshape(NB * T, Hl * D)
        ok_fill = bool(torch.equal(f_rows[got_rows], fill[:, None].expand_as(f_rows)[got_rows]))
        ok_keep = bool(torch.equal(f_rows[~got_rows], rows[:, lo:lo + Hl * D][~got_rows]))
        # lists: small side then large side, each ascending, unused slots 0xFFFF / value bits 0
        idx = (oidx.view(NB * T, 2 * k).to(torch.int64) & 0xFFFF)
        ok_lists = True
        for r in range(rows.shape[0]):
            for side in (0, 1):
                ent = idx[r, side * k:(side + 1) * k]
                used = ent[ent != 0xFFFF]
                ok_lists &= bool((ent[len(used):] == 0xFFFF).all())
                ok_lists &= bool(torch.equal(used, torch.nonzero(want_side[side][r, lo:lo + Hl * D]).flatten()))   # ascending
                vals = oval.view(NB * T, 2 * k)[r, side * k:side * k + len(used)]
                ok_lists &= bool(torch.equal(vals.view(torch.int16), rows[r, lo + used].view(torch.int16)))
        ret[rank] = (ok_mask, ok_fill, ok_keep, ok_lists, int((idx != 0xFFFF).sum()))
    finally:
        dist.destroy_process_group()


def test_exact_cross_shard_v_selection_world2():
    """SURVEY 8(e)'s candidate exchange (parallel.exact_v_selection): every rank ends up with exactly the elements of ITS heads that
    the reference's whole-row top-k / bottom-k picks (compress_function.py:304-311), ties by lower global column, the global fp16
    fill value, and sentinel-padded sorted lists; the ranks' list entries add up to 2 k per row."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_sel_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        assert all(ret[r][:4]), (r, ret[r])
    assert ret[0][4] + ret[1][4] == 2 * 24 * 2 * 5
