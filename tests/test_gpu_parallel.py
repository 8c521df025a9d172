"""Head-sharded decode on ONE GPU box: two processes (gloo, both on cuda:0) run the sharded FastGearDecoder; rank 0 also runs
the unsharded one.  Checks: the shards' K cache == the matching head slice of the unsharded cache (exact, outlier lists
included -- every K quantity lives inside a head), V backbone exact when no outliers are selected across heads, and the
decode logits of the sharded run track the unsharded run.  (The 8-GPU RCCL run is the driver's; this covers the control
flow and the arithmetic of the sharded path.)"""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, left, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from gear_amd.fast_decode import FastGearDecoder
        from gear_amd.modeling_llamagear import LlamaConfigLite, LlamaForCausalLM_GEARKIVI
        cfg = LlamaConfigLite(vocab_size=1000, hidden_size=512, intermediate_size=1024, num_hidden_layers=2,
                              num_attention_heads=4, num_key_value_heads=2, k_bits=2, v_bits=2)
        cc = dict(compress_method="KIVI", group_size=64, residual=64, quantize_bit=2, rank=0, rankv=0, loop=0, left=left)
        torch.manual_seed(0)
        model = LlamaForCausalLM_GEARKIVI(cfg, cc).half().cuda().eval()
        torch.manual_seed(1)
        ids = torch.randint(0, 1000, (1, 150)).cuda()
        sh = FastGearDecoder(model, 512, seed=5, tp_rank=rank, tp_world=world)
        full = FastGearDecoder(model, 512, seed=5)
        ls, lf = sh.prefill(ids), full.prefill(ids)
        worst = float(torch.nn.functional.cosine_similarity(ls.float(), lf.float()).min())
        tok = lf.argmax(-1, keepdim=True)
        for _ in range(80):                                   # crosses a block boundary
            ls, lf = sh.step(tok), full.step(tok)
            worst = min(worst, float(torch.nn.functional.cosine_similarity(ls.float(), lf.float()).min()))
            tok = lf.argmax(-1, keepdim=True)
        cs, cf = sh.layers[1]["cache"], full.layers[1]["cache"]
        h0 = rank * cs.H
        ok = {}
        # with outliers the V reconstructions of the two runs differ (k / world per shard), so the hidden states and with
        # them the K / V of the DECODED tokens drift apart: compare the prompt segment, which is computed replicated
        n0 = cs.seg0 if left else cs.Tmax
        sl = {"kcode": n0 // cs.fpi, "kscale": n0 // cs.group, "kmn": n0 // cs.group, "koidx": cs.kk0, "koval": cs.kk0}
        for name in ("kcode", "kscale", "kmn") + (("koidx", "koval") if left else ()):
            ok[name] = bool(torch.equal(getattr(cs, name)[..., :sl[name]], getattr(cf, name)[:, h0:h0 + cs.H][..., :sl[name]]))
        if not left:
            for name in ("vcode", "vscale", "vmn"):
                ok[name] = bool(torch.equal(getattr(cs, name), getattr(cf, name)[:, h0:h0 + cs.H]))
        ret[rank] = (worst, ok, (cs.n_comp, cs.n_win, cs.kk0, cs.kv), (cf.n_comp, cf.n_win, cf.kk0, cf.kv))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("left", [0.0, 0.04])
def test_head_sharded_decoder_matches_unsharded(left):
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), left, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        worst, ok, st_s, st_f = ret[r]
        assert all(ok.values()), (r, ok)
        assert st_s[:3] == st_f[:3], (st_s, st_f)              # same block structure, same K outlier count per channel row
        if left:
            assert st_s[3] == max(1, st_f[3] // world)         # V rows: k / world inside a shard's heads (documented divergence)
        assert worst > (0.999 if left else 0.9999), (r, worst)


# ------------------------------------------------------------------------------------------------ peer exchange (gear_xchg_*)
def _xchg_worker(rank, world, port, ret):
    import time
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from gear_amd.parallel import PeerHeadGather
        dev = torch.device("cuda:0")
        B, width = 3, 256
        g = PeerHeadGather(world, rank, B, width, torch.float16, dev)
        assert g.ok, g.error
        bad = 0
        gen = torch.Generator().manual_seed(7)                     # the same stream of inputs on every rank
        for it in range(300):
            xs = torch.randn((world, B, width), generator=gen).half()
            if (it + rank) % 7 == 0:
                time.sleep(0.002)                                  # uneven arrival: the waiting side really waits
            got = g(xs[rank].to(dev)).clone()
            want = xs.permute(1, 0, 2).reshape(B, world * width).to(dev)
            bad += int(not torch.equal(got, want))
        # back-to-back exchanges on the stream without a host sync in between (what a graph replay issues)
        xs = torch.randn((64, world, B, width), generator=gen).half().to(dev)
        outs = [g(xs[i, rank]).clone() for i in range(64)]
        torch.cuda.synchronize()
        for i in range(64):
            bad += int(not torch.equal(outs[i], xs[i].permute(1, 0, 2).reshape(B, world * width)))
        g.check()
        ret[rank] = bad
        g.close()
    finally:
        dist.destroy_process_group()


def test_peer_exchange_two_processes_one_gpu():
    """gear_xchg_allgather between two processes that map each other's exchange area with hipIpc (both on cuda:0 here; on an
    MI355X node the same handles map xGMI peer memory): every rank gets every rank's rows, under uneven arrival and for
    exchanges queued back to back."""
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_xchg_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: 0, 1: 0}


def _graph_worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from gear_amd.fast_decode import FastGearDecoder
        from gear_amd.modeling_llamagear import LlamaConfigLite, LlamaForCausalLM_GEARKIVI
        cfg = LlamaConfigLite(vocab_size=1000, hidden_size=512, intermediate_size=1024, num_hidden_layers=2,
                              num_attention_heads=4, num_key_value_heads=2, k_bits=2, v_bits=2)
        cc = dict(compress_method="gearslKIVI", group_size=64, residual=64, quantize_bit=2, rank=4, rankv=4, loop=3, left=0.0)
        torch.manual_seed(0)
        model = LlamaForCausalLM_GEARKIVI(cfg, cc).half().cuda().eval()
        torch.manual_seed(1)
        ids = torch.randint(0, 1000, (1, 150)).cuda()
        eager = FastGearDecoder(model, 512, seed=5, tp_rank=rank, tp_world=world)
        graph = FastGearDecoder(model, 512, seed=5, tp_rank=rank, tp_world=world)
        full = FastGearDecoder(model, 512, seed=5)
        assert eager.gather.capturable and graph.gather.capturable, (eager.exchange_error, graph.exchange_error)
        n_new = 90                                                # crosses a block boundary inside the replayed part
        a = eager.generate(ids, 150 + n_new, graph=False)
        b = graph.generate(ids, 150 + n_new, graph=True)
        c = full.generate(ids, 150 + n_new, graph=False)
        graph.gather.check()
        eager.gather.check()
        ret[rank] = (bool(torch.equal(a, b)), float((a == c).float().mean()), graph.graph is not None)
        eager.close()
        graph.close()
    finally:
        dist.destroy_process_group()


def test_head_sharded_graph_decode():
    """The sharded token step replayed as ONE hipGraph per token (the exchange is a launch of the graph): same tokens as the
    sharded eager steps; the unsharded decoder's tokens agree (no V outliers: the shards' payload is the unsharded payload)."""
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_graph_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    for r in range(world):
        same, agree, captured = ret[r]
        assert captured
        assert same, r
        assert agree > 0.9, (r, agree)
