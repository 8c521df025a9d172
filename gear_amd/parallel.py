"""Head-wise sharding of the GEAR cache + attention across GPUs (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on MI355X, "gloo" in the CPU tests).

The reference has no distributed code at all; this is the build's design for
BASELINE.json's 1/2/4/8-GPU rows:
  * KV heads are split contiguously across ranks; every quantization group, low-rank factor pair and K outlier row
    lives inside one head, so compress / decompress need no communication.  The one exception is the simulated
    path's V outlier selection, a top-k over the whole token row ACROSS heads (compress_function.py:304-311): the shards
    run the EXACT selection (exact_v_selection below: one small all-gather of per-row candidates, after which the
    concatenated shard payloads are the unsharded payload bit for bit); v_selection="per_shard" (k / world inside the
    shard's own heads, rounds 1-3) remains as an option and as the only mode without a process group.
  * softmax is per head, so attention is local; the only exchange is an all-gather of the per-rank attention output
    [B, q, H_local*D] (a few KiB per layer per token: latency-bound single hop over xGMI), after which every rank
    applies the replicated o_proj.
"""
from __future__ import annotations

from typing import Tuple

import torch


def shard_heads(n_heads: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [begin, end) head range of `rank`."""
    if n_heads % world:
        raise ValueError(f"{n_heads} heads do not divide across {world} ranks")
    per = n_heads // world
    return rank * per, (rank + 1) * per


def outliers_per_shard(k_full: int, world: int) -> int:
    """Per-side outlier count of a V token row restricted to one shard's heads (k scales with the row length)."""
    return max(1, k_full // world) if k_full > 0 else 0


def all_gather_stack(t: torch.Tensor, world: int, group=None) -> torch.Tensor:
    """t [...] on every rank -> [world, ...] (rank r's tensor at slot r).  gloo cannot gather device tensors: staged through the host
    there (the one-GPU tests); RCCL gathers in place."""
    import torch.distributed as dist
    t = t.contiguous()
    shp = (world * t.shape[0],) + tuple(t.shape[1:])      # (ranks concatenated on dim 0: the one output shape gloo accepts too)
    if dist.get_backend(group) == "gloo" and t.is_cuda:
        out = torch.empty(shp, dtype=t.dtype)
        dist.all_gather_into_tensor(out, t.cpu(), group=group)
        return out.view((world,) + tuple(t.shape)).to(t.device)
    out = torch.empty(shp, dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t, group=group)
    return out.view((world,) + tuple(t.shape))


def _order_key(bits: torch.Tensor) -> torch.Tensor:
    """fp16 bit patterns (int64, 0 .. 65535) -> ascending 16-bit order keys, -0 == +0 (csrc/ktile.h sort_key16)."""
    bits = torch.where(bits == 0x8000, torch.zeros_like(bits), bits)
    return torch.where((bits & 0x8000) != 0, (~bits) & 0xFFFF, bits | 0x8000)


def exact_v_selection(v: torch.Tensor, k: int, rank: int, world: int, group=None):
    """Exact cross-shard outlier selection of V token rows (the simulated path's gears_tokenQ: the k smallest and k largest of a
    token's row ACROSS ALL heads, compress_function.py:297-333; ties: lower index first, the build's rule).

    v fp16 [NB, H_local, T, 128]: this rank's heads (ranks hold contiguous head ranges in rank order).  Every rank contributes, per
    row and side, its k best candidates as unique composites (order key, then lower GLOBAL column first) and its exact fp64 row sum;
    after ONE all-gather every rank finds the row's k-th composite per side and keeps the local elements at or beyond it.
    Exchange: 8 (2 k + 1) bytes per row and rank -- 64 rows per layer at a decode-time block boundary; a prompt's rows go in one go
    too, layer by layer (Llama-2-7B, 4k tokens, 8 GPUs: 2.7 MB sent and 21 MB received per rank and layer, once per prompt).

    Returns (filled [NB,H,T,128] fp16: outliers replaced by the fp16 global row mean, mask [NB,H,T,128] bool,
             oidx int16 [NB,T,2k]: local column h*128 + d of this rank's outliers, small side then large side, each ascending,
             unused slots 0xFFFF (lies beyond every head bound, so the chunk index's terminal entry becomes the count),
             oval fp16 [NB,T,2k])."""
    NB, H, T, D = v.shape
    Ll = H * D
    if k > Ll:
        raise ValueError(f"exact_v_selection: k = {k} outliers per side exceed the shard's row length {Ll} (sparsity > 1 / world)")
    rows = v.permute(0, 2, 1, 3).reshape(NB * T, Ll)                       # row (nb, t): this rank's segment
    bits = rows.view(torch.int16).to(torch.int64) & 0xFFFF
    key = _order_key(bits)
    col = torch.arange(Ll, device=v.device, dtype=torch.int64)
    inv_g = 0xFFFFF - (rank * Ll + col)                                    # lower global column first (20 bits: <= 8192 * 128 columns)
    cl = (key << 20) | inv_g                                               # large side: bigger composite = selected first
    cs = ((0xFFFF - key) << 20) | inv_g                                    # small side
    kc = min(k, Ll)
    cand = torch.full((NB * T, 2 * k + 1), -1, dtype=torch.int64, device=v.device)
    cand[:, :kc] = torch.topk(cl, kc, dim=1).values
    cand[:, k:k + kc] = torch.topk(cs, kc, dim=1).values
    cand[:, 2 * k] = rows.double().sum(1).view(torch.int64)                # exact: fp16 values add exactly in fp64
    allc = all_gather_stack(cand, world, group) if world > 1 else cand[None]
    thr_l = torch.topk(allc[:, :, :k].permute(1, 0, 2).reshape(NB * T, world * k), k, dim=1).values[:, k - 1]
    thr_s = torch.topk(allc[:, :, k:2 * k].permute(1, 0, 2).reshape(NB * T, world * k), k, dim=1).values[:, k - 1]
    total = allc[:, :, 2 * k].contiguous().view(torch.float64).sum(0)
    fill = (total / float(Ll * world)).to(torch.float32).to(torch.float16)  # the kernels' fill: fp16(float(exact sum / length))
    m_l, m_s = cl >= thr_l[:, None], cs >= thr_s[:, None]
    mask_rows = m_l | m_s
    filled = torch.where(mask_rows, fill[:, None], rows)
    big = torch.full_like(col, 0xFFFF)
    i_s = torch.topk(torch.where(m_s, col, big), k, dim=1, largest=False).values if k <= Ll else None
    i_l = torch.topk(torch.where(m_l, col, big), k, dim=1, largest=False).values
    idx = torch.cat([i_s, i_l], 1)                                         # [rows, 2k], 0xFFFF = unused
    val = torch.gather(bits, 1, idx.clamp(max=Ll - 1))
    val = torch.where(idx == 0xFFFF, torch.zeros_like(val), val)
    to4 = lambda t: t.view(NB, T, H, D).permute(0, 2, 1, 3).contiguous()
    u16 = lambda t: ((t + 0x8000) % 0x10000 - 0x8000).to(torch.int16)      # 0 .. 65535 -> the int16 with the same bits
    oidx = u16(idx).view(NB, T, 2 * k)
    oval = u16(val).view(torch.float16).view(NB, T, 2 * k)
    return to4(filled), to4(mask_rows), oidx, oval


def v_candidates(v: torch.Tensor, k: int, rank: int, T: int = None) -> torch.Tensor:
    """gear_vsel_candidates on this rank's heads: v fp16 [NB, H_local, Tp, 128] contiguous (rows = the first T <= Tp tokens) ->
    int32 [NB*T, 2k + 2] (k best global composites per side + the fp64 bit pattern of the local row sum in two words)."""
    from . import _lib as L
    NB, H, Tp, D = v.shape
    T = Tp if T is None else T
    L.require_gpu(v)
    assert v.is_contiguous() and v.dtype == torch.float16 and D == 128
    rows = NB * T
    cand = torch.empty((rows, 2 * k + 2), dtype=torch.int32, device=v.device)
    L.check(L.load().gear_vsel_candidates(L.ptr(v), rows, T, H * Tp * D, D, H, D, Tp * D, k, rank * H * D, L.ptr(cand),
                                          L.stream_ptr(v)), "gear_vsel_candidates")
    return cand


def v_thresholds(cand_all: torch.Tensor, k: int, row_len_total: int, mode: int = 0):
    """gear_vsel_thresholds: cand_all int32 [world, rows, 2k + 2] -> (thr int32 [rows, 2], fill float32 [rows])."""
    from . import _lib as L
    world, rows = cand_all.shape[0], cand_all.shape[1]
    cand_all = cand_all.contiguous()
    thr = torch.empty((rows, 2), dtype=torch.int32, device=cand_all.device)
    fill = torch.empty((rows,), dtype=torch.float32, device=cand_all.device)
    L.check(L.load().gear_vsel_thresholds(L.ptr(cand_all), world, rows, k, row_len_total, mode, L.ptr(thr), L.ptr(fill),
                                          L.stream_ptr(cand_all)), "gear_vsel_thresholds")
    return thr, fill


def exact_v_thresholds(v: torch.Tensor, k: int, rank: int, world: int, group=None, mode: int = 0, T: int = None):
    """The exact cross-shard V outlier selection on the GPU (csrc/vsel.hip; round 5): two launches + ONE all-gather where
    exact_v_selection above runs ~25 torch launches incl. two topk.  v fp16 [NB, H_local, Tp, 128] contiguous (the first T <= Tp
    tokens of every head are the rows); returns (thr uint32-as-int32 [NB*T, 2], fill float32 [NB*T]) for
    gear_compress_value_sharded: per row the k-th largest composite per side over ALL ranks' candidates and the full row's mean
    (mode 0: rounded to fp16).  Exchange: 4 (2k + 2) bytes per row and rank."""
    cand = v_candidates(v, k, rank, T)
    allc = all_gather_stack(cand, world, group) if world > 1 else cand[None]
    return v_thresholds(allc, k, world * v.shape[1] * v.shape[3], mode)


def all_gather_heads(x: torch.Tensor, world: int, group=None) -> torch.Tensor:
    """x [B, q, H_local*D] on every rank -> [B, q, world*H_local*D] with rank r's heads at slot r."""
    import torch.distributed as dist
    x = x.contiguous()
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)   # ranks concatenated on dim 0
    with torch.no_grad():                                 # (inference path; gloo's implementation writes through views)
        dist.all_gather_into_tensor(out, x.detach(), group=group)   # one collective into one buffer (no tensor list, no cat)
    return out.view((world,) + tuple(x.shape)).movedim(0, -2).reshape(*x.shape[:-1], world * x.shape[-1])


class HeadGather:
    """All-gather of the per-rank attention output of a decode step into a PRE-ALLOCATED buffer with
    all_gather_into_tensor: one collective per layer per token, no list of tensors, no torch.cat (the exchange is a few KiB
    and latency-bound: a single hop over xGMI on MI355X, where backend "nccl" is RCCL).
    x [B, H_local*D] on every rank -> [B, world*H_local*D] with rank r's heads at slot r."""

    capturable = False      # (class default; an instance on the RCCL backend probes whether the collective can be captured)

    def __init__(self, world: int, batch: int, width: int, dtype, device, group=None, probe_capture: bool = True):
        import torch.distributed as dist
        self.world, self.B, self.width, self.group = world, batch, width, group
        self.buf = torch.empty((world * batch, width), dtype=dtype, device=device)       # ranks concatenated on dim 0
        # gloo cannot gather device tensors: the one-GPU debug mode of bench.py / the tests stage through the host
        backend = dist.get_backend(group)
        self.stage = backend == "gloo" and torch.device(device).type == "cuda"
        self.capture_error = None
        if backend == "nccl" and torch.device(device).type == "cuda" and probe_capture:
            self.capturable = self._probe_capture(dtype, device)

    def _probe_capture(self, dtype, device) -> bool:
        """RCCL collectives can be recorded into a hipGraph; whether THIS build / communicator does it is found out once, on
        every rank alike: capture two gathers, replay twice, compare with what the peers must have sent, agree across ranks.
        Any exception or mismatch -> eager steps only (the state before round 4)."""
        import torch.distributed as dist
        fine = True
        try:
            rank = dist.get_rank(self.group)
            src = torch.full((self.B, self.width), float(rank + 1), dtype=dtype, device=device)
            self(src)                                            # warm-up: communicator set-up must not happen under capture
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = self(src).clone()
                out2 = self(src * 2).clone()
            for _ in range(2):
                g.replay()
            torch.cuda.synchronize()
            want = torch.arange(1, self.world + 1, dtype=dtype, device=device).view(1, self.world, 1)
            fine = bool(torch.equal(out.view(self.B, self.world, self.width), want.expand(self.B, self.world, self.width)))
            fine = fine and bool(torch.equal(out2.view(self.B, self.world, self.width), (2 * want).expand(self.B, self.world, self.width)))
        except Exception as e:                                   # (capture refused: not an error of the exchange itself)
            self.capture_error = e
            fine = False
        good = [None] * self.world
        dist.all_gather_object(good, fine, group=self.group)
        return all(good)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        import torch.distributed as dist
        if self.stage:
            hb = torch.empty(self.buf.shape, dtype=self.buf.dtype)
            dist.all_gather_into_tensor(hb, x.contiguous().cpu(), group=self.group)
            self.buf.copy_(hb)
        else:
            dist.all_gather_into_tensor(self.buf, x.contiguous(), group=self.group)
        if self.B == 1:
            return self.buf.view(1, self.world * self.width)          # rank-major == head-major for one row
        return self.buf.view(self.world, self.B, self.width).permute(1, 0, 2).reshape(self.B, self.world * self.width)


class PeerHeadGather:
    """The same exchange without a collective call: every rank's slice is STORED into every rank's exchange area (uncached
    device memory mapped across processes with hipIpc -- xGMI peer memory on an MI355X node) by one single-workgroup launch per
    layer per token (gear_xchg_allgather, include/gear_hip.h), flags and epoch counter on the device.  No host involvement, so
    the launch sits inside the hipGraph of the token step (FastGearDecoder.step_graph with tp_world > 1).
    x [B, H_local*D] on every rank -> [B, world*H_local*D] with rank r's heads at slot r.

    The handles travel through all_gather_object of whatever process group is up (gloo or RCCL); nothing else uses the group.
    Construction is collective; `ok` says whether the areas could be set up AND a first exchange returned what the peers sent,
    agreed across ranks, so every rank takes the same decision when it falls back to HeadGather."""

    capturable = True

    def __init__(self, world: int, rank: int, batch: int, width: int, dtype, device, group=None):
        import ctypes as C
        import torch.distributed as dist
        from . import _lib as L
        self.world, self.rank, self.B, self.width, self.group = world, rank, batch, width, group
        self.row_bytes = width * torch.empty((), dtype=dtype).element_size()
        self.lib = L.load()
        self.base, self.mapped, self.ok = None, [], False
        self._closed = False
        err = None
        try:
            area = self.lib.gear_xchg_bytes(world, batch * self.row_bytes)
            if area == 0 or self.row_bytes % 16:
                raise ValueError(f"exchange of {batch} x {self.row_bytes} bytes over {world} ranks is not supported")
            with torch.cuda.device(device):
                base = C.c_void_p()
                L.check(self.lib.gear_xchg_alloc(area, C.byref(base)), "gear_xchg_alloc")
                self.base = base.value
                h = C.create_string_buffer(64)
                L.check(self.lib.gear_xchg_export(self.base, h), "gear_xchg_export")
        except Exception as e:                                  # the decision below is collective: no early exit
            err, h = e, None
        handles = [None] * world
        dist.all_gather_object(handles, None if h is None else h.raw, group=group)
        ptrs = []
        if err is None and all(x is not None for x in handles):
            try:
                with torch.cuda.device(device):
                    for r in range(world):
                        if r == rank:
                            ptrs.append(self.base)
                            continue
                        p = C.c_void_p()
                        L.check(self.lib.gear_xchg_open(handles[r], C.byref(p)), "gear_xchg_open")
                        self.mapped.append(p.value)
                        ptrs.append(p.value)
            except Exception as e:
                err = e
        self.error = err
        self.table = torch.tensor(ptrs if len(ptrs) == world else [0] * world, dtype=torch.int64, device=device)
        self.out = torch.empty((batch, world * width), dtype=dtype, device=device)
        self.status = torch.zeros(1, dtype=torch.int32, device=device)
        good = [None] * world
        dist.all_gather_object(good, err is None, group=group)          # (also: every area is mapped before anyone stores)
        if all(good):
            # first exchange: rank r sends the value r + 1 everywhere
            probe = torch.full((batch, width), float(rank + 1), dtype=dtype, device=device)
            got = self(probe).view(batch, world, width)
            want = torch.arange(1, world + 1, dtype=dtype, device=device).view(1, world, 1).expand_as(got)
            fine = bool(torch.equal(got, want)) and int(self.status.item()) == 0
            dist.all_gather_object(good, fine, group=group)
            self.ok = all(good)
            if not fine and self.error is None:
                self.error = RuntimeError(f"first exchange returned wrong data (status {int(self.status.item())})")
        if not self.ok:
            self.close()

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        from . import _lib as L
        x = x.contiguous()
        assert x.shape == (self.B, self.width) and x.dtype == self.out.dtype
        L.check(self.lib.gear_xchg_allgather(L.ptr(x), self.B, self.row_bytes, self.world, self.rank, L.ptr(self.table),
                                             L.ptr(self.out), L.ptr(self.status), L.stream_ptr(x)), "gear_xchg_allgather")
        return self.out

    def check(self):
        """Raise if an exchange ever gave up waiting for a peer (reads one word from the device: not for every token --
        FastGearDecoder calls it at block boundaries and at the end of generate()).  The kernel latches the status: after a
        timeout every later exchange copies out without waiting, so everything since the last clean check is suspect."""
        if int(self.status.item()):
            raise RuntimeError("gear_xchg_allgather: a peer did not deliver its slice within the time limit; the tokens "
                               "produced since the last check are not trustworthy")

    def close(self):
        """Unmap the peers' areas and free the own one.  Collective in spirit: call it on every rank once no exchange is in
        flight (the owner's free comes after a barrier so no peer still has stores under way)."""
        import torch.distributed as dist
        if self._closed:
            return
        # (the barrier below is entered by EVERY rank that constructed the object, also one whose allocation failed and
        # has nothing to free: an asymmetric early return would leave the healthy ranks waiting in it)
        self._closed = True
        torch.cuda.synchronize()
        for p in self.mapped:
            self.lib.gear_xchg_close(p)
        self.mapped = []
        try:
            dist.barrier(group=self.group)
        except Exception:
            pass
        if self.base is not None:
            self.lib.gear_xchg_free(self.base)
            self.base = None


def shard_attention_weights(full_attn, local_attn):
    """Copy the rank-local slices of a full LlamaAttention_GEAR's projections into a head-sharded instance
    (q/k/v are column-parallel over heads, o_proj is replicated)."""
    D = full_attn.head_dim
    r, w = local_attn.tp_rank, local_attn.tp_world
    qb, qe = shard_heads(full_attn.total_heads, w, r)
    kb, ke = shard_heads(full_attn.config.num_key_value_heads, w, r)
    with torch.no_grad():
        local_attn.q_proj.weight.copy_(full_attn.q_proj.weight[qb * D:qe * D])
        local_attn.k_proj.weight.copy_(full_attn.k_proj.weight[kb * D:ke * D])
        local_attn.v_proj.weight.copy_(full_attn.v_proj.weight[kb * D:ke * D])
        local_attn.o_proj.weight.copy_(full_attn.o_proj.weight)
        if full_attn.q_proj.bias is not None:
            local_attn.q_proj.bias.copy_(full_attn.q_proj.bias[qb * D:qe * D])
            local_attn.k_proj.bias.copy_(full_attn.k_proj.bias[kb * D:ke * D])
            local_attn.v_proj.bias.copy_(full_attn.v_proj.bias[kb * D:ke * D])
            local_attn.o_proj.bias.copy_(full_attn.o_proj.bias)
    return local_attn
