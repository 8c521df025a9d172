"""Fast single-token decode over GearKVCache: 6 launches per layer instead of the ~60 eager torch ops the attention
hook's forward issues per layer per token (cuda_supported_gear/modeling_llamagear.py:177-484 + decoder layer :502-560).

Takes the weights of a LlamaForCausalLM_GEARKIVI (any loader that fills that module works), fuses q/k/v and gate/up
projections into single GEMV weights, keeps the KV cache in GearKVCache (pre-allocated, kernel-native) and runs per layer
    [RMSNorm + qkv GEMV + RoPE + window append]  ->  fused attention over the compressed cache (2 launches)
    ->  [o_proj GEMV + residual add]  ->  [RMSNorm + gate/up GEMV + SwiGLU]  ->  [down GEMV + residual add]
with each bracketed group being one launch (gear_gemv_qkv_rope / gear_gemv_f16_add / gear_gemv_f16_norm) for batch <= 4;
larger batches use the library GEMM through torch.  Everything touching the cache is HIP.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import _lib as L
from .cache import Fp16KVCache, GearKVCache, GearKVCachePool
from .modeling_llamagear import apply_rotary_pos_emb


class FastGearDecoder:
    def __init__(self, model, max_tokens: int, batch: int = 1, seed: int = 0, tp_rank: int = 0, tp_world: int = 1,
                 tp_group=None, tp_exchange: str = "collective", v_selection: str = "exact", cache_kind: str = "gear"):
        """tp_world > 1: the cache and the attention are sharded head-wise: this rank owns
        Hq / tp_world query heads with their KV heads -- local q/k/v projection rows, local compressed cache, local attention --
        and all-gathers the per-rank attention output (parallel.HeadGather, pre-allocated) in front of the replicated
        o_proj / MLP, which every rank computes in full.  The model passed in holds the full (replicated) weights.
        tp_exchange: "collective" (default since round 5: the RCCL all-gather north_star names) = parallel.HeadGather
        (all_gather_into_tensor into a pre-allocated buffer; captured in the token-step graph when the RCCL capture probe passes),
        "peer" = parallel.PeerHeadGather (stores into the peers' hipIpc-mapped memory from one launch per layer, always capturable;
        falls back to the collective, on every rank alike, when the peer mappings cannot be set up).  Neither has run on more than
        one GPU yet (no multi-GPU node was ever available to this build): the collective is the conservative default.
        v_selection: "exact" = the V outliers of a token row are selected over ALL ranks' heads (the reference's row spans the heads:
        compress_function.py:304-311; csrc/vsel.hip through parallel.exact_v_thresholds: two launches + one small all-gather per
        compress call), "per_shard" = k / world inside the shard's own heads (rounds 1-3).
        cache_kind: "gear" (the compressed streaming cache) or "fp16" = cache.Fp16KVCache, the uncompressed baseline the reference's
        harness times beside the compressed models (cuda_supported_gear/test.py:41-62: model "None"): same weights, same fused GEMVs,
        same attention split / merge kernels over fp16 rows; eager steps on one GPU only."""
        self.model = model
        cfg = model.config
        self.cfg = cfg
        self.tp_rank, self.tp_world = tp_rank, tp_world
        self.Hq_full, self.Hkv_full = cfg.num_attention_heads, cfg.num_key_value_heads
        if self.Hq_full % tp_world or self.Hkv_full % tp_world:
            raise ValueError(f"{self.Hq_full} query / {self.Hkv_full} KV heads do not divide across {tp_world} ranks")
        self.Hq, self.Hkv = self.Hq_full // tp_world, self.Hkv_full // tp_world          # local heads
        self.D = cfg.hidden_size // cfg.num_attention_heads
        self.eps = cfg.rms_norm_eps
        self.theta = float(cfg.rope_theta)
        dev = model.lm_head.weight.device
        self.dev = dev
        self.layers = []
        cc0 = model.model.layers[0].self_attn.compress_config
        # pooled cache storage: block boundaries compress every layer's window in one launch sequence
        if v_selection not in ("exact", "per_shard"):
            raise ValueError(f"v_selection {v_selection!r}: 'exact' or 'per_shard'")
        if cache_kind not in ("gear", "fp16"):
            raise ValueError(f"cache_kind {cache_kind!r}: 'gear' or 'fp16'")
        if cache_kind == "fp16" and tp_world > 1:
            raise NotImplementedError("the fp16-cache baseline runs on one GPU")
        self.cache_kind = cache_kind
        tp = dict(rank=tp_rank, world=tp_world, group=tp_group, exact=v_selection == "exact") if tp_world > 1 else None
        self.pool = None if cache_kind == "fp16" else GearKVCachePool(len(model.model.layers), batch, self.Hkv, max_tokens, cc0, dev,
                                                                      self.D, seed=seed, heads_total=self.Hkv_full, tp=tp)
        for i, layer in enumerate(model.model.layers):
            at, mlp = layer.self_attn, layer.mlp
            assert at.q_proj.bias is None, "attention_bias is not supported by the fused qkv GEMV"
            # RMSNorm weights are folded into the columns of the projection that follows them (the GEMV applies the
            # row scale rsqrt(mean(h^2) + eps) to its finished dot products), gate/up rows are interleaved so a GEMV
            # block finishes whole SwiGLU pairs
            n1, n2 = layer.input_layernorm.weight, layer.post_attention_layernorm.weight
            wqkv_full = torch.cat([at.q_proj.weight, at.k_proj.weight, at.v_proj.weight], 0) * n1[None, :]
            if tp_world > 1:
                D, r = self.D, tp_rank
                wqkv = torch.cat([at.q_proj.weight[r * self.Hq * D:(r + 1) * self.Hq * D],
                                  at.k_proj.weight[r * self.Hkv * D:(r + 1) * self.Hkv * D],
                                  at.v_proj.weight[r * self.Hkv * D:(r + 1) * self.Hkv * D]], 0) * n1[None, :]
            else:
                wqkv = wqkv_full
            wgu = torch.stack([mlp.gate_proj.weight, mlp.up_proj.weight], 1).reshape(-1, n2.shape[0]) * n2[None, :]
            self.layers.append(dict(
                wqkv=wqkv.contiguous(), wqkv_full=wqkv_full if tp_world > 1 else None, wo=at.o_proj.weight, wgu=wgu.contiguous(), wd=mlp.down_proj.weight,
                cache=(Fp16KVCache(batch, self.Hkv, max_tokens, dev, self.D) if cache_kind == "fp16" else
                       GearKVCache(batch, self.Hkv, max_tokens, at.compress_config, dev, self.D, seed=seed + i,
                                   pool=self.pool, layer=i, heads_total=self.Hkv_full, tp=tp)),
                rotary=at.rotary_emb))
        self.w_head = (model.lm_head.weight * model.model.norm.weight[None, :]).contiguous()
        self.pos = 0
        self.use_gemv = True
        self.batch = batch
        # hipGraph mode: device-side {pos, slot, T, W} shared by every layer's cache, static token / logits buffers
        self.state = torch.zeros(4, dtype=torch.int32, device=dev)
        for lw in self.layers:
            lw["cache"].state = self.state
        self.graph = None
        self._state_dirty = True      # host counters moved without the device-side state (eager step / prefill)
        self.tok = torch.zeros((batch, 1), dtype=torch.long, device=dev)
        self.logits_static = None
        self.gather = None
        self.exchange_error = None
        if tp_world > 1:
            from .parallel import HeadGather, PeerHeadGather
            if tp_exchange not in ("peer", "collective"):
                raise ValueError(f"tp_exchange {tp_exchange!r}: 'peer' or 'collective'")
            if tp_exchange == "peer":
                g = PeerHeadGather(tp_world, tp_rank, batch, self.Hq * self.D, torch.float16, dev, tp_group)
                if g.ok:
                    self.gather = g
                else:
                    self.exchange_error = g.error
            if self.gather is None:
                self.gather = HeadGather(tp_world, batch, self.Hq * self.D, torch.float16, dev, tp_group)

    def check_block_kernel(self):
        """Raise if a V tile of gear_compress_block ever gave up waiting for its row duties (csrc/block_fused.hip: the poll is
        bounded so that a shared / preempted GPU yields a status word instead of a hang; a tile that gave up has quantized with
        stale masks).  One device word: called every 16th block boundary and at the end of generate()."""
        from . import cache as gc
        st = gc.block_kernel_status(self.dev)
        if st:
            raise RuntimeError(f"gear_compress_block: hand-off timed out (status {st}); the cache contents since the last "
                               "check are not trustworthy -- set gear_amd.cache.USE_BLOCK_KERNEL = False to use the kernel chain")

    def _after_boundary(self):
        self._nbound = getattr(self, "_nbound", 0) + 1
        self.check_exchange()
        if self._nbound % 16 == 0:
            self.check_block_kernel()

    def check_exchange(self):
        """Sharded decoder: raise if the peer exchange ever timed out (one device word; called at block boundaries, at the end
        of generate() and by bench.py -- a latched timeout makes every later token silently wrong otherwise)."""
        if self.gather is not None and hasattr(self.gather, "check"):
            self.gather.check()

    def _attn_out(self, a):
        """[B, Hq_local, 1, D] of this rank -> [B, Hq_full * D] (all-gather over the head shards when sharded)."""
        a = a.view(a.shape[0], self.Hq * self.D)
        return a if self.gather is None else self.gather(a)

    # ------------------------------------------------------------------------------------------------ helpers
    def _fused(self, B, K):
        return self.use_gemv and B <= 4 and K % 8 == 0

    def _unit_rms(self, h):
        """RMSNorm without its weight (folded into the next projection), fp32 statistics like LlamaRMSNorm."""
        hf = h.float()
        return (hf * torch.rsqrt(hf.pow(2).mean(-1, keepdim=True) + self.eps)).to(h.dtype)

    def _linear_add(self, x, w, res):
        """res + x @ w^T: the residual-stream update after o_proj / down_proj (gear_gemv_f16_add for B <= 4)."""
        B, K = x.shape
        if not self._fused(B, K):
            return res + F.linear(x, w)
        y = torch.empty_like(res)
        rc = L.load().gear_gemv_f16_add(L.ptr(x), L.ptr(w), B, K, w.shape[0], L.ptr(res), L.ptr(y), L.stream_ptr())
        L.check(rc, "gear_gemv_f16_add")
        return y

    def _norm_linear(self, res, w, swiglu=False):
        """RMSNorm(res) -> w (norm weight folded into w), optionally through SwiGLU (w rows interleaved).
        One launch (gear_gemv_f16_norm) for B <= 4; library GEMM otherwise."""
        B, K = res.shape
        N = w.shape[0]
        if self._fused(B, K):
            y = torch.empty((B, N // 2 if swiglu else N), dtype=res.dtype, device=res.device)
            rc = L.load().gear_gemv_f16_norm(L.ptr(res), None, None, self.eps, L.ptr(w), B, K, N, 1 if swiglu else 0,
                                             None, L.ptr(y), L.stream_ptr())
            L.check(rc, "gear_gemv_f16_norm")
            return y
        y = F.linear(self._unit_rms(res), w)
        return F.silu(y[:, 0::2]) * y[:, 1::2] if swiglu else y

    def _norm_qkv_rope(self, res, lw, dyn):
        """RMSNorm(res) -> fused q/k/v projection -> RoPE; k, v land in the cache window.  Returns q [B,Hq,1,128].
        dyn: position / slot come from the device state (graph replay)."""
        cache = lw["cache"]
        B, K = res.shape
        if self._fused(B, K):
            if not dyn and cache.n_win >= cache.R:
                raise L.GearError(f"FastGearDecoder: the cache window is full ({cache.n_win} of {cache.R} slots): capacity reached")
            q = torch.empty((B, self.Hq, 1, self.D), dtype=res.dtype, device=res.device)
            rc = L.load().gear_gemv_qkv_rope(
                L.ptr(res), None, None, self.eps, L.ptr(lw["wqkv"]), B, K, self.Hq, self.Hkv, self.D,
                0 if dyn else self.pos, 0 if dyn else cache.n_win, cache.R, self.theta,
                L.ptr(self.state) if dyn else None, None, L.ptr(q), L.ptr(cache.kwin), L.ptr(cache.vwin), L.stream_ptr())
            L.check(rc, "gear_gemv_qkv_rope")
            if not dyn:
                cache.n_win += 1
            return q
        qkv = F.linear(self._unit_rms(res), lw["wqkv"])
        return cache.append_rope_dyn(qkv, self.Hq, self.theta) if dyn else cache.append_rope(qkv, self.Hq, self.pos, self.theta)

    # ------------------------------------------------------------------------------------------------ prefill
    @torch.no_grad()
    def prefill(self, input_ids: torch.Tensor) -> torch.Tensor:
        """Dense causal attention over the prompt (torch SDPA), compress each layer's K/V into its cache.
        Returns the logits of the last position [B, vocab]."""
        B, T = input_ids.shape
        m = self.model.model
        h = m.embed_tokens(input_ids)
        pos = torch.arange(T, device=self.dev).unsqueeze(0)
        n_rep = self.Hq_full // self.Hkv_full
        for lw, layer in zip(self.layers, m.layers):
            # (the prompt is processed replicated, with every head; each rank keeps its own heads' K / V)
            wq = lw["wqkv"] if lw["wqkv_full"] is None else lw["wqkv_full"]
            qkv = F.linear(self._unit_rms(h), wq)              # input_layernorm's weight lives in wqkv's columns
            Hq, Hkv = self.Hq_full, self.Hkv_full
            q, k, v = qkv.split([Hq * self.D, Hkv * self.D, Hkv * self.D], dim=-1)
            q = q.view(B, T, Hq, self.D).transpose(1, 2)
            k = k.view(B, T, Hkv, self.D).transpose(1, 2)
            v = v.view(B, T, Hkv, self.D).transpose(1, 2)
            cos, sin = lw["rotary"](v, pos)
            q, k = apply_rotary_pos_emb(q, k, cos, sin)
            k0 = self.tp_rank * self.Hkv
            lw["cache"].prefill(k[:, k0:k0 + self.Hkv], v[:, k0:k0 + self.Hkv])
            kk = k if n_rep == 1 else k.repeat_interleave(n_rep, 1)
            vv = v if n_rep == 1 else v.repeat_interleave(n_rep, 1)
            a = F.scaled_dot_product_attention(q, kk, vv, is_causal=True)
            h = h + F.linear(a.transpose(1, 2).reshape(B, T, Hq * self.D), lw["wo"])
            h = h + layer.mlp(layer.post_attention_layernorm(h))
        self.pos = T
        self._state_dirty = True
        return self.model.lm_head(m.norm(h[:, -1]))

    # ------------------------------------------------------------------------------------------------ decode
    @torch.no_grad()
    def step(self, token_ids: torch.Tensor) -> torch.Tensor:
        """token_ids [B, 1] (or [B]) -> logits [B, vocab]; advances every layer's cache by one token."""
        m = self.model.model
        res = m.embed_tokens(token_ids.view(-1))            # [B, hidden]
        for lw in self.layers:
            cache = lw["cache"]
            q = self._norm_qkv_rope(res, lw, dyn=False)
            a = cache.attend(q)
            res = self._linear_add(self._attn_out(a), lw["wo"], res)
            act = self._norm_linear(res, lw["wgu"], swiglu=True)
            res = self._linear_add(act, lw["wd"], res)
        self.pos += 1
        self._state_dirty = True          # (the captured graph reads pos / slot / T / W from self.state)
        if self.pool is not None and self.layers[0]["cache"].n_win == self.layers[0]["cache"].R:
            self.pool.compress_all()
            self._after_boundary()
        return self._norm_linear(res, self.w_head)

    # ------------------------------------------------------------------------------------------------ hipGraph decode
    def _sync_state(self):
        c = self.layers[0]["cache"]
        self.state.copy_(torch.tensor([self.pos, c.n_win, c.n_comp, c.n_win + 1], dtype=torch.int32))

    def _step_body_dyn(self):
        """The token step with every per-token scalar read from self.state on the device: identical launches for every
        token, so the whole thing (embedding -> 32 layers -> lm_head -> argmax -> state advance) is one graph."""
        m = self.model.model
        res = m.embed_tokens(self.tok.view(-1))
        for lw in self.layers:
            q = self._norm_qkv_rope(res, lw, dyn=True)
            a = lw["cache"].attend_dyn(q)
            res = self._linear_add(self._attn_out(a), lw["wo"], res)
            act = self._norm_linear(res, lw["wgu"], swiglu=True)
            res = self._linear_add(act, lw["wd"], res)
        logits = self._norm_linear(res, self.w_head)
        self.tok.copy_(logits.argmax(-1, keepdim=True))
        L.check(L.load().gear_decode_state_advance(L.ptr(self.state), L.stream_ptr()), "gear_decode_state_advance")
        return logits

    @torch.no_grad()
    def step_graph(self, token_ids: torch.Tensor = None) -> torch.Tensor:
        """One greedy decode token by replaying the captured graph.  token_ids (optional) overrides the token the graph
        produced itself; returns the NEXT token [B,1] (the graph's own argmax).  Block compression (every `residual`
        tokens) runs eagerly between replays."""
        if self.pool is None:
            raise NotImplementedError("step_graph(): the fp16-cache baseline has eager steps only")
        if self.gather is not None and not self.gather.capturable and not getattr(self, "_peer_tried", False):
            # the collective cannot be captured here (its probe failed, on every rank alike): a graph step needs the peer exchange
            # (csrc/xchg.hip, always capturable) -- set it up now, collectively, and keep the collective if that fails too
            self._peer_tried = True
            from .parallel import PeerHeadGather
            g = PeerHeadGather(self.tp_world, self.tp_rank, self.batch, self.Hq * self.D, torch.float16, self.dev,
                               getattr(self.gather, "group", None))
            if g.ok:
                self.gather = g
            else:
                self.exchange_error = g.error
        if self.gather is not None and not self.gather.capturable:
            raise NotImplementedError("step_graph() with tp_world > 1 needs a capturable exchange: tp_exchange='peer', or the "
                                      "collective on the RCCL backend when its capture probe passed (the one-GPU gloo staging "
                                      "path copies through the host) -- use step()")
        if token_ids is not None:
            self.tok.copy_(token_ids.view(self.batch, 1))
        if self.graph is None:
            self._sync_state()
            self._state_dirty = False
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.logits_static = self._step_body_dyn()
            self.graph = g
        elif self._state_dirty:          # an eager step() / prefill() ran since the last replay: bring the device state up to date
            self._sync_state()
        self._state_dirty = False
        self.graph.replay()
        self.pos += 1
        full = False
        for lw in self.layers:
            c = lw["cache"]
            c.n_win += 1
            full = c.n_win == c.R
        if full:
            self.pool.compress_all()
            self._sync_state()
            self._after_boundary()
        return self.tok

    def close(self):
        """Release the peer mappings of a sharded decoder (collective: every rank calls it)."""
        if self.gather is not None and hasattr(self.gather, "close"):
            self.gather.close()

    @torch.no_grad()
    def generate(self, input_ids: torch.Tensor, max_length: int, graph: bool = False) -> torch.Tensor:
        logits = self.prefill(input_ids)
        out = [input_ids]
        nxt = logits.argmax(-1, keepdim=True)
        out.append(nxt)
        if graph and sum(t.shape[1] for t in out) < max_length:
            nxt = self.step(nxt).argmax(-1, keepdim=True)      # one eager step warms up every library handle
            out.append(nxt)
            self.tok.copy_(nxt)
        while sum(t.shape[1] for t in out) < max_length:
            if graph:
                nxt = self.step_graph().clone()
            else:
                nxt = self.step(nxt).argmax(-1, keepdim=True)
            out.append(nxt)
        self.check_exchange()
        self.check_block_kernel()
        return torch.cat(out, dim=1)
