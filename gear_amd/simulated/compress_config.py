"""CompressionConfig with the reference's field surface (GenerationBench/.../Simulated/compress_config.py:1-86).

Scalars given to the constructor are broadcast to per-layer lists by copy_for_all_attention(), which is what the
attention hook indexes with layer_idx (compress_function.py:433-438, :441-555)."""
from __future__ import annotations

_PER_LAYER_FIELDS = (
    "compress_method", "quantize_bit", "group_num", "rank", "prefill_rank", "loop", "top_k", "device_num", "left",
    "stage", "rankv", "prefill_rankv", "start_saving", "locality_saving", "token_preserving", "iter", "heavy_size",
    "recent_size", "streaming", "streaming_gap", "group_size", "stream_grouping",
)


class CompressionConfig(dict):
    def __init__(self, compress_method=None, attention_number=12, quantize_bit=0, group_num=0, group_size=0, rank=0.0,
                 rankv=0.0, prefill_rank=0.0, prefill_rankv=0.0, loop=0, top_k=0.0, left=0.0, stage=1, device_num=0,
                 batch_num=1, start_saving=0, locality_saving=0, token_preserving=False, streaming=False,
                 streaming_gap=0, stream_grouping=False, iter=0, heavy_size=0, recent_size=0):
        super().__init__()
        loc = dict(locals())
        for name in ("compress_method", "attention_number", "quantize_bit", "group_num", "group_size", "rank", "rankv",
                     "prefill_rank", "prefill_rankv", "loop", "top_k", "left", "stage", "device_num", "batch_num",
                     "start_saving", "locality_saving", "token_preserving", "streaming", "streaming_gap",
                     "stream_grouping", "iter", "heavy_size", "recent_size"):
            setattr(self, name, loc[name])
        self.ranv = rankv  # the reference sets this alias too (compress_config.py:36)
        self._broadcast = False

    def create_attention_config(self, config):
        return [config for _ in range(self.attention_number)]

    def copy_for_all_attention(self):
        """Broadcast every per-layer field to a list of length attention_number (compress_config.py:63-85)."""
        for name in _PER_LAYER_FIELDS:
            setattr(self, name, self.create_attention_config(getattr(self, name)))
        self._broadcast = True

    # ---- the reference's analytic ratio bookkeeping (compress_config.py:87-281; called by the GenerationBench drivers, e.g.
    #      evaluation_gsm8k.py:407 `calculate_compress_ratio_list(4095, 4096)`).  It knows the LEGACY method names only; for the
    #      methods of today's dispatcher (KIVI_V2, KCVT, GEAR*, GEARL*) the reference's if / elif chain matches nothing: the
    #      function returns None and the list gets no entry -- kept as is (payload_ratio below is the build's own figure for them).
    @staticmethod
    def _ratio_terms(seqlen, model_dim, batch_num):
        n_tok = seqlen * batch_num                       # token rows of the [n_tok, model_dim] matrix being approximated
        return n_tok, n_tok * model_dim, model_dim + n_tok

    def compress_ratio(self, compress_method, seqlen, model_dim, rank=0, rankv=0, quantize_bit=0, top_k=0, left=0.0, stage=1,
                       batch_num=1):
        """fp16 elements per stored element for one layer (None for a method name the table does not know, as in the
        reference).  Same arguments, same arithmetic order of magnitude by magnitude as compress_config.py:87-186."""
        if compress_method is None:
            return 1.0
        n_tok, dense, fac = self._ratio_terms(seqlen, model_dim, batch_num)
        m = compress_method
        if m == "Picache":               # K and V each as rank-r factors, quantized (:104-148)
            if seqlen > rank and seqlen > rankv:
                return 2 * dense / (fac * (rank + rankv) * quantize_bit / 16)
            if seqlen <= rank:           # K kept dense
                return 2 * dense / (fac * rankv + dense) * 16 / quantize_bit
            if seqlen <= rankv:          # V kept dense
                return 2 * dense / (fac * rank + dense) * 16 / quantize_bit
            return None
        if m == "poweriteration":
            return dense / (fac * rank)
        if m == "stagept":
            return dense / (model_dim * rank + n_tok * (rank / stage))
        if m in ("uniformquantization", "groupquantization", "sortquantization"):
            return 16 / quantize_bit
        if m == "pruning":
            return 1 / top_k
        if m in ("densesparseuniformquantization", "densesparsesortquantization"):
            return 1 / (quantize_bit / 16 + left)
        if m == "pt+outlier":
            return dense * 16 / quantize_bit / (fac * rank)
        return None

    # which per-layer fields each legacy method hands to compress_ratio (compress_config.py:188-278)
    _RATIO_ARGS = {
        "Picache": ("rank", "rankv", "quantize_bit", "left"), "poweriteration": ("rank",), "stagept": ("rank", "stage"),
        "uniformquantization": ("quantize_bit",), "groupquantization": ("quantize_bit",), "sortquantization": ("quantize_bit",),
        "pruning": ("top_k",), "densesparseuniformquantization": ("quantize_bit", "left"),
        "densesparsesortquantization": ("quantize_bit", "left"), "pt+outlier": ("rank", "quantize_bit", "left"),
    }
    _RATIO_BATCHED = ("Picache", "poweriteration", "stagept", "pt+outlier")      # (these also receive batch_num)

    def calculate_compress_ratio_list(self, seqlen, model_dim):
        """Per-layer ratios into self.compress_ratio_list (needs copy_for_all_attention() first, like the reference)."""
        self.compress_ratio_list = []
        for i, method in enumerate(self.compress_method):
            if method is None:
                self.compress_ratio_list.append(self.compress_ratio(method, seqlen, model_dim))
            elif method in self._RATIO_ARGS:
                kw = {f: getattr(self, f)[i] for f in self._RATIO_ARGS[method]}
                if method in self._RATIO_BATCHED:
                    kw["batch_num"] = self.batch_num
                self.compress_ratio_list.append(self.compress_ratio(method, seqlen, model_dim, **kw))
            # (any other name: no entry -- the reference's chain has no else)

    def calculate_compress_ratio_total(self):
        return sum(self.compress_ratio_list) / len(self.compress_ratio_list)

    # ---- bookkeeping: bytes of the REAL payload the build stores, per fp16 KV byte (the reference's
    #      compress_ratio above covers only its legacy method names)
    def payload_ratio(self, layer_idx, B, H, T, D):
        g = lambda f: getattr(self, f)[layer_idx] if self._broadcast else getattr(self, f)  # noqa: E731
        method, bits, gs = g("compress_method"), g("quantize_bit"), g("group_size")
        if method is None or not bits:
            return 1.0
        n = B * H * T * D
        scale_bytes = 2 if method in ("KIVI_V2", "KCVT") else 4
        total = 0.0
        for rank_field in ("rank", "rankv"):
            by = n * bits / 8 + 2 * scale_bytes * n / max(gs, 1)
            if method.startswith("GEAR"):
                by += 2 * int(g(rank_field)) * (T + D) * B * H
            if method in ("GEAR", "GEAR-KCVT"):
                k = int(int(n * g("left")) / B / T / 2)
                rows = B * T if rank_field == "rankv" else B * H * D
                by += rows * 2 * k * 4
            total += by
        return (2 * 2 * n) / total

    def __str__(self):
        return ("compress_method:%s,\nquantize_bit:%s,\ngroup_size:%s,\nrank:%s,\nrankv:%s,\nloop:%s,\nleft:%s" %
                (self.compress_method, self.quantize_bit, self.group_size, self.rank, self.rankv, self.loop, self.left))
