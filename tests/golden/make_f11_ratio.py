"""Fixture F11: the reference's analytic compress-ratio bookkeeping (GenerationBench/.../Simulated/compress_config.py:87-281), made
by EXECUTING the reference's CompressionConfig in the build container.  Data only: constructor arguments, (seqlen, model_dim), the
per-layer list and the total it produced ("zde" where calculate_compress_ratio_total divides by an empty list's length).
usage: PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_f11_ratio.py"""
import importlib.util
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location(
    "ref_cc", "/root/reference/GenerationBench/GenerationTest/GEARLM/Simulated/compress_config.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

METHODS = [None, "Picache", "poweriteration", "stagept", "uniformquantization", "groupquantization", "sortquantization", "pruning",
           "densesparseuniformquantization", "densesparsesortquantization", "pt+outlier", "GEAR", "GEARL", "KIVI_V2", "KCVT"]
cases = []
for m in METHODS:
    for rank, rankv, seq, dim in [(4, 8, 4095, 4096), (5000, 8, 4095, 4096), (4, 5000, 4095, 4096), (5000, 6000, 100, 4096),
                                  (8, 8, 2048, 5120)]:
        kw = dict(compress_method=m, attention_number=3, quantize_bit=4, rank=rank, rankv=rankv, top_k=0.25, left=0.02, stage=2,
                  batch_num=2, loop=3, group_size=64)
        c = ref.CompressionConfig(**kw)
        c.copy_for_all_attention()
        c.calculate_compress_ratio_list(seq, dim)
        try:
            tot = c.calculate_compress_ratio_total()
        except ZeroDivisionError:
            tot = "zde"
        cases.append({"kwargs": kw, "seqlen": seq, "model_dim": dim, "list": c.compress_ratio_list, "total": tot})
json.dump(cases, open(os.path.join(HERE, "f11_compress_ratio.json"), "w"), indent=0)
print(len(cases), "cases")
