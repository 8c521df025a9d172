"""a14: the timing harness (harness.py, counterpart of cuda_supported_gear/test.py:25-102) actually runs -- all three variants the
reference compares (GEAR, KIVI, uncompressed), its knobs, one synchronize before the clock stops, peak memory and tokens/s -- on a
2-layer Llama-2-7B-shaped model with random weights (no network: the reference's weights and wikitext prompt cannot be loaded)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(*extra):
    cmd = [sys.executable, os.path.join(ROOT, "harness.py"), "--layers", "2", "--prompt_len", "200", "--max_length", "330",
           "--batch_size", "2", "--json", *extra]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    res = json.loads(p.stdout.strip().splitlines()[-1])
    return res, p.stdout


def test_harness_runs_all_variants():
    res, text = _run("--all")
    assert [r["variant"] for r in res] == ["gearl", "KIVI", "None"]
    assert text.count("Peak memory usage on GPU") == 3              # the line the reference prints (test.py:101)
    for r in res:
        assert r["batch_size"] == 2 and r["prompt_len"] == 200 and r["max_length"] == 330 and r["layers"] == 2
        assert r["time_s"] > 0 and r["new_tokens_per_s"] > 0 and r["peak_mem_MiB"] > 0
        assert abs(r["new_tokens_per_s"] * r["time_s"] - 2 * 130) < 1e-6 * 260 + 1e-3


def test_harness_fast_path_with_outliers():
    res, _ = _run("--model", "gearl", "--fast", "--left", "0.02", "--compress_method", "gearslKIVI", "--rank", "4")
    assert len(res) == 1 and res[0]["fast"] and res[0]["max_length"] == 330 and res[0]["new_tokens_per_s"] > 0


def test_harness_control_through_the_fast_decoder():
    """--model None --fast: the uncompressed control through the SAME decoder as the GEAR fast path (cache.Fp16KVCache,
    gear_attn_decode_f16), so that the harness's variants differ in the cache only."""
    res, _ = _run("--model", "None", "--fast")
    assert len(res) == 1 and res[0]["variant"] == "None" and res[0]["fast"] and res[0]["max_length"] == 330 and res[0]["new_tokens_per_s"] > 0
