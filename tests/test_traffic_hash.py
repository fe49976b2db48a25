"""CPU: `roofline.traffic` of the bench line comes from committed rocprofv3 counters, so it must go null the day the kernels change
and the counters are not re-collected (VERDICT r04 'weak' 8): the profile carries the sha256 of csrc/ and bench.py compares."""
from __future__ import annotations

import importlib.util
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def test_bench_and_collector_hash_the_same_sources_and_stale_counters_give_null(tmp_path, monkeypatch):
    bench = _load(ROOT / "bench.py", "bench_for_hash_test")
    tool = _load(ROOT / "tools" / "pmc_traffic.py", "pmc_traffic_for_hash_test")
    now = bench.kernel_source_hash()
    assert now == tool.kernel_source_hash() and len(now) == 64
    blk = {"hbm_bytes_per_launch": 1.0e9, "kernel_source_sha256": now}
    prof = tmp_path / "traffic.json"
    prof.write_text(json.dumps({"f16x3_unet": blk, "f16x3_resnet18": dict(blk, hbm_bytes_per_launch=3.0e9)}))
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    monkeypatch.setattr(bench, "PMC_FILE", "traffic.json")
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: now)
    val, why = bench.pmc_traffic("f16x3", 64, 16384, 3, 1)
    assert why is None and val == (3 * 1.0e9 + 3.0e9) / 4
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: "0" * 64)          # a kernel source changed since the counters were taken
    val, why = bench.pmc_traffic("f16x3", 64, 16384, 3, 1)
    assert val is None and "stale" in why
    val, why = bench.pmc_traffic("f16x3", 32, 16384, 3, 1)
    assert val is None and "chunks" in why
    val, why = bench.pmc_traffic("f16", 64, 16384, 3, 1)
    assert val is None and "no f16 counters" in why
