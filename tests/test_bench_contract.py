"""The bench line the driver reads: the record committed from the last GPU run of `python bench.py` (profiles/) must carry the
contract's fields with the right types, and the numbers must hang together (value = samples / time, frac = achieved / peak)."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_default.json")))
    assert files, "no committed default bench line under profiles/"
    line = [l for l in open(files[-1]) if l.startswith("{")][0]
    return json.loads(line)


def test_default_bench_line_contract():
    d = _latest()
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                 ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict),
                 ("cpu_baseline", dict), ("parity_gate", dict)):
        assert k in d and isinstance(d[k], t), k
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and d["n_gpus"] == 1 and d["data"] == "synthetic" and d["dtype"] == "f32"
    cfg = d["config"]
    assert cfg["workload"] == "target" and "model" not in cfg and "engine_boundary" in cfg
    # value = samples of the timed steps / their time
    samples = cfg["parts_per_block"] * cfg["nsamp_step"]
    assert abs(samples / (d["ms_per_step"] * 1e-3) / 1e6 / d["value"] - 1) < 1e-3
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] > r["algorithmic_bytes_per_part"] * cfg["max_parts"]       # PMC bytes per launch group
    assert abs(r["achieved"] - r["algorithmic_bytes_per_part"] * cfg["parts_per_block"] / (r["group_ms_per_block"] * 1e-3) / 1e9) < 1.0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)
    g = d["parity_gate"]
    assert g["status"] == "ok" and g["oracle_check"]["status"] == "ok" and g["oracle_check"]["profile_max_err_rel"] <= 1e-5
    eb = cfg["engine_boundary"]
    assert eb["raw_deferred"] >= 0.9 * d["value"] and eb["raw_deferred_fused_blocks"] > 0 and eb["float_eager_fused_blocks"] == 0
    assert d["subband_shard"]["workload"] == "cfg4" and d["subband_shard"]["value"] > 0
    # round 4: which transport carried the dumps is a top-level field; the PCIe-inclusive companion and the PMC traffic ratio of
    # EVERY workload are part of the default line
    assert d["exchange"] == "none" and cfg["exchange"] == "none" and cfg["reduce_ms_per_dump"] is None
    assert d["subband_shard"]["exchange"] == "none" and d["subband_shard"]["reduce_ms_per_dump"] is None
    pc = cfg["pcie_inclusive"]
    assert pc["unit"] == "Msamples/s" and 0 < pc["value"] < d["value"] and pc["steps"] >= 5
    assert r["traffic_ratio"] is not None and abs(r["traffic_ratio"] - r["traffic"] / (r["algorithmic_bytes_per_part"] * cfg["max_parts"])) < 1e-2
    assert "profiles/" in r["traffic_source"]
    assert d["subband_shard"]["roofline_traffic_ratio"] > 1.0 and cfg["transform_passes"] == 3
    for w in d["other_workloads"]:
        assert w["roofline_traffic_ratio"] is not None and w["roofline_traffic_ratio"] >= 1.0, w["workload"]
    assert {w["workload"] for w in d["other_workloads"]} >= {"cfg1", "cfg1opt", "cfg2", "cfg3", "cfg5"}
