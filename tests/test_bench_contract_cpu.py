"""The committed bench line (profiles/r03d_bench_final.json, produced by `python bench.py` on an MI355X)
carries every field the driver's contract asks for, and its numbers are self-consistent."""
import json
import os

from conftest import ROOT


def test_bench_line_contract():
    r = json.load(open(os.path.join(ROOT, "profiles", "r03d_bench_final.json")))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert r["metric"] == base["metric"] and r["unit"] == "frames/s"
    for key in ("value", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in r, key
    assert r["n_gpus"] == 1 and r["higher_is_better"] is True and r["scaling"] == "weak"
    assert r["vs_baseline"] is None            # BASELINE.md publishes no number for this metric
    assert r["dtype"].startswith("f64") and r["data"] == "synthetic" and "workload" in r["config"]
    assert "model" not in r["config"]
    # value = frames of one step / time per step
    assert abs(r["value"] - r["config"]["frames_per_step"] / (r["ms_per_step"] * 1e-3)) / r["value"] < 1e-6
    assert r["value"] >= 1e6                   # north_star target on 1x MI355X
    rf = r["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["launch_ms"] * 1e-3) / 1e9) < 1e-6 * rf["achieved"]
    assert rf["traffic"] is None or rf["traffic"] < rf["algorithmic_bytes_per_launch"]
    lm = rf["latency_model"]                   # the bound this configuration actually runs against
    assert abs(lm["cycles_per_step"] - rf["launch_ms"] * 1e-3 * lm["f_clk_hz"] / lm["steps_longest_stream"]) < 1e-6 * lm["cycles_per_step"]
    assert r["value"] <= lm["bound_frames_per_s"] * 1.02
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "frames/s" and cb["value"] > 0 and cb["sample"]
    legs = cb["legs"]
    assert legs["c_port_1core"]["cores"] == 1 and legs["numpy_1core"]["cores"] == 1 and legs["numpy_1core"]["path_equals_c_port"]
    assert cb["cores"] == legs["c_port_allcores"]["cores"] >= 2 and cb["value"] == legs["c_port_allcores"]["value"]
    assert legs["c_port_allcores"]["value"] > legs["c_port_1core"]["value"] > legs["numpy_1core"]["value"]
    assert r["parity"]["path_mismatches"] == 0 and r["parity"]["streams_checked"] == r["parity"]["streams_total"] == 64
    # provenance of the counter figure and of the clock (round-2 review): labelled, and tied to the kernel source
    ts = rf["traffic_source"]
    assert ts["measured_in_this_run"] is False and ts["kernel_source_sha16"] == ts["kernel_source_sha16_now"]
    assert "assumed" in lm["f_clk_source"]
    # the other BASELINE configs ride in the same line, each with its parity verdict and its roofline fraction
    sec = {e["key"]: e for e in r["secondary"]}
    assert set(sec) == {"dtw322", "dtw1289", "otw_b1", "otw_b64_f64", "chroma", "wtw20", "wtw100", "wtw10k"}
    for e in sec.values():
        assert e["ms"] > 0 and e["algorithmic_bytes"] > 0 and abs(e["frac"] - e["algorithmic_bytes"] / (e["ms"] * 1e-3) / 1e9 / 8000.0) < 1e-6 * e["frac"]
    assert sec["dtw322"]["parity"]["path_and_acc_equal_c_port"] and sec["dtw1289"]["parity"]["path_and_acc_equal_c_port"]
    assert sec["dtw322"]["cpu_port_ms"] > 0 and sec["dtw1289"]["cpu_port_ms"] > sec["dtw322"]["cpu_port_ms"]
    for k in ("otw_b1", "otw_b64_f64", "wtw20", "wtw100"):
        assert sec[k]["parity"]["path_mismatches"] == 0 and sec[k]["parity"]["streams_checked"] >= 1
    assert sec["chroma"]["parity"]["ok"] and sec["wtw10k"]["parity"]["equal_to_fixture"]
    assert r["secondary_wall_s"] < 60
