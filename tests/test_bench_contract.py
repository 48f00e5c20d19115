"""The committed bench line (profiles/r01_bench.json, written by `python bench.py` on an
MI355X) carries every field of the driver's contract, and its numbers are consistent
with one another and with the committed rocprofv3 / PMC summaries."""
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_matches_the_contract():
    d = json.loads(open(os.path.join(ROOT, "profiles", "r01_bench.json")).read())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
              "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "GFLOP/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["n_gpus"] == 1 and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # achieved = algorithmic bytes per launch / measured kernel time
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1.0
    # the value is 2 * nnz_full / t (bench/bench_spmv_mmf.cpp:168)
    nnz_full = int(d["config"]["workload"].split("nnz_full=")[1].split(",")[0])
    assert abs(d["value"] - 2.0 * nnz_full / (d["ms_per_step"] * 1e-3) / 1e9) < 0.5
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1


def test_rocprof_summary_agrees_with_the_bench_line():
    d = json.loads(open(os.path.join(ROOT, "profiles", "r01_bench.json")).read())
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "r01_kernel_stats.csv"))))
    tile = [r for r in rows if "cfs_sym_tile_kernel" in r["Name"]]
    assert tile, "tile kernel missing from the rocprofv3 summary"
    calls = sum(int(r["Calls"]) for r in tile)
    avg_ns = sum(float(r["TotalDurationNs"]) for r in tile) / calls
    # HIP events in bench.py vs rocprofv3's kernel trace of the same command: within 5 %
    assert abs(avg_ns * 1e-6 - d["roofline"]["kernel_ms"]) <= 0.05 * d["roofline"]["kernel_ms"]
    # PMC traffic (separate passes) is what the line reports, and it is below the
    # algorithmic figure: no re-reads
    t = json.load(open(os.path.join(ROOT, "profiles", "r01_hbm_traffic_pmc.json")))
    assert d["roofline"]["traffic"] is not None
    assert abs(t["cfs_sym_tile_kernel"]["hbm_bytes_per_launch"] - d["roofline"]["traffic"]) \
        <= 0.01 * d["roofline"]["traffic"]
    assert d["roofline"]["traffic"] < d["roofline"]["algorithmic_bytes_per_launch"]
