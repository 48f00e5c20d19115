"""copy the per-config evidence of tools/profile_round.sh from gpurun_out/<round>/<cfg>/ into
profiles/<round>_<cfg>_* (tracked) and rebuild profiles/hbm_traffic.json, the table bench.py
reads for roofline.traffic (keyed by workload, validated against the schedule that runs).
usage: collect_profiles.py [--merge] r02 flan=Flan_1565:1.0:f64:1 flan_w1024=Flan_1565:1.0:f64:1 pwtk=pwtk:1.0:f64:1 ...
(several sets may share a key: one entry per schedule the workload was profiled on)"""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if a != "--merge"]
merge = "--merge" in sys.argv[1:]  # keep the table's other keys (only the sets named here were re-taken)
rnd = args[0]
table_path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
table = json.load(open(table_path)) if merge and os.path.exists(table_path) else {}
for key in {a.split("=")[1] for a in args[1:]}:
    table.pop(key, None)
for arg in args[1:]:
    cfg, key = arg.split("=")
    src = os.path.join(ROOT, "gpurun_out", rnd, cfg)
    for name in ("bench.json", "kernel_stats.csv", "pmc_fetch.csv", "pmc_write.csv", "hbm_traffic.json"):
        shutil.copy(os.path.join(src, name), os.path.join(ROOT, "profiles", f"{rnd}_{cfg}_{name}"))
    t = json.load(open(os.path.join(src, "hbm_traffic.json")))
    if "cfs_sym_tile_kernel" not in t:  # a general-CSR set: matched on the kernel form and the bytes it streams
        kern = [k for k in t if k.startswith("cfs_csr_")][0]
        table.setdefault(key, []).append({
            "hbm_bytes_per_launch": t[kern]["hbm_bytes_per_launch"], "kernel": kern,
            "bytes_streamed": t["bytes_streamed"],
            "source": f"profiles/{rnd}_{cfg}_pmc_fetch.csv + _pmc_write.csv (rocprofv3 --pmc FETCH_SIZE / "
                      "WRITE_SIZE, separate passes); reads = 2 x FETCH_SIZE KiB (gfx950 correction), "
                      "writes = WRITE_SIZE KiB"})
        continue
    table.setdefault(key, []).append({
        "hbm_bytes_per_launch": t["cfs_sym_tile_kernel"]["hbm_bytes_per_launch"],
        "kernel": "cfs_sym_tile_kernel",
        # the schedule the counters were taken on: bench.py quotes the number only for the same
        "bytes_streamed": t["bytes_streamed"], "lds_bytes": t["lds_bytes"],
        "block_threads": t["block_threads"],
        "source": f"profiles/{rnd}_{cfg}_pmc_fetch.csv + _pmc_write.csv (rocprofv3 --pmc FETCH_SIZE / "
                  "WRITE_SIZE, separate passes); reads = 2 x FETCH_SIZE KiB (gfx950 correction), "
                  "writes = WRITE_SIZE KiB",
    })
json.dump(table, open(table_path, "w"), indent=1)
print(json.dumps(table, indent=1))
