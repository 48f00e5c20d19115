#!/bin/bash
# Round profile: run on the GPU box (gpurun).  Produces, under gpurun_out/<tag>/:
#   bench.json                 the contract line of `python bench.py <args>`
#   kernel_stats.csv           rocprofv3 --kernel-trace --stats of the same command
#   pmc_fetch.csv, pmc_write.csv   separate --pmc passes (never combined with tracing)
#   hbm_traffic.json           per-launch HBM bytes of the kernels:
#                              (2*FETCH_SIZE + WRITE_SIZE) KiB, the gfx950 correction of
#                              /opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE counts 128-B
#                              requests at 64 B for wide coalesced reads; WRITE_SIZE is exact),
#                              with the schedule it was measured on (bytes_streamed, lds_bytes,
#                              block_threads): bench.py quotes roofline.traffic only for the same
# usage: tools/profile_round.sh <tag> [bench args...]
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
python3 bench.py "$@" > $out/bench.json 2> $out/bench.err || { echo "bench failed"; tail -5 $out/bench.err; exit 1; }
# the profiler passes run the window shape tune() kept in the bench run above (the two
# shapes are within a few per cent of each other; under the profiler the clock may say otherwise)
export CFS_HIP_SHAPE=$(python3 -c "import json,sys; print(json.load(open('$out/bench.json'))['config'].get('block_threads', 0))")
# ... and the general CSR form the bench run measured to be faster
form=$(python3 -c "import json,sys; print(json.load(open('$out/bench.json'))['config'].get('kernel_form', '').split(' ')[0])")
if [ -n "$form" ]; then export CFS_HIP_CSR_KERNEL=$form; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline "$@" > $out/bench_traced.json 2> $out/trace.err
cp $out/trace/*/*_kernel_stats.csv $out/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_f -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > /dev/null 2> $out/pmc_f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_w -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > /dev/null 2> $out/pmc_w.err
cp $out/pmc_f/*/*_counter_collection.csv $out/pmc_fetch.csv
cp $out/pmc_w/*/*_counter_collection.csv $out/pmc_write.csv
rm -rf $out/trace $out/pmc_f $out/pmc_w
python3 - $out <<'PY'
import csv, json, sys
out = sys.argv[1]
b = json.load(open(out + '/bench.json'))
def mean(path, counter, kern):
    v = [float(r['Counter_Value']) for r in csv.DictReader(open(path))
         if r['Counter_Name'] == counter and kern in r['Kernel_Name']]
    return (sum(v) / len(v), len(v)) if v else (0.0, 0)
main = b["roofline"].get("kernel", "cfs_sym_tile_kernel")
res = {"bench_config": b["config"]["workload"][:120], "lds_bytes": b["config"].get("lds_bytes"),
       "block_threads": b["config"].get("block_threads"),
       "bytes_streamed": b["roofline"].get("bytes_streamed_by_format")}
for kern in ((main, "cfs_fold_kernel") if main == "cfs_sym_tile_kernel" else (main,)):
    f, nf = mean(out + '/pmc_fetch.csv', 'FETCH_SIZE', kern)
    w, nw = mean(out + '/pmc_write.csv', 'WRITE_SIZE', kern)
    res[kern] = {"FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB": w, "launches": [nf, nw],
                 "hbm_bytes_per_launch": int((2 * f + w) * 1024),
                 "correction": "reads = 2 x FETCH_SIZE (gfx950: 128-B requests tallied at 64 B), writes exact"}
json.dump(res, open(out + '/hbm_traffic.json', 'w'), indent=1)
# keep the per-dispatch csv small: the tile / fold kernels only
for name in ('pmc_fetch.csv', 'pmc_write.csv'):
    rows = list(csv.DictReader(open(out + '/' + name)))
    keep = [r for r in rows if 'cfs_' in r['Kernel_Name']]
    with open(out + '/' + name, 'w', newline='') as fo:
        w = csv.DictWriter(fo, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep[:400])
ks = [r for r in csv.DictReader(open(out + '/kernel_stats.csv'))]
t = [r for r in ks if main in r['Name']]
prod = max(t, key=lambda r: int(r['Calls']))  # the instantiation of the timed steps (not tune()'s trials)
avg = float(prod['TotalDurationNs']) / int(prod['Calls'])
print(json.dumps({"tag": out, "ms_per_step": b["ms_per_step"], "value": b["value"],
                  "kernel_ms_events": b["roofline"]["kernel_ms"], "kernel_ms_rocprof": round(avg * 1e-6, 5),
                  "frac": b["roofline"]["frac"], "hbm_bytes_tile": res[main]["hbm_bytes_per_launch"],
                  "alg_bytes": b["roofline"]["algorithmic_bytes_per_launch"], "preproc_s": b["config"]["preproc_s"],
                  "cpu": b.get("cpu_baseline", {}).get("value")}))
PY
