#!/bin/bash
# Round profile: run on the GPU box (gpurun).  Produces, under gpurun_out/<tag>/:
#   bench.json                 the contract line of `python bench.py`
#   kernel_stats.csv           rocprofv3 --kernel-trace --stats of the same command
#   pmc_fetch.csv, pmc_write.csv   separate --pmc passes (never combined with tracing)
#   hbm_traffic.json           per-launch HBM bytes of cfs_sym_tile_kernel:
#                              (2*FETCH_SIZE + WRITE_SIZE) KiB, the gfx950 correction of
#                              /opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE counts 128-B
#                              requests at 64 B for wide coalesced reads; WRITE_SIZE is exact)
# usage: tools/profile_round.sh <tag> [bench args...]
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
python3 bench.py "$@" > $out/bench.json 2> $out/bench.err || { echo "bench failed"; tail -5 $out/bench.err; exit 1; }
cat $out/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline "$@" > $out/bench_traced.json 2> $out/trace.err
cp $out/trace/*/*_kernel_stats.csv $out/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_f -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > /dev/null 2> $out/pmc_f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_w -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > /dev/null 2> $out/pmc_w.err
cp $out/pmc_f/*/*_counter_collection.csv $out/pmc_fetch.csv
cp $out/pmc_w/*/*_counter_collection.csv $out/pmc_write.csv
python3 - $out "$@" <<'PY'
import csv, json, sys
out = sys.argv[1]
def mean(path, counter, kern):
    v = [float(r['Counter_Value']) for r in csv.DictReader(open(path))
         if r['Counter_Name'] == counter and kern in r['Kernel_Name']]
    return sum(v) / len(v), len(v)
res = {}
for kern in ("cfs_sym_tile_kernel", "cfs_fold_kernel"):
    f, nf = mean(out + '/pmc_fetch.csv', 'FETCH_SIZE', kern)
    w, nw = mean(out + '/pmc_write.csv', 'WRITE_SIZE', kern)
    res[kern] = {"FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB": w, "launches": [nf, nw],
                 "hbm_bytes_per_launch": int((2 * f + w) * 1024),
                 "correction": "reads = 2 x FETCH_SIZE (gfx950: 128-B requests tallied at 64 B), writes exact"}
json.dump(res, open(out + '/hbm_traffic_raw.json', 'w'), indent=1)
print(json.dumps(res))
PY
head -4 $out/kernel_stats.csv | cut -c1-200
