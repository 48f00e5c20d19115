#!/bin/bash
# LDS counters of the tile kernel with and without the sibling-combined atomics
# (separate --pmc passes, never combined with tracing).  usage: tools/pmc_lds.sh <outdir>
export TMPDIR=/tmp
out=$1
mkdir -p $out
for m in "ldoor --matrix ldoor" "queen_f32 --matrix Queen_4147 --dtype f32 --block 1024 --max-slots 9984" "flan --matrix Flan_1565"; do
  set -- $m; name=$1; shift
  for comb in 0 1; do
    for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
      tag=${name}_comb${comb}_$(echo $grp | tr ' ' '+')
      CFS_HIP_COMBINE=$comb rocprofv3 --pmc $grp --output-format csv -d $out/$tag -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --tuning none "$@" > $out/$tag.log 2>&1 || echo "pass $tag failed"
    done
  done
done
python3 - $out <<'PY'
import csv, glob, sys, json, collections, os
out = sys.argv[1]
res = collections.defaultdict(dict)
for d in sorted(glob.glob(out + '/*_comb*')):
    if not os.path.isdir(d): continue
    tag = os.path.basename(d)
    name, comb = tag.split('_comb')[0], tag.split('_comb')[1][0]
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'cfs_sym_tile_kernel' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for c, v in acc.items():
            res[f"{name} combine={comb}"][c] = round(sum(v) / len(v))
json.dump(res, open(out + '/lds_counters.json', 'w'), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
PY
