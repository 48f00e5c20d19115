#!/bin/bash
# same-box A/B: old library vs new (no staging) vs new (LDS-DMA staging): tools/r3_ab2.sh <outdir>
out=${1:-gpurun_out/ab2}; mkdir -p $out
run() { local tag=$1; shift; local envs=$1; shift
  env $envs python3 bench.py --no-cpu-baseline --steps 300 --warmup 50 "$@" > $out/$tag.json 2> $out/$tag.err || { echo "$tag failed"; tail -3 $out/$tag.err; }
  python3 - $out/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]; c=d["config"]
    print(f"{sys.argv[2]:24s} step {d['ms_per_step']*1e3:7.2f} us  tile(ev) {r['kernel_ms']*1e3:7.2f}  inkernel {((r['kernel_ms_inkernel_clock'] or 0)*1e3):7.2f}  frac {r['frac']:.3f} whole {c['effective_GBps_whole_step']/8000:.3f} blk {c['block_threads']} lds {c['lds_bytes']} tiles {c['tiles']} halo {c['halo_slots']} {c['format']}", flush=True)
except Exception as e:
    print(sys.argv[2], "no result", e, flush=True)
PY
}
OLD="CFS_HIP_LIB=$PWD/_ab/libcfs_hip_old.so"
for rep in 1 2; do
for cfg in "pwtk --matrix pwtk" "ldoor --matrix ldoor" "pdb --matrix pdb1HYS" "flan8 --shard-of 8 --shard-rank 3" "queen8 --matrix Queen_4147 --dtype f32 --shard-of 8 --shard-rank 3"; do
  set -- $cfg; t=$1; shift
  # pin the shape so that the three builds run the same schedule (tuning none: no measured choices)
  run ${t}_old_$rep "$OLD" --tuning none --block 1024 --max-slots 9984 "$@"
  run ${t}_new0_$rep CFS_HIP_STAGE=0 --tuning none --block 1024 --max-slots 9984 "$@"
  run ${t}_stg_$rep CFS_X=1 --tuning none --block 1024 --max-slots 9984 "$@"
done
done
run flan_old "$OLD"
run flan_new CFS_HIP_STAGE=0
run queen_old "$OLD" --matrix Queen_4147 --dtype f32
run queen_new CFS_HIP_STAGE=0 --matrix Queen_4147 --dtype f32
