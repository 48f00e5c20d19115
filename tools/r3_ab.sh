#!/bin/bash
# same-box A/B of the tile kernel's staging (developer tool): tools/r3_ab.sh <outdir>
out=${1:-gpurun_out/ab}; mkdir -p $out
run() { # tag, env, args...
  tag=$1; shift; envs=$1; shift
  env $envs python3 bench.py --no-cpu-baseline --steps 200 --warmup 50 "$@" > $out/$tag.json 2> $out/$tag.err || { echo "$tag failed"; tail -3 $out/$tag.err; }
  python3 - $out/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]; c=d["config"]
    print(f"{sys.argv[2]:28s} step {d['ms_per_step']*1e3:7.2f} us  tile(ev) {r['kernel_ms']*1e3:7.2f} us  inkernel {((r['kernel_ms_inkernel_clock'] or 0)*1e3):7.2f}  frac {r['frac']:.3f}  whole {c['effective_GBps_whole_step']/8000:.3f}  blk {c['block_threads']} lds {c['lds_bytes']} tiles {c['tiles']} halo {c['halo_slots']} fmt {c['format']} pre {c['preproc_s']}")
except Exception as e:
    print(sys.argv[2], "no result", e)
PY
}
for rep in 1 2; do
for m in pwtk ldoor pdb1HYS; do
  run ${m}_stg0_$rep CFS_HIP_STAGE=0 --matrix $m
  run ${m}_stg_$rep CFS_X=1 --matrix $m
done
run flan8_stg0_$rep CFS_HIP_STAGE=0 --shard-of 8 --shard-rank 3
run flan8_stg_$rep CFS_X=1 --shard-of 8 --shard-rank 3
done
run flan_stg0 CFS_HIP_STAGE=0
run flan_stg CFS_X=1
run queen_stg0 CFS_HIP_STAGE=0 --matrix Queen_4147 --dtype f32
run queen_stg CFS_X=1 --matrix Queen_4147 --dtype f32
run queen8_stg0 CFS_HIP_STAGE=0 --matrix Queen_4147 --dtype f32 --shard-of 8 --shard-rank 3
run queen8_stg CFS_X=1 --matrix Queen_4147 --dtype f32 --shard-of 8 --shard-rank 3
