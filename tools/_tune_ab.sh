#!/bin/bash
# same-box A/B of tune() (bench.py --tuning none, preproc_s): values uploaded beside the device work (new) vs before it
for rep in 1 2 3 4; do
for lib in new old; do
  if [ $lib = old ]; then export CFS_HIP_LIB=$PWD/_ab/libcfs_hip_natural.so; else unset CFS_HIP_LIB; fi
  for m in "Flan_1565 f64" "Queen_4147 f32"; do
  set -- $m
  echo -n "$1 $lib rep=$rep: "
  CFS_PLAN_VERBOSE=1 python bench.py --tuning none --no-cpu-baseline --steps 30 --warmup 10 --matrix $1 --dtype $2 2>gpurun_out/r3s/tn_${1}_${lib}_$rep.err | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['config']['preproc_s'], d['ms_per_step'], d['roofline']['frac'])"
  grep "create: schedule" gpurun_out/r3s/tn_${1}_${lib}_$rep.err | head -1
  done
done; done
