#!/bin/bash
# same-box A/B: the block form with the narrow blocks in natural order (previous build) against lane order
for rep in 1 2 3; do
for cfg in "Flan_1565 f64" "Queen_4147 f64" "Queen_4147 f32" "pwtk f64"; do
for lib in new natural; do
  set -- $cfg
  if [ $lib = natural ]; then export CFS_HIP_LIB=$PWD/_ab/libcfs_hip_natural.so; else unset CFS_HIP_LIB; fi
  echo -n "$1 $2 $lib rep=$rep: "
  CFS_HIP_CSR_KERNEL=block python bench.py --format csr --no-cpu-baseline --steps 300 --warmup 50 --matrix $1 --dtype $2 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print(d['ms_per_step'], r['kernel'], r['kernel_ms'], r['frac'], r['nnz_with_16bit_columns'])"
done; done; done
