#!/bin/bash
# A/B of the general CSR kernel (block form): entries per lane and load (1 / 2 / 4), fp64 and fp32
for rep in 1 2; do
for cfg in "Flan_1565 f64" "Queen_4147 f32" "Flan_1565 f32" "pwtk f64"; do
for w in 4 2; do
  set -- $cfg
  echo -n "$1 $2 wide=$w rep=$rep: "
  CFS_HIP_CSR_WIDE=$w CFS_HIP_CSR_KERNEL=block python bench.py --format csr --no-cpu-baseline --steps 300 --warmup 50 --matrix $1 --dtype $2 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print(d['ms_per_step'], r['kernel'], r['kernel_ms'], r['frac'], r['nnz_with_16bit_columns'])"
done; done; done
