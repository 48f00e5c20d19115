#!/bin/bash
out=${1:-gpurun_out/csr2}; mkdir -p $out
for rep in 1 2; do
for m in Flan_1565 ldoor; do
  for w in 0 1; do
    CFS_HIP_CSR_KERNEL=block CFS_HIP_CSR_WIDE=$w python3 bench.py --format csr --matrix $m --no-cpu-baseline --steps 100 --warmup 20 > $out/${m}_w${w}_$rep.json 2> $out/${m}_w${w}_$rep.err || tail -3 $out/${m}_w${w}_$rep.err
    python3 -c "
import json
d=json.load(open('$out/${m}_w${w}_$rep.json')); r=d['roofline']
print('$m wide=$w: step %.1f us kernel %.1f us frac %.3f' % (d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['frac']), flush=True)"
  done
done
done
