#!/bin/bash
# general CSR kernel: wave-stream form vs block form, same box
out=${1:-gpurun_out/csr}; mkdir -p $out
for rep in 1 2; do
for m in Flan_1565 ldoor pwtk; do
  for form in wave block; do
    CFS_HIP_CSR_KERNEL=$form python3 bench.py --format csr --matrix $m --no-cpu-baseline --steps 100 --warmup 20 > $out/${m}_${form}_$rep.json 2> $out/${m}_${form}_$rep.err || tail -3 $out/${m}_${form}_$rep.err
    python3 -c "
import json,sys
d=json.load(open('$out/${m}_${form}_$rep.json')); r=d['roofline']
print('$m $form: step %.1f us kernel %.1f us frac %.3f gflops %.0f' % (d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['frac'], d['value']), flush=True)"
  done
done
done
