#!/bin/bash
# dump the memory/sync skeleton of a kernel's ISA: tools/isa.sh <mangled-substring>
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fopenmp -I/root/repo/include -I/root/repo/cfs_spmv_amd/csrc --cuda-device-only -S /root/repo/cfs_spmv_amd/csrc/cfs_hip.hip -o /tmp/probe/cfs_hip.s 2>&1 | grep -E "error|warning"
python3 - "$1" <<'PY'
import re,sys
s=open('/tmp/probe/cfs_hip.s').read()
pat=sys.argv[1]
m=re.search(r'^(_Z\w*'+pat+r'\w*):(.*?)\.end_amdhsa_kernel', s, re.S|re.M)
print(m.group(1))
lines=m.group(2).split('\n')
print(len(lines),"lines")
for i,l in enumerate(lines):
    if re.search(r'global_load|buffer_load|ds_add|ds_read|ds_write|s_waitcnt|s_barrier|s_cbranch|^\.LBB|global_store|s_load|sched_barrier', l):
        print(i, l.strip())
for k in ('vgpr_count','sgpr_count','lds_size'):
    print(k, re.findall(r'\.'+k+r':\s+(\d+)', s)[:1])
PY
