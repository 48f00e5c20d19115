for b in 4096 2048 8192; do for m in Flan_1565 ldoor; do echo "csr block $b $m: $(CFS_HIP_CSR_BLOCK=$b QB_CSR=1 python tools/quick_bench.py $m 1.0 0,0,0,32 2>&1 | grep '"csr"')"; done; done
echo "f32 queen: $(QB_DTYPE=f32 QB_CSR=1 python tools/quick_bench.py Queen_4147 1.0 0,0,0,32 2>&1 | grep '"csr"')"
python -m pytest tests/test_gpu_parity.py -m gpu -q -k csr 2>&1 | tail -1
