python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1
echo "ldoor hyb: $(python tools/quick_bench.py ldoor 1.0 0,0,0,160 2>&1 | grep cfg | cut -c1-200)"
echo "pdb1HYS hyb: $(python tools/quick_bench.py pdb1HYS 1.0 0,0,0,160 2>&1 | grep cfg | cut -c1-140)"
echo "flan: $(python tools/quick_bench.py Flan_1565 1.0 0,0,0,32 2>&1 | grep cfg | cut -c1-140)"
