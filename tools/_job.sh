python -m pytest tests/test_gpu_round2.py -m gpu -x -q -k "hyb" 2>&1 | tail -1
python tools/quick_bench.py ldoor 1.0 0,0,0,160 0,0,0,32 2>&1 | grep -E "cfg"
python tools/quick_bench.py pdb1HYS 1.0 0,0,0,160 0,0,0,32 2>&1 | grep -E "cfg"
