python -m pytest tests/test_gpu_round2.py -m gpu -x -q -k "hyb or update_values" 2>&1 | tail -1
python tools/quick_bench.py ldoor 1.0 0,0,0,160 2>&1 | grep -E "cfg" | cut -c1-200
python tools/quick_bench.py pdb1HYS 1.0 0,0,0,160 2>&1 | grep -E "cfg" | cut -c1-200
python tools/quick_bench.py Flan_1565 1.0 0,0,0,32 2>&1 | grep -E "cfg" | cut -c1-200
python tools/quick_bench.py Flan_1565 1.0 0,0,0,32 2>&1 | grep -E "cfg" | cut -c1-200
