mkdir -p gpurun_out/r02
python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r02/t_fr.log 2>&1; tail -2 gpurun_out/r02/t_fr.log
for m in pwtk ldoor Queen_4147; do
 if [ $m = Queen_4147 ]; then export QB_DTYPE=f32; fi
 echo "$m runs : $(CFS_PLAN_VERBOSE=1 python tools/quick_bench.py $m 1.0 0,0,0,32 2>&1 | grep -E 'cfg|cfs_hip\] fold' | cut -c1-230 | tr '\n' ' ')"
 echo "$m recs : $(CFS_HIP_NO_FOLD_RUNS=1 python tools/quick_bench.py $m 1.0 0,0,0,32 2>&1 | grep cfg | cut -c1-200)"
done
unset QB_DTYPE
echo "shard runs: $(python tools/shard_bench.py Flan_1565 1.0 8 3 2>&1 | tail -1 | cut -c1-300)"
echo "shard recs: $(CFS_HIP_NO_FOLD_RUNS=1 python tools/shard_bench.py Flan_1565 1.0 8 3 2>&1 | tail -1 | cut -c1-300)"
