nproc; cat /sys/fs/cgroup/cpu.max
bash tools/profile_round.sh r02/flan --steps 200 --warmup 50
bash tools/profile_round.sh r02/pwtk --matrix pwtk --steps 400 --warmup 100
bash tools/profile_round.sh r02/ldoor --matrix ldoor --steps 300 --warmup 100
bash tools/profile_round.sh r02/flan_shard8 --shard-of 8 --shard-rank 3 --steps 400 --warmup 100
bash tools/profile_round.sh r02/queen_f32 --matrix Queen_4147 --dtype f32 --steps 100 --warmup 30
bash tools/profile_round.sh r02/unstruct --matrix unstruct --steps 200 --warmup 50
bash tools/profile_round.sh r02/queen_f32_shard8 --matrix Queen_4147 --dtype f32 --shard-of 8 --shard-rank 3 --steps 400 --warmup 100
bash tools/profile_round.sh r02/pdb1HYS --matrix pdb1HYS --steps 400 --warmup 100
python3 bench.py --tuning none --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/r02/flan_tuning_none.json 2>/dev/null; python3 -c "import json; d=json.load(open('gpurun_out/r02/flan_tuning_none.json')); print('tuning none: preproc', d['config']['preproc_s'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'])"
CFS_PLAN_VERBOSE=1 python3 bench.py --tuning none --no-cpu-baseline --steps 5 --warmup 2 2>&1 >/dev/null | grep cfs_ | head -30
