bash tools/profile_round.sh r02/flan_shard8 --shard-of 8 --shard-rank 3 --steps 400 --warmup 100
bash tools/profile_round.sh r02/queen_f32_shard8 --matrix Queen_4147 --dtype f32 --shard-of 8 --shard-rank 3 --steps 400 --warmup 100
bash tools/profile_round.sh r02/pwtk --matrix pwtk --steps 400 --warmup 100
bash tools/profile_round.sh r02/ldoor --matrix ldoor --steps 300 --warmup 100
bash tools/profile_round.sh r02/queen_f32 --matrix Queen_4147 --dtype f32 --steps 100 --warmup 30
bash tools/profile_round.sh r02/pdb1HYS --matrix pdb1HYS --steps 400 --warmup 100
