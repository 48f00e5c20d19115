export CFS_HIP_KEEP_ALT=1
bash tools/profile_round.sh r02/flan_w1024 --steps 200 --warmup 50
