mkdir -p gpurun_out/r02
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python -m pytest tests -m gpu -x -q > gpurun_out/r02/t_final.log 2>&1; tail -2 gpurun_out/r02/t_final.log
python bench.py > gpurun_out/r02/bench_default.json 2> gpurun_out/r02/bench_default.err; tail -c 900 gpurun_out/r02/bench_default.json
