set -e
python - <<'PY'
import sys, time
sys.path.insert(0, ".")
from cfs_spmv_amd import synth
n, rp, ci, va, low = synth.generate("Flan_1565", 1.0)
t = time.time(); synth.write_mtx("gpurun_out/flan_full.mtx", n, rp, ci, va); print("write_mtx", round(time.time() - t, 1), "s")
PY
ls -la gpurun_out/flan_full.mtx
CFS_MMF_VERBOSE=1 timeout -k 10 600 build/bench_spmv_mmf gpurun_out/flan_full.mtx 1 128
rm -f gpurun_out/flan_full.mtx
