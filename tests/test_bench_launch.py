"""bench.py --gpus N without a torchrun environment: the parent must start the N ranks
itself as a CHILD process (it never touches HIP) and pass the child's exit code on.
On this CPU box the ranks exit with "needs a GPU" -- which is exactly what shows that
they were started, that they got a torchrun environment, and that a failing child makes
the parent fail."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_self_launches_ranks_and_propagates_failure():
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["CUDA_VISIBLE_DEVICES"] = ""  # no GPU for the children whatever the box has
    env["HIP_VISIBLE_DEVICES"] = ""
    env["ROCR_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--matrix", "pwtk", "--scale", "0.01"],
                       capture_output=True, text=True, env=env, timeout=600)
    err = r.stderr
    assert "launching 2 ranks" in err and "torch.distributed.run" in err, err[-2000:]
    assert "--nproc-per-node=2" in err and "--master-addr 127.0.0.1" in err
    # both ranks ran bench.py under torchrun and refused to run without a GPU
    assert err.count("bench.py needs a GPU") >= 1, err[-3000:]
    assert r.returncode != 0  # a child failure is the parent's failure
    assert r.stdout.strip() == ""  # and no JSON line was invented


def test_bench_does_not_relaunch_inside_torchrun(monkeypatch):
    """with WORLD_SIZE set (the driver's own torchrun launch) bench.py must NOT spawn"""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    called = []
    monkeypatch.setattr(bench, "self_launch", lambda a: called.append(a) or 0)
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    try:
        bench.main()
    except SystemExit:
        pass
    except Exception:
        pass  # no GPU / no process group here: irrelevant
    assert not called


def test_watchdog_prints_the_line_and_exits_nonzero(tmp_path):
    """a stalled exchange form (a collective that never returns) must end bench.py with a
    NON-ZERO exit code after the contract line was printed -- rc 0 would hide the hang from
    torchrun and the driver.  The stall is a sleeping callable under the same
    run_guarded() the N > 1 path uses; no GPU, no process group."""
    code = (
        "import sys, time, json; sys.path.insert(0, %r); import bench\n"
        "out = {'metric': 'm', 'value': 1.0}\n"
        "def on_timeout():\n"
        "    out['exchange_forms'] = {'none': 0.1, 'others': 'timed out'}\n"
        "    print(json.dumps(out), flush=True)\n"
        "bench.run_guarded(lambda: time.sleep(60), 0.5, on_timeout)\n"
        "print('NOT REACHED')\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    sys.path.insert(0, ROOT)
    import bench
    assert r.returncode == bench.WATCHDOG_EXIT != 0, (r.returncode, r.stderr[-500:])
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and "NOT REACHED" not in r.stdout
    import json
    assert json.loads(lines[0])["exchange_forms"]["others"] == "timed out"


def test_watchdog_is_silent_when_the_forms_return():
    sys.path.insert(0, ROOT)
    import bench
    fired = []
    assert bench.run_guarded(lambda: 42, 30.0, lambda: fired.append(1)) == 42
    assert not fired
