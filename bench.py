#!/usr/bin/env python3
"""bench.py -- the contract benchmark of the symmetric-SpMV hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one SpMV y <- A x of the hot path (tile kernel + halo fold; for N > 1
every rank does that for its row block -- mirrored shards need no exchange, the
exchange forms are selectable with --exchange) on the Flan_1565 stand-in
(BASELINE.json's headline config; the real .mtx is not available offline, the
generator is SURVEY.md section 8d's), fp64, x and y resident in HBM.  The matrix
is FIXED and sharded by 1-D row blocks, so scaling is "strong".

Metric (BASELINE.json): GFLOP/s = 2*nnz_full / t, the reference driver's own
formula (bench/bench_spmv_mmf.cpp:168), with the achieved algorithmic GB/s of
the dominant kernel against the 8 TB/s HBM3E peak in "roofline", and the CPU
oracle (the reference's OpenMP conflict-free path, restated) timed on this
box's host cores in "cpu_baseline" (rank 0, N = 1 only).

`python bench.py --gpus N` WITHOUT a torchrun environment (WORLD_SIZE unset) starts
the N ranks itself: the parent never touches HIP, it spawns
`python -m torch.distributed.run --nproc-per-node N ... bench.py <same args>` as a
child process and exits with its code.  At N > 1 the line also carries
`exchange_forms`: ms_per_step of the three forms of the off-block exchange (none =
mirrored shards, all_to_all, reduce_scatter = the RCCL collective the north-star
names), measured in the same run.

Only the cpu_baseline leg imports oracle/ -- as the reported baseline, never as
the thing measured.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6300.0  # same guide: what a streaming kernel reaches on the wire


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--matrix", default="Flan_1565")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--max-slots", type=int, default=0)
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-loops", type=int, default=128,
                    help="timed SpMVs of each cpu_baseline leg (the reference driver's protocol: "
                         "loops/2 warm-up, loops timed; SURVEY 8d asks for 128)")
    ap.add_argument("--cpu-bind", action="store_true",
                    help="OMP_PLACES=cores OMP_PROC_BIND=close for the cpu_baseline legs")
    ap.add_argument("--no-exchange-forms", action="store_true",
                    help="N>1: skip the extra timings of the other two exchange forms")
    ap.add_argument("--exchange", default="none", choices=["none", "all_to_all", "reduce_scatter"],
                    help="N>1: 'none' = mirrored shards, every off-block entry is stored by both "
                         "ranks it touches and no SpMV needs a collective (default); "
                         "'all_to_all' = packed contributions to lower ranks, one sparse "
                         "all-to-all; 'reduce_scatter' = the dense form over padded blocks")
    ap.add_argument("--tuning", default="aggressive", choices=["aggressive", "none"],
                    help="Tuning::Aggressive (the reference's default: tune() also measures "
                         "alternatives and keeps the fastest) or Tuning::None (one schedule build)")
    ap.add_argument("--flags", type=int, default=0, help="extra CFS_HIP_FLAG_* bits (hyb = 128, "
                                                         "deterministic = 1024, ...)")
    ap.add_argument("--shard-of", type=int, default=0,
                    help="N=1 only: time ONE rank's mirrored row block of an N-way split on this "
                         "GPU (what every rank of the driver's N-GPU run executes per step; no "
                         "process group).  value / roofline then refer to that block")
    ap.add_argument("--shard-rank", type=int, default=0)
    ap.add_argument("--format", default="sss", choices=["sss", "csr"],
                    help="sss (default): the symmetric hot path.  csr: Format::csr -- every stored entry "
                         "through the general CSR kernel (cpu_mv's role, csr_matrix.tpp:2683-2704), N = 1")
    ap.add_argument("--settle-ms", type=float, default=300.0,
                    help="untimed SpMVs before the W warm-up steps (set-up: clocks and power in the state "
                         "of a long run)")
    ap.add_argument("--event-every", type=int, default=4,
                    help="bracket the tile kernel with HIP events on every n-th timed step")
    return ap.parse_args()


def host_cpus():
    """CPUs this process may really use: affinity mask capped by the cgroup quota (a
    container may show every core of the host and still grant only a share)"""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = min(cores, max(1, -(-int(quota) // int(period))))
    except Exception:
        pass
    return max(1, cores)


def cpu_baseline(n, rp, ci, va, x_host, nnz_full, loops, threads, bound):
    """the oracle's restatement of the reference's two CPU paths on the host cores,
    same matrix, same x, reference protocol (loops/2 warm-up, loops timed,
    bench/bench_spmv_mmf.cpp:154-167): cpu_mv_sym_conflict_free_v2
    (csr_matrix.tpp:2965-3028; `value`) and plain CSR cpu_mv (:2683-2704; `csr`).
    `threads` = the CPUs this process may use, counted BEFORE any OpenMP runtime was
    loaded (a bound runtime shrinks the main thread's mask)."""
    import ctypes as C
    import numpy as np
    from oracle import oracle
    T = max(1, min(threads, 96))  # MaxThreads = 96, include/utils/runtime.hpp:15
    t0 = time.time()
    o = oracle.SymOracle(n, rp, ci, va, T)
    preproc = time.time() - t0
    y = o.spmv(x_host)
    for _ in range(loops // 2):
        o.spmv(x_host, y)
    t0 = time.time()
    for _ in range(loops):
        o.spmv(x_host, y)
    dt = (time.time() - t0) / loops
    info = o.info()
    o.close()
    # plain CSR leg: cpu_mv over partition_by_nnz (tune(Aggressive), csr_matrix.tpp:250-254)
    L = oracle.lib()
    suf = "f64" if va.dtype == np.float64 else "f32"
    rs = oracle.partition_by_nnz(n, rp, T)
    f = getattr(L, "orc_csr_spmv_" + suf)
    xs = np.ascontiguousarray(x_host, va.dtype)
    args = (n, rp.ctypes.data, ci.ctypes.data, va.ctypes.data, T, rs.ctypes.data,
            y.ctypes.data, xs.ctypes.data)
    for _ in range(loops // 2):
        f(*args)
    t0 = time.time()
    for _ in range(loops):
        f(*args)
    dt_csr = (time.time() - t0) / loops
    s = va.itemsize
    return {
        "value": round(2.0 * nnz_full / dt / 1e9, 2), "unit": "GFLOP/s", "cores": T,
        "kind": "port",
        "sample": f"whole workload, threads "
                  f"{'bound (OMP_PLACES=cores OMP_PROC_BIND=close)' if bound else 'unbound'}: "
                  f"{loops // 2} warm-up + "
                  f"{loops} timed SpMVs of the oracle's conflict-free v2 path "
                  f"({info['ncolors']} colours), {dt * 1e3:.2f} ms/SpMV; same protocol for the "
                  f"plain-CSR cpu_mv leg, {dt_csr * 1e3:.2f} ms/SpMV.  The oracle's preprocessing "
                  f"({preproc:.1f}s) skips the indirect-conflict scan of rows whose upper entries "
                  f"stay in one thread, an early-out the reference does not have: it is NOT the "
                  f"reference's preprocessing time.  Parity of the GPU path against this oracle "
                  f"(tests/) is |y - y_ref| <= 1e-12 (fp64) / 1e-5 (fp32) x max(|y_ref|, "
                  f"sum_j |a_ij||x_j|) -- the row scale, not plain |y_ref| -- plus the "
                  f"reference's own isEqual on rows that do not cancel",
        "ms_per_step": round(dt * 1e3, 3),
        "csr": {"value": round(2.0 * nnz_full / dt_csr / 1e9, 2), "unit": "GFLOP/s",
                "ms_per_step": round(dt_csr * 1e3, 3),
                "effective_GBps": round((nnz_full * (4 + s) + n * (4 + 2 * s)) / dt_csr / 1e9, 1)},
        "effective_GBps": round(((nnz_full - n) // 2 * (4 + s) + n * (4 + 3 * s)) / dt / 1e9, 1),
    }


def bench_csr(args, cfs, lib, n, rp, ci, va, x_host, nnz_full, dev, t_dt, data_kind, source, ncpus):
    """Format::csr: one step = y <- A x over EVERY stored entry with the general CSR kernel
    (cfs_csr_stream_kernel; the reference's cpu_mv, csr_matrix.tpp:2683-2704 -- the ground truth
    of its own self-check, test/test_spmv_mmf.cpp:85-89).  Same protocol and line as the
    symmetric path; algorithmic bytes per launch = nnz*(4+s) + n*(4+2s): colind + values, rowptr,
    x once, y once."""
    import ctypes as C
    import numpy as np
    import torch
    from cfs_spmv_amd import _lib
    t0 = time.time()
    A = cfs.CsrMatrix(n, n, rp, ci, va)
    preproc = time.time() - t0
    x = torch.from_numpy(x_host).to(dev)
    y = torch.full((n,), float("nan"), dtype=t_dt, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    f = lib.cfs_hip_csr_spmv_async
    hA, yp, xp, stp = A._h, C.c_void_p(y.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(stream)

    def step():
        rc = f(hA, yp, xp, stp)
        if rc != 0:
            _lib.check(rc)

    def new_event():
        e = C.c_void_p()
        _lib.check(lib.cfs_hip_event_create(C.byref(e)))
        return e
    K = args.steps
    every = max(1, min(args.event_every, K // 10 if K >= 10 else 1))
    sampled = [i for i in range(K) if i % every == 0]
    ev0 = {i: new_event() for i in sampled}
    ev1 = {i: new_event() for i in sampled}
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.settle_ms * 1e-3:
        for _ in range(16):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        if i in ev0:
            _lib.check(lib.cfs_hip_event_record(ev0[i], stream))
            step()
            _lib.check(lib.cfs_hip_event_record(ev1[i], stream))
        else:
            step()
    torch.cuda.synchronize()
    ms_per_step = (time.perf_counter() - t0) / K * 1e3
    ms = C.c_float()
    tot = 0.0
    for i in sampled:
        _lib.check(lib.cfs_hip_event_elapsed_ms(ev0[i], ev1[i], C.byref(ms)))
        tot += ms.value
    kern_ms = tot / len(sampled)
    for e in list(ev0.values()) + list(ev1.values()):
        lib.cfs_hip_event_destroy(e)
    # self-check against the symmetric path (another kernel, another format) on the same x
    S = cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(flags=32))
    y2 = torch.empty(n, dtype=t_dt, device=dev)
    S.dense_vector_multiply(y2, x)
    torch.cuda.synchronize()
    scale = torch.maximum(y2.abs(), torch.tensor(1.0, dtype=t_dt, device=dev))
    err = float(((y - y2).abs() / scale).max().item())
    S.close()
    if not err < (1e-9 if args.dtype == "f64" else 1e-3):
        raise SystemExit(f"self-check failed: max scaled |y_csr - y_sss| = {err}")
    s_ = va.itemsize
    form, meas = C.c_int(), C.c_int()
    _lib.check(lib.cfs_hip_csr_kernel_form(A._h, C.byref(form), C.byref(meas)))
    kname = "cfs_csr_wave_kernel" if form.value == 1 else "cfs_csr_stream_kernel"
    alg = int(nnz_full * (4 + s_) + n * (4 + 2 * s_))
    achieved = alg / (kern_ms * 1e-3) / 1e9
    streamed, narrow = C.c_int64(), C.c_int64()
    _lib.check(lib.cfs_hip_csr_stats(A._h, C.byref(streamed), C.byref(narrow)))
    # HBM bytes per launch from a committed PMC pass of the same kernel form over the same arrays
    traffic, traffic_source = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            key = f"{args.matrix}:{args.scale}:{args.dtype}:csr"
            for ent in json.load(f).get(key, []):
                if ent.get("kernel") == kname and ent.get("bytes_streamed") == streamed.value:
                    traffic = ent.get("hbm_bytes_per_launch")
                    traffic_source = {"file": "profiles/hbm_traffic.json", "key": key,
                                      "matched_on": {"kernel": kname, "bytes_streamed": streamed.value},
                                      "measured_by": ent.get("source"),
                                      "note": "a committed measurement of the SAME kernel form and arrays on "
                                              "another run, not taken inside this run"}
    except Exception:
        traffic = None
    out = {
        "metric": f"fp{s_ * 8} general CSR SpMV GFLOP/s", "value": round(2.0 * nnz_full / (ms_per_step * 1e-3) / 1e9, 2),
        "unit": "GFLOP/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": args.dtype, "data": data_kind,
        "config": {"workload": f"{source}: n={n}, nnz_full={nnz_full}, general CSR SpMV y=Ax over every "
                               f"stored entry (Format::csr)", "format": "csr",
                   "algorithmic_bytes_per_spmv": alg,
                   "effective_GBps_whole_step": round(alg / (ms_per_step * 1e-3) / 1e9, 1),
                   "preproc_s": round(preproc, 2),
                   "kernel_form": ("wave" if form.value == 1 else "block") +
                                  (" (pinned by CFS_HIP_CSR_KERNEL)" if os.environ.get("CFS_HIP_CSR_KERNEL")
                                   else " (the faster of the two forms, measured at the first SpMV)")},
        "roofline": {"bound": "hbm", "kernel": kname,
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "kernel_ms": round(kern_ms, 5), "kernel_samples": len(sampled),
                     "algorithmic_bytes_per_launch": alg,
                     "bytes_streamed_by_format": streamed.value,
                     "nnz_with_16bit_columns": narrow.value},
    }
    r_ = out["roofline"]
    if traffic:
        wire = traffic / (kern_ms * 1e-3) / 1e9
        r_.update({"traffic_source": traffic_source, "hbm_GBps_from_traffic": round(wire, 1),
                   "hbm_frac_of_peak_from_traffic": round(wire / HBM_PEAK_GBS, 4),
                   "hbm_frac_of_achievable": round(wire / HBM_ACHIEVABLE_GBS, 4),
                   "achievable_GBps": HBM_ACHIEVABLE_GBS})
    else:
        r_["traffic_source"] = None
    if not args.no_cpu_baseline:
        try:
            cb = cpu_baseline(n, rp, ci, va, x_host, nnz_full, args.cpu_loops, ncpus, args.cpu_bind)
            out["cpu_baseline"] = dict(cb["csr"], cores=cb["cores"], kind="port",
                                       sample=cb["sample"] + " (this line quotes the plain-CSR leg)")
        except Exception as e:
            out["cpu_baseline"] = {"value": None, "unit": "GFLOP/s", "cores": 0, "kind": "port",
                                   "sample": f"failed: {e}"}
    A.close()
    print(json.dumps(out), flush=True)


def time_other_forms(args, cfs, A, sh, n, rp, ci, va, N, rank, rs, dev, opt, backend, x, y,
                     stream, barrier, dist):
    """ms_per_step (max over ranks, K steps between barriers) of the exchange forms the
    main run did not use.  "none" = mirrored shards (no collective), "all_to_all" = packed
    contributions, one sparse all-to-all, "reduce_scatter" = one RCCL
    reduce_scatter_tensor(sum) over padded blocks (the north-star's form).  A form that
    cannot run reports its error text instead of a number; the main line is unaffected."""
    import torch
    from cfs_spmv_amd import _lib
    from cfs_spmv_amd.dist import ShardedSym, build_shard
    from cfs_spmv_amd.matrix import FLAG_SHARD_EXCHANGE
    out = {}
    K, W = args.steps, max(5, args.warmup // 2)
    handles = {}

    def timed(step):
        for _ in range(W):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(K):
            step()
        barrier()
        el = time.perf_counter() - t0
        t = torch.tensor([el], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return round(float(t.item()) / K * 1e3, 5)

    for form in ("none", "all_to_all", "reduce_scatter"):
        if form == args.exchange:
            continue
        try:
            if form == "none":
                A2, sh2, used = build_shard(n, rp, ci, va, N, rank, rs, dev, options=opt,
                                            exchange="none", stage_via_host=(backend != "nccl"))
                if used != "none":
                    out[form] = "not mirrorable: fell back to all_to_all"
                    A2.close()
                    continue
                handles["none"] = A2
            else:
                # both exchange forms share ONE handle built with CFS_HIP_FLAG_SHARD_EXCHANGE
                A2 = handles.get("x") or (A if args.exchange != "none" else None)
                if A2 is None:
                    A2 = cfs.SymMatrix(n, rp, ci, va, options=_lib.Options(
                        opt.max_slots, opt.max_tile_nnz, opt.block_threads,
                        opt.flags | FLAG_SHARD_EXCHANGE), row_splits=rs, rank=rank)
                    handles["x"] = A2
                sh2 = ShardedSym(A2, N, rank, va.dtype, dev, stage_via_host=(backend != "nccl"),
                                 exchange=form, row_splits=rs)
            out[form] = timed(lambda: sh2.spmv(y, x))
        except Exception as e:  # reported, never fatal for the main number
            out[form] = f"failed: {type(e).__name__}: {e}"[:200]
    torch.cuda.synchronize()
    for h in handles.values():
        h.close()
    return out


WATCHDOG_EXIT = 3  # exit code of a run whose extra exchange-form timings hung


def run_guarded(fn, timeout_s, on_timeout):
    """fn() under a watchdog.  When it has not returned after timeout_s the watchdog
    calls on_timeout() (which prints the contract line with what IS known) and ends
    the process with WATCHDOG_EXIT: a rank that sits in a hung collective has touched
    the GPU and cannot be unwound, and the caller -- torchrun, the driver -- must see
    the hang as a FAILURE, never as rc 0."""
    import threading

    def give_up():
        try:
            on_timeout()
        finally:
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(WATCHDOG_EXIT)
    dog = threading.Timer(timeout_s, give_up)
    dog.daemon = True
    dog.start()
    try:
        return fn()
    finally:
        dog.cancel()


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(args):
    """--gpus N without a torchrun environment: start the N ranks as a CHILD process
    (the parent has not touched the GPU and never does) and pass its exit code on"""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    print(f"[bench] WORLD_SIZE unset: launching {args.gpus} ranks: {' '.join(cmd)}",
          file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and os.environ.get("CFS_FORCE_DIST") != "1":
        raise SystemExit(self_launch(args))
    # the CPU baseline's binding (OMP_PLACES=cores OMP_PROC_BIND=close) is an
    # OPT-IN here (--cpu-bind): measured on the MI355X boxes (a 256-CPU host shared by
    # containers with a 16-CPU quota each) it binds every tenant's threads to the same
    # first cores -- the oracle's v2 path fell from 103 to 16 GFLOP/s and tune() from 1.2
    # to 3.0 s -- and it shrinks the main thread's affinity mask to one core.  It must
    # precede the first OpenMP runtime.
    if args.cpu_bind:
        os.environ.setdefault("OMP_PLACES", "cores")
        os.environ.setdefault("OMP_PROC_BIND", "close")
    # host-side setup (matrix generator, schedule build) is OpenMP code: give every
    # rank of this node an equal share of the CPUs (torchrun presets
    # OMP_NUM_THREADS=1 for N > 1, which would serialise it).  Set before any
    # OpenMP runtime is loaded.
    ncpus = host_cpus()
    share = max(1, ncpus // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))))
    os.environ["OMP_NUM_THREADS"] = str(share)
    os.environ["CFS_HOST_THREADS"] = str(share)
    import numpy as np
    import torch
    import cfs_spmv_amd as cfs
    from cfs_spmv_amd import _lib, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    N = args.gpus
    if world != N and not (N == 1 and world == 1):
        raise SystemExit(f"--gpus {N} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # CFS_DIST_BACKEND=gloo is a REHEARSAL mode for boxes with fewer GPUs than
    # ranks (ranks share devices, the exchange is staged through the host); the
    # driver's runs use the default: nccl (= RCCL over xGMI), one GPU per rank.
    backend = os.environ.get("CFS_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, ndev)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    lib = _lib.load()
    _lib.check(lib.cfs_hip_init(dev_index))
    dist = None
    # CFS_FORCE_DIST=1: rehearsal of the sharded code path (process group,
    # all-to-all, owner-side fold) with a single rank on a single GPU
    force_dist = N == 1 and os.environ.get("CFS_FORCE_DIST") == "1"
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if N > 1 or force_dist:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    np_dt = np.float64 if args.dtype == "f64" else np.float32
    t_dt = torch.float64 if args.dtype == "f64" else torch.float32
    # the real SuiteSparse file when the user provides it ($CFS_MTX_DIR/<name>.mtx,
    # through the C++ surface's reader); offline, SURVEY 8d's stand-in generator
    from cfs_spmv_amd import mmf
    real = mmf.find_real_matrix(args.matrix) if args.scale == 1.0 else None
    if real:
        A0 = mmf.load_mtx(real)
        if not A0["symmetric"] or A0["nrows"] != A0["ncols"]:
            raise SystemExit(f"{real}: not a symmetric matrix")
        n, rp, ci, va = A0["nrows"], A0["rowptr"], A0["colind"], A0["values"]
        rows_of = np.repeat(np.arange(n, dtype=np.int32), np.diff(rp))
        nnz_low = int(np.count_nonzero(ci < rows_of))
        del rows_of, A0
        data_kind, source = "file", f"{os.path.basename(real)} (SuiteSparse file)"
    else:
        n, rp, ci, va, nnz_low = synth.generate(args.matrix, args.scale)
        data_kind = "synthetic"
        source = f"{args.matrix}-like synthetic (SURVEY 8d generator), scale {args.scale}"
    va = va.astype(np_dt, copy=False)
    nnz_full = int(rp[-1])
    x_host = synth.make_x(n, 42, np_dt)
    opt = cfs.make_options(max_slots=args.max_slots, block_threads=args.block,
                           flags=args.flags | (32 if args.tuning == "none" else 0))

    if args.format == "csr":
        if N != 1:
            raise SystemExit("--format csr runs on one GPU")
        return bench_csr(args, cfs, lib, n, rp, ci, va, x_host, nnz_full, dev, t_dt, data_kind, source, ncpus)

    t0 = time.time()
    rs = None
    if N == 1 and not force_dist and args.shard_of > 1:
        rs = cfs.balanced_splits(n, rp, ci, args.shard_of)
        A = cfs.SymMatrix(n, rp, ci, va, options=opt, row_splits=rs, rank=args.shard_rank)
        sh = None
    elif N == 1 and not force_dist:
        A = cfs.SymMatrix(n, rp, ci, va, options=opt)
        sh = None
    else:
        from cfs_spmv_amd.dist import build_shard
        rs = cfs.balanced_splits(n, rp, ci, N)
        A, sh, args.exchange = build_shard(n, rp, ci, va, N, rank, rs, dev, options=opt,
                                           exchange=args.exchange,
                                           stage_via_host=(backend != "nccl"))
    preproc = time.time() - t0
    st = A.stats()
    rows = st["row_end"] - st["row_begin"]
    x = torch.from_numpy(x_host).to(dev)
    y = torch.full((rows,), float("nan"), dtype=t_dt, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    send = sh.send_buf if sh is not None else None
    # one GPU, or mirrored shards: an SpMV is the local launch sequence, no collective
    local_only = sh is None or args.exchange == "none"

    import ctypes as C

    def new_event():
        e = C.c_void_p()
        _lib.check(lib.cfs_hip_event_create(C.byref(e)))
        return e

    K = args.steps
    # the tile kernel is bracketed by HIP events on every n-th step of the timed
    # region only: an event between two kernels costs ~1-2 us of launch gap
    # (a short run -- the driver's --steps 20 -- still gets ten samples)
    every = max(1, min(args.event_every, K // 10 if K >= 10 else 1))
    sampled = [i for i in range(K) if i % every == 0]
    ev0 = {i: new_event() for i in sampled}
    ev1 = {i: new_event() for i in sampled}

    # the C ABI entry point with its arguments resolved once: at N = 8 a rank's
    # SpMV is ~20 us of GPU time and the host must enqueue faster than that
    phases_async = lib.cfs_hip_sym_spmv_phases_async
    hA, yp, xp = A._h, C.c_void_p(y.data_ptr()), C.c_void_p(x.data_ptr())
    sp, stp = C.c_void_p(send.data_ptr() if send is not None else 0), C.c_void_p(stream)

    def phases(ph):
        rc = phases_async(hA, yp, xp, sp, ph, stp)
        if rc != 0:
            _lib.check(rc)

    def step(i=None):
        # one SpMV = tile kernel (the roofline kernel) + halo fold (+ exchange).
        # In the timed region the tile kernel is bracketed by HIP events recorded
        # on the stream it is launched on.
        if i is not None and i in ev0:
            _lib.check(lib.cfs_hip_event_record(ev0[i], stream))
            phases(1)
            _lib.check(lib.cfs_hip_event_record(ev1[i], stream))
            phases(2 if local_only else 4)
        else:
            phases(3 if local_only else 1 | 4)
        if not local_only:
            sh.finish(y, x)   # exchange || local fold, then fold of what arrived

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up, untimed: bring clocks, power state and caches to the state of a long run before
    # the W warm-up steps.  Two transients were measured on MI355X: the clock ramp of an idle
    # GPU (a run with --warmup 5 alone read 5 % slow), and ~0.2 s of reduced clocks right after
    # the burst of tune()'s device kernels (radix sorts over 60 M keys): with 30 ms of set-up a
    # device-built schedule read 4-5 % slower than the bit-identical host-built one over 200
    # steps and equal (0.5 %) over 4 000 (profiles/r03_experiment_notes.md).  --settle-ms of
    # SpMVs (default 300), ending idle.
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.settle_ms * 1e-3:
        for _ in range(32):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    # ---- timed region: exactly K steps between barrier+synchronize pairs --------
    barrier()
    t0 = time.perf_counter()
    for i in range(K):
        step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64,
                            device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms_per_step = elapsed / K * 1e3

    # ---- roofline: average duration of the tile kernel over the timed region ----
    tile_total = 0.0
    ms = C.c_float()
    for i in sampled:
        _lib.check(lib.cfs_hip_event_elapsed_ms(ev0[i], ev1[i], C.byref(ms)))
        tile_total += ms.value
    tile_ms = tile_total / len(sampled)
    for e in list(ev0.values()) + list(ev1.values()):
        lib.cfs_hip_event_destroy(e)

    # cross-check with the kernel's own clock (developer timeline build of the same
    # launch: first workgroup start -> last workgroup end, 100 MHz wall clock).  A HIP
    # event bracket also holds the dispatch latency around the kernel (~2-3 us: 2 % of
    # a 110 us launch, 20 % of a 13 us one); rocprofv3's kernel duration is the
    # counterpart of this number.
    tl_ms = None
    try:
        tbuf = np.zeros(8 * 8192, dtype=np.uint64)
        ng = C.c_int()
        spans = []
        for _ in range(5):
            _lib.check(lib.cfs_hip_sym_debug_timeline(A._h, y.data_ptr(), x.data_ptr(),
                                                      tbuf.ctypes.data, tbuf.size, C.byref(ng)))
            tt = tbuf[:ng.value * 8].reshape(-1, 8).astype(np.int64)
            spans.append((tt[:, 3].max() - tt[:, 0].min()) / 100.0 * 1e-3)
        tl_ms = float(np.median(spans))
        phases(3 if local_only else 1 | 4)  # y of a complete step again for the self-check
        if not local_only:
            sh.finish(y, x)
        torch.cuda.synchronize()
    except Exception:
        tl_ms = None

    alg_bytes = st["bytes_algorithmic"]  # this rank's rows: nnz_low*(4+s) + rows*(4+3s)
    achieved = alg_bytes / (tile_ms * 1e-3) / 1e9
    # HBM bytes per launch from the PMC passes (profiles/hbm_traffic.json, written by
    # tools/profile_round.sh): only quoted when the entry was measured on the SAME
    # schedule that ran here (bytes the format streams, tiles, window, block) -- a
    # stale entry reads null, never a number of another kernel
    traffic = None
    traffic_source = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            with open(tpath) as f:
                tj = json.load(f)
            key = f"{args.matrix}:{args.scale}:{args.dtype}:{N}"
            if args.shard_of > 1:
                key += f":shard{args.shard_rank}of{args.shard_of}"
            # (tune() keeps the faster of two window shapes, and boxes differ: a workload
            # may hold one entry per schedule it was profiled on)
            ents = tj.get(key, [])
            for ent in (ents if isinstance(ents, list) else [ents]):
                if all(ent.get(k) == st[k] for k in ("bytes_streamed", "lds_bytes", "block_threads")):
                    traffic = ent.get("hbm_bytes_per_launch")
                    traffic_source = {
                        "file": "profiles/hbm_traffic.json", "key": key,
                        "matched_on": {k: st[k] for k in ("bytes_streamed", "lds_bytes",
                                                          "block_threads")},
                        "measured_by": ent.get("source", "tools/profile_round.sh: two rocprofv3 "
                                               "--pmc passes of this command (FETCH_SIZE, "
                                               "WRITE_SIZE), 2 x FETCH + WRITE per launch"),
                        "note": "a committed measurement of the SAME schedule on another run, "
                                "not taken inside this run"}
        except Exception:
            traffic = None

    # quick self-check of the timed result against the plain-CSR row sums (GPU CSR
    # kernel, not the oracle): guards against timing a broken kernel
    G = cfs.CsrMatrix(n, n, rp, ci, va)
    y2 = torch.empty(n, dtype=t_dt, device=dev)
    G.dense_vector_multiply(y2, x)
    torch.cuda.synchronize()
    y2 = y2[st["row_begin"]:st["row_end"]]
    scale = torch.maximum(y2.abs(), torch.tensor(1.0, dtype=t_dt, device=dev))
    err = float(((y - y2).abs() / scale).max().item()) if rows else 0.0
    G.close()
    if not err < (1e-9 if args.dtype == "f64" else 1e-3):
        raise SystemExit(f"rank {rank}: self-check failed: max scaled |y - y_csr| = {err}")

    # what one step multiplies: the whole matrix, or (--shard-of) the rows of one block
    shard_mode = args.shard_of > 1 and sh is None
    nnz_step = int(rp[st["row_end"]] - rp[st["row_begin"]]) if shard_mode else nnz_full
    alg_step = int(st["bytes_algorithmic"]) if shard_mode else \
        int(nnz_low * (4 + va.itemsize) + n * (4 + 3 * va.itemsize))
    out = None
    if rank == 0:
        out = {
            "metric": "fp64 symmetric SpMV GFLOP/s" if args.dtype == "f64"
                      else "fp32 symmetric SpMV GFLOP/s",
            "value": round(2.0 * nnz_step / (ms_per_step * 1e-3) / 1e9, 2),
            "unit": "GFLOP/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": args.dtype, "data": data_kind,
            "config": {
                "workload": f"{source}: n={n}, nnz_full={nnz_full}, "
                            f"nnz_low={nnz_low}, symmetric SSS SpMV y=Ax" +
                            (f"; ONLY the mirrored row block of rank {args.shard_rank} of "
                             f"{args.shard_of} (rows {st['row_begin']}..{st['row_end']}, "
                             f"{nnz_step} of the nonzeros): one rank's step of an "
                             f"{args.shard_of}-GPU run, timed alone" if shard_mode else ""),
                "format": "hyb" if st.get("far_entries", 0) else "sss",
                "far_entries": st.get("far_entries", 0), "halo_slots": st["halo_slots"],
                "sharding": f"1d-row-blocks x{N}",
                "exchange": (None if sh is None else
                             "none: off-block entries mirrored on both ranks, no collective "
                             "per SpMV" if args.exchange == "none" else args.exchange),
                "mirror_entries_rank0": st.get("mirror_entries", 0),
                "algorithmic_bytes_per_spmv": alg_step,
                "effective_GBps_whole_step": round(alg_step / (ms_per_step * 1e-3) / 1e9, 1),
                "tiles": st["ntiles"], "lds_bytes": st["lds_bytes"],
                "block_threads": st["block_threads"], "preproc_s": round(preproc, 2),
                "tuning": args.tuning, "flags": args.flags,
            },
            "roofline": {
                "bound": "hbm", "kernel": "cfs_sym_tile_kernel",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "kernel_ms": round(tile_ms, 5), "kernel_samples": len(sampled),
                "kernel_ms_inkernel_clock": round(tl_ms, 5) if tl_ms else None,
                "frac_inkernel_clock": (round(alg_bytes / (tl_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                        if tl_ms else None),
                "algorithmic_bytes_per_launch": int(alg_bytes),
                # `achieved` is ALGORITHMIC bytes / kernel time (the device format streams
                # fewer: 16-bit de-duplicated slots); what really crossed the HBM interface:
                "hbm_GBps_from_traffic": (round(traffic / (tile_ms * 1e-3) / 1e9, 1)
                                          if traffic else None),
                "traffic_source": traffic_source,
                # the same bytes against the peak and against what the guide calls achievable
                # on this part (~6.3 TB/s of streaming reads): `frac` above is the contract's
                # EFFECTIVE figure (algorithmic bytes), these two are the wire's
                "hbm_frac_of_peak_from_traffic": (round(traffic / (tile_ms * 1e-3) / 1e9 /
                                                        HBM_PEAK_GBS, 4) if traffic else None),
                "hbm_frac_of_achievable": (round(traffic / (tile_ms * 1e-3) / 1e9 /
                                                 HBM_ACHIEVABLE_GBS, 4) if traffic else None),
                "achievable_GBps": HBM_ACHIEVABLE_GBS,
                "bytes_streamed_by_format": int(st["bytes_streamed"]),
            },
        }

    if rank == 0 and N == 1 and not args.no_cpu_baseline and not shard_mode:
        try:
            out["cpu_baseline"] = cpu_baseline(n, rp, ci, va, x_host, nnz_full, args.cpu_loops,
                                               ncpus, args.cpu_bind)
        except Exception as e:  # the baseline is reported, never required
            out["cpu_baseline"] = {"value": None, "unit": "GFLOP/s", "cores": 0,
                                   "kind": "port", "sample": f"failed: {e}"}
    # ---- N > 1: the other forms of the off-block exchange, same run, same protocol.
    # Measured LAST, under a watchdog: whatever happens to them (a collective that
    # hangs on some stack), the ONE line of the contract is printed -- and a hang ends
    # the run with a non-zero exit code (run_guarded).
    if sh is not None and not args.no_exchange_forms:
        def on_timeout():
            if rank == 0:
                out["exchange_forms"] = {args.exchange: round(ms_per_step, 5),
                                         "others": f"timed out after {wd_s:.0f} s"}
                print(json.dumps(out), flush=True)
            print(f"[bench] rank {rank}: exchange-form timings hung; exiting {WATCHDOG_EXIT}",
                  file=sys.stderr, flush=True)
        wd_s = float(os.environ.get("CFS_BENCH_WATCHDOG_S", "120"))
        forms = {args.exchange: round(ms_per_step, 5)}
        forms.update(run_guarded(
            lambda: time_other_forms(args, cfs, A, sh, n, rp, ci, va, N, rank, rs, dev, opt,
                                     backend, x, y, stream, barrier, dist),
            wd_s, on_timeout))
        if rank == 0:
            out["exchange_forms"] = forms
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
