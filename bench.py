#!/usr/bin/env python3
"""bench.py -- leapfrog steps/sec of the fused WALNUTS/NUTS engine on the reference's headline
workload (BASELINE.json: 100-D ill-conditioned Gaussian, kappa=1e4, fixedLeapFrog, M=10,
H0=0.5, 100 iterations; SURVEY.md 8d cfg 2), with roofline and CPU-baseline legs.

  python bench.py --gpus N --steps K --warmup W     (N > 1: starts one child process per GPU itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one wn_run launch: `--iters` sampler iterations for `--chains` chains per GPU.
Chains shard over GPUs with no data-path collective (weak scaling: per-GPU chains fixed); the
only collective is the final gather of draws (reported separately as gather_ms) and the
step-count/time reductions.

The default invocation also runs BASELINE configs 3, 4 and 5 (all three runs of mainSW.py:41-80) at SURVEY 8d's
sizes, three timed launches each, outside the headline's timed region: `extras.other_configs` carries min / median /
max, a roofline, a CPU baseline and ESS/s for every one of them.  `--workload funnel | dense1000 | stockwatson |
walnuts10` prints the same object as a full contract line.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 20251121            # SURVEY.md 8d master seed
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X FP64 matrix = FP64 vector (v_mfma_f64_16x16x4_f64: 256 CU x 4 SIMD x 32 flop/clk
                              # x 2.4 GHz); measured on the box by tools/mfma_f64_rate.hip (profiles/r02_mfma_f64_rate.txt)
HBM_PEAK_GBS = 8000.0
FUNNEL_TICK_FLOOR_US = 0.68   # tools/wave_tick_floor.hip: the bare arithmetic of a funnel macro step on one wave


def make_dense_problem(D, kappa, C, seed, chain_offset=0):
    """Sigma = Q diag(lam) Q', lam = logspace(0, log10 kappa, D); precision A = Q diag(1/lam) Q'
    symmetrised; q0 ~ N(0, Sigma) per chain, keyed by 256-chain blocks of the GLOBAL chain id so
    that a chain's start is the same however chains are sharded."""
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.normal(size=(D, D)))
    lam = np.logspace(0.0, np.log10(kappa), D)
    A = (Q * (1.0 / lam)) @ Q.T
    A = 0.5 * (A + A.T)
    S = Q * np.sqrt(lam)
    q0 = np.empty((C, D))
    c = 0
    while c < C:
        g = chain_offset + c
        blk, off = divmod(g, 256)
        z = np.random.default_rng([seed, 7, blk]).normal(size=(256, D))
        n = min(256 - off, C - c)
        q0[c:c + n] = z[off:off + n] @ S.T
        c += n
    return A, q0


# ---------------------------------------------------------------------------------------------
# workloads: SURVEY 8d's configurations as data (the engine arguments, the oracle's arguments, the metric)
# ---------------------------------------------------------------------------------------------
def _sw_data():
    sw = np.load(os.path.join(ROOT, "tests", "golden", "stockwatson_data.npz"))
    T = int(sw["T"][0])
    return T, np.concatenate([[float(T)], sw["y"]])


def workload_spec(name, args=None, rank=0, sw_run="D"):
    """name in gauss100 | funnel | dense1000 | stockwatson.  Returns a dict: D, C, iters, engine kwargs, target, tpar, q0,
    unit, labels, roofline model."""
    from walnuts_amd import _lib as L
    a = args
    if name in ("gauss100", "dense1000"):
        big = name == "dense1000"
        D = 1000 if big else (a.dim if a else 100)
        kappa = 100.0 if big else (a.kappa if a else 1e4)
        C = (a.chains if a and a.chains else (1024 if big else 4096))
        iters = (a.iters if a and a.iters else (50 if big else 100))
        M, H0 = (a.M if a else 10), (a.H0 if a else 0.5)
        A, q0 = make_dense_problem(D, kappa, C, SEED, chain_offset=rank * C)
        ident = bool(a and a.gather == "identity")
        return dict(name=name, D=D, C=C, iters=iters, target=L.T_DENSEGAUSS, tpar=A, q0=q0, unit="leapfrog steps/s",
                    eng=dict(integrator=L.I_FIXED, H0=H0, stepSizeRandScale=0.2, M=M,
                             generated=L.G_IDENTITY if ident else L.G_Q0_SUMSQ), ident=ident,
                    oracle=dict(integrator=0, H0=H0, M=M),
                    metric="leapfrog steps/sec across chains, %d-D Gaussian" % D,
                    workload=f"{D}-D dense Gaussian kappa={kappa:g}, {C} chains/GPU, fixedLeapFrog, M={M}, H0={H0}, {iters} iterations/launch "
                             f"(SURVEY 8d cfg {4 if big else 2})",
                    bound="mfma", per_unit=2.0 * D * D + 12.0 * D, cfg=4 if big else 2, pyref="cfg2_dense100_fixedLeapFrog" if not big else None)
    if name == "funnel":
        C = a.chains if a and a.chains else 4096
        iters = a.iters if a and a.iters else 300
        rng = np.random.default_rng([SEED, 3, rank])
        q0 = np.zeros((C, 11)); q0[:, 0] = 3.0 * rng.normal(size=C)
        q0[:, 1:] = np.exp(0.5 * q0[:, :1]) * rng.normal(size=(C, 10))
        return dict(name=name, D=11, C=C, iters=iters, target=L.T_FUNNEL10, tpar=None, q0=q0, unit="gradient evaluations/s",
                    eng=dict(integrator=L.I_ADAPT_R2P, H0=0.3, delta0=0.3, M=12, generated=L.G_FIRSTK, gen_k=2), ident=False,
                    oracle=dict(integrator=L.I_ADAPT_R2P, H0=0.3, delta0=0.3, M=12),
                    metric="gradient evaluations/sec across chains, funnel10 (adaptLeapFrogR2P)",
                    workload=f"funnel10 D=11, {C} chains/GPU, adaptLeapFrogR2P, M=12, H0=0.3, delta0=0.3, {iters} iterations/launch (SURVEY 8d cfg 3)",
                    bound="hbm", per_unit=528.0, cfg=3, traffic_key="funnel", floor_us=FUNNEL_TICK_FLOOR_US, pyref="cfg3_funnel10_R2P")
    if name == "stockwatson":
        T, tpar = _sw_data()
        D = 3 * T
        C = a.chains if a and a.chains else 256
        iters = a.iters if a and a.iters else 20
        ig, H0, minC, nm = {"D": (L.I_ADAPT_D, 0.1, 3, "adaptLeapFrogD minC=3"), "fixed": (L.I_FIXED, 0.002, 0, "fixedLeapFrog"),
                            "R2P": (L.I_ADAPT_R2P, 0.1, 3, "adaptLeapFrogR2P minC=3")}[sw_run]
        return dict(name=f"stockwatson_{sw_run}", D=D, C=C, iters=iters, target=L.T_STOCKWATSON, tpar=tpar, q0=np.zeros((C, D)),
                    unit="gradient evaluations/s",
                    eng=dict(integrator=ig, H0=H0, delta0=0.3, M=14, minC=minC, generated=L.G_FIRSTK, gen_k=2), ident=False,
                    oracle=dict(integrator=ig, H0=H0, delta0=0.3, M=14, minC=minC),
                    metric=f"gradient evaluations/sec across chains, StockWatson ({nm.split()[0]})",
                    workload=f"StockWatson D={D}, {C} chains/GPU, {nm}, M=14, H0={H0}, delta0=0.3, q0=0, {iters} iterations/launch "
                             f"(SURVEY 8d cfg 5, mainSW.py:41-80 run '{sw_run}')",
                    bound="hbm", per_unit=48.0 * D, cfg=5, traffic_key="stockwatson" if sw_run == "D" else None, pyref=None)
    raise ValueError(name)


def _with_route(spec, a):
    """--route / --split-parts / --split-ks of the command line -> the engine's keyword arguments (A/B runs only)"""
    if a is not None and (getattr(a, "route", "") or getattr(a, "split_parts", 0) or getattr(a, "split_ks", 0)):
        spec["eng_extra"] = dict(route=[r for r in a.route.split(",") if r], split_parts=a.split_parts, split_ks=a.split_ks)
    return spec


# ---------------------------------------------------------------------------------------------
# CPU baseline: the oracle (validated C restatement of the reference's path) on the host cores
# ---------------------------------------------------------------------------------------------
def _cpu_worker(args):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    target, tpar, q0, chain, iters, okw = args
    t0 = time.perf_counter()
    o = ol.walnuts(target, q0, tpar=None if tpar is None else np.asarray(tpar).ravel(), numIter=iters, seed=SEED, chain=chain, **okw)
    dt = time.perf_counter() - t0
    d = o["diagnostics"]
    return float(d[:, 6].sum() + d[:, 7].sum()), dt


def python_reference_cpu(key):
    """The UNMODIFIED Python reference timed in the build container (tools/ref_cpu_baseline.py; it never travels to the
    GPU box): BASELINE.md section 3's "measured elsewhere" figure, quoted with where it was measured."""
    if not key:
        return None
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", "r02_python_reference_cpu.json")))
        r = j[key]
        return {"unit": r["unit"], "per_core": r["one_core"]["per_core"], "all_cores": r["all_cores"]["value"],
                "cores": r["all_cores"]["cores"], "measured": j["where"], "code": j["code"],
                "note": "not measured in this run: the Python reference does not travel to the GPU box"}
    except Exception:
        return None


def cpu_baselines(specs, budgets):
    """Time the CPU oracle on ALL host cores -- the same worker-process count for every configuration, so the GPU/CPU ratios
    of different configurations compare -- one chain per process at a time, on a bounded sample of each workload (`budgets`
    seconds each).  One pool serves every configuration.  Returns {spec name: record}; every record states the all-core rate
    (`value`, `cores`) and the per-core rate beside it."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    ol.build()
    cores = os.cpu_count() or 1
    out = {}
    with mp.get_context("fork").Pool(cores) as pool:
        sp0 = specs[0]
        pool.map(_cpu_worker, [(sp0["target"], sp0["tpar"], sp0["q0"][c % sp0["q0"].shape[0]], c % sp0["q0"].shape[0], 1,
                                dict(sp0["oracle"], M=min(3, sp0["oracle"].get("M", 10)))) for c in range(cores)])  # start the workers
        for spec, budget_s in zip(specs, budgets):
            try:
                q0, iters = spec["q0"], spec["iters"]
                okw = dict(spec["oracle"])
                # size the sample from a PILOT under full load (one chain per worker, a few iterations): a single chain timed
                # alone on an idle host ran up to ten times faster than 256 of them side by side, and the "12 s" sample took 84
                nproc = min(cores, q0.shape[0])
                s1, t1 = _cpu_worker((spec["target"], spec["tpar"], q0[0], 0, 1, okw))
                it0 = int(max(1, min(iters, 0.25 / max(t1, 1e-5))))
                t0 = time.perf_counter()
                res = pool.map(_cpu_worker, [(spec["target"], spec["tpar"], q0[c], c, it0, okw) for c in range(nproc)], chunksize=1)
                wall = time.perf_counter() - t0
                n_it, nchain = it0, nproc
                if wall < 0.5 * budget_s:
                    f = budget_s / max(wall, 1e-3)
                    n_it = int(max(it0, min(iters, it0 * f)))
                    f /= n_it / it0
                    k = int(max(1, min(f, (q0.shape[0] + cores - 1) // cores)))
                    nchain = min(q0.shape[0], nproc * k)
                    t0 = time.perf_counter()
                    res = pool.map(_cpu_worker, [(spec["target"], spec["tpar"], q0[c], c, n_it, okw) for c in range(nchain)], chunksize=1)
                    wall = time.perf_counter() - t0
                units = sum(r[0] for r in res)
                busy = sum(r[1] for r in res)  # summed in-worker seconds: units / busy = the rate of ONE core working alone on its chain
                C_all = q0.shape[0]
                out[spec["name"]] = {
                    "value": units / wall, "unit": spec["unit"], "cores": nproc, "kind": "port",
                    "per_core": units / busy, "per_core_note": "units / summed in-worker seconds (one chain at a time per worker process)",
                    "sample": f"the first {nchain} of the {C_all} chains ({100.0 * nchain / C_all:.2f} %) x {n_it} of the {iters} iterations "
                              f"({units:.3g} units, {wall:.1f} s), {nproc} worker processes = every host thread, one chain at a time each "
                              f"(oracle/libwalnuts_oracle.so, the C restatement pinned to the Python reference)",
                    "python_reference": python_reference_cpu(spec.get("pyref"))}
            except Exception as ex:
                out[spec["name"]] = {"error": repr(ex)}
    return out


def cpu_baseline(spec, budget_s=10.0):
    return cpu_baselines([spec], [budget_s])[spec["name"]]


# ---------------------------------------------------------------------------------------------
# measured HBM traffic (separate rocprofv3 --pmc passes, tools/prof_traffic.py) of the kernels
# ---------------------------------------------------------------------------------------------
def _kernel_base(name):
    """'void wn::wn_fast_kernel<25, 0, 3>' and 'wn_fast_kernel<KS=25,dense>' -> 'wn_fast_kernel'"""
    import re
    m = re.search(r"(wn_\w+?_kernel)", name or "")
    return m.group(1) if m else (name or "")


def headline_traffic(kernel_name, D, C, iters):
    """bytes per launch of the headline kernel from the newest profiles/rNN_traffic.json that is about this kernel (and,
    where the file says, this workload); None when there is none."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            tj = json.load(open(f))
            if _kernel_base(tj.get("kernel", "")) != _kernel_base(kernel_name):
                continue
            w = tj.get("workload")
            if w and (w.get("dim"), w.get("chains"), w.get("iters")) != (D, C, iters):
                continue
            return float(tj["bytes_per_launch"]), os.path.basename(f)
        except Exception:
            continue
    return None, None


def side_traffic(key, field):
    import glob
    if not key:
        return None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic_side.json")), reverse=True):
        try:
            rec = json.load(open(f))
            if key in rec and field in rec[key]:
                return float(rec[key][field])
        except Exception:
            continue
    return None


# ---------------------------------------------------------------------------------------------
def _self_launch(args):
    """`python bench.py --gpus N` without torchrun: this process has not touched torch or HIP yet, so it
    may start children.  Starts one fresh child per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, the
    same command line) and relays rank 0's JSON line.  Every child is polled: the first one that
    exits non-zero (or the time limit) ends the others -- a crashed rank would otherwise leave its siblings in
    init_process_group / all_reduce until the collective's own timeout -- and the parent exits non-zero.
    The N = 1 reference of `weak_scaling_eff` is `--n1-value`, or (`--n1-ref`) one more single-GPU child of the same
    workload started AFTER the N ranks have finished: the N-rank line never waits for it, and if it fails or runs
    out of time the line is printed without the two fields."""
    import socket
    import subprocess
    n = args.gpus
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    argv = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    t_begin = time.monotonic()
    procs = []
    for rank in range(n):
        env = dict(base, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), WN_BENCH_CHILD="1")
        procs.append(subprocess.Popen(argv, env=env, stdout=subprocess.PIPE if rank == 0 else sys.stderr, text=True))
    deadline = t_begin + args.child_timeout
    bad = None
    while True:
        rcs = [p_.poll() for p_ in procs]
        failed = [(k, rc) for k, rc in enumerate(rcs) if rc not in (None, 0)]
        if failed:
            bad = failed[0]
            break
        if all(rc == 0 for rc in rcs):
            break
        if time.monotonic() > deadline:
            bad = (-1, 124)
            break
        time.sleep(0.05)
    if bad is not None:
        for p_ in procs:  # fresh children only ever: nothing that touched a GPU is re-executed, the survivors are just ended
            if p_.poll() is None:
                p_.terminate()
        t_end = time.monotonic() + 5.0
        for p_ in procs:
            try:
                p_.wait(timeout=max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p_.kill()
        print(f"bench.py: rank {bad[0]} exited with code {bad[1]}; the other ranks were ended" if bad[0] >= 0
              else f"bench.py: the ranks did not finish within {args.child_timeout:.0f} s and were ended", file=sys.stderr)
        sys.exit(abs(bad[1]) or 1)
    out0 = procs[0].stdout.read()
    n1 = args.n1_value
    if n1 is None and args.n1_ref:
        left = deadline - time.monotonic()
        try:
            r = subprocess.run(argv + ["--gpus", "1", "--no-extras", "--no-cpu-baseline"], env=dict(base, WN_BENCH_CHILD="1"),
                               stdout=subprocess.PIPE, text=True, timeout=max(1.0, left))
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode == 0 and lines:
                n1 = float(json.loads(lines[-1])["value"])
            else:
                print(f"bench.py: the N=1 reference run failed (rc {r.returncode}); the line goes out without it", file=sys.stderr)
        except subprocess.TimeoutExpired:
            print("bench.py: the N=1 reference run timed out; the line goes out without it", file=sys.stderr)
    if n1 is not None:
        keep = []
        for l in out0.splitlines():
            if l.startswith("{"):
                try:
                    o = json.loads(l)
                    o["n1_value_same_invocation"] = n1
                    if isinstance(o.get("value"), (int, float)) and n1 > 0:
                        o["weak_scaling_eff"] = o["value"] / (n * n1)
                    l = json.dumps(o)
                except Exception:
                    pass
            keep.append(l)
        out0 = "\n".join(keep) + "\n"
    sys.stdout.write(out0)
    sys.stdout.flush()
    sys.exit(0)


def _dist_setup(force=False):
    """One process per GPU (torch.distributed.run env); returns rank, world, device, collective device, dist.
    `force`: initialise the process group (RCCL) also when WORLD_SIZE is 1 -- the one-GPU box then runs ncclCommInitRank and
    every collective of the N-rank path on device tensors."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("WN_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if ndev < 1:
        sys.exit("bench.py needs a ROCm GPU (the product has no CPU path)")
    if world > 1 or force:
        if backend == "nccl" and local_rank >= ndev:
            print(f"bench.py: LOCAL_RANK {local_rank} but only {ndev} device(s) are visible", file=sys.stderr)
            sys.exit(2)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if "MASTER_PORT" not in os.environ:  # plain `python bench.py --force-collectives`: a one-rank group of its own
            import socket
            with socket.socket() as s_:
                s_.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank % ndev)
        fail = os.environ.get("WN_BENCH_FAIL_RANK")  # test hook: this rank dies before the rendezvous (tests/test_host_logic.py)
        if fail is not None and int(fail) == rank:
            sys.exit(3)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank % ndev))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())
    return rank, world, dev, (dev if backend == "nccl" else torch.device("cpu")), dist


def _scaling_fields(dist, world, cdev, value, kernel_ms_avg, units_rank, coll=False):
    """Per-rank kernel time and work (one small all_gather outside the timed region) and, when bench.py
    launched the ranks itself, the weak-scaling efficiency against the single-GPU run of the same invocation."""
    import torch
    if world == 1 and not coll:
        return {}
    mine = torch.tensor([kernel_ms_avg, units_rank], dtype=torch.float64, device=cdev)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    out = {"per_rank_kernel_ms": [float(a[0].item()) for a in allr],
           "per_rank_units": [float(a[1].item()) for a in allr]}
    return out


# ---------------------------------------------------------------------------------------------
# the measurement core: W untimed + K timed launches of one workload on this rank's GPU
# ---------------------------------------------------------------------------------------------
def make_engine(spec, dev, rank):
    import walnuts_amd as wa
    e = wa.Engine(spec["D"], spec["C"], chain_offset=rank * spec["C"], device=dev.index, **spec["eng"], **spec.get("eng_extra", {}))
    return e.set_target(spec["target"], spec["tpar"])


def timed_launches(eng, spec, dev, steps, warmup, barrier):
    """-> dict(elapsed, units (this rank), kernel_ms [...], per_launch_units [...], busiest [...], ticks [...], draws)"""
    import torch
    q0d = torch.as_tensor(spec["q0"], device=dev)
    for w in range(warmup):
        _, dw = eng.run(q0d, SEED + 1000 + w, spec["iters"])
        float((dw[:, :, 6] + dw[:, :, 7]).sum(1).max().item())  # also loads torch's reduction kernels outside the timed region
    if not warmup:  # (first use of a torch kernel costs ~90 ms of code loading)
        float(torch.zeros((2, 2, 8), dtype=torch.float64, device=dev)[:, :, 6].sum(1).max().item())
    barrier()
    t0 = time.perf_counter()
    tot = torch.zeros((), dtype=torch.float64, device=dev)
    per = []; kms = []; ticks = []
    last = None
    for k in range(steps):
        s, d = eng.run(q0d, SEED + k, spec["iters"])
        pc = (d[:, :, 6] + d[:, :, 7]).sum(1)
        per.append(pc)
        tot += pc.sum()
        kms.append(eng.last_run_ms())  # HIP events on the launch stream (synchronises)
        ticks.append(eng.last_split_ticks())
        last = s
    barrier()
    el = time.perf_counter() - t0
    # tail: a T wave of the headline kernel owns four chains and lives until the longest of them is done, the launch until the last wave is;
    # mean over the 4-chain groups of (the group's longest chain) / the longest of all = the fraction of the launch the average wave is alive
    # for (PMC: SQ_WAVE_CYCLES x 4 / waves against GRBM_GUI_ACTIVE / 8 gave 0.946 in round 5) -- what separates the tick's cost from the
    # end-of-launch tail in `value`
    tf = []
    for p_ in per:
        g_ = p_[: (p_.numel() // 4) * 4].reshape(-1, 4).max(1).values
        tf.append(float((g_.mean() / g_.max()).item()) if g_.numel() else 1.0)
    return dict(elapsed=el, units=float(tot.item()), kernel_ms=kms, per_launch_units=[float(p_.sum().item()) for p_ in per],
                busiest=[float(p_.max().item()) for p_ in per], ticks=ticks, draws=last, tail_frac=float(np.mean(tf)) if tf else None)


def roofline_of(spec, units_per_launch, avg_ms, kernel_name, steps_info=None):
    """SURVEY 8d's roofline record of one workload (achieved = algorithmic bytes or flop per launch / launch duration)."""
    if spec["bound"] == "mfma":
        ach = units_per_launch * spec["per_unit"] / (avg_ms * 1e-3) / 1e12
        traffic, src = headline_traffic(kernel_name, spec["D"], spec["C"], spec["iters"])
        if traffic is None and spec["name"] == "dense1000":
            bps = side_traffic("dense1000", "bytes_per_leapfrog_step")
            traffic = None if bps is None else bps * units_per_launch
            src = "profiles/r*_traffic_side.json (bytes per leapfrog step x steps)" if traffic is not None else None
        return {"bound": "mfma", "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_MFMA_PEAK_TFLOPS,
                "traffic": traffic, "traffic_source": src, "flop_per_leapfrog_step": spec["per_unit"]}
    ach = units_per_launch * spec["per_unit"] / (avg_ms * 1e-3) / 1e9
    bpe = side_traffic(spec.get("traffic_key"), "bytes_per_gradient_evaluation")
    return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            # `achieved` is SURVEY 8d's NOMINAL streaming figure (48*D bytes per evaluation); the fused kernel keeps its state
            # on chip, so beside it: the measured HBM rate (PMC bytes), the FP64-vector fraction at 14*D flop per evaluation
            #  and -- what actually bounds these launches -- the serial time per evaluation of the busiest chain
            "measured_hbm_GBps": None if bpe is None else bpe * units_per_launch / (avg_ms * 1e-3) / 1e9,
            "fp64_vector_frac": units_per_launch * 14.0 * spec["D"] / (avg_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
            "traffic": None if bpe is None else bpe * units_per_launch,
            "bytes_per_gradient_evaluation": spec["per_unit"]}


def ess_of(spec, draws, per_launch_s):
    """ESS/sec of gen(q) = [q[0], sum(q*q)] (mainGaussESS.py:29-30; for the first-k kinds: of q[0] and q[1]) over all chains
    of one launch; rank-normalised split bulk ESS (walnuts_amd/ess.py), host side, outside the timing."""
    try:
        from walnuts_amd.ess import ess
        dr = draws.cpu().numpy()
        if spec.get("ident"):
            dr = np.stack([dr[:, :, 0], (dr * dr).sum(2)], axis=2)
        e0, e1 = ess(dr[:, 1:, 0]), ess(dr[:, 1:, 1])
        names = ("q0", "sum_q2") if spec["bound"] == "mfma" else ("q0", "q1")
        return {names[0]: e0 / per_launch_s, names[1]: e1 / per_launch_s, "ess_" + names[0]: e0, "ess_" + names[1]: e1,
                "draws": int(dr.shape[0] * (dr.shape[1] - 1)), "method": "rank-normalised split bulk ESS over all chains of one launch"}
    except Exception as ex:  # the throughput line must not depend on the ESS estimator
        return {"error": repr(ex)}


def side_record(spec, m, kernel_name, cpu):
    """min / median / max over the timed launches of one side configuration + its roofline, CPU baseline, ESS/s"""
    vals = [u / (ms * 1e-3) for u, ms in zip(m["per_launch_units"], m["kernel_ms"])]
    avg_ms = float(np.mean(m["kernel_ms"]))
    upl = float(np.mean(m["per_launch_units"]))
    rec = {"value": float(np.median(vals)), "min": float(np.min(vals)), "max": float(np.max(vals)), "unit": spec["unit"],
           "launches": len(vals), "launch_values": vals, "kernel_ms": m["kernel_ms"], "workload": spec["workload"], "kernel": kernel_name,
           "roofline": roofline_of(spec, upl, avg_ms, kernel_name), "ess_per_sec": ess_of(spec, m["draws"], avg_ms * 1e-3)}
    if spec["bound"] == "hbm":
        bus = float(np.mean(m["busiest"]))
        rec.update({"busiest_chain_evals": bus, "chain_occupancy": upl / spec["C"] / bus, "tick_us": avg_ms * 1e3 / bus})
        if spec.get("floor_us"):
            rec["tick_over_floor"] = rec["tick_us"] / spec["floor_us"]
    if any(m["ticks"]):
        tk = float(np.mean(m["ticks"]))
        rec.update({"split_ticks_per_launch": tk, "tick_us": avg_ms * 1e3 / tk, "kernel_launches_per_s": 2.0 * tk / (avg_ms * 1e-3),
                    "fp64_mfma_frac": rec["roofline"]["frac"]})
    if cpu is not None:
        rec["cpu_baseline"] = cpu
    return rec


# ---------------------------------------------------------------------------------------------
# SURVEY 8e on ONE GPU: the R shards of an R-rank job, one after another
# ---------------------------------------------------------------------------------------------
def emulate_ranks(name, R, dev, steps, warmup, args=None, sw_run="D"):
    """The R shards of an R-rank job of workload `name` (rank r: chains r*C .. (r+1)*C - 1 of the global chain numbering,
    i.e. `chain_offset = r*C`, its own starting points, the same launch seeds as the real job), run one after another on
    this one device.  Chains never interact (WALNUTS.py:701-712: even the adaptation is per chain) and the ranks of the real
    job exchange nothing inside the timed region, so rank r of an R-GPU job executes exactly shard r's launches: the job's
    time is the slowest shard's, and the only loss against N = 1 is the imbalance between shards.
    predicted_weak_scaling_eff = (sum of the shards' units / (R * slowest shard's time)) / (shard 0's units / its time):
    the figure the driver computes from its own N = 1 and N = R runs, here from R sequential runs on one GPU (it leaves
    out what only R GPUs can show: the one barrier + gather at the end, clock / thermal differences between devices)."""
    import torch
    shards = []
    for r in range(R):
        sp = _with_route(workload_spec(name, args, r, sw_run), args)
        e = make_engine(sp, dev, r)
        m = timed_launches(e, sp, dev, steps, warmup, lambda: torch.cuda.synchronize(dev))
        e.close()
        shards.append({"rank": r, "chain_offset": r * sp["C"], "kernel_ms": float(np.sum(m["kernel_ms"])), "units": m["units"],
                       "busiest_chain_units": float(np.max(m["busiest"]))})
    return {"ranks": R, "launches_per_shard": steps, "iters_per_launch": sp["iters"], "chains_per_shard": sp["C"],
            **scaling_prediction([s_["kernel_ms"] for s_ in shards], [s_["units"] for s_ in shards]),
            "busiest_chain_units": [s_["busiest_chain_units"] for s_ in shards], "workload": sp["workload"],
            "what": "R shards of an R-rank job run sequentially on one GPU; eff = sum(units) / (R * max shard time) / (shard 0 rate)"}


def scaling_prediction(shard_ms, shard_units):
    """What an R-rank weak-scaling job would report, from its R shards' times and units measured one after another on one device (pure
    arithmetic: tests/test_host_logic.py): the job's time is the slowest shard's, its value the units of all shards / that time, and its
    efficiency that value / (R x the N = 1 value), the N = 1 job being shard 0 alone."""
    t = np.asarray(shard_ms, dtype=np.float64); u = np.asarray(shard_units, dtype=np.float64)
    R = len(t)
    n1 = u[0] / t[0]
    return {"per_shard_kernel_ms": [float(x) for x in t], "per_shard_units": [float(x) for x in u],
            "per_shard_rate": [float(x) for x in u / t * 1e3],
            "n1_value": float(n1 * 1e3), "predicted_value": float(u.sum() / t.max() * 1e3),
            "predicted_weak_scaling_eff": float(u.sum() / (R * t.max()) / n1),
            # the same against the MEAN shard's rate instead of shard 0's (takes out whether shard 0 happens to be a quick or a slow one)
            "predicted_weak_scaling_eff_vs_mean_shard": float(u.sum() / (R * t.max()) / np.mean(u / t)),
            "time_max_over_mean": float(t.max() / t.mean()), "units_max_over_mean": float(u.max() / u.mean())}


EMULATED = (("cfg2", "gauss100", "D"), ("cfg3", "funnel", "D"), ("cfg4", "dense1000", "D"), ("cfg5D", "stockwatson", "D"))


def summary_line(out):
    """Every configuration's number (and its roofline fraction or serial time per unit) in one stderr line of at most 600 bytes, so that
    a log tail carries all of them (measurement hygiene)."""
    g = lambda d, *ks: (g(d.get(ks[0], {}), *ks[1:]) if len(ks) > 1 else d.get(ks[0])) if isinstance(d, dict) else None
    f = lambda v, spec="%.3g": (spec % v) if isinstance(v, (int, float)) else "n/a"
    ex = out.get("extras", {})
    oc = ex.get("other_configs", {})
    parts = ["cfg2@%d=%s(%s,tail_frac=%s)" % (out["config"].get("chains_per_gpu", 0), f(out["value"], "%.4g"), f(g(out, "roofline", "frac")), f(out.get("tail_frac")))]
    parts.append("cfg2@1024=%s(%s,med3)" % (f(g(ex, "dense_1024_chains", "value")), f(g(ex, "dense_1024_chains", "roofline", "frac"))))
    parts.append("diag=%s" % f(g(ex, "diag_variant", "value")))
    parts.append("cfg2adaptD/R2P=%s/%s" % (f(g(ex, "dense_adaptive", "adaptLeapFrogD", "value")), f(g(ex, "dense_adaptive", "adaptLeapFrogR2P", "value"))))
    parts.append("stdR2P=%s" % f(g(ex, "std_adaptive_R2P", "value")))
    c3 = oc.get("cfg3_funnel", {})
    parts.append("cfg3=%s/%s/%s,tick_us=%s" % (f(c3.get("min")), f(c3.get("value")), f(c3.get("max")), f(c3.get("tick_us"))))
    c4 = oc.get("cfg4_dense1000", {})
    parts.append("cfg4=%s(%s),tick_us=%s" % (f(c4.get("value")), f(g(c4, "roofline", "frac")), f(c4.get("tick_us"), "%.3g")))
    for k in ("D", "fixed", "R2P"):
        c5 = oc.get("cfg5_stockwatson_" + k, {})
        parts.append("cfg5%s=%s,tick_us=%s" % (k, f(c5.get("value")), f(c5.get("tick_us"))))
    parts.append("w10dev=%s" % f(g(ex, "walnuts10_device", "value")))
    parts.append("w_batch=%s" % f(g(ex, "walnuts_batch", "value")))
    em = ex.get("emulated_8_ranks", {})
    parts.append("eff8emu=" + "/".join(f(g(em, k, "predicted_weak_scaling_eff"), "%.3f") for k, _, _ in EMULATED))
    return ("bench.py summary (units/s (roofline frac)): " + " ".join(parts))[:600]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--chains", type=int, default=0, help="chains per GPU (default: the workload's SURVEY 8d size)")
    ap.add_argument("--iters", type=int, default=0, help="sampler iterations per launch (default: SURVEY 8d's)")
    ap.add_argument("--dim", type=int, default=100)
    ap.add_argument("--kappa", type=float, default=1e4)
    ap.add_argument("--M", type=int, default=10)
    ap.add_argument("--H0", type=float, default=0.5)
    ap.add_argument("--workload", default="gauss100", choices=["gauss100", "funnel", "dense1000", "stockwatson", "walnuts10"],
                    help="gauss100 = BASELINE configs[1] family (headline); funnel / dense1000 / stockwatson = configs 3 / 4 / 5 "
                         "(SURVEY 8d); walnuts10 = configs[0]'s target through walnuts_batch")
    ap.add_argument("--sw-run", default="D", choices=["D", "fixed", "R2P"],
                    help="stockwatson workload: which of mainSW.py:41-80's three runs (adaptLeapFrogD minC=3 H0=0.1 | "
                         "fixedLeapFrog H0=0.002 | adaptLeapFrogR2P minC=3 H0=0.1)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the side measurements (BASELINE configs 3, 4, 5 at SURVEY 8d size; C = 1024; diagonal precision)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--n1-ref", action="store_true",
                    help="self-launched --gpus N > 1: AFTER the N ranks have finished, also run the same workload on one GPU and add "
                         "n1_value_same_invocation / weak_scaling_eff to the line (off by default: it doubles the time of the invocation, "
                         "and the driver computes efficiency from its own N = 1 run)")
    ap.add_argument("--n1-value", type=float, default=None, help="the N = 1 value of this workload measured elsewhere (fills weak_scaling_eff)")
    ap.add_argument("--no-n1-ref", action="store_true", help="(accepted for compatibility; the N = 1 reference run is opt-in now: --n1-ref)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="initialise the process group (RCCL) and run every collective of the N-rank path -- barrier, MAX / SUM all_reduce, "
                         "all_gather of per-rank fields, the final gather of draws -- also with ONE rank: exercises RCCL on a one-GPU box")
    ap.add_argument("--gather", default="generated", choices=["generated", "identity"],
                    help="what the final gather moves: generated(q) = [q0, sum q^2] (kilobytes) or the identity draws "
                         "(SURVEY 8e: 0.83 GB per GPU for config 4 at 100 iterations)")
    ap.add_argument("--child-timeout", type=float, default=1800.0, help="self-launched ranks: seconds before they are ended")
    ap.add_argument("--dry-run", action="store_true",
                    help="ranks only rendezvous (gloo), reduce and gather a small tensor and print a line: the launcher's own test, no GPU")
    ap.add_argument("--route", default="", help="comma-separated WN_ROUTE_* names for the workload's engine (include/walnuts_amd.h; A/B runs: "
                                                "FAST4_WHOLE, ORBIT_LEGACY, ...); default: the library chooses")
    ap.add_argument("--split-parts", type=int, default=0, help="dense1000: wn_config.split_parts (0 / 1 = one part)")
    ap.add_argument("--split-ks", type=int, default=0, help="dense1000: wn_config.split_ks")
    ap.add_argument("--emulate-ranks", type=int, default=0,
                    help="run the R shards of an R-rank job of --workload (chain_offset = r * chains, the real job's seeds) one after another on "
                         "ONE GPU and print per-shard kernel_ms and predicted_weak_scaling_eff (SURVEY 8e: the only loss is imbalance); the "
                         "default invocation does this with R = 8 for configs 2-5 (extras.emulated_8_ranks)")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per launch from a separate rocprofv3 --pmc pass (see profiles/)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            return _self_launch(args)  # never returns
    elif int(os.environ["WORLD_SIZE"]) != args.gpus and "WN_BENCH_CHILD" not in os.environ:
        ap.error(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks")
    if args.dry_run:
        return main_dry_run()
    if args.workload == "walnuts10":
        return main_walnuts10(args)
    if args.emulate_ranks:
        import torch
        if torch.cuda.device_count() < 1:
            sys.exit("bench.py needs a ROCm GPU (the product has no CPU path)")
        torch.cuda.set_device(0)
        rec = emulate_ranks(args.workload, args.emulate_ranks, torch.device("cuda", 0), args.steps, args.warmup, args, args.sw_run)
        print(json.dumps({"metric": "predicted weak-scaling efficiency, %d emulated ranks on one GPU" % args.emulate_ranks,
                          "value": rec["predicted_weak_scaling_eff"], "unit": "fraction", "n_gpus": 1, "steps": args.steps,
                          "warmup": args.warmup, "higher_is_better": True, "scaling": "weak", "data": "synthetic", "dtype": "f64", **rec}), flush=True)
        return

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    spec = _with_route(workload_spec(args.workload, args, rank, args.sw_run), args)
    headline = args.workload == "gauss100"
    extras_on = headline and world == 1 and not args.no_extras
    side_specs = []
    if extras_on:
        side_specs = [workload_spec("funnel"), workload_spec("dense1000"), workload_spec("stockwatson", sw_run="D"),
                      workload_spec("stockwatson", sw_run="fixed"), workload_spec("stockwatson", sw_run="R2P")]
    cpu = None
    side_cpu = {}
    if rank == 0 and not args.no_cpu_baseline:  # forks workers: before any GPU init (N > 1: rank 0 only; the others wait in the rendezvous)
        t_ = time.perf_counter()
        # every configuration on every host thread (one pool); side configurations get a shorter sample: the default run must stay within minutes
        recs = cpu_baselines([spec] + side_specs, [12.0] + [3.0] * len(side_specs))
        cpu = recs[spec["name"]]
        side_cpu = {sp["name"]: recs[sp["name"]] for sp in side_specs}
        print(f"bench.py: CPU baselines took {time.perf_counter() - t_:.1f} s", file=sys.stderr)

    import torch
    rank, world, dev, cdev, dist = _dist_setup(args.force_collectives)
    coll = world > 1 or args.force_collectives  # the collectives run (RCCL also with one rank when forced)

    def barrier():
        if coll:
            dist.barrier()
        torch.cuda.synchronize(dev)

    t_ = time.perf_counter()
    eng = make_engine(spec, dev, rank)
    m = timed_launches(eng, spec, dev, args.steps, args.warmup, barrier)
    print(f"bench.py: rank {rank}: GPU set-up + {args.warmup} + {args.steps} launches took {time.perf_counter() - t_:.1f} s", file=sys.stderr)
    elapsed = m["elapsed"]
    t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    tot = torch.tensor([m["units"]], dtype=torch.float64, device=cdev)
    if coll:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    units_all = float(tot.item())
    avg_ms = float(np.mean(m["kernel_ms"]))
    scal = _scaling_fields(dist, world, cdev, units_all / elapsed, avg_ms, m["units"], coll)

    # final gather of draws to rank 0 (the only data collective; RCCL over xGMI), timed apart
    gather_ms = 0.0
    gather_bytes = 0
    draws = m["draws"]
    if coll:
        from walnuts_amd.dist import gather_draws
        torch.cuda.synchronize(dev)
        g0 = time.perf_counter()
        mine = m["draws"].to(cdev)
        draws = gather_draws(mine, dst=0)
        torch.cuda.synchronize(dev)
        gather_ms = (time.perf_counter() - g0) * 1e3
        gather_bytes = int(mine.numel() * mine.element_size())

    if rank == 0:
        upl = units_all / world / args.steps  # units of one launch of one rank
        roof = roofline_of(spec, upl, avg_ms, eng.kernel_name)
        if args.traffic_bytes is not None:
            roof["traffic"] = args.traffic_bytes
        out = {
            "metric": spec["metric"], "value": units_all / elapsed, "unit": spec["unit"], "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": spec["workload"], "chains_per_gpu": spec["C"], "dim": spec["D"], "iters_per_step": spec["iters"],
                       "kernel": eng.kernel_name, "sharding": f"chains x{world}, no data-path collective"},
            ("leapfrog_steps_per_launch_per_gpu" if spec["bound"] == "mfma" else "gradient_evaluations_per_launch_per_gpu"): upl,
            "kernel_ms_avg": avg_ms, "kernel_ms": m["kernel_ms"],
            # mean life of a 4-chain wave / the launch (from the per-chain step counts): value / tail_frac = the rate of a launch without its tail
            "tail_frac": m["tail_frac"], "value_without_tail": None if not m["tail_frac"] else units_all / elapsed / m["tail_frac"],
            "gather_ms": gather_ms, "gather_bytes_per_rank": gather_bytes,
            "collectives": (os.environ.get("WN_DIST_BACKEND", "nccl") + (" (forced with one rank)" if world == 1 else "")) if coll else None,
            **scal, "roofline": roof,
        }
        if not headline:
            out.update({k: v for k, v in side_record(spec, m, eng.kernel_name, None).items()
                        if k in ("min", "max", "launch_values", "busiest_chain_evals", "chain_occupancy", "tick_us", "tick_over_floor",
                                 "split_ticks_per_launch", "kernel_launches_per_s")})
        if world > 1 and args.n1_value:  # (the self-launcher's parent fills these itself; this is the torch.distributed.run form)
            out["n1_value_same_invocation"] = args.n1_value
            out["weak_scaling_eff"] = out["value"] / (world * args.n1_value)
        out["ess_per_sec"] = ess_of(spec, draws, elapsed / args.steps)
        import hashlib
        out["draws_sha256"] = hashlib.sha256(np.ascontiguousarray(draws.cpu().numpy()).tobytes()).hexdigest()  # of the gathered draws [C_total, iters+1, dg]
        eng.close()
        if extras_on:
            import walnuts_amd as wa
            from walnuts_amd import _lib as L
            extras = {}
            try:
                # SURVEY 8d cfg 2 side numbers, one launch each: BASELINE configs[1]'s literal chain count (1024) and the
                # diagonal-precision variant (HBM-style target)
                D, C = spec["D"], spec["C"]
                q0d = torch.as_tensor(spec["q0"], device=dev)
                e2 = wa.Engine(D, 1024, integrator=L.I_FIXED, H0=args.H0, stepSizeRandScale=0.2, M=args.M,
                               generated=L.G_Q0_SUMSQ, device=dev.index).set_target(L.T_DENSEGAUSS, spec["tpar"])
                e2.run(q0d[:1024], SEED + 77, spec["iters"])
                r2 = []  # one untimed + three timed launches (different seeds): the median rate
                for k in range(3):
                    s2, d2 = e2.run(q0d[:1024], SEED + k, spec["iters"])
                    st = float((d2[:, :, 6].sum() + d2[:, :, 7].sum()).item())
                    r2.append((st / (e2.last_run_ms() * 1e-3), e2.last_run_ms()))
                r2.sort()
                v2, ms2 = r2[1]
                tf2 = v2 * spec["per_unit"] / 1e12
                extras["dense_1024_chains"] = {"value": v2, "unit": "leapfrog steps/s", "kernel": e2.kernel_name, "ms_per_step": ms2,
                                               "value_min": r2[0][0], "value_max": r2[2][0], "launches": 3,
                                               "roofline": {"bound": "mfma", "achieved": tf2, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                                            "frac": tf2 / FP64_MFMA_PEAK_TFLOPS, "traffic": None,
                                                            "flop_per_leapfrog_step": spec["per_unit"]}}
                e2.close()
                lam = np.logspace(0.0, np.log10(args.kappa), D)
                e3 = wa.Engine(D, C, integrator=L.I_FIXED, H0=args.H0, stepSizeRandScale=0.2, M=args.M,
                               generated=L.G_Q0_SUMSQ, device=dev.index).set_target(L.T_DIAGGAUSS, 1.0 / lam)
                qd = torch.as_tensor(np.random.default_rng([SEED, 9]).normal(size=(C, D)) * np.sqrt(lam), device=dev)
                e3.run(qd, SEED + 77, spec["iters"])
                s3, d3 = e3.run(qd, SEED, spec["iters"])
                st = float((d3[:, :, 6].sum() + d3[:, :, 7].sum()).item())
                ms3 = e3.last_run_ms()
                extras["diag_variant"] = {"value": st / (ms3 * 1e-3), "unit": "leapfrog steps/s", "kernel": e3.kernel_name,
                                          "hbm_roofline_GBps_at_48D_bytes_per_step": st * 48.0 * D / (ms3 * 1e-3) / 1e9}
                e3.close()
                # the reference's adaptive integrators on the same problem (adaptiveIntegrators.py:65-137 / 361-475; H0 1.0, delta0 0.3,
                # 20 iterations per launch): one chain per wavefront, the matrix in LDS (wn_k_orbit.hip wide mode / wn_k_wavelds.hip)
                ad = {}
                for nm_, ig_ in (("adaptLeapFrogD", L.I_ADAPT_D), ("adaptLeapFrogR2P", L.I_ADAPT_R2P)):
                    e4 = wa.Engine(D, C, integrator=ig_, H0=1.0, delta0=0.3, stepSizeRandScale=0.2, M=args.M, generated=L.G_Q0_SUMSQ,
                                   device=dev.index).set_target(L.T_DENSEGAUSS, spec["tpar"])
                    e4.run(q0d, SEED + 77, 20)
                    s4, d4 = e4.run(q0d, SEED, 20)
                    st = float((d4[:, :, 6].sum() + d4[:, :, 7].sum()).item())
                    ad[nm_] = {"value": st / (e4.last_run_ms() * 1e-3), "unit": "gradient evaluations/s", "kernel": e4.kernel_name, "ms_per_step": e4.last_run_ms()}
                    e4.close()
                extras["dense_adaptive"] = ad
                # ... and on the standard Gaussian of the same dimension (the reference's Gaussian scaling experiments, mainGaussESS.py:47-48:
                # adaptLeapFrogR2P, M = 10, generated = [q0, sum q^2]): the trajectory-first kernel's wide mode
                e5 = wa.Engine(D, C, integrator=L.I_ADAPT_R2P, H0=1.4 * D ** -0.25, delta0=0.3, stepSizeRandScale=0.2, M=args.M, generated=L.G_Q0_SUMSQ,
                               device=dev.index).set_target(L.T_STDGAUSS, None)
                qs_ = torch.as_tensor(np.random.default_rng([SEED, 11]).normal(size=(C, D)), device=dev)
                e5.run(qs_, SEED + 77, 20)
                s5, d5 = e5.run(qs_, SEED, 20)
                st = float((d5[:, :, 6].sum() + d5[:, :, 7].sum()).item())
                extras["std_adaptive_R2P"] = {"value": st / (e5.last_run_ms() * 1e-3), "unit": "gradient evaluations/s", "kernel": e5.kernel_name, "ms_per_step": e5.last_run_ms(), "dim": D}
                e5.close()
            except Exception as ex:
                extras["error"] = repr(ex)
            # BASELINE configs 3, 4 (one GPU's shard), 5 (one GPU's shard; all three runs of mainSW.py:41-80) at SURVEY 8d's
            # sizes: one untimed + three timed launches each
            oc = {"note": "SURVEY 8d sizes, 1 untimed + 3 timed launches each, outside the headline's timed region; value = median of the "
                          "per-launch rates (units of a launch / its HIP-event time on the launch stream)"}
            for sp in side_specs:
                t_ = time.perf_counter()
                try:
                    e = make_engine(sp, dev, 0)
                    ms_ = timed_launches(e, sp, dev, 3, 1, lambda: torch.cuda.synchronize(dev))
                    oc[f"cfg{sp['cfg']}_{sp['name']}"] = side_record(sp, ms_, e.kernel_name, side_cpu.get(sp["name"]))
                    e.close()
                except Exception as ex:
                    oc[f"cfg{sp['cfg']}_{sp['name']}"] = {"error": repr(ex)}
                print(f"bench.py: side configuration {sp['name']} took {time.perf_counter() - t_:.1f} s", file=sys.stderr)
            extras["other_configs"] = oc
            try:  # BASELINE configs[0]'s sampler as whole transitions on the device (1 untimed + 2 timed calls)
                t_ = time.perf_counter()
                w = walnuts10_record(4096, 20, 2, 1, host_form=False)
                extras["walnuts10_device"] = {k: w[k] for k in ("value", "unit", "ms_per_step", "config", "note", "draws_mean_abs_max", "draws_var_mean",
                                                                   "reference_python_transitions_per_s_per_core")}
                extras["walnuts_batch"] = {"value": walnuts_batch_rate(4096, 4), "unit": "transitions/s",
                                           "what": "walnuts_batch, 4096 chains x 4 transitions, one numpy Generator per chain consumed on the host in the "
                                                   "reference's order through numpy's C distributions (csrc/wn_hostrng.c): the reference's contract"}
                print(f"bench.py: walnuts10 took {time.perf_counter() - t_:.1f} s", file=sys.stderr)
            except Exception as ex:
                extras["walnuts10_device"] = {"error": repr(ex)}
            # SURVEY 8e's only loss term, measured on the one GPU: the 8 shards of an 8-rank job one after another
            em = {}
            t_ = time.perf_counter()
            for key, nm_, run_ in EMULATED:
                try:
                    em[key] = emulate_ranks(nm_, 8, dev, 5, 1, args if nm_ == "gauss100" else None, run_)
                except Exception as ex:
                    em[key] = {"error": repr(ex)}
            print(f"bench.py: 8 emulated ranks x 4 configurations took {time.perf_counter() - t_:.1f} s", file=sys.stderr)
            extras["emulated_8_ranks"] = em
            out["extras"] = extras
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
        if extras_on:
            print(summary_line(out), file=sys.stderr, flush=True)  # LAST line of stderr: every configuration's number in <= 600 bytes
    if coll:
        dist.barrier()
        dist.destroy_process_group()


def main_dry_run():
    """The launcher's own test (tests/test_host_logic.py): rendezvous over gloo, one reduction, the final gather; no GPU."""
    import torch
    import torch.distributed as dist
    from walnuts_amd.dist import gather_draws
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    fail = os.environ.get("WN_BENCH_FAIL_RANK")
    if fail is not None and int(fail) == rank:
        sys.exit(3)
    slp = os.environ.get("WN_BENCH_SLEEP_RANK")  # test hook: this rank sleeps through the rendezvous, so nobody finishes
    if slp is not None and int(slp) == rank:
        time.sleep(600.0)
    if world > 1:
        dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    g = torch.full((3, 2), float(rank), dtype=torch.float64)
    g[:, 1] = torch.arange(rank * 3, rank * 3 + 3, dtype=torch.float64)  # the GLOBAL chain ids of this rank's shard (chain_offset = rank * C)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        g = gather_draws(g, dst=0)
    if rank == 0:
        print(json.dumps({"metric": "dry run", "value": float(t.item()), "n_gpus": world, "gathered_rows": int(g.shape[0]),
                          "gathered_in_global_chain_order": bool((g[:, 1] == torch.arange(g.shape[0], dtype=torch.float64)).all().item()),
                          "gathered_rank_of_row": [int(x) for x in g[:, 0].tolist()]}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def walnuts10_record(C, iters, steps, warmup, host_form=True):
    """BASELINE configs[0]'s target (10-D standard normal, test/test.py:11-18 parameters: macro_step 2.0, max_nuts_depth 10,
    max_error 0.1) as whole transitions on the device (`walnuts_device`: walnuts/walnuts.py:279-359 with the engine's random
    stream): transitions/s of C chains, `iters` transitions per call.  The time is the whole API call (handle set-up, H2D of the
    start, the launch, D2H of the draws).  Beside it, the host-Generator form (`walnuts_batch`, the reference's contract) on a
    small batch.  The reference runs ~1.2e3 transitions/s on one core."""
    import torch
    from walnuts_amd import targetDistr as td
    import walnuts_amd as wa
    D = 10
    theta = np.zeros((C, D))
    t_all = []
    d = None
    for k in range(warmup + steps):
        t0 = time.perf_counter()
        d = wa.walnuts_device(SEED + k, theta, td.stdGauss, np.ones(D), 2.0, 10, 0.1, 0, iters, transitions_per_launch=iters)
        torch.cuda.synchronize()
        if k >= warmup:
            t_all.append(time.perf_counter() - t0)
    el = float(np.sum(t_all))
    host_rate = None
    if host_form:  # the host-Generator form (the reference's contract: one numpy Generator per chain, consumed in the reference's order)
        host_rate = walnuts_batch_rate(C, 4)
    from walnuts_amd.ess import ess
    per = el / steps
    return {"metric": "transitions/sec across chains, walnuts() on the 10-D standard normal", "value": C * iters * steps / el,
            "unit": "transitions/s", "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": per * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"walnuts_device: {C} chains x {iters} transitions per step, D=10 standard normal, macro_step=2.0, "
                                   f"max_nuts_depth=10, max_error=0.1 (BASELINE configs[0]'s target, test/test.py:11-18)",
                       "kernel": "wstep_wave_kernel (wn_k_wstep.hip): one chain per wavefront, lane = element, whole transitions on the device, Philox stream of wn_rng.h"},
            "note": "a launch lasts as long as its slowest chain, and the cost of a transition of this sampler has a heavy tail (81 gradient evaluations in "
                    "the median, 75 332 at the maximum of 12 000 transitions of the reference itself: profiles/r04_walnuts_tail.txt), so more transitions "
                    "per launch average better; round 3's one-lane kernel made 2.7e4 transitions/s on this workload",
            "draws_mean_abs_max": float(np.abs(d.mean(axis=(0, 1))).max()), "draws_var_mean": float(d.var(axis=(0, 1)).mean()),
            "ess_per_sec": {"q0": ess(d[:, :, 0]) / per, "method": "rank-normalised split bulk ESS over all chains of one step"},
            "host_generator_form": None if host_rate is None else {"value": host_rate, "unit": "transitions/s", "what": f"walnuts_batch, {C} chains x 4 transitions, one numpy Generator per chain "
                                    "consumed on the host in the reference's order (the reference's contract)"},
            "reference_python_transitions_per_s_per_core": 1.2e3}


def walnuts_batch_rate(C, n_trans):
    """transitions/s of `walnuts_batch` -- walnuts(rng, ...)'s contract (walnuts/walnuts.py:362-408) for C chains at once, each with its
    own numpy Generator consumed on the host in the reference's order -- on BASELINE configs[0]'s target; one untimed call first."""
    from walnuts_amd import targetDistr as td
    import walnuts_amd as wa
    D = 10
    wa.walnuts_batch([np.random.default_rng(c) for c in range(C)], np.zeros((C, D)), td.stdGauss, td.stdGauss, np.ones(D), 2.0, 10, 0.1, 0, 1)
    rngs = [np.random.default_rng(1000 + c) for c in range(C)]
    t0 = time.perf_counter()
    wa.walnuts_batch(rngs, np.zeros((C, D)), td.stdGauss, td.stdGauss, np.ones(D), 2.0, 10, 0.1, 0, n_trans)
    return C * n_trans / (time.perf_counter() - t0)


def main_walnuts10(args):
    print(json.dumps(walnuts10_record(args.chains or 4096, args.iters or 20, args.steps, args.warmup)), flush=True)


if __name__ == "__main__":
    main()
