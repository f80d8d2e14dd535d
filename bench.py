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


def _cpu_worker(args):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    A, q0, chain, iters, M, H0 = args
    t0 = time.perf_counter()
    o = ol.walnuts(ol.T_DENSEGAUSS, q0, integrator=ol.I_FIXED, tpar=A.ravel(), H0=H0, numIter=iters,
                   M=M, seed=SEED, chain=chain)
    dt = time.perf_counter() - t0
    d = o["diagnostics"]
    return float(d[:, 6].sum() + d[:, 7].sum()), dt


def python_reference_cpu(key):
    """The UNMODIFIED Python reference timed in the build container (tools/ref_cpu_baseline.py; it never travels to the
    GPU box): BASELINE.md section 3's "measured elsewhere" figure, quoted with where it was measured."""
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", "r02_python_reference_cpu.json")))
        r = j[key]
        return {"unit": r["unit"], "per_core": r["one_core"]["per_core"], "all_cores": r["all_cores"]["value"],
                "cores": r["all_cores"]["cores"], "measured": j["where"], "code": j["code"],
                "note": "not measured in this run: the Python reference does not travel to the GPU box"}
    except Exception:
        return None


def cpu_baseline(A, q0, iters, M, H0, budget_s=12.0):
    """Time the CPU oracle (the validated C restatement of the reference's path) on the host
    cores, one chain per process, on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    ol.build()
    cores = os.cpu_count() or 1
    # calibrate on one chain, one iteration
    s1, t1 = _cpu_worker((A, q0[0], 0, 1, M, H0))
    rate1 = s1 / max(t1, 1e-6)
    per_iter = 540.0 / rate1 if rate1 > 0 else 1.0  # ~540 leapfrog steps per iteration of this workload; only sizes the sample
    n_it = int(max(1, min(iters, budget_s / max(per_iter, 1e-6))))
    nchain = min(cores, q0.shape[0])
    with mp.get_context("fork").Pool(nchain) as pool:
        pool.map(_cpu_worker, [(A, q0[c], c, 1, 3, H0) for c in range(nchain)])  # start the workers
        t0 = time.perf_counter()
        res = pool.map(_cpu_worker, [(A, q0[c], c, n_it, M, H0) for c in range(nchain)], chunksize=1)
        wall = time.perf_counter() - t0
    steps = sum(r[0] for r in res)
    C_all = q0.shape[0]
    return {"value": steps / wall, "unit": "leapfrog steps/s", "cores": nchain, "kind": "port",
            "sample": f"the first {nchain} of the {C_all} chains ({100.0 * nchain / C_all:.2f} %) x {n_it} of the {iters} iterations "
                      f"({steps:.3g} leapfrog steps, {wall:.1f} s), one chain per process (oracle/libwalnuts_oracle.so, the C "
                      f"restatement pinned to the Python reference)",
            "python_reference": python_reference_cpu("cfg2_dense100_fixedLeapFrog")}


def _self_launch(args):
    """`python bench.py --gpus N` without torchrun: this process has not touched torch or HIP yet, so it
    may start children.  Starts one fresh child per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, the
    same command line), after one single-GPU child of the same workload whose value becomes the N = 1
    reference of `weak_scaling_eff`; relays rank 0's JSON line and exits with the worst child's code."""
    import socket
    import subprocess
    n = args.gpus
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    argv = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    if not args.no_n1_ref:
        r = subprocess.run(argv + ["--gpus", "1", "--no-extras", "--no-cpu-baseline"], env=dict(base, WN_BENCH_CHILD="1"),
                           stdout=subprocess.PIPE, text=True)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode == 0 and lines:
            base["WN_BENCH_N1_VALUE"] = repr(float(json.loads(lines[-1])["value"]))
        else:
            print(f"bench.py: the N=1 reference run failed (rc {r.returncode}); continuing without it", file=sys.stderr)
    procs = []
    for rank in range(n):
        env = dict(base, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), WN_BENCH_CHILD="1")
        procs.append(subprocess.Popen(argv, env=env, stdout=subprocess.PIPE if rank == 0 else sys.stderr, text=True))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p_.wait() for p_ in procs[1:]]
    sys.stdout.write(out0)
    sys.stdout.flush()
    sys.exit(max(abs(rc) for rc in rcs))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--chains", type=int, default=4096, help="chains per GPU")
    ap.add_argument("--iters", type=int, default=100, help="sampler iterations per launch")
    ap.add_argument("--dim", type=int, default=100)
    ap.add_argument("--kappa", type=float, default=1e4)
    ap.add_argument("--M", type=int, default=10)
    ap.add_argument("--H0", type=float, default=0.5)
    ap.add_argument("--workload", default="gauss100", choices=["gauss100", "funnel", "dense1000", "stockwatson"],
                    help="gauss100 = BASELINE configs[1] family (headline); funnel = configs[2] "
                         "(funnel10, adaptLeapFrogR2P, M=12, H0=0.3, delta0=0.3; SURVEY 8d cfg 3)")
    ap.add_argument("--sw-run", default="D", choices=["D", "fixed", "R2P"],
                    help="stockwatson workload: which of mainSW.py:41-80's three runs (adaptLeapFrogD minC=3 H0=0.1 | "
                         "fixedLeapFrog H0=0.002 | adaptLeapFrogR2P minC=3 H0=0.1)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the SURVEY 8d side measurements (C=1024, diagonal-precision variant)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-n1-ref", action="store_true",
                    help="self-launched --gpus N > 1: skip the single-GPU run that weak_scaling_eff refers to")
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

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.workload == "funnel":
        return main_funnel(args)
    if args.workload == "stockwatson":
        return main_stockwatson(args)
    if args.workload == "dense1000":  # SURVEY 8d cfg 4: one GPU's shard (1024 chains) of the 8192-chain job
        args.dim, args.kappa, args.no_extras = 1000, 100.0, True
        if args.chains == 4096:
            args.chains = 1024
        if args.iters == 100:
            args.iters = 50
    C, D = args.chains, args.dim
    offset = rank * C
    A, q0 = make_dense_problem(D, args.kappa, C, SEED, chain_offset=offset)
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(A, q0, args.iters, args.M, args.H0)  # forks workers: before any GPU init

    import torch
    rank, world, dev, cdev, dist = _dist_setup()

    import walnuts_amd as wa
    from walnuts_amd import _lib as L

    eng = wa.Engine(D, C, integrator=L.I_FIXED, H0=args.H0, stepSizeRandScale=0.2, M=args.M,
                    generated=L.G_Q0_SUMSQ, chain_offset=offset, device=dev.index)
    eng.set_target(L.T_DENSEGAUSS, A)
    q0d = torch.as_tensor(q0, device=dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    warm_steps = torch.zeros((), dtype=torch.float64, device=dev)
    for w in range(max(args.warmup, 1) if args.warmup else 0):
        _, dw = eng.run(q0d, SEED + 1000 + w, args.iters)
        warm_steps += dw[:, :, 6].sum() + dw[:, :, 7].sum()  # also loads torch's reduction kernels outside the timed region
    if not args.warmup:  # (first use of a torch kernel costs ~90 ms of code loading)
        warm_steps += torch.zeros((2, 2, 8), dtype=torch.float64, device=dev)[:, :, 6].sum()
    barrier()
    t0 = time.perf_counter()
    total_steps = torch.zeros((), dtype=torch.float64, device=dev)
    kernel_ms = []
    last = None
    for k in range(args.steps):
        samples, diag = eng.run(q0d, SEED + k, args.iters)
        total_steps += diag[:, :, 6].sum() + diag[:, :, 7].sum()
        kernel_ms.append(eng.last_run_ms())  # HIP events on the launch stream (synchronises)
        last = samples
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    total_steps = total_steps.to(cdev).reshape(1)
    total_steps_rank = float(total_steps.item())
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(total_steps, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    steps_all = float(total_steps.item())
    scal = _scaling_fields(dist, world, cdev, steps_all / elapsed, float(np.mean(kernel_ms)), float(total_steps_rank))

    # final gather of draws to rank 0 (the only data collective; RCCL over xGMI), timed apart
    gather_ms = 0.0
    draws = last
    if world > 1:
        from walnuts_amd.dist import gather_draws
        torch.cuda.synchronize(dev)
        g0 = time.perf_counter()
        draws = gather_draws(last.to(cdev), dst=0)
        torch.cuda.synchronize(dev)
        gather_ms = (time.perf_counter() - g0) * 1e3

    if rank == 0:
        traffic = args.traffic_bytes
        if traffic is None:  # PMC traffic of this exact workload, measured in a separate rocprofv3 pass
            try:
                for name in ("r02_traffic.json", "r01_traffic.json"):
                    tj = json.load(open(os.path.join(ROOT, "profiles", name)))
                    w = tj["workload"]
                    if (w["dim"], w["chains"], w["iters"]) == (D, C, args.iters) and w["kernel"] == eng.kernel_name:
                        traffic = tj["bytes_per_launch"]
                        break
            except Exception:
                traffic = None
            if traffic is None and D == 1000 and "split" in eng.kernel_name:
                bps = _side_traffic("dense1000", "bytes_per_leapfrog_step")
                traffic = None if bps is None else bps * steps_all / world / args.steps
        flop_per_step = 2.0 * D * D + 12.0 * D  # matvec + leapfrog updates + 2 dots
        avg_ms = float(np.mean(kernel_ms))
        steps_per_launch_rank0 = steps_all / world / args.steps
        achieved = steps_per_launch_rank0 * flop_per_step / (avg_ms * 1e-3) / 1e12
        out = {
            "metric": "leapfrog steps/sec across chains, 100-D Gaussian",
            "value": steps_all / elapsed,
            "unit": "leapfrog steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{D}-D dense Gaussian kappa={args.kappa:g}, {C} chains/GPU, "
                                   f"fixedLeapFrog, M={args.M}, H0={args.H0}, {args.iters} iterations/launch "
                                   f"(SURVEY 8d cfg {4 if D >= 1000 else 2})",
                       "chains_per_gpu": C, "dim": D, "iters_per_step": args.iters,
                       "kernel": eng.kernel_name, "sharding": f"chains x{world}, no data-path collective"},
            "leapfrog_steps_per_launch_per_gpu": steps_per_launch_rank0,
            "kernel_ms_avg": avg_ms,
            "gather_ms": gather_ms,
            **scal,
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                         "traffic": traffic,
                         "flop_per_leapfrog_step": flop_per_step},
        }
        # ESS/sec of gen(q) = [q[0], sum(q*q)] (mainGaussESS.py:29-30) over all chains of the last
        # launch; rank-normalised split bulk ESS (walnuts_amd/ess.py), host side, outside the timing
        try:
            from walnuts_amd.ess import ess
            dr = draws.cpu().numpy()
            e0, e1 = ess(dr[:, 1:, 0]), ess(dr[:, 1:, 1])
            per_launch_s = elapsed / args.steps
            out["ess_per_sec"] = {"q0": e0 / per_launch_s, "sum_q2": e1 / per_launch_s,
                                  "ess_q0": e0, "ess_sum_q2": e1,
                                  "draws": int(dr.shape[0] * (dr.shape[1] - 1)),
                                  "method": "rank-normalised split bulk ESS over all chains of one launch"}
        except Exception as ex:  # the throughput line must not depend on the ESS estimator
            out["ess_per_sec"] = {"error": repr(ex)}
        if world == 1 and not args.no_extras and args.workload == "gauss100":
            # SURVEY 8d cfg 2 side numbers, one launch each, outside the timed region:
            # BASELINE configs[1]'s chain count (1024) and the diagonal-precision variant (HBM-style target)
            extras = {}
            try:
                e2 = wa.Engine(D, 1024, integrator=L.I_FIXED, H0=args.H0, stepSizeRandScale=0.2, M=args.M,
                               generated=L.G_Q0_SUMSQ, device=dev.index).set_target(L.T_DENSEGAUSS, A)
                e2.run(q0d[:1024], SEED + 77, args.iters)
                s2, d2 = e2.run(q0d[:1024], SEED, args.iters)
                st = float((d2[:, :, 6].sum() + d2[:, :, 7].sum()).item())
                extras["dense_1024_chains"] = {"value": st / (e2.last_run_ms() * 1e-3), "unit": "leapfrog steps/s"}
                e2.close()
                lam = np.logspace(0.0, np.log10(args.kappa), D)
                e3 = wa.Engine(D, C, integrator=L.I_FIXED, H0=args.H0, stepSizeRandScale=0.2, M=args.M,
                               generated=L.G_Q0_SUMSQ, device=dev.index).set_target(L.T_DIAGGAUSS, 1.0 / lam)
                qd = torch.as_tensor(np.random.default_rng([SEED, 9]).normal(size=(C, D)) * np.sqrt(lam), device=dev)
                e3.run(qd, SEED + 77, args.iters)
                s3, d3 = e3.run(qd, SEED, args.iters)
                st = float((d3[:, :, 6].sum() + d3[:, :, 7].sum()).item())
                ms3 = e3.last_run_ms()
                extras["diag_variant"] = {"value": st / (ms3 * 1e-3), "unit": "leapfrog steps/s", "kernel": e3.kernel_name,
                                          "hbm_roofline_GBps_at_48D_bytes_per_step": st * 48.0 * D / (ms3 * 1e-3) / 1e9}
                e3.close()
            except Exception as ex:
                extras["error"] = repr(ex)
            # the other BASELINE configs, one short launch each (outside the timed region; the full lines
            # are `--workload funnel | dense1000 | stockwatson`)
            try:
                extras["other_configs"] = side_configs(wa, L, torch, dev)
            except Exception as ex:
                extras["other_configs"] = {"error": repr(ex)}
            out["extras"] = extras
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def side_configs(wa, L, torch, dev):
    """One launch each of BASELINE configs 3, 4 (one GPU's shard) and 5 (one GPU's shard)."""
    res = {}
    rng = np.random.default_rng([SEED, 3])
    C = 4096
    q0 = np.zeros((C, 11)); q0[:, 0] = 3.0 * rng.normal(size=C)
    q0[:, 1:] = np.exp(0.5 * q0[:, :1]) * rng.normal(size=(C, 10))
    e = wa.Engine(11, C, integrator=L.I_ADAPT_R2P, H0=0.3, delta0=0.3, M=12, generated=L.G_FIRSTK, gen_k=2,
                  device=dev.index).set_target(L.T_FUNNEL10)
    e.run(torch.as_tensor(q0, device=dev), SEED + 1, 2)  # untimed first launch (code load)
    s, d = e.run(torch.as_tensor(q0, device=dev), SEED, 30)
    ev = float((d[:, :, 6] + d[:, :, 7]).sum().item())
    res["note"] = ("one SHORT launch each, outside the timed region (iteration counts below, not SURVEY 8d's): a smoke reading only; "
                   "the contract lines of these configurations are `bench.py --workload funnel | dense1000 | stockwatson` "
                   "(profiles/r02_bench_*.json)")
    res["cfg3_funnel10_R2P_4096_chains"] = {"value": ev / (e.last_run_ms() * 1e-3), "unit": "gradient evaluations/s",
                                            "kernel": e.kernel_name, "iterations": 30}
    e.close()
    A, q1 = make_dense_problem(1000, 100.0, 1024, SEED)
    e = wa.Engine(1000, 1024, integrator=L.I_FIXED, H0=0.5, M=10, generated=L.G_Q0_SUMSQ, device=dev.index)
    e.set_target(L.T_DENSEGAUSS, A)
    e.run(torch.as_tensor(q1, device=dev), SEED + 1, 2)
    s, d = e.run(torch.as_tensor(q1, device=dev), SEED, 30)
    st = float((d[:, :, 6] + d[:, :, 7]).sum().item()); ms = e.last_run_ms()
    res["cfg4_dense1000_1024_chains"] = {"value": st / (ms * 1e-3), "unit": "leapfrog steps/s", "kernel": e.kernel_name, "iterations": 30,
                                         "fp64_mfma_frac": st * (2.0e6 + 12.0e3) / (ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS}
    e.close()
    sw = np.load(os.path.join(ROOT, "tests", "golden", "stockwatson_data.npz"))
    T = int(sw["T"][0])
    e = wa.Engine(3 * T, 256, integrator=L.I_ADAPT_D, H0=0.1, delta0=0.3, M=14, minC=3, generated=L.G_FIRSTK, gen_k=2,
                  device=dev.index).set_target(L.T_STOCKWATSON, np.concatenate([[float(T)], sw["y"]]))
    e.run(torch.zeros((256, 3 * T), dtype=torch.float64, device=dev), SEED + 1, 1)
    s, d = e.run(torch.zeros((256, 3 * T), dtype=torch.float64, device=dev), SEED, 10)
    ev = float((d[:, :, 6] + d[:, :, 7]).sum().item())
    res["cfg5_stockwatson_adaptD_256_chains"] = {"value": ev / (e.last_run_ms() * 1e-3), "unit": "gradient evaluations/s",
                                                 "kernel": e.kernel_name, "iterations": 10}
    e.close()
    return res


def _dist_setup():
    """One process per GPU (torch.distributed.run env); returns rank, world, device, collective device, dist."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("WN_DIST_BACKEND", "nccl")
    if torch.cuda.device_count() < 1:
        sys.exit("bench.py needs a ROCm GPU (the product has no CPU path)")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())
    return rank, world, dev, (dev if backend == "nccl" else torch.device("cpu")), dist


def _scaling_fields(dist, world, cdev, value, kernel_ms_avg, units_rank):
    """Per-rank kernel time and work (one small all_gather outside the timed region) and, when bench.py
    launched the ranks itself, the weak-scaling efficiency against the single-GPU run of the same invocation."""
    import torch
    if world == 1:
        return {}
    mine = torch.tensor([kernel_ms_avg, units_rank], dtype=torch.float64, device=cdev)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    out = {"per_rank_kernel_ms": [float(a[0].item()) for a in allr],
           "per_rank_units": [float(a[1].item()) for a in allr]}
    n1 = os.environ.get("WN_BENCH_N1_VALUE")
    if n1:
        out["n1_value_same_invocation"] = float(n1)
        out["weak_scaling_eff"] = value / (world * float(n1))
    return out


def _side_traffic(key, field):
    try:
        for name in ("r02_traffic_side.json", "r01_traffic_side.json"):
            f = os.path.join(ROOT, "profiles", name)
            if os.path.exists(f):
                rec = json.load(open(f))
                if key in rec and field in rec[key]:
                    return float(rec[key][field])
        return None
    except Exception:
        return None


def _run_evals_workload(args, make_engine, q0_of, metric, workload, bytes_per_eval, traffic_key=None):
    """Common driver of the gradient-evaluation workloads (configs 3 and 5): chains shard over ranks with no
    data-path collective; barrier + synchronize around exactly `steps` launches; max time over ranks."""
    import torch
    rank, world, dev, cdev, dist = _dist_setup()
    eng, C = make_engine(rank, dev)
    q0d = torch.as_tensor(q0_of(rank, C), device=dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for w in range(args.warmup):
        _, dw = eng.run(q0d, SEED + 1000 + w, args.iters)
        float((dw[:, :, 6] + dw[:, :, 7]).sum(1).max().item())  # loads torch's reduction kernels outside the timed region
    if not args.warmup:
        float(torch.zeros((2, 2, 8), dtype=torch.float64, device=dev)[:, :, 6].sum(1).max().item())
    barrier()
    t0 = time.perf_counter()
    tot = 0.0; kms = []; mx = []
    for k in range(args.steps):
        s, d = eng.run(q0d, SEED + k, args.iters)
        per_chain = (d[:, :, 6] + d[:, :, 7]).sum(1)
        tot += float(per_chain.sum().item())
        mx.append(float(per_chain.max().item()))
        kms.append(eng.last_run_ms())
    barrier()
    el = time.perf_counter() - t0
    t = torch.tensor([el, tot], dtype=torch.float64, device=cdev)
    if world > 1:
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        el, tot_all = float(tmax[0].item()), float(t[1].item())
    else:
        tot_all = tot
    scal = _scaling_fields(dist, world, cdev, tot_all / el, float(np.mean(kms)), tot)
    if rank == 0:
        ach = tot / args.steps * bytes_per_eval / (np.mean(kms) * 1e-3) / 1e9  # rank 0's kernel
        print(json.dumps({"metric": metric, "value": tot_all / el, "unit": "gradient evaluations/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
                          "data": "synthetic",
                          "config": {"workload": workload(C), "kernel": eng.kernel_name, "chains_per_gpu": C,
                                     "sharding": f"chains x{world}, no data-path collective"},
                          "kernel_ms_avg": float(np.mean(kms)), **scal,
                          # a launch lasts as long as its busiest chain (that chain's evaluations are sequential)
                          "busiest_chain_evals": float(np.mean(mx)),
                          "chain_occupancy": tot / args.steps / C / float(np.mean(mx)),
                          "tick_us": float(np.mean(kms)) * 1e3 / float(np.mean(mx)),
                          "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": ach / HBM_PEAK_GBS,
                                       # `achieved` is SURVEY 8d's NOMINAL streaming figure (48*D bytes per evaluation); the fused
                                       # kernel keeps its state on chip, so beside it: the measured HBM rate (PMC bytes) and the
                                       # FP64-vector fraction at 14*D flop per evaluation
                                       "measured_hbm_GBps": (None if traffic_key is None or _side_traffic(traffic_key, "bytes_per_gradient_evaluation") is None
                                                             else _side_traffic(traffic_key, "bytes_per_gradient_evaluation") * tot / args.steps / (np.mean(kms) * 1e-3) / 1e9),
                                       "fp64_vector_frac": tot / args.steps * 14.0 * (bytes_per_eval / 48.0) / (np.mean(kms) * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                                       # measured HBM bytes of one launch of rank 0 (PMC bytes per evaluation x evaluations)
                                       "traffic": (None if traffic_key is None or _side_traffic(traffic_key, "bytes_per_gradient_evaluation") is None
                                                   else _side_traffic(traffic_key, "bytes_per_gradient_evaluation") * tot / args.steps),
                                       "bytes_per_gradient_evaluation": bytes_per_eval}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main_funnel(args):
    """SURVEY 8d cfg 3: funnel10 (D=11), adaptLeapFrogR2P, M=12, H0=0.3, delta0=0.3, warmupIter=0,
    q0[0]~N(0,9), q0[1:]~N(0,e^{q0[0]}); metric = gradient evaluations (nevalF+nevalB) per second.
    Roofline per SURVEY 8d: HBM, 48*D = 528 B per gradient evaluation (unfused streaming figure)."""
    import walnuts_amd as wa
    from walnuts_amd import _lib as L
    if args.iters == 100:
        args.iters = 300  # SURVEY 8d cfg 3

    def make_engine(rank, dev):
        C = args.chains
        e = wa.Engine(11, C, integrator=L.I_ADAPT_R2P, H0=0.3, delta0=0.3, M=12, generated=L.G_FIRSTK, gen_k=2,
                      chain_offset=rank * C, device=dev.index)
        return e.set_target(L.T_FUNNEL10), C

    def q0_of(rank, C):
        rng = np.random.default_rng([SEED, 3, rank])
        q0 = np.zeros((C, 11)); q0[:, 0] = 3.0 * rng.normal(size=C)
        q0[:, 1:] = np.exp(0.5 * q0[:, :1]) * rng.normal(size=(C, 10))
        return q0

    _run_evals_workload(args, make_engine, q0_of, "gradient evaluations/sec across chains, funnel10 (adaptLeapFrogR2P)",
                        lambda C: f"funnel10 D=11, {C} chains/GPU, adaptLeapFrogR2P, M=12, H0=0.3, delta0=0.3, "
                                  f"{args.iters} iterations/launch (SURVEY 8d cfg 3)", 528.0, "funnel")


def main_stockwatson(args):
    """SURVEY 8d cfg 5: StockWatson (D = 756), one GPU's shard of the 2048-chain job per rank, the first of
    mainSW.py:41-80's three runs (adaptLeapFrogD, minC=3, H0=0.1, delta0=0.3, M=14), q0 = 0.
    Metric = gradient evaluations/s; roofline per SURVEY 8d: HBM, 48*D bytes per evaluation."""
    import walnuts_amd as wa
    from walnuts_amd import _lib as L
    sw = np.load(os.path.join(ROOT, "tests", "golden", "stockwatson_data.npz"))
    T = int(sw["T"][0]); D = 3 * T
    if args.chains == 4096:
        args.chains = 256
    if args.iters == 100:
        args.iters = 20

    ig, H0, minC, name = {"D": (L.I_ADAPT_D, 0.1, 3, "adaptLeapFrogD minC=3"), "fixed": (L.I_FIXED, 0.002, 0, "fixedLeapFrog"),
                          "R2P": (L.I_ADAPT_R2P, 0.1, 3, "adaptLeapFrogR2P minC=3")}[args.sw_run]

    def make_engine(rank, dev):
        C = args.chains
        e = wa.Engine(D, C, integrator=ig, H0=H0, delta0=0.3, M=14, minC=minC, generated=L.G_FIRSTK, gen_k=2,
                      chain_offset=rank * C, device=dev.index)
        return e.set_target(L.T_STOCKWATSON, np.concatenate([[float(T)], sw["y"]])), C

    _run_evals_workload(args, make_engine, lambda rank, C: np.zeros((C, D)),
                        f"gradient evaluations/sec across chains, StockWatson ({name.split()[0]})",
                        lambda C: f"StockWatson D={D}, {C} chains/GPU, {name}, M=14, H0={H0}, delta0=0.3, "
                                  f"q0=0, {args.iters} iterations/launch (SURVEY 8d cfg 5)", 48.0 * D, "stockwatson")


if __name__ == "__main__":
    main()
