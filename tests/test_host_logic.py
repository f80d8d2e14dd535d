"""Host-side logic without a GPU: ESS estimator, sharding helpers (2-process gloo), bench
problem construction."""
import os
import subprocess
import sys
import textwrap
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ess_on_ar1_series():
    from walnuts_amd.ess import ess
    rng = np.random.default_rng(0)
    for phi in (0.0, 0.5, 0.9):
        m, n = 8, 4000
        x = np.zeros((m, n)); e = rng.normal(size=(m, n))
        x[:, 0] = e[:, 0] / np.sqrt(1 - phi ** 2)
        for t in range(1, n):
            x[:, t] = phi * x[:, t - 1] + e[:, t]
        want = m * n * (1 - phi) / (1 + phi)
        got = ess(x)
        assert 0.8 * want < got < 1.25 * want, (phi, got, want)
    # anticorrelated draws (what NUTS produces on Gaussians) exceed the number of draws
    x = np.zeros((4, 2000)); e = rng.normal(size=(4, 2000))
    for t in range(1, 2000):
        x[:, t] = -0.4 * x[:, t - 1] + e[:, t]
    assert ess(x) > 4 * 2000


def test_shard_range_partitions():
    from walnuts_amd.dist import shard_range
    for total in (1, 7, 4096, 8192 + 5):
        for world in (1, 2, 3, 8):
            r = [shard_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_bench_problem_is_sharding_invariant():
    sys.path.insert(0, ROOT)
    from bench import make_dense_problem
    A, q = make_dense_problem(20, 1e4, 700, 5)
    A2, qa = make_dense_problem(20, 1e4, 300, 5, chain_offset=0)
    A3, qb = make_dense_problem(20, 1e4, 400, 5, chain_offset=300)
    np.testing.assert_array_equal(A, A2); np.testing.assert_array_equal(A, A3)
    np.testing.assert_array_equal(q, np.concatenate([qa, qb]))
    ev = np.linalg.eigvalsh(A)
    assert abs(ev.max() / ev.min() - 1e4) / 1e4 < 1e-6


WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    from walnuts_amd.dist import shard_range, gather_draws, total_steps
    from bench import make_dense_problem
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    total = 37
    lo, hi = shard_range(total, rank, world)
    A, q0 = make_dense_problem(6, 100.0, hi - lo, 11, chain_offset=lo)
    # stand-in for per-rank draws: a deterministic function of the GLOBAL chain id and q0
    draws = torch.tensor(q0[:, :2] + np.arange(lo, hi)[:, None])
    diag = torch.zeros(hi - lo, 3, 24, dtype=torch.float64); diag[..., 6] = 5.0; diag[..., 7] = 1.0
    out = gather_draws(draws, dst=0)
    steps = total_steps(diag)
    if rank == 0:
        A1, qall = make_dense_problem(6, 100.0, total, 11)
        want = qall[:, :2] + np.arange(total)[:, None]
        assert out.shape == (total, 2) and np.array_equal(out.numpy(), want), "gather mismatch"
        assert steps == total * 3 * 6.0, steps
        print("OK")
    dist.destroy_process_group()
''')


def test_two_process_gloo_sharding_and_gather(tmp_path):
    """N > 1 path on CPU: contiguous sharding keyed by global chain id + the final gather."""
    script = tmp_path / "w.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29731", str(script)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


def _bench(args, env_extra=None, timeout=120):
    env = dict(os.environ, **(env_extra or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    return r, time.monotonic() - t0


def test_bench_self_launcher_starts_ranks_and_relays_rank0():
    """`python bench.py --gpus N` with no launcher: one fresh child per rank, rank 0's line relayed; with --n1-ref the N = 1
    reference child runs AFTER the ranks and the parent adds the two efficiency fields to the line."""
    import json
    r, dt = _bench(["--gpus", "2", "--dry-run"])
    assert r.returncode == 0, r.stdout + r.stderr
    o = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert o["n_gpus"] == 2 and o["value"] == 3.0 and o["gathered_rows"] == 6 and "n1_value_same_invocation" not in o
    r, dt = _bench(["--gpus", "2", "--dry-run", "--n1-ref"])
    assert r.returncode == 0, r.stdout + r.stderr
    o = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert o["n_gpus"] == 2 and o["value"] == 3.0 and o["n1_value_same_invocation"] == 1.0 and o["weak_scaling_eff"] == 1.5
    r, dt = _bench(["--gpus", "2", "--dry-run", "--n1-value", "2.0"])
    o = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert o["n1_value_same_invocation"] == 2.0 and o["weak_scaling_eff"] == 0.75


def test_scaling_prediction_arithmetic():
    """bench.py --emulate-ranks: the prediction is the contract's own arithmetic (value = units of all ranks / max time over ranks;
    efficiency = value / (R x the N = 1 value)) applied to the R shards' separately measured times."""
    from bench import scaling_prediction
    p = scaling_prediction([100.0] * 8, [5e7] * 8)  # balanced shards: efficiency 1
    assert p["predicted_weak_scaling_eff"] == 1.0 and p["predicted_value"] == 8 * 5e8 and p["n1_value"] == 5e8
    p = scaling_prediction([100.0, 100.0, 100.0, 200.0], [5e7] * 4)  # one shard twice as slow: the job takes its time
    assert p["predicted_value"] == 4 * 5e7 / 0.2 and p["predicted_weak_scaling_eff"] == 0.5
    assert abs(p["predicted_weak_scaling_eff_vs_mean_shard"] - (1e9 / 4) / np.mean([5e8, 5e8, 5e8, 2.5e8])) < 1e-12
    assert p["time_max_over_mean"] == 200.0 / 125.0
    p = scaling_prediction([200.0, 100.0], [5e7, 5e7])  # shard 0 (the N = 1 job) is the slow one: the others do not make the job faster than it
    assert p["predicted_weak_scaling_eff"] == 1.0


def test_bench_self_launcher_eight_ranks():
    """The node's full rank count (8 GPUs): eight fresh children rendezvous over gloo, reduce, and the final gather
    lands every rank's shard at its global chain offset (`chain_offset = rank * C`) on rank 0."""
    import json
    r, dt = _bench(["--gpus", "8", "--dry-run"], timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    o = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert o["n_gpus"] == 8 and o["value"] == 36.0 and o["gathered_rows"] == 24
    assert o["gathered_in_global_chain_order"] and o["gathered_rank_of_row"] == [k // 3 for k in range(24)]


def test_bench_self_launcher_ends_the_job_when_a_rank_dies():
    """A rank that exits before the rendezvous leaves its sibling waiting in init_process_group (gloo's store timeout is
    half an hour): the parent polls every child, ends the survivors and exits non-zero at once."""
    r, dt = _bench(["--gpus", "2", "--dry-run"], {"WN_BENCH_FAIL_RANK": "1"})
    assert r.returncode == 3, (r.returncode, r.stdout, r.stderr)
    assert dt < 60.0, dt
    assert "rank 1 exited with code 3" in r.stderr


def test_bench_self_launcher_time_limit():
    """Ranks that never finish are ended at --child-timeout: rank 1 sleeps through the rendezvous, rank 0 waits for it in
    init_process_group; the parent ends both at the limit and exits 124."""
    r, dt = _bench(["--gpus", "2", "--dry-run", "--child-timeout", "6"], {"WN_BENCH_SLEEP_RANK": "1"})
    assert r.returncode == 124, (r.returncode, r.stdout, r.stderr)
    assert "did not finish within 6 s" in r.stderr, r.stderr
    assert 5.0 < dt < 30.0, dt
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
