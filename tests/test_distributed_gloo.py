"""World-size-2 CPU test (gloo) of the one collective on the path: the per-sub-integration reduce of
per-sub-band folded profiles (dspsr_amd.pipeline.reduce_subbands), i.e. the code bench.py runs over
RCCL when launched with --gpus N."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nfloat, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dspsr_amd.pipeline import reduce_subbands
    rng = np.random.default_rng(100 + rank)
    gather = torch.zeros(world * nfloat, dtype=torch.float32)
    results = []
    for subint in range(3):                       # several dumps re-use the buffer
        prof = torch.from_numpy(rng.standard_normal(nfloat).astype(np.float32))
        res = reduce_subbands(prof, dist, rank, world, gather)
        if rank == 0:
            results.append(res.clone().numpy())
        else:
            assert res is None
    if rank == 0:
        np.save(out_path, np.stack(results))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])          # 8: the node the sub-band config is written for
def test_subband_reduce_gloo(tmp_path, world):
    nfloat = 4 * 3 * 16 * 4          # nchan*npol*nbin*ndim of one sub-band
    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(world, _free_port(), nfloat, out), nprocs=world, join=True)
    got = np.load(out)
    for subint in range(3):
        want = []
        gens = [np.random.default_rng(100 + r) for r in range(world)]
        for r in range(world):
            for _ in range(subint + 1):
                v = gens[r].standard_normal(nfloat).astype(np.float32)
            want.append(v)
        # every rank's slice arrives bit-exact: the other ranks contribute exact zeros to the sum
        assert np.array_equal(got[subint], np.concatenate(want))


def _replica_worker(rank, world, port, nbin, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dspsr_amd.pipeline import DspsrAmdError, check_identical_hits, reduce_replicas
    rng = np.random.default_rng(7 + rank)
    prof = torch.from_numpy(rng.standard_normal(3 * nbin * 4).astype(np.float32))
    hits = rng.integers(0, 50, nbin).astype(np.uint32)
    res = reduce_replicas(prof, hits, 0.25 * (rank + 1), 1000 + rank, dist, rank, world)
    # sub-band ranks: identical hits pass, diverging hits raise on EVERY rank (no rank is left waiting)
    check_identical_hits(np.arange(nbin), dist, rank, world)
    raised = False
    try:
        check_identical_hits(np.arange(nbin) + (rank == world - 1), dist, rank, world)
    except DspsrAmdError:
        raised = True
    assert raised
    if rank == 0:
        np.savez(out_path, prof=res[0].numpy(), hits=res[1], length=res[2], ndat=res[3])
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_time_slice_replica_reduce_gloo(tmp_path, world):
    """PhaseSeries::combine semantics over the collective (PhaseSeries.C:442-484): profiles, hits, integration_length and
    ndat_total of all replicas ADD onto rank 0."""
    nbin = 16
    out = str(tmp_path / "res.npz")
    mp.spawn(_replica_worker, args=(world, _free_port(), nbin, out), nprocs=world, join=True)
    got = np.load(out)
    prof = np.zeros(3 * nbin * 4, np.float32)
    hits = np.zeros(nbin, np.uint32)
    for r in range(world):
        rng = np.random.default_rng(7 + r)
        prof = prof + rng.standard_normal(3 * nbin * 4).astype(np.float32)
        hits += rng.integers(0, 50, nbin).astype(np.uint32)
    assert np.array_equal(got["hits"], hits) and int(got["ndat"]) == sum(1000 + r for r in range(world))
    assert float(got["length"]) == sum(0.25 * (r + 1) for r in range(world))
    assert np.abs(got["prof"] - prof).max() <= 1e-6 * np.abs(prof).max()


def test_bench_gpus_n_is_one_command_and_fails_loudly():
    """bench.py --gpus 2 spawns its own ranks; without a HIP device every rank refuses to run (no CPU path) and the
    parent returns non-zero instead of hanging or printing a number."""
    import subprocess
    import sys
    import sys
    if torch.cuda.is_available():
        pytest.skip("a GPU is present (the GPU suite runs the real thing)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "needs a HIP device" in p.stderr and not any(ln.startswith("{") for ln in p.stdout.splitlines())


def test_reduce_single_rank_is_identity():
    from dspsr_amd.pipeline import reduce_subbands
    p = torch.arange(10, dtype=torch.float32)
    assert reduce_subbands(p) is p


def test_bench_bounded_wait_exits_instead_of_hanging():
    import subprocess
    import sys
    """bench._bounded: a collective set-up step whose peers never arrive (ncclCommInitRank with a rank missing) must end the
    rank with a non-zero exit code and a message, not hang the 8-GPU run; a step that returns in time hands back its result
    or its exception."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "r, e = bench._bounded('fast step', lambda: 7, 0, timeout=5.0); assert r == 7 and e is None\n"
            "r, e = bench._bounded('failing step', lambda: 1 / 0, 0, timeout=5.0); assert r is None and isinstance(e, ZeroDivisionError)\n"
            "print('before', flush=True)\n"
            "bench._bounded('stuck step', lambda: time.sleep(60), 3, timeout=0.3)\n"
            "print('not reached', flush=True)\n" % root)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == 3, (p.returncode, p.stderr[-500:])
    assert "before" in p.stdout and "not reached" not in p.stdout
    assert "rank 3: stuck step did not return" in p.stderr
