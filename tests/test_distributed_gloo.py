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


@pytest.mark.parametrize("world", [2, 3])
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


def test_reduce_single_rank_is_identity():
    from dspsr_amd.pipeline import reduce_subbands
    p = torch.arange(10, dtype=torch.float32)
    assert reduce_subbands(p) is p
