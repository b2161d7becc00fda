"""Multi-GPU semantics checked on ONE GPU.

  * sub-band sharding: the 8 "ranks" of BASELINE cfg 4, run one after the other, concatenated == ONE full-band run of the
    NCHAN-8 complex file (dspsr -F 4096:D), bit for bit, with identical hits on every rank;
  * time-slice replicas (single-channel input): blocks dealt round robin to replicas and combined like
    PhaseSeries::combine == one pipeline folding every block: identical hits, profile equal to float rounding;
  * bench.py --gpus 2 as ONE command (self-spawned ranks, gloo on the single device) for both modes, with the parity gate.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    import dspsr_amd
    return dspsr_amd


def _raw(n, seed):
    rng = np.random.default_rng(seed)
    return np.clip(np.rint(rng.standard_normal(n) * 24.0), -128, 127).astype(np.int8)


@pytest.mark.parametrize("force_fused", [False, True])
def test_subband_shards_equal_fullband(oracle, gpu, force_fused):
    """(force_fused False: 32 channel tiles per sub-band -> the segmented fused fold; True: one workgroup per tile, exact.)"""
    from dspsr_amd import pipeline
    nsub, nblocks = 8, 2
    info = pipeline.InputInfo(centre_frequency=1382.0, bandwidth=-400.0, nchan=nsub, npol=2, ndim=2, tsamp_us=0.02,
                              machine="DADA")
    cfg = pipeline.Config(nchan=4096, dispersion_measure=1000.0, nbin=1024, folding_period=0.0893, freq_res=512, ndim=4,
                          parts_per_block=6, max_parts=4, force_fused=force_fused)
    stream = torch.cuda.current_stream().cuda_stream
    full = pipeline.LoadToFold(cfg, info, device=0, stream=stream)
    # SURVEY Appendix B cfg 4: the common 27/27 of the full-band kernel (worst channel), nkeep 458, step 234 496
    assert (full.response.impulse_pos, full.response.impulse_neg, full.nkeep, full.nsamp_step) == (27, 27, 458, 234496)
    # the oracle's Dedispersion on the whole band gives the same kernel and the same common discard counts
    obs = oracle.Observation(centre_frequency=1382.0, bandwidth=-400.0, nchan=nsub, npol=2, ndim=2, tsamp_us=0.02,
                             dispersion_measure=1000.0)
    oresp = oracle.Dedispersion()
    oresp.set_frequency_resolution(512)
    oresp.match(obs, 4096)
    assert (oresp.impulse_pos, oresp.impulse_neg) == (27, 27)
    assert np.abs(full.response.kernel - oresp.buffer).max() <= 1.2e-7
    step = cfg.parts_per_block * full.nsamp_step
    ndat = nblocks * step + full.nsamp_overlap
    raw = _raw(ndat * nsub * 4, 77)                      # generic order ((t*8 + c)*2 + p)*2 + d
    d_full = torch.from_numpy(raw).cuda()
    bps = nsub * 4                                       # bytes per time sample of the 8-channel stream
    for b in range(nblocks):
        full.process_block(d_full[b * step * bps:(b * step + step + full.nsamp_overlap) * bps])
    full.finish_subint()
    full.synchronize()
    want = full.subints[0]
    wprof = want["profile_dev"].view(4096, -1)
    assert int(want["hits"].sum()) == nblocks * cfg.parts_per_block * full.nkeep and float(wprof.abs().max()) > 0
    # the folded profile of the full-band run (== the concatenated shards, below) against the float64 oracle on a few channels
    # of three sub-bands: Filterbank.C:561-662 on the NCHAN-8 complex input, cross_detect, Fold.C:835-891; hits identical
    import dataclasses
    oplan = oracle.filterbank_plan(obs, 4096, oresp)
    fobs = oracle.filterbank_output_observation(obs, oplan)
    assert (oplan.nkeep, oplan.nsamp_step, fobs.rate, fobs.start_seconds) == (full.nkeep, full.nsamp_step, full.out_rate, full.out_start)
    unpacked = oracle.unpack_8bit(raw, obs)                               # [8][2][ndat*2]
    npart = nblocks * cfg.parts_per_block
    nblk = cfg.parts_per_block * full.nkeep
    for g, chans in ((0, [0, 1, 255]), (3, [17, 511]), (7, [0, 510, 511])):
        sub_plan = dataclasses.replace(oplan, nchan=512, input_nchan=1)
        fbk = oracle.filterbank(unpacked[g:g + 1].astype(np.float64), sub_plan, full.response.kernel[g * 512 * 512:(g + 1) * 512 * 512],
                                npart=npart, dtype=np.float64)
        det = oracle.detect_layout(oracle.detect_products(fbk[chans], "Coherence"), 4)       # [chan][1][ndat][4] float64
        ps = oracle.PhaseSeries(len(chans), 1, 4, 1024, data=np.zeros((len(chans), 1, 1024, 4), np.float64))
        for b in range(nblocks):
            oracle.fold(det, fobs, oracle.FoldConfig(nbin=1024, folding_period=0.0893), ps, idat_start=b * nblk, ndat_fold=nblk)
        assert np.array_equal(want["hits"], ps.hits), g
        got = wprof.view(4096, 1024, 4)[[g * 512 + c for c in chans]].cpu().numpy()
        err = np.abs(got - ps.data[:, 0]).max() / np.abs(ps.data[..., :2]).max()
        assert err <= 1e-5, (g, err)
    for g in range(nsub):
        lt = pipeline.LoadToFold(cfg, info, device=0, stream=stream, subband=g)
        assert lt.fused_mode == (1 if force_fused else 2)
        assert (lt.nkeep, lt.nsamp_step, lt.out_start, lt.out_rate) == (full.nkeep, full.nsamp_step, full.out_start, full.out_rate)
        mine = torch.from_numpy(np.ascontiguousarray(raw.reshape(ndat, nsub, 4)[:, g, :]).reshape(-1)).cuda()   # what rank g reads
        for b in range(nblocks):
            lt.process_block(mine[b * step * 4:(b * step + step + lt.nsamp_overlap) * 4])
        lt.finish_subint()
        lt.synchronize()
        got = lt.subints[0]
        assert np.array_equal(got["hits"], want["hits"]), g                       # sample aligned: identical hits
        assert got["integration_length"] == want["integration_length"] and got["ndat_total"] == want["ndat_total"]
        assert torch.equal(got["profile_dev"].view(512, -1), wprof[g * 512:(g + 1) * 512]), g
        lt.close()
    full.close()


def test_time_slice_replicas_combine(gpu):
    """Single-channel input: 2 replicas take alternate blocks (MultiThread.C:120-148) and are combined with
    PhaseSeries::combine semantics; one pipeline folds all blocks in order.  hits / ndat_total identical, profile equal
    to rounding (the partial sums associate differently, exactly as in the reference's own -t N runs)."""
    from dspsr_amd import pipeline, synth
    freq, bw, tsamp, dm, period, nchan, nbin = 1382.0, -16.0, 1.0 / 32.0, 30.0, 0.004, 16, 64
    info = pipeline.InputInfo(centre_frequency=freq, bandwidth=bw, tsamp_us=tsamp, machine="DADA")
    cfg = pipeline.Config(nchan=nchan, dispersion_measure=dm, nbin=nbin, folding_period=period, ndim=4, parts_per_block=3,
                          max_parts=2)
    stream = torch.cuda.current_stream().cuda_stream
    one = pipeline.LoadToFold(cfg, info, device=0, stream=stream)
    nblocks = 6
    step = cfg.parts_per_block * one.nsamp_step
    raw = torch.from_numpy(synth.voltages(nblocks * step + one.nsamp_overlap, freq, bw, tsamp, dm, period)).cuda()
    blk = lambda b: raw[2 * b * step: 2 * (b * step + step + one.nsamp_overlap)]
    for b in range(nblocks):
        one.process_block(blk(b))
    one.finish_subint()
    want = one.subints[0]
    reps = [pipeline.LoadToFold(cfg, info, device=0, stream=stream) for _ in range(2)]
    for b in range(nblocks):
        r = reps[b % 2]
        r.seek_block(b)
        r.process_block(blk(b))
    acc = None
    for r in reps:
        r.finish_subint()
        acc = pipeline.combine_phase_series(acc, r.subints[0])
        r.close()
    assert np.array_equal(acc["hits"], want["hits"]) and acc["ndat_total"] == want["ndat_total"]
    assert abs(acc["integration_length"] - want["integration_length"]) <= 1e-12 * want["integration_length"]
    w = want["profile_dev"].cpu().numpy()
    assert np.abs(acc["profile"].reshape(-1) - w).max() <= 2e-6 * np.abs(w).max()
    one.close()


def test_rccl_communicator_world1(gpu):
    """The product's exchange (dspsr_amd_comm_* / dspsr_amd_reduce_profiles_*, csrc/comm.hip) through RCCL with a one-rank
    communicator -- what a single GPU can execute of it: library load, ncclCommInitRank, the snapshot on the compute stream,
    ncclGather / the packed ncclReduce on the communicator's stream, the MIN/MAX hits check, the copy to pinned memory.
    (i) engine level, padded profile rows, both modes, profile zeroed right after start() (the snapshot must already hold);
    (ii) pipeline level: finish_subint through the communicator == finish_subint without, asynchronous dumps collected in
    order."""
    import dspsr_amd
    from dspsr_amd import pipeline, synth
    ctx = dspsr_amd.Context(0, torch.cuda.current_stream().cuda_stream)
    uid = dspsr_amd.Communicator.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = dspsr_amd.Communicator(ctx, 1, 0, uid)
    assert dspsr_amd.lib.dspsr_amd_comm_size(comm.handle) == 1 and dspsr_amd.lib.dspsr_amd_comm_rank(comm.handle) == 0
    rng = np.random.default_rng(5)
    nrow, row, span, nbin = 24, 40, 48, 10
    prof = torch.from_numpy(rng.standard_normal((nrow, span)).astype(np.float32)).cuda()
    want = prof[:, :row].cpu().numpy().reshape(-1).copy()
    hits = rng.integers(0, 1 << 31, nbin).astype(np.uint32)
    for mode in (comm.SUM, comm.GATHER):
        p = prof.clone()
        comm.start(mode, p.data_ptr(), span, nrow, row, hits, 12.625, (1 << 40) + 7, check_hits=True)
        p.zero_()                                                    # stream ordered behind the snapshot
        got, h, length, ndat, same = comm.finish()
        assert same and np.array_equal(got, want) and np.array_equal(h, hits) and length == 12.625 and ndat == (1 << 40) + 7
    comm.start(comm.SUM, prof.data_ptr(), span, nrow, row, hits, 1.0, 5, check_hits=False)
    view = comm.finish(copy=False)[0]                                # in place: the communicator's pinned buffer
    assert not view.flags.owndata and np.array_equal(view, want)
    with pytest.raises(dspsr_amd.DspsrAmdError):
        comm.finish()                                                # nothing in flight
    # ---- pipeline level
    freq, bw, tsamp, dm, period, nchan, nbin = 1382.0, -16.0, 1.0 / 32.0, 30.0, 0.004, 16, 64
    info = pipeline.InputInfo(centre_frequency=freq, bandwidth=bw, tsamp_us=tsamp, machine="DADA")
    cfg = pipeline.Config(nchan=nchan, dispersion_measure=dm, nbin=nbin, folding_period=period, ndim=4, parts_per_block=3, max_parts=2)
    stream = torch.cuda.current_stream().cuda_stream
    res = {}
    for replicas in (False, True):
        for use_comm in (False, True):
            lt = pipeline.LoadToFold(cfg, info, device=0, stream=stream)
            step = cfg.parts_per_block * lt.nsamp_step
            raw = torch.from_numpy(synth.voltages(4 * step + lt.nsamp_overlap, freq, bw, tsamp, dm, period)).cuda()
            c2 = dspsr_amd.Communicator(lt.ctx, 1, 0, dspsr_amd.Communicator.unique_id()) if use_comm else None
            if use_comm:
                lt.set_rccl_communicator(c2)
            for b in range(4):
                lt.process_block(raw[2 * b * step: 2 * (b * step + step + lt.nsamp_overlap)])
                if b % 2 == 1:
                    lt.finish_subint(None, 0, 1, None, replicas=replicas, wait=False)     # in flight while the next block runs
            lt.collect_subint()
            lt.synchronize()
            assert len(lt.subints) == 2
            res[use_comm] = [(s_["hits"].copy(), s_["integration_length"], s_["ndat_total"],
                              s_["profile"].reshape(-1) if "profile" in s_ else s_["profile_dev"].cpu().numpy().reshape(-1)) for s_ in lt.subints]
            if c2 is not None:
                c2.close()
            lt.close()
        for a, b in zip(res[False], res[True]):
            assert np.array_equal(a[0], b[0]) and a[1] == b[1] and a[2] == b[2] and np.array_equal(a[3], b[3])
            assert int(a[0].sum()) == a[2] and float(np.abs(a[3]).max()) > 0
    comm.close()
    ctx.close()


def test_bench_rccl_setup_and_agreed_fallback(gpu, monkeypatch):
    """bench.open_rccl_exchange: communicator + trial exchange over a (one-rank) process group; when the communicator cannot
    be made every rank falls back together and the reason is kept for the bench line."""
    import torch.distributed as dist
    import dspsr_amd
    sys.path.insert(0, ROOT)
    import bench
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29617", rank=0, world_size=1)
    try:
        ctx = dspsr_amd.Context(0, torch.cuda.current_stream().cuda_stream)
        comm, note = bench.open_rccl_exchange(ctx, torch, dist, 0, 1)
        assert comm is not None and note is None
        comm.close()

        def refuse(*a, **k):
            raise dspsr_amd.DspsrAmdError("no librccl here")
        monkeypatch.setattr(dspsr_amd.Communicator, "unique_id", staticmethod(refuse))
        comm, note = bench.open_rccl_exchange(ctx, torch, dist, 0, 1)
        assert comm is None and "no librccl here" in note and note.startswith("dlopen")
        monkeypatch.undo()
        # a rank that fails in comm_create: the ranks agree on that BEFORE anyone enters the trial collective
        real_init = dspsr_amd.Communicator.__init__

        def refuse_init(self, *a, **k):
            raise dspsr_amd.DspsrAmdError("ncclCommInitRank refused")
        monkeypatch.setattr(dspsr_amd.Communicator, "__init__", refuse_init)
        comm, note = bench.open_rccl_exchange(ctx, torch, dist, 0, 1)
        assert comm is None and note.startswith("comm_create") and "refused" in note
        monkeypatch.setattr(dspsr_amd.Communicator, "__init__", real_init)
        # ... and one whose trial exchange fails
        def bad_finish(self, *a, **k):
            raise dspsr_amd.DspsrAmdError("collective failed")
        monkeypatch.setattr(dspsr_amd.Communicator, "finish", bad_finish)
        comm, note = bench.open_rccl_exchange(ctx, torch, dist, 0, 1)
        assert comm is None and note.startswith("trial exchange") and "collective failed" in note
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("workload,extra", [("cfg4", ["--parts-per-block", "16"]),
                                            ("target", ["--parts-per-block", "4", "--max-parts", "2"]),
                                            (None, [])])
def test_bench_two_ranks_one_command(gpu, workload, extra):
    """bench.py --gpus 2 without a launcher: it spawns its own ranks (gloo, both on this GPU), runs the sub-integration
    exchange inside the timed region, passes the parity gate and prints ONE JSON line.  workload None = the driver's own
    command (no --workload): the headline as time-slice replicas in `value` AND the sub-band sharded cfg4 run -- the
    north-star scaling curve -- in `subband_shard`, both from the one process group."""
    env = dict(os.environ, DSPSR_AMD_SINGLE_DEVICE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--dump-steps", "2",
           "--no-cpu-baseline"] + (["--workload", workload] if workload else []) + extra
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["value"] > 0 and res["parity_gate"]["status"] == "ok"
    assert ("sub-band" if workload == "cfg4" else "replicas") in res["config"]["parallelism"]
    assert res["config"]["reduce_ms_per_dump"] > 0
    # which transport carried the dumps is a top-level field (here: the one-device gloo rehearsal, never "rccl-c-abi")
    assert res["exchange"] == "gloo-rehearsal" and res["config"]["exchange"] == "gloo-rehearsal"
    if workload is None:
        assert res["subband_shard"]["exchange"] == "gloo-rehearsal"
        assert res["config"]["workload"] == "target"
        sh = res["subband_shard"]
        assert sh["workload"] == "cfg4" and sh["value"] > 0 and sh["ms_per_step"] > 0 and sh["reduce_ms_per_dump"] > 0
        assert sh["identical_hits"] is True and sh["parity_gate"] == "ok" and "sub-band" in sh["parallelism"]


def test_subband_shards_with_interchannel_dedispersion(gpu):
    """-K (SampleDelay + fractional-delay phase) on a sub-band sharded run: every rank applies its channels' delays relative
    to the zero of the WHOLE band and gives up the band's total delay, so the ranks stay sample aligned -- profiles, hits and
    start time equal one full-band run bit for bit."""
    from dspsr_amd import pipeline
    nsub, nblocks = 4, 3
    info = pipeline.InputInfo(centre_frequency=1400.0, bandwidth=-64.0, nchan=nsub, npol=2, ndim=2, tsamp_us=1.0 / 16.0,
                              machine="DADA")
    cfg = pipeline.Config(nchan=64, dispersion_measure=3.0, nbin=128, folding_period=0.0123, freq_res=256, ndim=4,
                          parts_per_block=5, max_parts=4, interchan_dedispersion=True)
    stream = torch.cuda.current_stream().cuda_stream
    full = pipeline.LoadToFold(cfg, info, device=0, stream=stream)
    assert full.sample_delay is not None and full.sd_head > 8            # the band really is delayed by several samples
    step = cfg.parts_per_block * full.nsamp_step
    ndat = nblocks * step + full.nsamp_overlap
    raw = _raw(ndat * nsub * 4, 5)
    d_full = torch.from_numpy(raw).cuda()
    bps = nsub * 4
    for b in range(nblocks):
        full.process_block(d_full[b * step * bps:(b * step + step + full.nsamp_overlap) * bps])
    full.finish_subint()
    full.synchronize()
    want = full.subints[0]
    wprof = want["profile_dev"].view(64, -1)
    assert want["ndat_total"] == nblocks * cfg.parts_per_block * full.nkeep - full.sd_head and float(wprof.abs().max()) > 0
    shorts = []
    for g in range(nsub):
        lt = pipeline.LoadToFold(cfg, info, device=0, stream=stream, subband=g)
        assert (lt.sd_head, lt.out_start) == (full.sd_head, full.out_start)
        shorts.append(lt.sd_short)
        mine = torch.from_numpy(np.ascontiguousarray(raw.reshape(ndat, nsub, 4)[:, g, :]).reshape(-1)).cuda()
        for b in range(nblocks):
            lt.process_block(mine[b * step * 4:(b * step + step + lt.nsamp_overlap) * 4])
        lt.finish_subint()
        lt.synchronize()
        got = lt.subints[0]
        assert np.array_equal(got["hits"], want["hits"]) and got["ndat_total"] == want["ndat_total"], g
        assert torch.equal(got["profile_dev"].view(16, -1), wprof[g * 16:(g + 1) * 16]), g
        lt.close()
    assert min(shorts) == 0 and max(shorts) > 0          # one rank holds the band's extreme channel, the others give up more
    full.close()
