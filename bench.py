#!/usr/bin/env python3
"""Benchmark of the hot path: 8-bit dual-pol voltages -> coherent-dedispersion filterbank (-F N:D)
-> detection -> fold, on N MI355X GPUs (one process per GPU).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload target|cfg3|cfg2|cfg4]

A "step" is one pass of the hot path over one block (parts_per_block overlap-save parts) of synthetic
8-bit input already resident in HBM.  metric = Msamples/s = unique input time samples per polarisation
consumed per second, summed over all ranks.  With N > 1 every rank holds one
frequency sub-band of the same geometry (weak scaling) and ONE RCCL reduce of the folded profiles is
done per sub-integration dump (every --dump-steps steps), inside the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) including
  roofline     : algorithmic bytes of the FFT+chirp launch group / its measured duration (HIP events)
  cpu_baseline : the numpy oracle timed on a bounded sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md

# name -> (freq MHz, bw MHz, input nchan, ndim, tsamp_us, nchan out, DM, -x freq_res, nbin)
WORKLOADS = {
    # north-star headline: -F 1024:D -x 4096, DM 1000 on Benchmark/header.dada (SURVEY Appendix B "target")
    "target": dict(freq=1382.0, bw=-400.0, in_nchan=1, ndim=1, tsamp_us=0.00125, nchan=1024, dm=1000.0,
                   freq_res=4096, nbin=1024, machine="CASPSR",
                   cmd="dspsr -F 1024:D -x 4096 -D 1000 -b 1024 (header.dada band, 8-bit dual-pol real)"),
    # BASELINE cfg 1 (the reference's CPU-runnable case): -F 64:D on header.dada, vela.par DM, minimum response
    # length 16384 -> the four-pass path (two-pass inverse), no fused fold
    "cfg1": dict(freq=1382.0, bw=-400.0, in_nchan=1, ndim=1, tsamp_us=0.00125, nchan=64, dm=67.99,
                 freq_res=16384, nbin=512, machine="CASPSR",
                 cmd="dspsr -F 64:D -x 16384 -D 67.99 -b 512 (header.dada band, vela.par DM)"),
    "cfg1opt": dict(freq=1382.0, bw=-400.0, in_nchan=1, ndim=1, tsamp_us=0.00125, nchan=64, dm=67.99,
                    freq_res=262144, nbin=512, machine="CASPSR",
                    cmd="dspsr -F 64:D -D 67.99 -b 512 with the optimal response length 262144 (N = 2^24)"),
    "cfg3": dict(freq=1382.0, bw=-400.0, in_nchan=1, ndim=1, tsamp_us=0.00125, nchan=1024, dm=2000.0,
                 freq_res=4096, nbin=1024, machine="CASPSR",
                 cmd="dspsr -F 1024:D -x 4096 -D 2000 -b 1024"),
    "cfg2": dict(freq=2000.0, bw=-400.0, in_nchan=1, ndim=1, tsamp_us=0.00125, nchan=256, dm=500.0,
                 freq_res=4096, nbin=1024, machine="CASPSR",
                 cmd="dspsr -F 256:D -x 4096 -D 500 -b 1024 (band moved to 2000 MHz, SURVEY Appendix B)"),
    # one 50 MHz complex sub-band per rank of an 8-channel 400 MHz band
    "cfg4": dict(freq=1382.0, bw=-50.0, in_nchan=1, ndim=2, tsamp_us=0.02, nchan=512, dm=1000.0,
                 freq_res=512, nbin=1024, machine="DADA",
                 cmd="dspsr -F 512:D -D 1000 -b 1024 per 50 MHz complex sub-band"),
    # SURVEY 8f-1 / BASELINE config 5: search-mode front end, detect only (no fold)
    "cfg5": dict(freq=1382.0, bw=-400.0, in_nchan=1, ndim=1, tsamp_us=0.00125, nchan=4096, dm=0.0, freq_res=1,
                 nbin=0, machine="DADA", tscrunch=16, nparts=32768,
                 cmd="digifil -F 4096 -t 16 -b 8 (TFP filterbank + square law + tscrunch, Rescale, PScrunch, 8-bit digitizer; no fold)"),
}


def bench_search_mode(args, wl, torch):
    """cfg5: one step = one block of nparts FFT blocks (2*nchan samples each) through digifil's chain
    (dspsr_amd.pipeline.LoadToFil): TFP filterbank + square law + tscrunch [one kernel, the roofline kernel] ->
    Rescale -> PScrunch -> 8-bit SigProcDigitizer, everything resident in HBM."""
    from dspsr_amd import pipeline
    nchan, sf, npart = wl["nchan"], wl["tscrunch"], wl["nparts"]
    info = pipeline.InputInfo(centre_frequency=wl["freq"], bandwidth=wl["bw"], nchan=1, npol=2, ndim=1, tsamp_us=wl["tsamp_us"],
                              machine=wl["machine"])
    lf = pipeline.LoadToFil(pipeline.SearchConfig(nchan=nchan, tscrunch=sf, nbit=8, parts_per_block=npart), info, device=0,
                            stream=torch.cuda.current_stream().cuda_stream)
    ctx = lf.ctx
    nbytes = lf.block_bytes()
    gen = torch.Generator(device="cuda").manual_seed(20100413)
    raw = torch.randn(nbytes, generator=gen, device="cuda").mul_(24.0).round_().clamp_(-128, 127).to(torch.int8)
    out = lf.detected

    def step():
        lf.process_block(raw)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        step()
        b.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    k_ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
    samples = npart * 2 * nchan
    # the front-end kernel alone, timed on extra calls (the step above also runs Rescale, PScrunch and the digitizer on
    # the 16x smaller scrunched block)
    import dspsr_amd
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(max(3, args.steps // 4))]
    for a, b in ev:
        a.record()
        dspsr_amd.tfp_filterbank(ctx, raw, nchan, npart, out, False, sf, lf.layout, lf.scale8)
        b.record()
    torch.cuda.synchronize()
    k_ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
    b_alg = nbytes + out.numel() * 4
    achieved = b_alg / (k_ms * 1e-3) / 1e9
    res = {"metric": "Msamples/s dedispersed+folded", "value": round(samples * args.steps / elapsed / 1e6, 2),
           "unit": "Msamples/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "cfg5", "command": wl["cmd"], "nchan": nchan, "tscrunch": sf,
                      "parts_per_block": npart, "input": "8-bit dual-pol, resident in HBM",
                      "chain": "TFPFilterbank+detect+TScrunch -> Rescale(10 s) -> PScrunch -> SigProcDigitizer(8 bit)",
                      "note": "search mode: detected, scrunched and digitised, NOT folded"},
           "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "kernel": "k_tfp<13>",
                        "algorithmic_bytes_per_launch": b_alg, "kernel_ms": round(k_ms, 4)}}
    if not args.no_cpu_baseline:
        import oracle.dspsr_oracle as o
        n = 512
        rr = raw[: n * 4 * nchan].cpu().numpy()
        t1 = time.perf_counter()
        det = o.tscrunch_tfp(o.tfp_filterbank(o.unpack_8bit(rr, o.Observation()), nchan, False), sf)
        o.sigproc_digitize(o.pscrunch_tfp(o.Rescale().transform(det)), 8, flip_band=wl["bw"] > 0)
        dt = time.perf_counter() - t1
        res["cpu_baseline"] = {"value": n * 2 * nchan / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
                               "sample": "%d FFT blocks of the same workload, numpy oracle, %.1f s" % (n, dt)}
    print(json.dumps(res), flush=True)
    lf.close()


def algorithmic_bytes_per_part(npol, nsamp_fft, nbit, N, nchan_subband, nkeep):
    """each input byte read once, chirp read once, each kept output written once."""
    return npol * nsamp_fft * nbit // 8 + 8 * N + npol * nchan_subband * nkeep * 8


def measured_traffic(workload, max_parts, key="hbm_bytes_per_launch_group"):
    """HBM bytes per launch group from the committed PMC profile (rocprofv3 cannot run inside the bench);
    None when the profile was taken for another workload / grouping."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):   # newest round first
        try:
            d = json.load(open(path))
            if d["workload"] == workload and d["parts_per_launch_group"] == max_parts and key in d:
                measured_traffic.source = os.path.relpath(path, ROOT)
                return d[key]
        except Exception:
            pass
    return None


measured_traffic.source = None


def _cpu_port_run(args):
    """One CPU worker of the baseline: the numpy oracle (oracle/dspsr_oracle.py, a 'port' of the reference's CPU
    path) on `nparts` overlap-save parts of the workload: unpack -> filterbank+chirp -> detect -> fold.
    Runs in a spawned process (no torch / HIP in the children); nparts == 0 is the warm-up call."""
    wl, nparts, seed = args
    import oracle.dspsr_oracle as o
    obs = o.Observation(centre_frequency=wl["freq"], bandwidth=wl["bw"], nchan=wl["in_nchan"], npol=2,
                        ndim=wl["ndim"], tsamp_us=wl["tsamp_us"], machine=wl["machine"],
                        dispersion_measure=wl["dm"])
    resp = o.Dedispersion()
    resp.set_frequency_resolution(wl["freq_res"])
    resp.match(obs, wl["nchan"])
    plan = o.filterbank_plan(obs, wl["nchan"], resp)
    if nparts == 0:
        np.fft.rfft(np.zeros(1024, np.float32))
        return 0, 0.0
    ndat = nparts * plan.nsamp_step + plan.nsamp_overlap
    rng = np.random.default_rng(seed)
    raw = np.clip(np.rint(rng.standard_normal(ndat * wl["in_nchan"] * 2 * wl["ndim"]) * 24), -128, 127).astype(np.int8)
    t0 = time.perf_counter()
    unpacked = o.unpack_8bit(raw, obs)
    fb = o.filterbank(unpacked, plan, resp.buffer, dtype=np.float32)
    det = o.detect_layout(o.detect_products(fb, "Coherence"), 4)
    fobs = o.filterbank_output_observation(obs, plan)
    # fold with the vectorised equivalent of the sequential loop (np.add.at keeps time order per bin)
    phi, pfold = o.fold_phase(o.FoldConfig(nbin=wl["nbin"], folding_period=0.089), fobs, fobs.start_seconds)
    binplan = o.fold_binplan(phi, (1.0 / fobs.rate) / pfold, wl["nbin"], det.shape[2])
    prof = np.zeros((det.shape[0], wl["nbin"], 4), np.float32)
    np.add.at(prof, (slice(None), binplan), det[:, 0])
    return nparts * plan.nsamp_step, time.perf_counter() - t0


def cpu_baseline(wl, lt, parts_per_worker=2):
    """Times the CPU port on the host cores of this node with the reference's own parallelisation model --
    one worker per time block (dspsr -t <ncores>, MultiThread.C:65-82) -- on a bounded sample of the workload."""
    import multiprocessing as mp
    try:
        ncore = len(os.sched_getaffinity(0))
    except AttributeError:
        ncore = os.cpu_count() or 1
    ncore = max(1, min(ncore, 16))
    wl = {k: v for k, v in wl.items() if k != "cmd"}
    ctx = mp.get_context("spawn")         # the parent holds a HIP context: never fork it
    with ctx.Pool(ncore) as pool:
        pool.map(_cpu_port_run, [(wl, 0, 0)] * ncore)                      # start-up, imports, response build
        t0 = time.perf_counter()
        res = pool.map(_cpu_port_run, [(wl, parts_per_worker, 1 + i) for i in range(ncore)], chunksize=1)
        wall = time.perf_counter() - t0
    samples = sum(r[0] for r in res)
    return {"value": samples / wall / 1e6, "unit": "Msamples/s", "cores": ncore, "kind": "port",
            "sample": "%d workers x %d overlap-save part(s) of the same workload (%.1f Msamples/pol in all), numpy "
                      "oracle with pocketfft float32, one process per time block, %.1f s wall, %.1f core-seconds"
                      % (ncore, parts_per_worker, samples / 1e6, wall, sum(r[1] for r in res))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="target", choices=sorted(WORKLOADS))
    ap.add_argument("--parts-per-block", type=int, default=0,
                    help="overlap-save parts per block; 0 = 64 for the headline geometry (2^23 samples per part), "
                         "more for smaller parts so that a block stays near 5e8 samples (capped at 256)")
    ap.add_argument("--max-parts", type=int, default=0, help="parts per launch group; 0 = parts_per_block/2, at most 64")
    ap.add_argument("--dump-steps", type=int, default=8, help="steps per sub-integration dump")
    ap.add_argument("--ndim", type=int, default=4, choices=[1, 2, 4])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--h2d", action="store_true",
                    help="also measure the PCIe-inclusive rate: every block copied from pinned host memory on a second "
                         "stream, double buffered, overlapped with the kernels (reported as config.pcie_inclusive)")
    ap.add_argument("--no-fused-fold", action="store_true",
                    help="Detection and Fold as separate operations (detected time series through HBM)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from dspsr_amd import pipeline

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device (the product has no CPU path)")
    # test hook: DSPSR_AMD_SINGLE_DEVICE=1 maps every rank to GPU 0 with the gloo backend, so the N>1 code path
    # can be exercised on a one-GPU box (the real run is one rank per GPU over RCCL)
    single = os.environ.get("DSPSR_AMD_SINGLE_DEVICE") == "1"
    if single:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if single:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    wl = WORKLOADS[args.workload]
    if args.workload == "cfg5":
        if world > 1:
            sys.exit("bench.py: the search-mode workload runs as independent replicas; use --gpus 1")
        return bench_search_mode(args, wl, torch)
    # sub-band sharding: rank g holds the g-th band of the same geometry (centre frequencies stacked downwards)
    freq = wl["freq"] + rank * wl["bw"]
    info = pipeline.InputInfo(centre_frequency=freq, bandwidth=wl["bw"], nchan=wl["in_nchan"], npol=2,
                              ndim=wl["ndim"], tsamp_us=wl["tsamp_us"], machine=wl["machine"])
    if not args.parts_per_block:
        n_fft = (wl["nchan"] // wl["in_nchan"]) * wl["freq_res"]
        nsamp_fft = 2 * n_fft if wl["ndim"] == 1 else n_fft
        args.parts_per_block = max(16, min(256, (1 << 29) // nsamp_fft))
    if not args.max_parts:
        args.max_parts = max(1, min(64, args.parts_per_block // 2))
    cfg = pipeline.Config(nchan=wl["nchan"], dispersion_measure=wl["dm"], nbin=wl["nbin"],
                          folding_period=0.0893, freq_res=wl["freq_res"], ndim=args.ndim,
                          parts_per_block=args.parts_per_block, max_parts=args.max_parts,
                          fused_fold=not args.no_fused_fold)
    lt = pipeline.LoadToFold(cfg, info, device=local_rank, stream=torch.cuda.current_stream().cuda_stream)

    # synthetic block resident in HBM: seeded Gaussian noise, sigma = 24 LSB (content does not change the work)
    nbytes = lt.block_bytes()
    gen = torch.Generator(device="cuda").manual_seed(20100413 + rank)
    raw = torch.empty(nbytes, dtype=torch.int8, device="cuda")
    chunk = 1 << 26
    for s in range(0, nbytes, chunk):
        e = min(nbytes, s + chunk)
        raw[s:e] = torch.randn(e - s, generator=gen, device="cuda").mul_(24.0).round_().clamp_(-128, 127).to(torch.int8)

    gather = None
    if world > 1:
        gather = torch.zeros(world * cfg.nchan * lt.npol_out * cfg.nbin * cfg.ndim, dtype=torch.float32, device="cuda")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step(i, ev=None):
        lt.process_block(raw, events=ev)
        if (i + 1) % args.dump_steps == 0:
            lt.finish_subint(dist if world > 1 else None, rank, world, gather)
            lt.subints.clear()

    for i in range(args.warmup):
        step(i)
    barrier()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t_start = time.perf_counter()
    for i in range(args.steps):
        step(i, events[i])
    lt.finish_subint(dist if world > 1 else None, rank, world, gather)
    barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    samples_per_step = cfg.parts_per_block * lt.nsamp_step            # per rank, per pol
    value = world * samples_per_step * args.steps / elapsed / 1e6

    if rank == 0:
        timed_ms = sum(a.elapsed_time(b) for a, b in events) / len(events)   # launch group of the timed region
        fused = lt.fused_fold
        fb_ms = timed_ms
        extra = 0
        if fused:
            # The timed region ran the fused kernels (the detected time series never reaches HBM).  The roofline
            # of the FFT+chirp(+detect) pass as SURVEY 8(d) defines it -- input once, chirp once, kept output once --
            # is measured on extra blocks right after the timed region, with Detection and Fold as separate
            # operations; the fused launch group is reported beside it with its own (smaller) algorithmic bytes.
            lt.fused_fold = False
            extra = max(4, args.steps // 4)
            ev2 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(extra)]
            lt.process_block(raw)
            for e in ev2:
                lt.process_block(raw, events=e)
            torch.cuda.synchronize()
            fb_ms = sum(a.elapsed_time(b) for a, b in ev2) / len(ev2)
            lt.fused_fold = True
        r = lt.response
        nchan_subband = cfg.nchan // info.nchan
        N = nchan_subband * r.ndat
        nsamp_fft = 2 * N if info.ndim == 1 else N
        b_alg = algorithmic_bytes_per_part(2, nsamp_fft * info.ndim, 8, N, nchan_subband, lt.nkeep)
        achieved = b_alg * cfg.parts_per_block * info.nchan / (fb_ms * 1e-3) / 1e9
        out = {
            "metric": "Msamples/s dedispersed+folded", "value": round(value, 2), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "command": wl["cmd"], "nchan": cfg.nchan, "freq_res": r.ndat,
                       "n_fft": N, "nfilt_pos": r.impulse_pos, "nfilt_neg": r.impulse_neg, "nkeep": lt.nkeep,
                       "nsamp_step": lt.nsamp_step, "dm": cfg.dispersion_measure, "nbin": cfg.nbin,
                       "parts_per_block": cfg.parts_per_block, "input": "8-bit dual-pol, resident in HBM",
                       "detected_ndim": cfg.ndim, "fused_fold": bool(fused),
                       "parallelism": "sub-band per GPU x%d" % world,
                       "realtime_factor": round(value / world / (info.rate / 1e6), 3)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": measured_traffic(args.workload, cfg.max_parts),
                         "traffic_unit": "HBM bytes per launch group of %d parts (PMC, profiles/r*_traffic.json, newest); "
                                         "algorithmic bytes for the same group: %d" % (cfg.max_parts, b_alg * cfg.max_parts),
                         "kernel": "filterbank launch group k_raw_transpose+k_fwd_cols+k_fwd_rows+k_inv_chan<.,false> "
                                   "(FFT+chirp+detect, detected output written)",
                         "algorithmic_bytes_per_part": b_alg, "group_ms_per_block": round(fb_ms, 4),
                         "region": ("%d extra blocks right after the timed region with Detection and Fold as "
                                    "separate operations (HIP events on the launch stream)" % extra) if fused
                                   else "the timed region (HIP events on the launch stream)"},
        }
        if fused:
            b_fused = b_alg - 2 * nchan_subband * lt.nkeep * 8         # no detected output: input once + chirp once
            ach_f = b_fused * cfg.parts_per_block * info.nchan / (timed_ms * 1e-3) / 1e9
            out["roofline_fused"] = {
                "bound": "hbm", "achieved": round(ach_f, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach_f / HBM_PEAK_GBS, 4),
                "traffic": measured_traffic(args.workload, cfg.max_parts, "hbm_bytes_per_launch_group_fused"),
                "kernel": "k_raw_transpose+k_fwd_cols+k_fwd_rows+k_inv_chan<.,true> (FFT+chirp+detect+fold in one "
                          "launch group, the timed region)",
                "algorithmic_bytes_per_part": b_fused, "group_ms_per_block": round(timed_ms, 4),
                "note": "the fused group also does the fold; its algorithmic bytes have no output term "
                        "(SURVEY 8(d)), so this fraction is not comparable with roofline.frac"}
        if args.h2d:
            # host-buffer hand-over: block i+1 is copied H2D on a side stream while block i is processed
            host = raw.cpu().pin_memory()
            bufs = [torch.empty_like(raw), torch.empty_like(raw)]
            copy_stream = torch.cuda.Stream()
            ready = [torch.cuda.Event(), torch.cuda.Event()]
            done = [torch.cuda.Event(), torch.cuda.Event()]
            main = torch.cuda.current_stream()

            def h2d_run(nsteps):
                with torch.cuda.stream(copy_stream):
                    bufs[0].copy_(host, non_blocking=True)
                    ready[0].record(copy_stream)
                for i in range(nsteps):
                    b = i & 1
                    if i + 1 < nsteps:
                        with torch.cuda.stream(copy_stream):
                            if i >= 1:
                                copy_stream.wait_event(done[b ^ 1])     # the kernels of block i-1 are finished with it
                            bufs[b ^ 1].copy_(host, non_blocking=True)
                            ready[b ^ 1].record(copy_stream)
                    main.wait_event(ready[b])
                    lt.process_block(bufs[b])
                    done[b].record(main)
                torch.cuda.synchronize()
            h2d_run(3)
            t1 = time.perf_counter()
            h2d_run(args.steps)
            dt = time.perf_counter() - t1
            out["config"]["pcie_inclusive"] = {
                "value": round(samples_per_step * args.steps / dt / 1e6, 1), "unit": "Msamples/s",
                "note": "blocks copied from pinned host memory (%.1f MB each) on a second stream, double buffered"
                        % (raw.numel() / 1e6)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(wl, lt)
            except Exception as e:  # the oracle is only a reported baseline
                out["cpu_baseline"] = {"value": None, "unit": "Msamples/s", "cores": 1, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)
    lt.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
