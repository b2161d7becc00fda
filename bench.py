#!/usr/bin/env python3
"""Benchmark of the hot path: 8-bit dual-pol voltages -> coherent-dedispersion filterbank (-F N:D)
-> detection -> fold, on N MI355X GPUs (one process per GPU).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload target|cfg3|cfg2|cfg4|...]

A "step" is one pass of the hot path over one block (parts_per_block overlap-save parts) of synthetic
8-bit input already resident in HBM.  metric = Msamples/s = unique input time samples per polarisation
consumed per second, each rank's own samples summed over all ranks.

--gpus N is ONE command: without WORLD_SIZE in the environment the N ranks are spawned here (before anything touches
the GPU); under torch.distributed.run the ranks are taken from the environment.  One process per GPU, weak scaling:
  * multi-channel input (cfg4: one 400 MHz band as NCHAN 8 complex 50 MHz sub-bands, dspsr -F 4096:D): SUB-BAND
    SHARDING as SURVEY 8(e) -- rank g = input channel g, the g-th 512-channel slice of the ONE full-band kernel, the
    common nfilt_pos/neg, identical hits on every rank (asserted in the parity gate); no exchange until the
    sub-integration dump, then ONE reduce (RCCL) delivers every slice of the band to rank 0.
  * single-channel input (target, cfg1-3: no exchange-free frequency split): TIME-SLICE REPLICAS, the reference's own
    strategy (MultiThread.C:65-82): rank r takes blocks r, r+N, ... and the dump SUMS profiles, hits and
    integration length (PhaseSeries::combine).
The dump (every --dump-steps steps) is inside the timed region.

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
FP32_VECTOR_PEAK_TFLOPS = 157.3   # MI355X fp32 vector peak (same guide, chip-level parameters)

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
    # BASELINE cfg 4: the 400 MHz band as NCHAN 8 complex 50 MHz sub-bands, one sub-band (input channel) per rank
    # (= dspsr -F 4096:D on the 8-channel file; 8 ranks x -F 512:D), common 27/27 of the full-band kernel
    "cfg4": dict(freq=1382.0, bw=-400.0, in_nchan=8, ndim=2, tsamp_us=0.02, nchan=4096, dm=1000.0,
                 freq_res=512, nbin=1024, machine="DADA",
                 cmd="dspsr -F 4096:D -x 512 -D 1000 -b 1024 on NCHAN 8 complex 50 MHz sub-bands, one per GPU (8 x -F 512:D)"),
    # SURVEY 8f-1 / BASELINE config 5: search-mode front end, detect only (no fold)
    "cfg5": dict(freq=1382.0, bw=-400.0, in_nchan=1, ndim=1, tsamp_us=0.00125, nchan=4096, dm=0.0, freq_res=1,
                 nbin=0, machine="DADA", tscrunch=16, nparts=32768,
                 cmd="digifil -F 4096 -t 16 -b 8 (TFP filterbank + square law + tscrunch, Rescale, PScrunch, 8-bit digitizer; no fold)"),
    # transform lengths with an odd factor (Filterbank.C:107-155 plans whatever FFTW plans; here 2^k * {3, 5, 7, 9, 15}): nchan_subband
    # = 96 = 3 * 32 (three interleaved sub-sequences forward + one radix-3 pass) and freq_res = 12288 = 3 * 4096 (three pseudo-channels
    # per channel + one radix-3 pass in time, Detection and Fold as launches of their own), with their power-of-two neighbours
    "odd_nchan": dict(freq=1382.0, bw=-400.0, in_nchan=1, ndim=1, tsamp_us=0.00125, nchan=96, dm=8.0, freq_res=4096, nbin=1024,
                      machine="CASPSR", cmd="dspsr -F 96:D -x 4096 -D 8 -b 1024 (nchan_subband = 3 * 32)"),
    "odd_nchan_nb": dict(freq=1382.0, bw=-400.0, in_nchan=1, ndim=1, tsamp_us=0.00125, nchan=128, dm=8.0, freq_res=4096, nbin=1024,
                         machine="CASPSR", cmd="dspsr -F 128:D -x 4096 -D 8 -b 1024 (power-of-two neighbour of -F 96:D)"),
    "odd_fres": dict(freq=1382.0, bw=-400.0, in_nchan=1, ndim=1, tsamp_us=0.00125, nchan=1024, dm=1000.0, freq_res=12288, nbin=1024,
                     machine="CASPSR", cmd="dspsr -F 1024:D -x 12288 -D 1000 -b 1024 (freq_res = 3 * 4096)"),
    "odd_fres_nb": dict(freq=1382.0, bw=-400.0, in_nchan=1, ndim=1, tsamp_us=0.00125, nchan=1024, dm=1000.0, freq_res=8192, nbin=1024,
                        machine="CASPSR", cmd="dspsr -F 1024:D -x 8192 -D 1000 -b 1024 (power-of-two neighbour of -x 12288)"),
    # the other branch of digifil (LoadToFil.C:185-222): convolving filterbank with coherent dedispersion in search mode -- the headline
    # geometry, detected (Intensity), time scrunched, rescaled and digitised instead of folded
    "cfg5c": dict(freq=1382.0, bw=-400.0, in_nchan=1, ndim=1, tsamp_us=0.00125, nchan=1024, dm=1000.0, freq_res=4096, nbin=0,
                  machine="DADA", tscrunch=16, nparts=64, max_parts=32,
                  cmd="digifil -F 1024:D -x 4096 -D 1000 -t 16 -b 8 (convolving filterbank + square law + tscrunch in one launch group, "
                      "Rescale + 8-bit digitizer; no fold)"),
    # `dspsr -F 128` without `:D` (Filterbank::Config::After, the default of FilterbankConfig.C:56): the non-convolving filterbank
    # (freq_res = 1, csrc/fb_plain.hip), then dsp::Convolution with the dedispersion response on its 128 channels (the filterbank
    # object with nchan_subband = 1: four tile passes at this response length), Detection, Fold
    "after": dict(freq=1382.0, bw=-400.0, in_nchan=1, ndim=1, tsamp_us=0.00125, nchan=128, dm=67.99, freq_res=65536, nbin=512,
                  machine="CASPSR", when="after", parts_per_block=32, max_parts=32,
                  cmd="dspsr -F 128 -x 65536 -D 67.99 -b 512 (filterbank, THEN convolution: Config::After; header.dada band, vela.par DM)"),
    # the same with the shortest response the smearing allows a power of two above (3657 of 8192 points dropped): the convolution of a
    # (channel, part) sequence then fits ONE workgroup tile -- forward transform, response, backward transform, Detection in one pass
    # over the rows (csrc/fb_conv1.hip) instead of four
    "after8k": dict(freq=1382.0, bw=-400.0, in_nchan=1, ndim=1, tsamp_us=0.00125, nchan=128, dm=67.99, freq_res=8192, nbin=512,
                    machine="CASPSR", when="after", parts_per_block=256, max_parts=256,
                    cmd="dspsr -F 128 -x 8192 -D 67.99 -b 512 (filterbank, THEN convolution in one tile pass: Config::After)"),
    # the HEADLINE's band, channel count, DM and response length through Config::After (`-F 1024` without `:D`): k_fb_plain + the one-pass
    # convolution of 4096 points per channel (79 % of each transform kept) + Fold -- not the BASELINE metric (the channel filters differ
    # from the convolving filterbank's), a measure of what the other route costs on the same data
    "after1k": dict(freq=1382.0, bw=-400.0, in_nchan=1, ndim=1, tsamp_us=0.00125, nchan=1024, dm=1000.0, freq_res=4096, nbin=1024,
                    machine="CASPSR", when="after", parts_per_block=64, max_parts=64,
                    cmd="dspsr -F 1024 -x 4096 -D 1000 -b 1024 (filterbank, THEN convolution in one tile pass: Config::After on the headline's band)"),
    # few wide channels, long responses: `dspsr -F 8` on the same band smears 885k samples per 50 MHz channel -- 2^21 points, the
    # three-pass convolution at the other end of its range (csrc/fb_conv3.hip: 256-point columns, 8192-point rows; 58 % of each
    # transform kept)
    "after8c": dict(freq=1382.0, bw=-400.0, in_nchan=1, ndim=1, tsamp_us=0.00125, nchan=8, dm=67.99, freq_res=1 << 21, nbin=512,
                    machine="CASPSR", when="after", parts_per_block=8, max_parts=8,
                    cmd="dspsr -F 8 -x 2097152 -D 67.99 -b 512 (filterbank, THEN convolution in three tile passes: Config::After)"),
    # the filterbank alone (Filterbank::Config::Never: `dspsr -F 128` with coherent dedispersion switched off, or any DM-0 source):
    # k_fb_plain writing the detected rows, then Fold -- the roofline of the non-convolving filterbank kernel itself
    "plain": dict(freq=1382.0, bw=-400.0, in_nchan=1, ndim=1, tsamp_us=0.00125, nchan=128, dm=0.0, freq_res=1, nbin=512,
                  machine="DADA", when="never", parts_per_block=1 << 21, max_parts=1 << 21,
                  cmd="dspsr -F 128 without coherent dedispersion (Config::Never: non-convolving filterbank, Detection, Fold)"),
    # the reference's fold benchmark (Benchmark/fold.csh on Benchmark/fold_header.dada): already-detected input, 1024
    # channels x 4 polarisation products at 32 us, folded with vela.polyco -- dsp::Fold alone
    "fold": dict(freq=1382.0, bw=-400.0, in_nchan=1024, ndim=1, tsamp_us=32.0, nchan=1024, dm=0.0, freq_res=1, nbin=0,
                 machine="Dummy", npol=4, ndat=1 << 17,
                 cmd="dspsr -E vela.par -P vela.polyco fold_header.dada (Benchmark/fold.csh: NCHAN 1024, NPOL 4, STATE Coherence, "
                     "TSAMP 32 us; fold only)"),
}


def bench_fold_only(args, wl, torch):
    """Benchmark/fold.csh: one step = one block of `ndat` detected samples (1024 channels x 4 products, FPT floats resident
    in HBM -- the reference's Dummy unpacker does not touch the data either) folded with the vela.polyco predictor:
    phase and period from the polynomial at the block's first sample, the double-precision bin plan on the host
    (Fold.C:718-787), k_fold_dense / k_fold_chunked on the device (Fold.C:835-891).  nbin as dsp::Fold::choose_nbin picks it."""
    import dspsr_amd
    from dspsr_amd import pipeline
    nchan, npol, ndat = wl["nchan"], wl["npol"], wl["ndat"]
    rate = 1e6 / wl["tsamp_us"]
    text = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "vela_polyco.json")))["text"]
    polyco = pipeline.Polyco(text)
    day, sec0 = 55299, 7545.0                                  # UTC_START 2010-04-13-02:05:45 (fold_header.dada)
    nbin = pipeline.choose_nbin(1.0 / polyco.frequency(day, sec0), rate)
    ctx = dspsr_amd.Context(0, torch.cuda.current_stream().cuda_stream)
    fold = dspsr_amd.FoldEngine(ctx)
    fold.set_shape(nchan, npol, 1, nbin)
    gen = torch.Generator(device="cuda").manual_seed(20100413)
    det = torch.empty((nchan, npol, ndat), dtype=torch.float32, device="cuda")
    for c0 in range(0, nchan, 64):
        det[c0:c0 + 64] = torch.randn((min(64, nchan - c0), npol, ndat), generator=gen, device="cuda").square_()
    hits = np.zeros(nbin, np.uint32)
    state = {"blocks": 0, "length": 0.0}

    def step(ev=None):
        t0 = (state["blocks"] * ndat + 0.5) / rate
        phi = polyco.phase_frac(day, sec0 + t0)
        pfold = 1.0 / polyco.frequency(day, sec0 + t0)
        fold.set_nbin(nbin)
        fold.set_ndat(ndat, 0)
        folded = fold.set_bins(phi, (1.0 / rate) / pfold, ndat, 0, hits)
        if ev is not None:
            ev[0].record()
        fold.fold(det)
        if ev is not None:
            ev[1].record()
        state["blocks"] += 1
        state["length"] += folded / rate

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for e in ev:
        step(e)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    k_ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
    # parity gate (one more block on a zeroed profile): every sample in exactly one bin; the profile equals the float64
    # index_add of the same samples by the same plan to float rounding
    fold.zero()
    hits[:] = 0
    step()
    prof = torch.from_numpy(fold.synch()).cuda().view(nchan * npol, nbin).double()
    plan, _ = dspsr_amd.fold_binplan(polyco.phase_frac(day, sec0 + ((state["blocks"] - 1) * ndat + 0.5) / rate),
                                  (1.0 / rate) * polyco.frequency(day, sec0 + ((state["blocks"] - 1) * ndat + 0.5) / rate), nbin, ndat)
    want = torch.zeros((nchan * npol, nbin), dtype=torch.float64, device="cuda")
    want.index_add_(1, torch.from_numpy(plan.astype(np.int64)).cuda(), det.view(nchan * npol, ndat).double())
    rel = float((prof - want).abs().max() / want.abs().max())
    if int(hits.astype(np.int64).sum()) != ndat or not rel <= 2e-6:
        raise ParityGateError("bench.py parity gate FAILED (fold): hits.sum()=%d ndat=%d, profile rel %.3g"
                              % (int(hits.sum()), ndat, rel))
    b_alg = det.numel() * 4
    achieved = b_alg / (k_ms * 1e-3) / 1e9
    res = {"metric": "Msamples/s folded", "value": round(ndat * args.steps / elapsed / 1e6, 3), "unit": "Msamples/s",
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "fold", "command": wl["cmd"], "nchan": nchan, "npol": npol, "ndim": 1, "nbin": nbin,
                      "ndat_per_block": ndat, "input": "detected float32 FPT rows, resident in HBM",
                      "realtime_factor": round(ndat * args.steps / elapsed / rate, 1),
                      "note": "time samples per second; every sample carries nchan*npol = %d floats" % (nchan * npol)},
           "parity_gate": {"status": "ok", "checks": ["hits.sum() == ndat", "profile == float64 index_add (rel %.1e)" % rel]},
           "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": measured_traffic("fold", None, "hbm_bytes_per_launch"),
                        "traffic_source": measured_traffic.source, "kernel": "k_fold_dense<1,4> (dense per-chunk run table; k_fold_chunked for plans with two runs of a bin per chunk)",
                        "algorithmic_bytes_per_launch": b_alg, "kernel_ms": round(k_ms, 4)}}
    if res["roofline"]["traffic"]:
        res["roofline"]["traffic_ratio"] = round(res["roofline"]["traffic"] / b_alg, 3)
    if not args.no_cpu_baseline:
        import oracle.dspsr_oracle as o
        nc = nchan                                              # one whole block (about 1.5 s on one core)
        x = det[:nc].cpu().numpy()
        t1 = time.perf_counter()
        acc = np.zeros((nc, npol, nbin), np.float32)
        bins = o.fold_binplan(0.25, 1.0 / 2794.0, nbin, ndat)
        for c in range(nc):
            for q in range(npol):
                np.add.at(acc[c, q], bins, x[c, q])
        dt = time.perf_counter() - t1
        res["cpu_baseline"] = {"value": round(ndat * nc / nchan / dt / 1e6, 4), "unit": "Msamples/s", "cores": 1, "kind": "port",
                               "sample": "%d of %d channels of one block, numpy restatement of the Fold.C:835-891 loop, %.2f s"
                                         % (nc, nchan, dt)}
    fold.close()
    ctx.close()
    return res


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
    lg = int(np.log2(nchan))
    flops_part = 2 * (5 * nchan * lg + 10 * nchan) + 2 * (4 * nchan)
    res = {"metric": "Msamples/s dedispersed+folded", "value": round(samples * args.steps / elapsed / 1e6, 2),
           "unit": "Msamples/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "cfg5", "command": wl["cmd"], "nchan": nchan, "tscrunch": sf,
                      "parts_per_block": npart, "input": "8-bit dual-pol, resident in HBM",
                      "chain": "TFPFilterbank+detect+TScrunch [k_tfp] -> Rescale statistics -> Rescale apply + PScrunch + SigProcDigitizer(8 bit) "
                               "[one pass, dspsr_amd_rescale_pscrunch_digitize]",
                      "note": "search mode: detected, scrunched and digitised, NOT folded"},
           # The kernel reads every input byte once and writes 1/16 of a float per sample (traffic ratio 1.00): at 33 flop per
           # algorithmic byte it sits above the 20 flop/B ridge of the chip, so what bounds it is the vector unit (and the LDS
           # exchanges of the transform, which do not overlap with vector issue) -- both fractions are reported.
           "roofline": {"bound": "valu", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": measured_traffic("cfg5", None, "hbm_bytes_per_launch"),
                        "traffic_source": measured_traffic.source,
                        "kernel": "k_tfp4k (TFP filterbank + detection + tscrunch; post-processing on the last stage's registers)",
                        "algorithmic_bytes_per_launch": b_alg, "kernel_ms": round(k_ms, 4),
                        "valu": {"flops_per_part": flops_part, "achieved": round(flops_part * npart / (k_ms * 1e-3) / 1e12, 2),
                                 "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                                 "frac": round(flops_part * npart / (k_ms * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS, 4),
                                 "flops_per_byte": round(flops_part * npart / b_alg, 1),
                                 "model": "npol * (5 C log2 C [C = nchan complex points per pol and part] + 10 C [real-transform split, "
                                          "power]) + 2 per decoded sample"}}}
    if res["roofline"]["traffic"]:
        res["roofline"]["traffic_ratio"] = round(res["roofline"]["traffic"] / b_alg, 3)
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
    lf.close()
    return res


def make_search_coherent(wl, torch):
    """The pipeline object and the synthetic block of the cfg5c workload (shared with tools/pmc_workload.py)."""
    from dspsr_amd import pipeline
    info = pipeline.InputInfo(centre_frequency=wl["freq"], bandwidth=wl["bw"], nchan=1, npol=2, ndim=1, tsamp_us=wl["tsamp_us"],
                              machine=wl["machine"])
    cfg = pipeline.SearchConfig(nchan=wl["nchan"], tscrunch=wl["tscrunch"], nbit=8, dispersion_measure=wl["dm"], freq_res=wl["freq_res"],
                                parts_per_block=wl["nparts"], max_parts=wl["max_parts"], rescale_seconds=10.0)
    lf = pipeline.LoadToFilCoherent(cfg, info, device=0, stream=torch.cuda.current_stream().cuda_stream)
    nbytes = lf.block_bytes()
    gen = torch.Generator(device="cuda").manual_seed(20100413)
    raw = torch.empty(nbytes, dtype=torch.int8, device="cuda")
    for s0 in range(0, nbytes, 1 << 26):
        e = min(nbytes, s0 + (1 << 26))
        raw[s0:e] = torch.randn(e - s0, generator=gen, device="cuda").mul_(24.0).round_().clamp_(-128, 127).to(torch.int8)
    return lf, raw, cfg


def bench_search_coherent(args, wl, torch):
    """cfg5c: one step = one block of nparts overlap-save parts through digifil's convolving branch (dspsr_amd.pipeline.LoadToFilCoherent):
    Filterbank + chirp + Detection::square_law + TScrunch [one launch group, the roofline group] -> Rescale + 8-bit SigProcDigitizer
    [one pass over the 16x smaller scrunched rows], everything resident in HBM."""
    import dspsr_amd
    lf, raw, cfg = make_search_coherent(wl, torch)
    npart = wl["nparts"]
    for _ in range(args.warmup):
        lf.process_block(raw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lf.process_block(raw)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(max(3, args.steps // 2))]
    for a, b in ev:                                   # the filterbank + detection + scrunch launch group alone
        a.record()
        lf.detect_scrunch(raw)
        b.record()
    torch.cuda.synchronize()
    g_ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
    r = lf.response
    N = cfg.nchan * r.ndat
    # SURVEY 8(d) with the output term of this path: each input byte once, the chirp once, each SCRUNCHED sample written once
    b_alg = 2 * (2 * N) + 8 * N + cfg.nchan * cfg.npol * 4 * lf.nkeep / cfg.tscrunch
    achieved = b_alg * npart / (g_ms * 1e-3) / 1e9
    samples = npart * lf.nsamp_step
    res = {"metric": "Msamples/s dedispersed+folded", "value": round(samples * args.steps / elapsed / 1e6, 2), "unit": "Msamples/s",
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "cfg5c", "command": wl["cmd"], "nchan": cfg.nchan, "freq_res": r.ndat, "n_fft": N, "nkeep": lf.nkeep,
                      "tscrunch": cfg.tscrunch, "parts_per_block": npart, "max_parts": cfg.max_parts, "input": "8-bit dual-pol, resident in HBM",
                      "search_fused": bool(lf.fused and lf.fb.search_is_fused()), "library_build": dspsr_amd.build_id(),
                      "note": "search mode: dedispersed coherently, detected, scrunched and digitised, NOT folded"},
           "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                        "traffic": measured_traffic("cfg5c", cfg.max_parts), "traffic_source": measured_traffic.source,
                        "kernel": "filterbank launch group k_raw_transpose+k_fwd_cols+k_fwd_rows+k_inv_chan<.,2> (FFT + chirp + square law + "
                                  "time scrunch)",
                        "algorithmic_bytes_per_part": int(b_alg), "group_ms_per_block": round(g_ms, 4)}}
    if res["roofline"]["traffic"]:
        res["roofline"]["traffic_ratio"] = round(res["roofline"]["traffic"] / (b_alg * cfg.max_parts), 3)
    lf.close()
    return res


def algorithmic_bytes_per_part(npol, nsamp_fft, nbit, N, nchan_subband, nkeep):
    """each input byte read once, chirp read once, each kept output written once."""
    return npol * nsamp_fft * nbit // 8 + 8 * N + npol * nchan_subband * nkeep * 8


def measured_traffic(workload, max_parts, key="hbm_bytes_per_launch_group"):
    """HBM bytes per launch group (or, cfg5 / fold: per launch of the roofline kernel) from the committed PMC profile
    (tools/pmc_traffic.sh; rocprofv3 cannot run inside the bench).  Only a profile taken with THIS library counts: the file
    carries the build id of the library it was collected on (dspsr_amd_build_id(): sha256 of the sources), and a file whose
    id differs from the loaded library's -- or that has none -- is refused (traffic: null, the reason in traffic_source)."""
    import glob
    import dspsr_amd
    have = dspsr_amd.build_id()
    measured_traffic.source = "no profiles/r*_traffic.json for workload %s taken with library build %s" % (workload, have)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):   # newest round first
        try:
            d = json.load(open(path))
            if d["workload"] == workload and (max_parts is None or d["parts_per_launch_group"] == max_parts) and key in d:
                if d.get("build_id") != have:
                    measured_traffic.source = "%s refused: taken with library build %s, loaded build %s" % (
                        os.path.relpath(path, ROOT), d.get("build_id"), have)
                    continue
                measured_traffic.source = "%s (library build %s)" % (os.path.relpath(path, ROOT), d["build_id"])
                return d[key]
        except Exception:
            pass
    return None


measured_traffic.source = None


_CPU_SETUP = {}


def _cpu_port_run(args):
    """One CPU worker of the baseline: the numpy oracle (oracle/dspsr_oracle.py, a 'port' of the reference's CPU
    path) on `nparts` overlap-save parts of the workload: unpack -> filterbank+chirp -> detect -> fold.
    Runs in a spawned process (no torch / HIP in the children); nparts == 0 is the warm-up call."""
    wl, nparts, seed = args
    import oracle.dspsr_oracle as o
    key = tuple(sorted(wl.items()))
    if key not in _CPU_SETUP:                    # built once per worker (the warm-up call), like dspsr builds its response once
        obs = o.Observation(centre_frequency=wl["freq"], bandwidth=wl["bw"], nchan=wl["in_nchan"], npol=2,
                            ndim=wl["ndim"], tsamp_us=wl["tsamp_us"], machine=wl["machine"],
                            dispersion_measure=wl["dm"])
        resp = o.Dedispersion()
        resp.set_frequency_resolution(wl["freq_res"])
        resp.match(obs, wl["nchan"])
        _CPU_SETUP[key] = (obs, resp, o.filterbank_plan(obs, wl["nchan"], resp))
    obs, resp, plan = _CPU_SETUP[key]
    if nparts == 0:
        np.fft.rfft(np.zeros(1024, np.float32))
        return 0, 0.0
    # one overlap-save part at a time (a block of one part): the memory of a worker does not grow with the sample
    ndat = plan.nsamp_step + plan.nsamp_overlap
    rng = np.random.default_rng(seed)
    busy = 0.0
    fobs = o.filterbank_output_observation(obs, plan)
    prof = np.zeros((wl["nchan"], wl["nbin"], 4), np.float32)
    raw = np.clip(np.rint(rng.standard_normal(ndat * wl["in_nchan"] * 2 * wl["ndim"]) * 24), -128, 127).astype(np.int8)
    for part in range(nparts):                   # (the same noise every time: the content does not change the work)
        t0 = time.perf_counter()
        unpacked = o.unpack_8bit(raw, obs)
        fb = o.filterbank(unpacked, plan, resp.buffer, dtype=np.float32)
        det = o.detect_layout(o.detect_products(fb, "Coherence"), 4)
        # fold with the vectorised equivalent of the sequential loop (np.add.at keeps time order per bin)
        phi, pfold = o.fold_phase(o.FoldConfig(nbin=wl["nbin"], folding_period=0.089), fobs,
                                  fobs.start_seconds + part * plan.nkeep / fobs.rate)
        binplan = o.fold_binplan(phi, (1.0 / fobs.rate) / pfold, wl["nbin"], det.shape[2])
        np.add.at(prof, (slice(None), binplan), det[:, 0])
        busy += time.perf_counter() - t0
    return nparts * plan.nsamp_step * wl["in_nchan"], busy


def _host_cores():
    """Cores this process may really use: the affinity mask, capped by the cgroup CPU quota when there is one (a GPU
    box hands out a share of the host: the mask can list every core of the machine while the quota is 16)."""
    try:
        cores = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cores = list(range(os.cpu_count() or 1))
    quota, src = None, ""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]            # cgroup v2
        if q != "max":
            quota, src = float(q) / float(per), "cgroup cpu.max"
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())     # cgroup v1
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota, src = q / per, "cgroup cfs quota"
        except (OSError, ValueError):
            pass
    n = len(cores)
    note = "%d cores in the affinity mask" % n
    if quota is not None and quota < n:
        n = max(1, int(quota + 0.5))
        note += ", %s = %.1f cores -> %d workers" % (src, quota, n)
    return n, note


def cpu_baseline_c(wl, geom, ncore, parts_per_worker=4):
    """The plain-C restatement (oracle/oracle_c.c: float32, its own radix-2 FFT) on the same workload, one thread per
    time block (ctypes releases the GIL): unpack -> filterbank+chirp -> cross_detect -> bin plan -> fold."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    lib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle_c.so"))
    vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint
    lib.oracle_unpack8.argtypes = [vp, u64, u32, u32, u32, C.c_int, C.c_float, vp]
    lib.oracle_filterbank.argtypes = [vp, u64, u32, u32, C.c_int, u32, u32, u32, u32, u64, u64, vp, vp]
    lib.oracle_cross_detect.argtypes = [u32, vp, vp, vp, vp, vp, vp, u32]
    lib.oracle_fold_binplan.argtypes = [C.c_double, C.c_double, u32, u64, vp, vp]
    lib.oracle_fold.argtypes = [vp, u64, u32, u32, u32, u64, u64, vp, u32, vp]
    in_nchan, ndim, nchan, M, nbin = wl["in_nchan"], wl["ndim"], wl["nchan"], geom["freq_res"], wl["nbin"]
    C_sub, nkeep, step, ovl = nchan // in_nchan, geom["nkeep"], geom["nsamp_step"], geom["nsamp_overlap"]
    kernel = np.ascontiguousarray(geom["kernel"], dtype=np.complex64)
    caspsr = 1 if wl["machine"] == "CASPSR" else 0
    scale = float(geom["scale8"])

    def work(seed):
        ndat = step + ovl                        # one overlap-save part at a time: bounded memory per thread
        rng = np.random.default_rng(seed)
        unp = np.empty((in_nchan, 2, ndat * ndim), np.float32)
        fb = np.empty((nchan, 2, nkeep), np.complex64)
        det = np.empty((nchan, 1, nkeep, 4), np.float32)
        plan = np.empty(nkeep, np.uint32)
        hits = np.zeros(nbin, np.uint32)
        prof = np.zeros((nchan, 1, nbin, 4), np.float32)
        busy = 0.0
        raw = np.clip(np.rint(rng.standard_normal(ndat * in_nchan * 2 * ndim) * 24), -128, 127).astype(np.int8)
        for part in range(parts_per_worker):
            t0 = time.perf_counter()
            lib.oracle_unpack8(raw.ctypes.data, ndat, in_nchan, 2, ndim, caspsr, scale, unp.ctypes.data)
            lib.oracle_filterbank(unp.ctypes.data, ndat * ndim, in_nchan, 2, 1 if ndim == 1 else 0, C_sub, M, geom["nfilt_pos"],
                                  nkeep, step, 1, kernel.ctypes.data, fb.ctypes.data)
            for c in range(nchan):
                d = det[c, 0]
                lib.oracle_cross_detect(nkeep, fb[c, 0].ctypes.data, fb[c, 1].ctypes.data, d.ctypes.data, d.ctypes.data + 4,
                                        d.ctypes.data + 8, d.ctypes.data + 12, 4)
            lib.oracle_fold_binplan(0.123 + part * 0.37, (1.0 / geom["out_rate"]) / 0.089, nbin, nkeep, plan.ctypes.data,
                                    hits.ctypes.data)
            lib.oracle_fold(det.ctypes.data, nkeep * 4, nchan, 1, 4, 0, nkeep, plan.ctypes.data, nbin, prof.ctypes.data)
            busy += time.perf_counter() - t0
        return parts_per_worker * step * in_nchan, busy

    with ThreadPoolExecutor(ncore) as ex:
        t0 = time.perf_counter()
        res = list(ex.map(work, [101 + i for i in range(ncore)]))
        wall = time.perf_counter() - t0
    samples = sum(r[0] for r in res)
    return {"value": samples / wall / 1e6, "unit": "Msamples/s", "cores": ncore, "kind": "port",
            "sample": "%d threads x %d overlap-save part(s) of the same workload (%.1f Msamples/pol in all), plain-C oracle "
                      "(oracle/oracle_c.c: float32, own radix-2 FFT, no SIMD FFT library), one thread per time block, "
                      "%.1f s wall, %.1f core-seconds" % (ncore, parts_per_worker, samples / 1e6, wall, sum(r[1] for r in res))}


def oracle_check(wl, cfg, info, torch, nparts=2):
    """Part of the cpu_baseline leg (N = 1, after the timed region): the product path against the float64 oracle on the SAME
    bytes, beside the number -- `nparts` overlap-save parts of a dispersed, pulse-modulated 8-bit signal (dspsr_amd.synth, the
    band's own DM) through the workload's launch path (same kernels as the timed region, a block of `nparts`
    parts) and through oracle/dspsr_oracle.py: unpack -> filterbank x chirp -> coherency products -> fold.  Every channel.
    hits[] must be identical and the profile within 1e-5 of its maximum (the tolerance of tests/test_gpu_parity.py); a
    failure ends the run like the parity gate."""
    import dataclasses
    import oracle.dspsr_oracle as o
    from dspsr_amd import pipeline, synth
    small = dataclasses.replace(cfg, parts_per_block=nparts, max_parts=nparts)
    lt = pipeline.LoadToFold(small, info, device=torch.cuda.current_device(), stream=torch.cuda.current_stream().cuda_stream)
    try:
        ndat = nparts * lt.nsamp_step + lt.nsamp_overlap
        period = 0.25 * ndat * wl["tsamp_us"] * 1e-6                 # four pulses in the sample
        rawh = synth.voltages(ndat, wl["freq"], wl["bw"], wl["tsamp_us"], wl["dm"], period, npol=2, ndim=wl["ndim"],
                              nchan=wl["in_nchan"], layout="caspsr" if wl["machine"] == "CASPSR" else "generic")
        lt.process_block(torch.from_numpy(rawh).cuda())
        lt.finish_subint()
        lt.synchronize()
        sub = lt.subints[-1]
        prof = pipeline.subint_profile(sub).reshape(lt.nchan_out, small.nbin, 4).astype(np.float64)
        t0 = time.perf_counter()
        obs = o.Observation(centre_frequency=wl["freq"], bandwidth=wl["bw"], nchan=wl["in_nchan"], npol=2, ndim=wl["ndim"],
                            tsamp_us=wl["tsamp_us"], machine=wl["machine"], dispersion_measure=wl["dm"])
        resp = o.Dedispersion()
        resp.set_frequency_resolution(wl["freq_res"])
        resp.match(obs, wl["nchan"])
        plan = o.filterbank_plan(obs, wl["nchan"], resp)
        # (the chirp the device multiplies by -- host C++ build, compared with the oracle's own below -- so that the figure is
        #  about the transforms, detection and fold, not about the last bit of a sincos)
        fb = o.filterbank(o.unpack_8bit(rawh, obs).astype(np.float64), plan, np.asarray(lt.response.kernel).astype(np.complex128),
                          dtype=np.float64)
        det = o.detect_layout(o.detect_products(fb, "Coherence"), 4)                 # [chan][1][ndat][4]
        fobs = o.filterbank_output_observation(obs, plan)
        ps = o.PhaseSeries(det.shape[0], 1, 4, small.nbin, data=np.zeros((det.shape[0], 1, small.nbin, 4), np.float64))
        o.fold(det, fobs, o.FoldConfig(nbin=small.nbin, folding_period=small.folding_period), ps)
        same_hits = bool(np.array_equal(sub["hits"], ps.hits)) and int(ps.hits.sum()) == nparts * lt.nkeep
        err = float(np.abs(prof - ps.data[:, 0]).max() / np.abs(ps.data[..., :2]).max())
        kerr = float(np.abs(lt.response.kernel - resp.buffer).max())
        rec = {"status": "ok" if same_hits and err <= 1e-5 else "FAILED", "profile_max_err_rel": err, "tolerance": 1e-5,
               "hits_identical": same_hits, "chirp_max_abs_diff": kerr, "channels": int(det.shape[0]), "parts": nparts,
               "input": "dispersed pulse-modulated noise (dspsr_amd.synth.voltages: DM %g, four pulses in %d samples/pol), 8-bit"
                        % (wl["dm"], ndat),
               "oracle": "oracle/dspsr_oracle.py in float64, %.1f s" % (time.perf_counter() - t0)}
    finally:
        lt.close()
    if rec["status"] != "ok":
        raise ParityGateError("bench.py oracle check FAILED: %r" % (rec,))
    return rec


def cpu_baseline(wl, geom, parts_per_worker=8):
    """Times the CPU port on ALL host cores of this process's affinity mask with the reference's own parallelisation
    model -- one worker per time block (dspsr -t <ncores>, MultiThread.C:65-82) -- on a bounded sample of the workload.
    Two stated baselines: the numpy oracle (pocketfft float32) as `cpu_baseline`, the plain-C oracle beside it."""
    import multiprocessing as mp
    ncore, note = _host_cores()
    wlc = {k: v for k, v in wl.items() if k != "cmd"}
    ctx = mp.get_context("spawn")         # the parent holds a HIP context: never fork it
    with ctx.Pool(ncore) as pool:
        pool.map_async(_cpu_port_run, [(wlc, 0, 0)] * ncore).get(timeout=180)   # start-up, imports, response build
        t0 = time.perf_counter()
        # bounded: a host that cannot finish the sample in 3 minutes is reported as such instead of stalling the bench
        res = pool.map_async(_cpu_port_run, [(wlc, parts_per_worker, 1 + i) for i in range(ncore)], chunksize=1).get(timeout=180)
        wall = time.perf_counter() - t0
    samples = sum(r[0] for r in res)
    out = {"value": samples / wall / 1e6, "unit": "Msamples/s", "cores": ncore, "kind": "port", "affinity": note,
           "sample": "%d workers x %d overlap-save part(s) of the same workload (%.1f Msamples/pol in all), numpy "
                     "oracle with pocketfft float32, one process per time block, %.1f s wall, %.1f core-seconds"
                     % (ncore, parts_per_worker, samples / 1e6, wall, sum(r[1] for r in res))}
    try:
        out["c_oracle"] = cpu_baseline_c(wl, geom, ncore, max(1, parts_per_worker // 2))
    except Exception as e:  # reported baseline only
        out["c_oracle"] = {"value": None, "sample": "failed: %r" % (e,)}
    return out


def spawn_ranks(argv, ngpu):
    """bench.py --gpus N as ONE command: start the N ranks as child processes (one per GPU) BEFORE this process touches
    the GPU, relay rank 0's JSON line, exit non-zero if any rank fails."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(ngpu):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(ngpu), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    try:
        for p in procs:
            p.wait()
            if p.returncode != 0:
                rc = rc or p.returncode or 1
                for q in procs:                  # a rank died: the others would wait in a collective for ever
                    if q.poll() is None:
                        q.terminate()
    except KeyboardInterrupt:
        for q in procs:
            q.kill()
        rc = 130
    return rc


class ParityGateError(SystemExit):
    pass


def parity_gate(lt, raw, torch, dist, rank, world, sharded):
    """Outside the timed region, on one more block: (i) the fused launch group and Detection + Fold as separate
    operations give the same profile bit for bit, (ii) every output sample landed in exactly one phase bin,
    (iii) the profile holds the power of the detected time series, (iv) sub-band ranks agree on hits[].
    Any failure ends the run with a non-zero exit code and no result line."""
    from dspsr_amd import pipeline
    was = lt.fused_fold
    ndat = lt.cfg.parts_per_block * lt.nkeep
    res = {}
    lt.subints.clear()
    first = lt.ndat_out
    for fused in ([True, False] if was else [False]):
        lt.fused_fold = fused
        lt.ndat_out = first                       # the same block of the stream both times: same phases, same bin plan
        lt.process_block(raw)
        if fused is False:
            det = lt.detected.view(lt.nchan_out, lt.npol_out, -1)
        hits = lt.hits.copy()
        prof = lt.profiles_tensor().clone()
        lt.fold.zero()
        lt.hits[:] = 0
        lt.integration_length, lt.ndat_total = 0.0, 0
        res[fused] = (hits, prof)
    lt.fused_fold = was
    hits, prof = res[False]
    fails = []
    # mode 2: the parts of a launch are folded in runs (re-associated sums); bins of 64 samples or more: the separate Fold is
    # the long-run fold (re-associated micro-block sums) while the fused kernel keeps the exact order -- equal to rounding
    wide = lt.cfg.folding_period > 0 and lt.out_rate * lt.cfg.folding_period / lt.cfg.nbin >= 63.0
    exact = getattr(lt, "fused_mode", 1) != 2 and not wide
    if was:
        same_hits = np.array_equal(res[True][0], hits)
        if exact:
            same = bool(torch.equal(res[True][1], prof))
        else:
            same = float((res[True][1].double() - prof.double()).abs().max()) <= 2e-6 * float(prof.abs().max())
        if not (same_hits and same):
            fails.append("fused fold differs from Detection + Fold")
    if int(hits.astype(np.int64).sum()) != ndat:
        fails.append("hits.sum()=%d != ndat=%d" % (int(hits.sum()), ndat))
    nd = lt.cfg.ndim
    if nd == 4:                                   # PP and QQ (or I and Q) of every channel: sum over bins == sum over time
        want = det[:, 0, :ndat * 4].view(lt.nchan_out, ndat, 4)[:, :, :2].double().sum(dim=1)
        got = prof.view(lt.nchan_out, lt.cfg.nbin, 4)[:, :, :2].double().sum(dim=1)
        rel = float((got - want).abs().max() / want.abs().max())
        if not rel <= 2e-5:
            fails.append("profile power differs from the detected time series by %.3g" % rel)
        res["power_rel"] = rel
    if sharded and world > 1:
        try:
            pipeline.check_identical_hits(hits, dist, rank, world)
        except pipeline.DspsrAmdError as e:
            fails.append(str(e))
    if fails:
        raise ParityGateError("bench.py parity gate FAILED on rank %d: %s" % (rank, "; ".join(fails)))
    return {"status": "ok", "checks": (["fused == Detection+Fold " + ("bit for bit" if exact else "to 2e-6 (segmented fused fold / long-run fold)")] if was else []) +
            ["hits.sum() == ndat", "profile power == detected power (rel %.1e)" % res.get("power_rel", 0.0)] +
            (["identical hits on all sub-band ranks"] if sharded and world > 1 else [])}


def engine_boundary(torch):
    """The headline geometry through the C++ Engine adaptors in DSPSR's own call order (tools/engine_boundary_bench.cpp,
    built by dspsr_amd/csrc/Makefile): a child process, started after this process has finished its own timing."""
    import subprocess
    exe = os.path.join(ROOT, "dspsr_amd", "host", "engine_boundary_bench")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(ROOT, "dspsr_amd", "csrc"), "../host/engine_boundary_bench"], capture_output=True)
    try:
        p = subprocess.run([exe, "64", "6", "32"], capture_output=True, text=True, timeout=300)
        rec = json.loads(p.stdout.strip().splitlines()[-1])
    except Exception as e:  # a reported companion figure only
        return {"error": repr(e)}
    rec["how"] = ("C++ adaptors dspsr_amd/host/dspsr_amd_engines.h driven as Filterbank.C:547-553 -> Detection.C:325-334 (ndim 2, in "
                  "place) -> Fold.C:724-741,817-829 per block; eager = every Engine call launches at once, deferred = HIP::Chain "
                  "(one fused launch group at Fold::Engine::fold); float = unpacked float32 rows, raw = the packed 8-bit "
                  "BitSeries handed over (set_raw_input)")
    return rec


RCCL_SETUP_TIMEOUT_S = 120.0


def _bounded(what, fn, rank, timeout=None, device=None):
    """Runs fn() on a helper thread and waits at most RCCL_SETUP_TIMEOUT_S for it.  The HIP (and torch) current device is a
    per-thread setting and a new thread starts on device 0: `device` is made current on the helper first, so that allocations
    and launches of rank r >= 1 land on ITS GPU, where its stream and communicator live.  ncclCommInitRank and the trial exchange
    are collectives: a rank whose peers never arrive would block for ever (ctypes releases the GIL, so this thread can watch).
    A rank that is stuck EXITS non-zero with a message -- the launcher then tears the job down -- instead of hanging the run.
    Returns (result, exception)."""
    import threading
    box = {}

    def run():
        try:
            if device is not None:
                import torch
                torch.cuda.set_device(device)
            box["r"] = fn()
        except BaseException as e:                                        # noqa: BLE001 -- handed to the caller
            box["e"] = e
    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(RCCL_SETUP_TIMEOUT_S if timeout is None else timeout)
    if th.is_alive():
        print("bench: rank %d: %s did not return within %.0f s (a peer failed or never entered it): giving up"
              % (rank, what, RCCL_SETUP_TIMEOUT_S if timeout is None else timeout), file=sys.stderr, flush=True)
        os._exit(3)
    return box.get("r"), box.get("e")


def open_rccl_exchange(ctx, torch, dist, rank, world):
    """The C-ABI communicator (dspsr_amd_comm_*) for this rank's pipeline context plus one tiny trial exchange.
    Every collective step (ncclCommInitRank inside comm_create, the trial's start / finish) is entered only after ALL ranks
    have agreed, over the launcher's process group, that the step before it succeeded everywhere -- a rank that failed
    early (no librccl to dlopen, comm_create refused) can therefore not leave the others blocked inside a collective -- and
    each such step runs under a bounded wait (_bounded).  If any rank cannot, ALL ranks run the exchange through the
    launcher's process group instead -- which is RCCL too -- and the bench line says so at top level (`exchange`).
    Returns (communicator or None, reason or None)."""
    import dspsr_amd
    state = {"ok": 1, "note": None}
    dev = torch.cuda.current_device()               # this rank's GPU: made current on the helper threads below as well

    def agree(stage):
        t = torch.tensor([state["ok"]], dtype=torch.int32, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if int(t.item()) == 1:
            return True
        notes = [None] * world
        dist.all_gather_object(notes, state["note"])
        state["note"] = "%s: %s" % (stage, "; ".join(sorted({n_ for n_ in notes if n_})) or "unknown")
        return False

    def give_up(rccl):
        if rank == 0:
            print("bench: C-ABI RCCL communicator unavailable (%s): exchange through torch.distributed" % state["note"], file=sys.stderr)
        if rccl is not None:
            try:
                rccl.close()
            except Exception:                                             # noqa: BLE001
                pass
        return None, state["note"]

    # 1. librccl loads and answers on EVERY rank (each asks for an id of its own; rank 0's is the one that is used)
    uid = None
    try:
        uid = dspsr_amd.Communicator.unique_id()
    except Exception as e:                                                # noqa: BLE001 -- reported in the line
        state["ok"], state["note"] = 0, "%s" % e
    if not agree("dlopen / ncclGetUniqueId"):
        return give_up(None)
    ids = [uid if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    # 2. ncclCommInitRank: collective, all ranks enter it (they all passed step 1)
    rccl, err = _bounded("dspsr_amd_comm_create (ncclCommInitRank)", lambda: dspsr_amd.Communicator(ctx, world, rank, ids[0]), rank, device=dev)
    if err is not None:
        state["ok"], state["note"] = 0, "%s" % err
    if not agree("comm_create"):
        return give_up(rccl)
    # 3. one tiny exchange: collective again, entered by all (they all hold a communicator)
    def trial():
        t = torch.full((4, 8), float(rank + 1), dtype=torch.float32, device="cuda:%d" % dev)
        rccl.start(rccl.SUM, t.data_ptr(), 8, 4, 8, np.ones(4, np.uint32), 1.0, 4, check_hits=True)
        got = rccl.finish()
        if rank == 0 and (float(got[0][0]) != world * (world + 1) / 2 or int(got[1][0]) != world or not got[4]):
            raise RuntimeError("trial exchange returned %r" % (got[0][:2],))
    _, err = _bounded("the trial exchange", trial, rank, device=dev)
    if err is not None:
        state["ok"], state["note"] = 0, "%s" % err
    if not agree("trial exchange"):
        return give_up(rccl)
    return rccl, None


def make_fold_pipeline(name, args, torch, rank, world, local_rank):
    """The pipeline object and the synthetic 8-bit block (resident in HBM) of one filterbank + detect + fold workload, in
    the launch shape the bench times (parts per block / per launch group).  Shared by run_fold_workload and
    tools/pmc_workload.py, so the PMC counters are collected on exactly the timed launches."""
    import dspsr_amd
    from dspsr_amd import pipeline
    wl = WORKLOADS[name]
    sharded = wl["in_nchan"] > 1                 # sub-band sharding; otherwise time-slice replicas
    if sharded and world > wl["in_nchan"]:
        sys.exit("bench.py: workload %s has %d sub-bands, --gpus %d is more" % (name, wl["in_nchan"], world))
    info = pipeline.InputInfo(centre_frequency=wl["freq"], bandwidth=wl["bw"], nchan=wl["in_nchan"], npol=2,
                              ndim=wl["ndim"], tsamp_us=wl["tsamp_us"], machine=wl["machine"])
    parts_per_block, max_parts = args.parts_per_block or wl.get("parts_per_block", 0), args.max_parts or wl.get("max_parts", 0)
    n_fft = (wl["nchan"] // wl["in_nchan"]) * wl["freq_res"]
    if not parts_per_block:
        nsamp_fft = 2 * n_fft if wl["ndim"] == 1 else n_fft
        parts_per_block = max(16, min(256, (1 << 29) // nsamp_fft))
    if not max_parts:
        # parts per launch group: half a block, or -- small parts -- as many as keep each scratch buffer near 2 GB
        # (the persistent kernels amortise ramp-up and tail over the group and the launch gaps shrink: the 50 MHz sub-band
        # geometry 42.1k / 45.6k / 46.0k Msamples/s at 64 / 128 / 256 parts per group)
        part_bytes = 2 * n_fft * 8          # L = 2N points of real input, or two sequences of N (complex dual-pol)
        max_parts = max(1, parts_per_block // 2, min(parts_per_block, (2 << 30) // part_bytes))
    cfg = pipeline.Config(nchan=wl["nchan"], dispersion_measure=wl["dm"], nbin=wl["nbin"],
                          folding_period=0.0893, freq_res=wl["freq_res"], ndim=args.ndim,
                          parts_per_block=parts_per_block, max_parts=max_parts,
                          fused_fold=not args.no_fused_fold, two_pass=not getattr(args, "no_two_pass", False),
                          convolve_when=wl.get("when", "during"))
    lt = pipeline.LoadToFold(cfg, info, device=local_rank, stream=torch.cuda.current_stream().cuda_stream,
                             subband=rank if sharded else None)

    # synthetic block resident in HBM: seeded Gaussian noise, sigma = 24 LSB (content does not change the work; the
    # comparison with the oracle on a dispersed pulsed signal is oracle_check() below, beside the number, and the -m gpu
    # tests, tests/test_gpu_headline.py)
    nbytes = lt.block_bytes()
    gen = torch.Generator(device="cuda").manual_seed(20100413 + rank)
    raw = torch.empty(nbytes, dtype=torch.int8, device="cuda")
    chunk = 1 << 26
    for s in range(0, nbytes, chunk):
        e = min(nbytes, s + chunk)
        raw[s:e] = torch.randn(e - s, generator=gen, device="cuda").mul_(24.0).round_().clamp_(-128, 127).to(torch.int8)

    return lt, raw, cfg, info, wl, sharded


def run_fold_workload(name, args, torch, dist, rank, world, local_rank, single, steps, warmup, full=True):
    """One filterbank + detect + fold workload on this process group: the timed region (barrier, `steps` steps with a
    sub-integration dump every args.dump_steps, barrier), then -- outside it -- the unfused roofline blocks and the
    parity gate.  Returns the result record on rank 0 (None elsewhere).  full: cpu_baseline / --h2d companions too."""
    import dspsr_amd
    from dspsr_amd import pipeline
    lt, raw, cfg, info, wl, sharded = make_fold_pipeline(name, args, torch, rank, world, local_rank)
    nbytes = raw.numel()
    # the exchange: RCCL behind the C-ABI (dspsr_amd_comm_*), one communicator per pipeline context; the unique id travels
    # over the launcher's process group.  One-device rehearsals (gloo) keep the torch.distributed form of the same exchange.
    gather, rccl = None, None
    rccl_note = None
    if world > 1 and not single:
        rccl, rccl_note = open_rccl_exchange(lt.ctx, torch, dist, rank, world)
        if rccl is not None:
            lt.set_rccl_communicator(rccl)
            lt.copy_subints = False      # no archive writer here: each merged sub-integration is left in the pinned buffer it arrived in
        elif sharded:
            gather = torch.zeros(world * lt.nchan_out * lt.npol_out * cfg.nbin * cfg.ndim, dtype=torch.float32, device="cuda")
    elif world > 1 and sharded:
        gather = torch.zeros(world * lt.nchan_out * lt.npol_out * cfg.nbin * cfg.ndim, dtype=torch.float32, device="cuda")
    comm = dist if world > 1 else None

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def dump(wait=False, check=False):
        # (the dump of block i overlaps the kernels of block i+1: snapshot on the compute stream, collective on the
        #  communicator's stream; hits are checked in the parity gate and in the timed dumps of `reduce_ms_per_dump`)
        lt.finish_subint(comm, rank, world, gather, replicas=not sharded, check_hits=check, wait=wait)
        lt.subints.clear()

    blocks_done = [0]

    def step(i, ev=None):
        if not sharded:
            lt.seek_block(blocks_done[0] * world + rank)       # replica r takes blocks r, r + N, ... of the stream
        blocks_done[0] += 1
        lt.process_block(raw, events=ev)
        if (i + 1) % args.dump_steps == 0:
            dump()

    for i in range(warmup):
        step(i)
    lt.collect_subint()
    barrier()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    nsamp0 = lt.nsamples_in
    t_start = time.perf_counter()
    for i in range(steps):
        step(i, events[i])
    dump()
    lt.collect_subint()                                                   # the last exchange has arrived on the root
    lt.subints.clear()
    barrier()
    elapsed = time.perf_counter() - t_start
    samples = float((lt.nsamples_in - nsamp0) * lt.in_nchan)          # this rank's own samples (per pol)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        t = torch.tensor([samples], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        samples = float(t.item())
    value = samples / elapsed / 1e6

    # ---- outside the timed region: the dump alone, roofline blocks, parity gate, baselines ---------------------
    reduce_ms, hits_ok = None, None
    if world > 1:
        # the exchange by itself, start to arrival on the root, with the hits check in the same group (max over ranks)
        lt.process_block(raw)
        barrier()
        nd = 4
        t1 = time.perf_counter()
        for _ in range(nd):
            dump(wait=True, check=True)                                   # raises if the ranks disagree on hits[]
        torch.cuda.synchronize()
        t = torch.tensor([(time.perf_counter() - t1) / nd * 1e3], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        reduce_ms, hits_ok = float(t.item()), True
        lt.fold.zero()
        lt.hits[:] = 0
        lt.integration_length, lt.ndat_total = 0.0, 0
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
        extra = max(4, steps // 4)
        ev2 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(extra)]
        lt.process_block(raw)
        for e in ev2:
            lt.process_block(raw, events=e)
        torch.cuda.synchronize()
        fb_ms = sum(a.elapsed_time(b) for a, b in ev2) / len(ev2)
        lt.fused_fold = True
        lt.fold.zero()
        lt.hits[:] = 0
        lt.integration_length, lt.ndat_total = 0.0, 0
    lt.set_rccl_communicator(None)                                       # the gate's hits check goes through the process group
    gate = parity_gate(lt, raw, torch, dist, rank, world, sharded)       # every rank; raises on failure
    out = None
    if rank == 0:
        r = lt.response
        if r is None:                                   # Config::Never: the filterbank alone (freq_res = 1, nothing dropped)
            import types
            r = types.SimpleNamespace(ndat=1, impulse_pos=0, impulse_neg=0, kernel=None)
        nchan_subband = cfg.nchan // info.nchan
        N = nchan_subband * r.ndat
        nsamp_fft = 2 * N if info.ndim == 1 else N
        b_alg = algorithmic_bytes_per_part(2, nsamp_fft * info.ndim, 8, N, nchan_subband, lt.nkeep)
        if lt.response is None:
            b_alg -= 8 * N                              # no response: no chirp to read
        achieved = b_alg * cfg.parts_per_block * lt.in_nchan / (fb_ms * 1e-3) / 1e9
        npass = lt.fb.npass(True)
        group = {2: ("k_raw_cols+k_fwd_col1+k_rows_inv<.,.,false>", "k_raw_cols+k_fwd_col1+k_rows_inv<.,.,true>"),
                 3: ("k_raw_transpose+k_fwd_cols+k_fwd_rows+k_inv_chan<.,false>", "k_raw_transpose+k_fwd_cols+k_fwd_rows+k_inv_chan<.,true>"),
                 4: ("k_raw_transpose+k_fwd_cols+k_fwd_rows+k_inv_a+k_inv_b", "k_raw_transpose+k_fwd_cols+k_fwd_rows+k_inv_a+k_inv_b<.,true>+k_fold_segsum"),
                 }.get(npass, ("%d tile passes" % npass,) * 2)
        if getattr(cfg, "convolve_when", "during") in ("after", "never"):
            # Config::After / Never: the non-convolving filterbank, then the convolution on its float rows (one tile pass or four)
            group = {1: ("k_fb_plain",) * 2, 2: ("k_fb_plain+k_conv1",) * 2, 4: ("k_fb_plain+k_conv3_a+k_conv3_b+k_conv3_c",) * 2,
                     5: ("k_fb_plain+k_fwd_cols+k_fwd_rows+k_inv_a+k_inv_b",) * 2}.get(npass, ("k_fb_plain + %d tile passes" % (npass - 1),) * 2)
        exch = ("dspsr_amd_reduce_profiles_* (RCCL behind the C-ABI, csrc/comm.hip), snapshot on the compute stream, collective "
                "on its own stream" if rccl is not None else
                "torch.distributed (gloo rehearsal on one device)" if world > 1 and single else
                "torch.distributed over RCCL -- the C-ABI communicator was unavailable: %s" % rccl_note if world > 1 else
                "one rank: no exchange")
        # which transport carried the dumps, at a glance (a SCALE record must not mistake the fallback for the C-ABI path)
        exchange = ("rccl-c-abi" if rccl is not None else "gloo-rehearsal" if world > 1 and single else
                    "fallback" if world > 1 else "none")
        par = ("sub-band per GPU x%d (input channel g of %d, slice g of the full-band kernel; per dump ONE ncclGather of the "
               "ranks' slices; %s)" % (world, info.nchan, exch)) if sharded else \
              ("time-slice replicas x%d (blocks dealt round robin; per dump ONE packed ncclReduce(SUM) of profile + hits + "
               "lengths; %s)" % (world, exch) if world > 1 else "single GPU")
        out = {
            "metric": "Msamples/s dedispersed+folded", "value": round(value, 2), "unit": "Msamples/s",
            "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "exchange": exchange,
            "config": {"workload": name, "command": wl["cmd"], "nchan": cfg.nchan, "freq_res": r.ndat,
                       "n_fft": N, "nfilt_pos": r.impulse_pos, "nfilt_neg": r.impulse_neg, "nkeep": lt.nkeep,
                       "nsamp_step": lt.nsamp_step, "dm": cfg.dispersion_measure, "nbin": cfg.nbin,
                       "parts_per_block": cfg.parts_per_block, "max_parts": cfg.max_parts,
                       "input": "8-bit dual-pol, resident in HBM", "library_build": dspsr_amd.build_id(),
                       "detected_ndim": cfg.ndim, "fused_fold": bool(fused), "transform_passes": npass, "parallelism": par,
                       "realtime_factor": round(value / world / (info.rate / 1e6), 3)},
            "parity_gate": gate,
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": measured_traffic(name, cfg.max_parts),
                         "traffic_source": measured_traffic.source,
                         "traffic_unit": "HBM bytes per launch group of %d parts (PMC counters of a separate rocprofv3 run on the "
                                         "same library build, see traffic_source -- not measured in this run); "
                                         "algorithmic bytes for the same group: %d" % (cfg.max_parts, b_alg * cfg.max_parts),
                         "kernel": "filterbank launch group %s (FFT+chirp+detect, detected output written)" % group[0],
                         "algorithmic_bytes_per_part": b_alg, "group_ms_per_block": round(fb_ms, 4),
                         "traffic_ratio": None,
                         "region": ("%d extra blocks right after the timed region with Detection and Fold as "
                                    "separate operations (HIP events on the launch stream)" % extra) if fused
                                   else "the timed region (HIP events on the launch stream)"},
        }
        if out["roofline"]["traffic"]:
            out["roofline"]["traffic_ratio"] = round(out["roofline"]["traffic"] / (b_alg * cfg.max_parts), 3)
        out["config"]["exchange"] = exchange
        if exchange == "fallback":
            out["config"]["exchange_fallback_reason"] = rccl_note
        out["config"]["reduce_ms_per_dump"] = round(reduce_ms, 4) if world > 1 else None      # one rank: no exchange
        if world > 1:
            out["config"]["identical_hits"] = hits_ok
        if fused:
            b_fused = b_alg - 2 * nchan_subband * lt.nkeep * 8         # no detected output: input once + chirp once
            ach_f = b_fused * cfg.parts_per_block * lt.in_nchan / (timed_ms * 1e-3) / 1e9
            out["roofline_fused"] = {
                "bound": "hbm", "achieved": round(ach_f, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach_f / HBM_PEAK_GBS, 4),
                "traffic": measured_traffic(name, cfg.max_parts, "hbm_bytes_per_launch_group_fused"),
                "kernel": "%s (FFT+chirp+detect+fold in one launch group, the timed region)" % group[1],
                "algorithmic_bytes_per_part": b_fused, "group_ms_per_block": round(timed_ms, 4),
                "note": "the fused group also does the fold; its algorithmic bytes have no output term "
                        "(SURVEY 8(d)), so this fraction is not comparable with roofline.frac"}
        if full and (args.h2d or (world == 1 and not args.no_h2d)):
            # The PCIe-inclusive companion of `value` (the reference's path always starts in host memory,
            # TransferCUDA.C:24-85 / TransferBitSeriesCUDA.C:23-70): host-buffer hand-over through the product's own feeder
            # (pipeline.LoadToFold.process_host_blocks): block i+1 is copied H2D on a second stream while block i is
            # processed.  By default a short leg (5 blocks) beside the headline; --h2d times `steps` blocks.
            hsteps = steps if args.h2d else min(steps, 5)
            host = raw.cpu().pin_memory()
            lt.process_host_blocks(host for _ in range(2))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            lt.process_host_blocks(host for _ in range(hsteps))
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            out["config"]["pcie_inclusive"] = {
                "value": round(cfg.parts_per_block * lt.nsamp_step * hsteps / dt / 1e6, 1), "unit": "Msamples/s", "steps": hsteps,
                "effective_GBps": round(raw.numel() * hsteps / dt / 1e9, 1),
                "note": "the same blocks copied from pinned host memory (%.1f MB each) on a second stream, double buffered, "
                        "overlapped with the kernels; `value` above is for blocks already resident in HBM" % (raw.numel() / 1e6)}
            del host
        if full and world == 1 and not args.no_cpu_baseline and "when" not in wl:
            geom = {"freq_res": r.ndat, "nkeep": lt.nkeep, "nsamp_step": lt.nsamp_step, "nsamp_overlap": lt.nsamp_overlap,
                    "nfilt_pos": r.impulse_pos, "kernel": r.kernel, "scale8": lt.scale8, "out_rate": lt.out_rate}
            try:
                out["cpu_baseline"] = cpu_baseline(wl, geom)
            except Exception as e:  # the oracle is only a reported baseline
                out["cpu_baseline"] = {"value": None, "unit": "Msamples/s", "cores": 1, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
            if not sharded and args.ndim == 4:
                out["parity_gate"]["oracle_check"] = oracle_check(wl, cfg, info, torch)    # raises on a mismatch
    if rccl is not None:
        rccl.close()
    lt.close()
    del raw, gather
    torch.cuda.empty_cache()
    return out


def brief(rec):
    """One line of `other_workloads` / `subband_shard` from a full result record."""
    b = {"workload": rec["config"]["workload"], "value": rec["value"], "unit": rec["unit"], "ms_per_step": rec["ms_per_step"],
         "steps": rec["steps"], "roofline_frac": rec["roofline"]["frac"], "roofline_kernel": rec["roofline"]["kernel"].split(" (")[0],
         "command": rec["config"]["command"]}
    for k in ("parts_per_block", "max_parts", "fused_fold", "parallelism", "exchange", "exchange_fallback_reason",
              "reduce_ms_per_dump", "identical_hits"):
        if k in rec["config"]:
            b[k] = rec["config"][k]
    b["roofline_traffic_ratio"] = rec["roofline"].get("traffic_ratio")
    b["roofline_traffic_source"] = rec["roofline"].get("traffic_source")
    if "parity_gate" in rec:
        b["parity_gate"] = rec["parity_gate"]["status"]
    return b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="one workload only.  Default: the headline (`target`) as the result line, plus -- companions in the same "
                         "line -- the sub-band sharded run of cfg4 (the north-star scaling curve), and at --gpus 1 the other "
                         "BASELINE configurations in short runs and the headline through the C++ Engine adaptors")
    ap.add_argument("--parts-per-block", type=int, default=0,
                    help="overlap-save parts per block; 0 = 64 for the headline geometry (2^23 samples per part), "
                         "more for smaller parts so that a block stays near 5e8 samples (capped at 256)")
    ap.add_argument("--max-parts", type=int, default=0, help="parts per launch group; 0 = parts_per_block/2, more for small parts (2 GB of scratch)")
    ap.add_argument("--dump-steps", type=int, default=8, help="steps per sub-integration dump")
    ap.add_argument("--ndim", type=int, default=4, choices=[1, 2, 4])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-companions", action="store_true", help="default run without the companion measurements")
    ap.add_argument("--h2d", action="store_true",
                    help="also measure the PCIe-inclusive rate: every block copied from pinned host memory on a second "
                         "stream, double buffered, overlapped with the kernels (reported as config.pcie_inclusive)")
    ap.add_argument("--no-h2d", action="store_true", help="skip the short PCIe-inclusive leg of the default single-GPU run")
    ap.add_argument("--no-two-pass", action="store_true",
                    help="short-response geometries (cfg4) through the three-pass kernels instead of the two-pass path (A/B runs)")
    ap.add_argument("--no-fused-fold", action="store_true",
                    help="Detection and Fold as separate operations (detected time series through HBM)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(sys.argv[1:], args.gpus))          # nothing has touched the GPU yet

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
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

    def finish(rec):
        if rank == 0 and rec is not None:
            print(json.dumps(rec), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()

    name = args.workload or "target"
    if name in ("cfg5", "cfg5c", "fold"):
        if world > 1:
            sys.exit("bench.py: the %s workload runs as independent replicas; use --gpus 1" % name)
        fn = {"cfg5": bench_search_mode, "cfg5c": bench_search_coherent, "fold": bench_fold_only}[name]
        return finish(fn(args, WORKLOADS[name], torch))
    ctx = (torch, dist, rank, world, local_rank, single)
    out = run_fold_workload(name, args, *ctx, steps=args.steps, warmup=args.warmup, full=True)
    if args.workload is None and not args.no_companions and not (args.parts_per_block or args.max_parts):
        # ---- companions of the default line (the scaling curve is quoted on the sub-band sharded
        #      configuration; every BASELINE configuration gets a driver-run number) ---------------------------------
        short = argparse.Namespace(**vars(args))
        ssteps, swarm = max(5, min(args.steps, 10)), 3
        if world <= WORKLOADS["cfg4"]["in_nchan"]:
            # (blocks of 1.4 ms: enough of them that the first block's host preparation and the last dump do not weigh)
            shard = run_fold_workload("cfg4", short, *ctx, steps=max(4 * ssteps, 8 * args.dump_steps), warmup=swarm, full=False)
            if rank == 0:
                out["subband_shard"] = brief(shard)
                out["subband_shard"]["note"] = ("the north-star scaling curve: rank g = sub-band g of the NCHAN-8 band (weak scaling, "
                                                "%d of 8 sub-bands in this run); `value` above stays the headline geometry as time-slice "
                                                "replicas so that N = 1 agrees with the single-GPU record" % world)
        if world == 1:
            others = []
            for w in ("cfg1", "cfg1opt", "cfg2", "cfg3", "odd_nchan", "odd_fres", "after", "after8k", "after1k", "after8c", "plain"):
                others.append(brief(run_fold_workload(w, short, *ctx, steps=2 * ssteps, warmup=swarm, full=False)))
            sm = argparse.Namespace(**vars(args))
            sm.steps, sm.warmup, sm.no_cpu_baseline = 2 * ssteps, swarm, True
            rec5 = bench_search_mode(sm, WORKLOADS["cfg5"], torch)
            others.append({"workload": "cfg5", "value": rec5["value"], "unit": rec5["unit"], "ms_per_step": rec5["ms_per_step"], "steps": rec5["steps"],
                           "roofline_frac": rec5["roofline"]["frac"], "roofline_kernel": rec5["roofline"]["kernel"],
                           "roofline_traffic_ratio": rec5["roofline"].get("traffic_ratio"), "roofline_traffic_source": rec5["roofline"].get("traffic_source"),
                           "command": rec5["config"]["command"]})
            rec5c = bench_search_coherent(sm, WORKLOADS["cfg5c"], torch)
            others.append({"workload": "cfg5c", "value": rec5c["value"], "unit": rec5c["unit"], "ms_per_step": rec5c["ms_per_step"], "steps": rec5c["steps"],
                           "roofline_frac": rec5c["roofline"]["frac"], "roofline_kernel": rec5c["roofline"]["kernel"],
                           "roofline_traffic_ratio": rec5c["roofline"].get("traffic_ratio"), "roofline_traffic_source": rec5c["roofline"].get("traffic_source"),
                           "command": rec5c["config"]["command"]})
            fo = argparse.Namespace(**vars(args))
            fo.steps, fo.warmup, fo.no_cpu_baseline = 2 * ssteps, swarm, True
            recf = bench_fold_only(fo, WORKLOADS["fold"], torch)                 # BASELINE.md benchmark B (Benchmark/fold.csh): dsp::Fold alone
            others.append({"workload": "fold", "value": recf["value"], "unit": recf["unit"], "ms_per_step": recf["ms_per_step"],
                           "steps": recf["steps"], "roofline_frac": recf["roofline"]["frac"], "roofline_kernel": recf["roofline"]["kernel"],
                           "roofline_traffic_ratio": recf["roofline"].get("traffic_ratio"), "roofline_traffic_source": recf["roofline"].get("traffic_source"),
                           "command": recf["config"]["command"]})
            out["other_workloads"] = others
            out["config"]["engine_boundary"] = engine_boundary(torch)
    finish(out)


if __name__ == "__main__":
    main()
