"""tests/fuzz_fold.py [ncases] [seed] : random fold shapes and bin plans against the CPU loop (Fold.C:835-891 restated with
numpy, strict time order): bit-identical for plans without long runs, <= 2e-6 of the profile maximum otherwise; hits identical.
Also random LoadToFold configurations, fused against Detection + Fold."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import dspsr_amd
from dspsr_amd import pipeline, synth

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = dspsr_amd.Context(0, torch.cuda.current_stream().cuda_stream)
bad = 0
for i in range(ncases):
    ndim = int(rng.choice([1, 2, 4]))
    npol = 4 // ndim if rng.integers(0, 3) else 1
    nchan = int(rng.choice([1, 3, 17, 64, 300, 700]))
    nbin = int(rng.choice([8, 31, 64, 100, 500, 1024, 2048, 5000]))
    ndat = int(rng.integers(50, 40000))
    spb = float(rng.choice([0.3, 1.7, 9.0, 40.0, 90.0, 700.0, 5000.0]))          # samples per bin
    pps = 1.0 / (spb * nbin)
    phi = float(rng.random())
    idat_start = int(rng.integers(0, 40))
    ndat_fold = ndat - idat_start - int(rng.integers(0, 10))
    det = (rng.standard_normal((nchan, npol, ndat, ndim)).astype(np.float32)) ** 2
    d = torch.from_numpy(det.reshape(nchan, npol, ndat * ndim)).cuda()
    eng = dspsr_amd.FoldEngine(ctx)
    eng.set_shape(nchan, npol, ndim, nbin)
    hits = np.zeros(nbin, np.uint32)
    eng.set_nbin(nbin)
    eng.set_ndat(ndat_fold, idat_start)
    eng.set_bins(phi, pps, ndat_fold, idat_start, hits)
    eng.fold(d)
    got = eng.synch()
    eng.close()
    plan, want_hits = dspsr_amd.fold_binplan(phi, pps, nbin, ndat_fold)
    want = np.zeros((nchan, npol, nbin, ndim), np.float32)
    seg = det[:, :, idat_start:idat_start + ndat_fold, :]
    for b in np.unique(plan):                                    # per bin, time order: cumulative float32 sum
        idx = np.nonzero(plan == b)[0]
        acc = np.zeros((nchan, npol, ndim), np.float32)
        for t in idx:
            acc += seg[:, :, t, :]
        want[:, :, b, :] = acc
    desc = "nchan=%d npol=%d ndim=%d nbin=%d ndat=%d samples/bin=%g" % (nchan, npol, ndim, nbin, ndat_fold, spb)
    ok_hits = np.array_equal(hits, want_hits)
    exact = np.array_equal(got, want)
    # (float32 sums of n samples in two association orders differ by about eps*sqrt(n): the bound grows with the run length)
    close = np.abs(got - want).max() <= 2e-6 * max(1.0, (float(hits.max()) / 100.0) ** 0.5) * max(np.abs(want).max(), 1e-30)
    long_runs = spb >= 60
    if ok_hits and (exact or (long_runs and close)):
        print("ok   ", desc, "exact" if exact else "rounding", flush=True)
    else:
        bad += 1
        print("FAIL ", desc, "hits", ok_hits, "exact", exact, "close", close, flush=True)
# pipeline: fused against separate launches
for i in range(max(4, ncases // 3)):
    freq = float(rng.choice([1382.0, 400.0, 3100.0]))
    bw = float(rng.choice([-16.0, 16.0, -64.0, 8.0]))
    tsamp = 1.0 / (2.0 * abs(bw))                      # real sampling of the band
    dm = float(rng.choice([0.0, 3.0, 30.0, 120.0]))
    nchan = int(2 ** rng.integers(1, 10))
    nbin = int(rng.choice([16, 64, 256, 1024]))
    period = float(rng.choice([0.0007, 0.004, 0.0371]))
    ppb, mp = int(rng.integers(1, 6)), int(rng.integers(1, 4))
    sub = float(rng.choice([0.0, 0.0, 0.0021]))
    info = pipeline.InputInfo(centre_frequency=freq, bandwidth=bw, tsamp_us=tsamp, machine="DADA")
    res = []
    refused = None
    for fused in (True, False):
        cfg = pipeline.Config(nchan=nchan, dispersion_measure=dm, nbin=nbin, folding_period=period, ndim=4, parts_per_block=ppb,
                              max_parts=mp, fused_fold=fused, force_fused=fused, subint_seconds=sub)
        try:
            lt = pipeline.LoadToFold(cfg, info, device=0, stream=torch.cuda.current_stream().cuda_stream)
            if 3 * ppb * lt.nsamp_step + lt.nsamp_overlap > (1 << 25):
                lt.close()
                raise dspsr_amd.DspsrAmdError("block too long for a sweep")
        except dspsr_amd.DspsrAmdError as e:
            refused = str(e)
            break
        step = ppb * lt.nsamp_step
        raw = torch.from_numpy(synth.voltages(3 * step + lt.nsamp_overlap, freq, bw, tsamp, max(dm, 1.0), period, seed=7 + i)).cuda()
        for b in range(3):
            lt.process_block(raw[2 * b * step: 2 * (b * step + step + lt.nsamp_overlap)])
        if lt.ndat_total:
            lt.finish_subint()
        lt.synchronize()
        res.append([(s["hits"].copy(), s["profile_dev"].cpu().numpy(), s["ndat_total"]) for s in lt.subints])
        mode = lt.fused_mode if fused else mode
        freq_res = lt.response.ndat
        lt.close()
    if refused is not None:
        print("refused pipeline freq=%g bw=%g DM=%g nchan=%d -- %s" % (freq, bw, dm, nchan, refused[:90]), flush=True)
        continue
    desc = "pipeline freq=%g bw=%g DM=%g nchan=%d freq_res=%d nbin=%d period=%g parts/block=%d max_parts=%d subint=%g (%d sub-integrations, mode %d)" % (
        freq, bw, dm, nchan, freq_res, nbin, period, ppb, mp, sub, len(res[0]), mode)
    good = len(res[0]) == len(res[1])
    for a, b in zip(res[0], res[1]):
        # (float32 sums of N samples per bin, associated differently by the two paths: the same sqrt(N) allowance as above --
        #  186 000 hits per bin gave 3.8e-6, identical for every launch shape)
        scale = max(np.abs(b[1]).max(), 1e-30) * max(1.0, (float(b[0].max()) / 100.0) ** 0.5)
        good = good and np.array_equal(a[0], b[0]) and a[2] == b[2] and np.abs(a[1] - b[1]).max() <= 2e-6 * scale
    if good:
        print("ok   ", desc, flush=True)
    else:
        bad += 1
        print("FAIL ", desc, flush=True)
ctx.close()
print("%d failures" % bad)
sys.exit(1 if bad else 0)
