"""tests/fuzz_misc.py [ncases] [seed] : invariants between code paths that must give the SAME bits.
  * search-mode front end: a 16-byte aligned block (whole-range loads through LDS) against the same bytes at an offset of
    2 bytes (element-wise loads); random nchan / tscrunch / parts / byte order / pscrunch;
  * filterbank on float32 rows: aligned rows (regrouped per tile first) against rows shifted by one float (read in place);
  * pipeline: the detected layouts ndim 4 / 2 / 1 fold the same products in the same order -- identical sums, with and
    without -K, with sub-integrations."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import importlib.util

spec = importlib.util.spec_from_file_location("oracle_mod", os.path.join(ROOT, "oracle", "dspsr_oracle.py"))
oracle = importlib.util.module_from_spec(spec)
sys.modules["oracle_mod"] = oracle
spec.loader.exec_module(oracle)
import dspsr_amd
from dspsr_amd import pipeline, synth

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = dspsr_amd.Context(0, torch.cuda.current_stream().cuda_stream)
bad = 0


def report(ok, desc):
    global bad
    print(("ok    " if ok else "FAIL  ") + desc, flush=True)
    bad += 0 if ok else 1


for i in range(ncases):
    nchan = int(2 ** rng.integers(4, 13))
    L = 2 * nchan
    T = 16384 // L
    sf = int(rng.choice([1, 2, 4, 8, 16, 32, 64]))
    if not (sf % T == 0 or T % sf == 0):
        continue
    npart = int(rng.integers(1, 5)) * max(sf, 1) * int(rng.integers(1, 4)) + int(rng.integers(0, sf))
    caspsr = bool(rng.integers(0, 2))
    psc = bool(rng.integers(0, 2))
    nbytes = npart * L * 2
    buf = torch.randint(-128, 128, (nbytes + 32,), dtype=torch.int8, device="cuda")
    outs = []
    for off in (0, 16 if caspsr else 2):                    # (a CASPSR stream can only move by whole groups: both aligned)
        raw = buf[16:16 + nbytes] if off == 0 else None
        if off:
            shifted = torch.empty(nbytes + 64, dtype=torch.int8, device="cuda")
            shifted[off:off + nbytes] = buf[16:16 + nbytes]
            raw = shifted[off:off + nbytes]
        out = torch.zeros((max(1, npart // sf), nchan, 1 if psc else 2), dtype=torch.float32, device="cuda")
        dspsr_amd.tfp_filterbank(ctx, raw, nchan, npart, out, psc, sf, dspsr_amd.RAW_CASPSR if caspsr else dspsr_amd.RAW_GENERIC, 0.01)
        outs.append(out)
    torch.cuda.synchronize()
    good = bool(torch.equal(outs[0], outs[1])) and float(outs[0].abs().max()) > 0 or npart < sf
    if good and npart >= sf and npart * L <= (1 << 20):      # and against the float64 restatement (TFPFilterbank.C:27-101, TScrunch.C:180-206)
        obs = oracle.Observation(machine="CASPSR" if caspsr else "DADA")
        un = oracle.unpack_8bit(buf[16:16 + nbytes].cpu().numpy(), obs, scale=0.01)
        want = oracle.tscrunch_tfp(oracle.tfp_filterbank(un, nchan, psc, dtype=np.float64), sf)
        got = outs[0].cpu().numpy().astype(np.float64)[:want.shape[0]]
        good = np.abs(got - want).max() <= 2e-5 * np.abs(want).max()
    report(good, "tfp nchan=%d tscrunch=%d npart=%d caspsr=%s pscrunch=%s" % (nchan, sf, npart, caspsr, psc))

for i in range(max(3, ncases // 4)):
    logN = int(rng.integers(10, 19))
    logM = int(rng.integers(4, min(logN - 5, 13) + 1))
    C, M = 1 << (logN - logM), 1 << logM
    real = bool(rng.integers(0, 3))
    pos, neg = int(rng.integers(0, M // 4)), int(rng.integers(0, M // 4))
    kern = np.exp(2j * np.pi * rng.random(C * M)).astype(np.complex64)
    npart = int(rng.integers(1, 4))
    fb = dspsr_amd.FilterbankEngine(ctx).setup(C, M, pos, neg, 1, 2, real, kern, max_parts=int(rng.integers(1, 3)),
                                               force_four_pass=bool(rng.integers(0, 3) == 0))
    nd = 1 if real else 2
    nfl = (npart * fb.nsamp_step + fb.nsamp_overlap) * nd
    rows = ((nfl + 8 + 3) // 4) * 4
    base = torch.randn((1, 2, rows), dtype=torch.float32, device="cuda")
    outs = []
    for shift in (0, 1):
        store = torch.empty((1, 2, rows + 4), dtype=torch.float32, device="cuda")
        store[:, :, shift:shift + nfl] = base[:, :, :nfl]
        x = store[:, :, shift:shift + nfl]
        out = torch.zeros((C, 2, 2 * npart * fb.nkeep), dtype=torch.float32, device="cuda")
        fb.perform(x, out, npart, fb.nsamp_step * nd, 2 * fb.nkeep)
        outs.append(out)
    torch.cuda.synchronize()
    fb.close()
    report(bool(torch.equal(outs[0], outs[1])) and float(outs[0].abs().max()) > 0,
           "float rows aligned == shifted  C=%d M=%d real=%s nfilt=(%d,%d) npart=%d" % (C, M, real, pos, neg, npart))

freq, bw, tsamp, dm = 1382.0, -16.0, 1.0 / 32.0, 30.0
for i in range(max(3, ncases // 6)):
    nchan = int(rng.choice([16, 64, 128]))
    nbin = int(rng.choice([16, 64, 256]))
    period = float(rng.choice([0.0007, 0.004, 0.0371]))
    ppb, mp = int(rng.integers(1, 5)), int(rng.integers(1, 4))
    sub = float(rng.choice([0.0, 0.0031]))
    K = bool(rng.integers(0, 2))
    stokes = bool(rng.integers(0, 2))
    info = pipeline.InputInfo(centre_frequency=freq, bandwidth=bw, tsamp_us=tsamp, machine="DADA")
    res = {}
    for ndim in (4, 2, 1):
        cfg = pipeline.Config(nchan=nchan, dispersion_measure=dm, nbin=nbin, folding_period=period, ndim=ndim, parts_per_block=ppb,
                              max_parts=mp, subint_seconds=sub, interchan_dedispersion=K, stokes=stokes)
        try:
            lt = pipeline.LoadToFold(cfg, info, device=0, stream=torch.cuda.current_stream().cuda_stream)
        except dspsr_amd.DspsrAmdError as e:
            res = None
            print("refused pipeline", str(e)[:90], flush=True)
            break
        step = ppb * lt.nsamp_step
        raw = torch.from_numpy(synth.voltages(4 * step + lt.nsamp_overlap, freq, bw, tsamp, dm, period, seed=70 + i)).cuda()
        for b in range(4):
            lt.process_block(raw[2 * b * step: 2 * (b * step + step + lt.nsamp_overlap)])
        if lt.ndat_total:
            lt.finish_subint()
        lt.synchronize()
        # [chan][npol][bin][ndim] -> [chan][bin][4 products]
        res[ndim] = [(s["hits"].copy(), s["profile_dev"].cpu().numpy().reshape(nchan, 4 // ndim, nbin, ndim).transpose(0, 2, 1, 3)
                      .reshape(nchan, nbin, 4), s["ndat_total"]) for s in lt.subints]
        lt.close()
    if res is None:
        continue
    ok = len(res[4]) == len(res[2]) == len(res[1]) and len(res[4]) >= 1
    # bins of 64 samples or more take the long-run fold, whose time segments depend on the number of rows: equal to rounding
    wide = (32e6 / 2 / nchan) * period / nbin >= 60
    for a, b, c in zip(res[4], res[2], res[1]):
        ok = ok and np.array_equal(a[0], b[0]) and np.array_equal(a[0], c[0]) and a[2] == b[2] == c[2]
        if wide:
            tol = 2e-6 * max(1.0, (float(a[0].max()) / 100.0) ** 0.5) * np.abs(a[1]).max()
            ok = ok and np.abs(a[1] - b[1]).max() <= tol and np.abs(a[1] - c[1]).max() <= tol
        else:
            ok = ok and np.array_equal(a[1], b[1]) and np.array_equal(a[1], c[1])
        ok = ok and np.abs(a[1]).max() > 0
    report(ok, "pipeline ndim 4 == 2 == 1  nchan=%d nbin=%d period=%g parts/block=%d max_parts=%d subint=%g -K=%s stokes=%s (%d sub-ints)"
           % (nchan, nbin, period, ppb, mp, sub, K, stokes, len(res[4])))
ctx.close()
print("%d failures" % bad)
sys.exit(1 if bad else 0)
