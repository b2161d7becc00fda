"""tools/fold_host_probe.py : host time of one stand-alone fold step at Benchmark/fold.csh's shape (set_bins = the double-precision
bin plan on the host; fold = dense-table build + uploads + launch) against the kernel time: is the fold benchmark host bound?"""
import sys, os, json, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dspsr_amd
from dspsr_amd import pipeline
nchan, npol, rate = 1024, 4, 1e6 / 32.0
text = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "vela_polyco.json")))["text"]
polyco = pipeline.Polyco(text)
day, sec0 = 55299, 7545.0
nbin = pipeline.choose_nbin(1.0 / polyco.frequency(day, sec0), rate)
ctx = dspsr_amd.Context(0, torch.cuda.current_stream().cuda_stream)
for ndat in (1 << 17, 1 << 18):
    fold = dspsr_amd.FoldEngine(ctx)
    fold.set_shape(nchan, npol, 1, nbin)
    det = torch.rand((nchan, npol, ndat), dtype=torch.float32, device="cuda")
    hits = np.zeros(nbin, np.uint32)
    f = polyco.frequency(day, sec0)
    for _ in range(3):
        fold.set_nbin(nbin); fold.set_ndat(ndat, 0); fold.set_bins(0.25, (1.0 / rate) * f, ndat, 0, hits); fold.fold(det)
    torch.cuda.synchronize()
    n = 20
    tb = tf = 0.0
    t0 = time.perf_counter()
    for _ in range(n):
        a = time.perf_counter()
        fold.set_nbin(nbin); fold.set_ndat(ndat, 0); fold.set_bins(0.25, (1.0 / rate) * f, ndat, 0, hits)
        b = time.perf_counter()
        fold.fold(det)
        c = time.perf_counter()
        tb += b - a; tf += c - b
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("ndat 2^%d: host set_bins %.0f us, host fold() %.0f us per step; enqueue loop %.0f us per step, with final sync %.0f us per step"
          % (int(np.log2(ndat)), tb / n * 1e6, tf / n * 1e6, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6), flush=True)
    fold.close()
ctx.close()
