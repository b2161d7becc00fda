"""tools/fold_stride_probe.py : Benchmark/fold.csh shape (1024 channels x 4 products x 2^17 samples) with the rows of the detected
block at their natural power-of-two stride and padded by PAD floats -- does the row stride (512 KB: every workgroup's stream at
the same offset modulo a large power of two) cost HBM channel conflicts?  Prints ms per block and TB/s."""
import sys, os, json
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
for pad, ndat in ((0, 1 << 17), (2112, 1 << 17), (0, 1 << 15), (0, 1 << 16), (0, 1 << 18), (0, 1 << 19), (0, 1 << 17)):
    fold = dspsr_amd.FoldEngine(ctx)
    fold.set_shape(nchan, npol, 1, nbin)
    buf = torch.rand((nchan, npol, ndat + pad), dtype=torch.float32, device="cuda")
    det = buf[:, :, :ndat]
    hits = np.zeros(nbin, np.uint32)
    def step():
        fold.set_nbin(nbin); fold.set_ndat(ndat, 0)
        fold.set_bins(0.25, (1.0 / rate) * polyco.frequency(day, sec0), ndat, 0, hits)
        fold.fold(det)
    for _ in range(3): step()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:
        fold.set_nbin(nbin); fold.set_ndat(ndat, 0)
        fold.set_bins(0.25, (1.0 / rate) * polyco.frequency(day, sec0), ndat, 0, hits)
        a.record(); fold.fold(det); b.record()
    torch.cuda.synchronize()
    ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
    print("ndat 2^%d pad %5d floats: %.4f ms per block, %.2f TB/s" % (int(np.log2(ndat)), pad, ms, nchan * npol * ndat * 4 / ms / 1e9), flush=True)
    fold.close()
    del buf, det
ctx.close()
