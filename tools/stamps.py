"""Diagnostic (FB_STAMPS build): per-phase cycle shares of the pass-3 kernel.  usage: DSPSR_AMD_LIB=build/lib_st.so python tools/stamps.py"""
import ctypes as C, os, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dspsr_amd
from dspsr_amd import pipeline
import bench
lib = C.CDLL(os.environ["DSPSR_AMD_LIB"])
wl = bench.WORKLOADS[os.environ.get("WL", "target")]           # WL=cfg2|cfg3|target (three-pass, single-channel input)
info = pipeline.InputInfo(centre_frequency=wl["freq"], bandwidth=wl["bw"], nchan=1, npol=2, ndim=1, tsamp_us=wl["tsamp_us"], machine="CASPSR")
cfg = pipeline.Config(nchan=wl["nchan"], dispersion_measure=wl["dm"], nbin=wl["nbin"], folding_period=0.0893, freq_res=wl["freq_res"],
                      parts_per_block=int(os.environ.get("PPB", "16")), max_parts=int(os.environ.get("MAXP", "8")),
                      fused_fold=os.environ.get("FUSED", "1") == "1")
lt = pipeline.LoadToFold(cfg, info, device=0, stream=torch.cuda.current_stream().cuda_stream)
raw = (torch.randn(lt.block_bytes(), device="cuda") * 24).round().clamp(-128, 127).to(torch.int8)
for _ in range(2):
    lt.process_block(raw)
torch.cuda.synchronize()
lib.dspsr_amd_debug_stamps(None, 1)
n = 6
for _ in range(n):
    lt.process_block(raw)
torch.cuda.synchronize()
buf = np.zeros((1024, 8), np.uint64)
lib.dspsr_amd_debug_stamps(buf.ctypes.data_as(C.c_void_p), 0)
b = buf[buf[:, 5] > 0].astype(np.float64)
tiles = b[:, 5]
names = {"3": ["wait prefetched tile", "chirp load+convert", "issue next prefetch", "transform+staging", "fold phase"],
         "1": ["wait prefetched tile", "decode", "issue next prefetch", "transform+twiddle+staging", "copy-out stores"],
         "2": ["wait prefetched tile", "regroup", "issue next prefetch", "transform+staging", "copy-out stores"]}[os.environ.get("PASS", "3")]
tot = b[:, :5].sum(axis=1) / tiles
print("workgroups %d, tiles per workgroup %.1f, cycles per tile %.0f (100 MHz ticks? no: shader cycles)" % (len(b), tiles.mean(), tot.mean()))
for q, nm in enumerate(names):
    print("  %-22s %8.0f cycles  %5.1f %%" % (nm, (b[:, q] / tiles).mean(), 100 * (b[:, q] / tiles).mean() / tot.mean()))
