"""Diagnostic (FB_STAMPS=6|7 build): per-phase cycle shares of the two-pass kernels at the cfg4 shard.
usage: DSPSR_AMD_LIB=build/lib_st7.so PASS=7 python tools/stamps_cfg4.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dspsr_amd
from dspsr_amd import pipeline
lib = C.CDLL(os.environ["DSPSR_AMD_LIB"])
info = pipeline.InputInfo(centre_frequency=1382.0, bandwidth=-400.0, nchan=8, npol=2, ndim=2, tsamp_us=0.02, machine="DADA")
cfg = pipeline.Config(nchan=4096, dispersion_measure=1000.0, nbin=1024, folding_period=0.0893, freq_res=512, parts_per_block=256, max_parts=256,
                      fused_fold=os.environ.get("FUSED", "1") == "1")
lt = pipeline.LoadToFold(cfg, info, device=0, stream=torch.cuda.current_stream().cuda_stream, subband=3)
raw = (torch.randn(lt.block_bytes(), device="cuda") * 24).round().clamp(-128, 127).to(torch.int8)
for _ in range(2):
    lt.process_block(raw)
torch.cuda.synchronize()
lib.dspsr_amd_debug_stamps(None, 1)
for _ in range(6):
    lt.process_block(raw)
torch.cuda.synchronize()
buf = np.zeros((1024, 8), np.uint64)
lib.dspsr_amd_debug_stamps(buf.ctypes.data_as(C.c_void_p), 0)
b = buf[buf[:, 5] > 0].astype(np.float64)
tiles = b[:, 5]
p = os.environ.get("PASS", "7")
names = {"6": [(0, "wait prefetched tile"), (1, "decode"), (2, "issue next prefetch"), (3, "transform + radix-2 step + staging"), (4, "copy-out stores")],
         "7": [(0, "wait prefetched tile"), (1, "chirp, twiddle, rows FFT"), (2, "issue next prefetch"), (3, "rows -> bins exchange"),
               (4, "inverse transform (+ staging)"), (6, "fold phase")]}[p]
tot = sum(b[:, q] for q, _ in names) / tiles
print("pass %s: workgroups %d, tiles per workgroup %.1f, cycles per tile %.0f" % (p, len(b), tiles.mean(), tot.mean()))
for q, nm in names:
    print("  %-36s %8.0f cycles  %5.1f %%" % (nm, (b[:, q] / tiles).mean(), 100 * (b[:, q] / tiles).mean() / tot.mean()))
