"""Diagnostic (FB_STAMPS=4 build): per-phase cycle shares of k_inv_a at cfg1opt (four-pass geometry, real 8-bit input).
usage: DSPSR_AMD_LIB=build/lib_st4.so python tools/stamps_cfg1.py"""
import argparse, ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
lib = C.CDLL(os.environ["DSPSR_AMD_LIB"])
args = argparse.Namespace(parts_per_block=0, max_parts=0, ndim=4, no_fused_fold=os.environ.get("FUSED", "1") != "1", dump_steps=8)
lt, raw, cfg, info, wl, sharded = bench.make_fold_pipeline(os.environ.get("WL", "cfg1opt"), args, torch, 0, 1, 0)
for _ in range(2):
    lt.process_block(raw)
torch.cuda.synchronize()
lib.dspsr_amd_debug_stamps(None, 1)
for _ in range(4):
    lt.process_block(raw)
torch.cuda.synchronize()
buf = np.zeros((1024, 8), np.uint64)
lib.dspsr_amd_debug_stamps(buf.ctypes.data_as(C.c_void_p), 0)
b = buf[buf[:, 5] > 0].astype(np.float64)
tiles = b[:, 5]
names = [(0, "wait prefetched tile"), (1, "chirp load, split, multiply"), (2, "issue next prefetch"), (3, "memory -> butterfly order exchange"),
         (4, "transform + twiddle + staging"), (6, "copy-out stores")]
tot = sum(b[:, q] for q, _ in names) / tiles
print("k_inv_a: workgroups %d, tiles per workgroup %.1f, cycles per tile %.0f" % (len(b), tiles.mean(), tot.mean()))
for q, nm in names:
    print("  %-36s %8.0f cycles  %5.1f %%" % (nm, (b[:, q] / tiles).mean(), 100 * (b[:, q] / tiles).mean() / tot.mean()))
