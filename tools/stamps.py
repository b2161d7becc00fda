"""Per-phase cycle shares of one persistent kernel (diagnostic library built with -DFB_STAMPS=<id>, csrc/stamps.h).

    make -C dspsr_amd/csrc OBJDIR=../../build/obj_st8 OUT=../../build/lib_st8.so BBENCH= EXTRA=-DFB_STAMPS=8
    DSPSR_AMD_LIB=build/lib_st8.so python tools/stamps.py 8 [workload]          (on the GPU box)

ids / default workloads: 1 k_fwd_cols, 2 k_fwd_rows, 3 k_inv_chan fused (target | cfg2 | cfg3), 4 k_inv_a (cfg1opt | cfg1),
6 k_fwd_col1q, 7 k_rows_inv (cfg4), 8 k_tfp (cfg5).  FUSED=0: the launch group that writes its detected output.
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from dspsr_amd import pipeline  # noqa: E402

KERNELS = {
    1: ("fwd_cols", "target", ["wait prefetched tile", "decode", "issue next prefetch", "transform + staging", "copy-out stores"]),
    2: ("fwd_rows", "target", ["wait prefetched tile", "regroup + pass twiddle", "issue next prefetch", "transform + staging", "copy-out stores"]),
    3: ("inv_chan_fold", "target", ["wait prefetched tile", "split + chirp", "issue next prefetch", "transform + staging", "fold phase"]),
    4: ("four_pass", "cfg1opt", ["wait prefetched tile", "chirp + split", "issue next prefetch", "order exchange", "transform + staging", "copy-out stores"]),
    6: ("two_pass", "cfg4", ["wait prefetched tile", "decode", "issue next prefetch", "transform + radix-4 step + staging", "copy-out stores"]),
    7: ("two_pass", "cfg4", ["wait prefetched tile", "chirp, twiddle, rows FFT", "issue next prefetch", "rows -> bins exchange",
                             "inverse transform (+ staging)", "fold phase / stores"]),
    8: ("tfp", "cfg5", ["wait prefetched tile", "image through LDS + decode", "issue next prefetch", "transform + staging",
                        "split, powers, tscrunch, stores"]),
}


def main():
    kid = int(sys.argv[1])
    unit, wl_name, names = KERNELS[kid]
    if len(sys.argv) > 2:
        wl_name = sys.argv[2]
    if kid == 3 and os.environ.get("FUSED", "1") != "1":
        unit = "inv_chan"
    lib = C.CDLL(os.environ["DSPSR_AMD_LIB"])
    reader = getattr(lib, "dspsr_amd_debug_stamps_" + unit)
    wl = bench.WORKLOADS[wl_name]
    if wl_name == "cfg5":
        info = pipeline.InputInfo(centre_frequency=wl["freq"], bandwidth=wl["bw"], nchan=1, npol=2, ndim=1, tsamp_us=wl["tsamp_us"], machine=wl["machine"])
        p = pipeline.LoadToFil(pipeline.SearchConfig(nchan=wl["nchan"], tscrunch=wl["tscrunch"], nbit=8, parts_per_block=wl["nparts"]), info, device=0,
                               stream=torch.cuda.current_stream().cuda_stream)
    else:
        args = argparse.Namespace(parts_per_block=0, max_parts=0, ndim=4, no_fused_fold=os.environ.get("FUSED", "1") != "1", dump_steps=8)
        p, raw = bench.make_fold_pipeline(wl_name, args, torch, 0, 1, 0)[:2]
    if wl_name == "cfg5":
        raw = torch.randn(p.block_bytes(), generator=torch.Generator(device="cuda").manual_seed(20100413), device="cuda").mul_(24.0).round_().clamp_(-128, 127).to(torch.int8)
    for _ in range(2):
        p.process_block(raw)
    torch.cuda.synchronize()
    reader(None, 1)
    for _ in range(6):
        p.process_block(raw)
    torch.cuda.synchronize()
    buf = np.zeros((1024, 8), np.uint64)
    reader(buf.ctypes.data_as(C.c_void_p), 0)
    b = buf[buf[:, 7] > 0].astype(np.float64)
    if not len(b):
        sys.exit("no stamps recorded: was the library built with -DFB_STAMPS=%d?" % kid)
    tiles = b[:, 7]
    per = b[:, :len(names)] / tiles[:, None]
    tot = per.sum(axis=1).mean()
    print("kernel id %d (%s), workload %s: workgroups %d, tiles per workgroup %.1f, cycles per tile %.0f" % (kid, unit, wl_name, len(b), tiles.mean(), tot))
    for q, nm in enumerate(names):
        print("  %-36s %8.0f cycles  %5.1f %%" % (nm, per[:, q].mean(), 100 * per[:, q].mean() / tot))


if __name__ == "__main__":
    main()
