"""tools/boundary_bench.py : the headline geometry through the LITERAL plug-in boundary, i.e. what DSPSR reaches with the
three Engine adaptors and nothing else: Filterbank::Engine::perform on already-unpacked float32 input (complex output),
Detection::Engine::polarimetry (ndim 2, in place, as LoadToFold1.C:1105-1109 configures the GPU path), Fold::Engine::fold.
Against it: perform_detect on floats (the fused detection side channel), and the raw 8-bit side channels bench.py times.
Prints Msamples/s for each (block resident in HBM)."""
import sys

import numpy as np
import torch

import dspsr_amd
from dspsr_amd import _lib, pipeline

npart = int(sys.argv[1]) if len(sys.argv) > 1 else 32
info = pipeline.InputInfo(machine="CASPSR")
cfg = pipeline.Config(nchan=1024, dispersion_measure=1000.0, nbin=1024, folding_period=0.0893, freq_res=4096, parts_per_block=npart,
                      max_parts=npart)
lt = pipeline.LoadToFold(cfg, info, device=0, stream=torch.cuda.current_stream().cuda_stream)
fb, fold, ctx = lt.fb, lt.fold, lt.ctx
nkeep, step, ovl = lt.nkeep, lt.nsamp_step, lt.nsamp_overlap
ndat = npart * step + ovl
x = torch.randn((1, 2, ndat), dtype=torch.float32, device="cuda")
out = torch.empty((1024, 2, 2 * npart * nkeep), dtype=torch.float32, device="cuda")
det = dspsr_amd.DetectionEngine(ctx)
f2 = dspsr_amd.FoldEngine(ctx)
f2.set_shape(1024, 2, 2, 1024)
hits = np.zeros(1024, np.uint32)


def plan(eng):
    eng.set_nbin(1024)
    eng.set_ndat(npart * nkeep, 0)
    eng.set_bins(0.1, (1.0 / lt.out_rate) / 0.0893, npart * nkeep, 0, hits)


def boundary():
    fb.perform(x, out, npart, step, 2 * nkeep)
    det.polarimetry(2, out, out)
    plan(f2)
    f2.fold(out)


d4 = torch.empty((1024, 1, 4 * npart * nkeep), dtype=torch.float32, device="cuda")
fold.set_shape(1024, 1, 4, 1024)


def fused_detect_float():
    fb.perform_detect(d4, npart, _lib.COHERENCE, 4, inp=x, in_step=step)
    plan(fold)
    fold.fold(d4)


raw = torch.randint(-100, 100, (lt.block_bytes(npart),), dtype=torch.int8, device="cuda")


def raw_fused_fold():
    plan(fold)
    fb.perform_fold(fold, npart, _lib.COHERENCE, raw=raw, layout=lt.layout, scale=lt.scale8)


for name, fn in (("Engine boundary: perform(float32) + polarimetry(ndim 2, in place) + fold", boundary),
                 ("float32 input, detection fused (perform_detect) + fold", fused_detect_float),
                 ("8-bit input, everything fused (perform_fold, bench.py's timed path)", raw_fused_fold)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    n = 10
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / n
    print("%-80s %8.3f ms per %d parts  %9.0f Msamples/s" % (name, ms, npart, npart * step / ms / 1e3), flush=True)
lt.close()
