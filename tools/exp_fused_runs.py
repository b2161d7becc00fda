"""tools/exp_fused_runs.py (experiment build, DSPSR_AMD_LIB): headline geometry folded into few, wide phase bins -- the fused
kernel's exact time-order chains against Detection + long-run Fold as separate launches.  ms per block of 64 parts."""
import os
import sys
import torch
from dspsr_amd import pipeline

info = pipeline.InputInfo(machine="CASPSR")
for nbin in (1024, 512, 256, 128, 64, 32):
    res = {}
    for mode, thr in (("unfused", 0), ("fused", 1 << 20)):
        os.environ["DSPSR_AMD_FUSED_MAX_RUN"] = str(thr) if thr else "1"
        cfg = pipeline.Config(nchan=1024, dispersion_measure=1000.0, nbin=nbin, folding_period=0.0893, freq_res=4096,
                              parts_per_block=64, max_parts=32, fused_fold=True)
        lt = pipeline.LoadToFold(cfg, info, device=0, stream=torch.cuda.current_stream().cuda_stream)
        raw = torch.randint(-100, 100, (lt.block_bytes(),), dtype=torch.int8, device="cuda")
        for _ in range(3):
            lt.process_block(raw)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(8):
            lt.process_block(raw)
        b.record()
        torch.cuda.synchronize()
        res[mode] = a.elapsed_time(b) / 8
        lt.close()
    print("nbin %5d  run ~%4d samples   unfused %.3f ms   fused %.3f ms" % (nbin, 34883 // nbin, res["unfused"], res["fused"]), flush=True)
