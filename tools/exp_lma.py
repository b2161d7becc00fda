"""tools/exp_lma.py : experiment (needs an -DDSPSR_AMD_EXPERIMENT build selected with DSPSR_AMD_LIB): time
perform_detect of four-pass geometries for every split freq_res = 2^lma * 2^lmb (DSPSR_AMD_LMA) and both spectrum layouts
(DSPSR_AMD_X_NATURAL).  Prints ms per call (8-bit real dual-pol input resident, N = C*freq_res = 2^logN)."""
import os
import sys

import numpy as np
import torch

import dspsr_amd
from dspsr_amd import _lib

logN = int(sys.argv[1]) if len(sys.argv) > 1 else 24
logms = [int(a) for a in sys.argv[2:]] or [15, 16, 17, 18, 19, 20, 21, 22]
ctx = dspsr_amd.Context(0, torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(1)
for logm in logms:
    M, C = 1 << logm, 1 << (logN - logm)
    pos = M // 32
    npart = max(1, (1 << 27) >> logN)
    kern = np.exp(2j * np.pi * rng.random(C * M)).astype(np.complex64)
    res = {}
    for nat in (0, 1):
        for lma in range(max(4, logm - 13), min(13, logm - 4) + 1):
            os.environ["DSPSR_AMD_LMA"] = str(lma)
            if nat:
                os.environ["DSPSR_AMD_X_NATURAL"] = "1"
            else:
                os.environ.pop("DSPSR_AMD_X_NATURAL", None)
            try:
                fb = dspsr_amd.FilterbankEngine(ctx).setup(C, M, pos, pos, 1, 2, True, kern, max_parts=npart)
            except dspsr_amd.DspsrAmdError as e:
                continue
            raw = torch.randint(-100, 100, (fb._raw_bytes(npart),), dtype=torch.int8, device="cuda")
            det = torch.empty((C, 1, npart * fb.nkeep * 4), dtype=torch.float32, device="cuda")
            for _ in range(2):
                fb.perform_detect(det, npart, raw=raw, layout=_lib.RAW_GENERIC, scale=1.0)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(4):
                fb.perform_detect(det, npart, raw=raw, layout=_lib.RAW_GENERIC, scale=1.0)
            b.record()
            torch.cuda.synchronize()
            res[(nat, lma)] = a.elapsed_time(b) / 4
            fb.close()
            del raw, det
    if not res:
        print('logM %2d C %5d: not tileable' % (logm, C), flush=True)
        continue
    best = min(res, key=res.get)
    print("logM %2d C %5d parts %d | " % (logm, C, npart) +
          "  ".join("%s%d:%.2f" % ("n" if k[0] else "b", k[1], v) for k, v in sorted(res.items())) +
          " | best %s%d" % ("n" if best[0] else "b", best[1]), flush=True)
ctx.close()
