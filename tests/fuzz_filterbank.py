"""tests/fuzz_filterbank.py [ncases] [seed] [log2 N min] [log2 N max] : random filterbank geometries against the float64 oracle (test infrastructure:
uses tests/test_gpu_parity._fb_case).  Every combination the C-ABI accepts is fair game: 1 or 2 polarisations, real or complex
input, 8-bit or float32 input, generic or CASPSR byte order, 1-3 input channels, three- or four-pass, 1-3 parts per launch."""
import os
import sys
import traceback

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib.util

spec = importlib.util.spec_from_file_location("oracle_mod", os.path.join(ROOT, "oracle", "dspsr_oracle.py"))
oracle = importlib.util.module_from_spec(spec)
sys.modules["oracle_mod"] = oracle
spec.loader.exec_module(oracle)
import dspsr_amd
from test_gpu_parity import _fb_case

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (7, 21)     # log2 of N = nchan_subband * freq_res
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = dspsr_amd.Context(0, torch.cuda.current_stream().cuda_stream)
bad = 0
for i in range(ncases):
    logN = int(rng.integers(lo, hi + 1))
    logM = int(rng.integers(2, min(logN, 22) + 1))
    C, M = 1 << (logN - logM), 1 << logM
    real = bool(rng.integers(0, 4) != 0)
    npol = 2 if not real or rng.integers(0, 5) else 1
    pos = int(rng.integers(0, max(1, M // 3)))
    neg = int(rng.integers(0, max(1, M // 3)))
    if pos + neg >= M:
        pos, neg = 0, 0
    kw = dict(npol=npol, real=real, use_raw=bool(rng.integers(0, 3) != 0), max_parts=int(rng.integers(1, 4)),
              four_pass=bool(rng.integers(0, 3) == 0), seed=int(rng.integers(1, 1000)))
    if real and npol == 2 and kw["use_raw"] and rng.integers(0, 2):      # (whole 4-sample groups: _fb_case pads the stream)
        kw["layout"] = "caspsr"
    if rng.integers(0, 4) == 0 and logN <= 18 and "layout" not in kw:      # (the CASPSR byte order is single channel)
        kw["input_nchan"] = int(rng.integers(2, 4))
    npart = int(rng.integers(1, 4)) if logN <= 19 else int(rng.integers(1, 3))
    if rng.integers(0, 6) == 0 and logN - logM >= 3 and logM <= 13:
        # nchan_subband = 3 * 2^k / 5 * 2^k: interleaved sub-sequences + one radix-3 / radix-5 step (three-pass geometries)
        R = int(rng.choice([3, 5, 7, 9, 15, 11, 13, 21, 25, 27, 33, 45, 63, 75, 99, 125, 127]))
        sh = int(np.ceil(np.log2(R)))
        C = R * (C >> sh) if (C >> sh) >= 1 else C
        kw["four_pass"] = 0
    elif rng.integers(0, 6) == 0 and 3 <= logM <= 15 and logN - logM <= 12:
        # freq_res = 3 * 2^k / 5 * 2^k: pseudo-channels of freq_res / R bins + one radix-R step in time (k_time_combine)
        R = int(rng.choice([3, 5, 7, 9, 15, 11, 13, 21, 25, 27, 33, 45, 63, 75, 99, 125, 127]))
        sh = int(np.ceil(np.log2(R)))
        M = R * (M >> sh) if (M >> sh) >= 2 else M
        pos, neg = int(rng.integers(0, max(1, M // 3))), int(rng.integers(0, max(1, M // 3)))
        kw["four_pass"] = 0
        kw["max_parts"] = int(rng.integers(1, 8))                          # (sub-groups of parts q, q + rp, ...)
        npart = int(rng.integers(1, 8)) if logN <= 17 else npart
    if rng.integers(0, 6) == 0:
        # the two-pass family of short responses (complex dual-pol, nchan_subband * freq_res^2 == 2^27): 8-bit blocks take the
        # two-pass kernels (or, four_pass = 2, the three-pass ones), float32 rows always the three-pass ones
        logM = int(rng.integers(9, 13))
        logC = int(rng.integers(13 - logM, 27 - 2 * logM + 1))            # Fb <= C <= 2^27 / M^2
        C, M = 1 << logC, 1 << logM
        pos, neg = int(rng.integers(0, M // 3)), int(rng.integers(0, M // 3))
        kw = dict(npol=2, real=False, use_raw=bool(rng.integers(0, 4) != 0), max_parts=int(rng.integers(1, 4)),
                  four_pass=int(rng.choice([0, 0, 2, 1])), seed=int(rng.integers(1, 1000)))
        if rng.integers(0, 4) == 0 and logM <= 10 and logC + logM <= 17:
            kw["input_nchan"] = 2
        npart = int(rng.integers(1, 5))
    fam = rng.integers(0, 8)
    if fam == 0:
        # the non-convolving filterbank (freq_res = 1, csrc/fb_plain.hip): any power-of-two channel count, every input form, part
        # counts from a fraction of a tile to several tiles per workgroup
        C, M, pos, neg = 1 << int(rng.integers(1, 14)), 1, 0, 0
        real = bool(rng.integers(0, 3) != 0)
        kw = dict(npol=2 if rng.integers(0, 4) else 1, real=real, use_raw=bool(rng.integers(0, 3) != 0), seed=int(rng.integers(1, 1000)))
        if real and kw["npol"] == 2 and kw["use_raw"] and rng.integers(0, 3) == 0:
            kw["layout"] = "caspsr"
        elif rng.integers(0, 3) == 0:
            kw["input_nchan"] = int(rng.integers(2, 6))
        npart = int(rng.integers(1, max(2, min(6000, (1 << 19) // (C * kw.get("input_nchan", 1))))))
    elif fam == 1:
        # dsp::Convolution on many channels (nchan_subband = 1, complex float rows): groups of channels as one launch group
        logM = int(rng.integers(4, 23))          # (one tile pass up to 2^13, three for 2^14 ... 2^21: csrc/fb_conv1.hip, fb_conv3.hip)
        C, M = 1, 1 << logM
        pos, neg = int(rng.integers(0, max(1, M // 3))), int(rng.integers(0, max(1, M // 3)))
        kw = dict(npol=2, real=False, use_raw=False, max_parts=int(rng.integers(1, 5)), seed=int(rng.integers(1, 1000)),
                  input_nchan=int(rng.choice([1, 3, 4, 6, 8, 12, 16, 20, 32, 64] if logM <= 14 else [1, 2, 3, 4, 6, 8] if logM <= 18 else [1, 2, 3])), four_pass=int(rng.choice([0, 0, 0, 2])))
        npart = int(rng.integers(1, 7)) if logM + int(np.log2(kw["input_nchan"])) <= 17 else int(rng.integers(1, 3))
    desc = "C=%d M=%d nfilt=(%d,%d) npart=%d %s" % (C, M, pos, neg, npart, kw)
    try:
        _fb_case(oracle, (dspsr_amd, ctx), C, M, (pos, neg), npart, **kw)
        print("ok   ", desc, flush=True)
    except dspsr_amd.DspsrAmdError as e:
        print("refused", desc, "--", str(e)[:100], flush=True)
    except AssertionError as e:
        bad += 1
        print("FAIL ", desc, "--", str(e)[:200], flush=True)
    except Exception:
        bad += 1
        print("ERROR", desc, flush=True)
        traceback.print_exc()
ctx.close()
print("%d cases, %d failures" % (ncases, bad))
sys.exit(1 if bad else 0)
