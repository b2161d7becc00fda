"""dsp::Convolution on many channels (nchan_subband = 1, complex float32 rows) by itself: one tile pass (csrc/fb_conv1.hip, n_fft <= 8192)
and three (csrc/fb_conv3.hip, 2^14 ... 2^21) against the four tile passes (grouped channels), per response length.  HIP events on the launch stream.
  python tools/conv_probe.py [log2 points per polarisation] [n_fft,n_fft,...] [rows,detected,none]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dspsr_amd  # noqa: E402


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    ctx = dspsr_amd.Context(0, torch.cuda.current_stream().cuda_stream)
    nchan = 128
    print("build", dspsr_amd.build_id(), "complex samples per polarisation 2^%d in %d channels" % (lg, nchan))
    for M in [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else (256, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1048576, 2097152):
        nfilt = (M // 10, M // 10)
        step = M - sum(nfilt)
        npart = (1 << lg) // nchan // step
        n = npart * step + sum(nfilt)
        x = torch.randn((nchan, 2, 2 * n), dtype=torch.float32, device="cuda")
        rng = np.random.default_rng(1)
        kernel = np.exp(1j * rng.uniform(-np.pi, np.pi, nchan * M)).astype(np.complex64)
        out = torch.empty((nchan, 2, 2 * npart * step), dtype=torch.float32, device="cuda")
        det = torch.empty((nchan, 1, 4 * npart * step), dtype=torch.float32, device="cuda")
        for ffp, name in ((0, "one pass" if M <= 8192 else "three" if M <= 2097152 else "auto"), (1, "four passes")):
            eng = dspsr_amd.FilterbankEngine(ctx).setup(1, M, nfilt[0], nfilt[1], nchan, 2, False, kernel, max_parts=min(npart, 256),
                                                        force_four_pass=ffp)
            for what in (sys.argv[3].split(",") if len(sys.argv) > 3 else ("rows", "detected", "none")):
                def run():
                    if what == "rows":
                        eng.perform(x, out, npart, 2 * step, 2 * step)
                    elif what == "none":
                        eng.perform(x, None, npart, 2 * step, 2 * step)
                    else:
                        eng.perform_detect(det, npart, dspsr_amd.COHERENCE, 4, inp=x, in_step=2 * step)
                run()
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(3):
                    run()
                b.record()
                torch.cuda.synchronize()
                ms = a.elapsed_time(b) / 3
                kept = nchan * npart * step
                print("M=%5d %-11s %-9s %8.3f ms  %8.1f M kept samples/s per pol  (%d parts)" % (M, name, what, ms, kept / ms / 1e3, npart))
            eng.close()
        del x, out, det
    ctx.close()


if __name__ == "__main__":
    main()
