"""Non-convolving filterbank (freq_res = 1, csrc/fb_plain.hip) by itself: time per launch and bytes moved for a few channel counts,
input layouts and output forms (complex rows, detected rows, none).  HIP events on the launch stream.
  python tools/plain_probe.py [nsamp_log2]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dspsr_amd  # noqa: E402


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 29
    nsamp = 1 << lg                                       # real samples per polarisation
    ctx = dspsr_amd.Context(0, torch.cuda.current_stream().cuda_stream)
    raw = torch.randint(-100, 100, (2 * nsamp,), dtype=torch.int8, device="cuda")
    print("build", dspsr_amd.build_id(), "samples/pol 2^%d" % lg)
    for C in (16, 32, 128, 512, 1024, 2048, 8192):
        npart = nsamp // (2 * C)
        out = torch.empty((C, 2, 2 * npart), dtype=torch.float32, device="cuda")
        det = torch.empty((C, 1, 4 * npart), dtype=torch.float32, device="cuda")
        for layout, lname in ((dspsr_amd.RAW_GENERIC, "generic"), (dspsr_amd.RAW_CASPSR, "caspsr")):
            eng = dspsr_amd.FilterbankEngine(ctx).setup(C, 1, 0, 0, 1, 2, True, None)
            for what in ("rows", "detected", "none"):
                def run():
                    if what == "rows":
                        eng.perform_raw(raw, layout, 1.0, out, npart)
                    elif what == "none":
                        eng.perform_raw(raw, layout, 1.0, None, npart)
                    else:
                        eng.perform_detect(det, npart, dspsr_amd.COHERENCE, 4, raw=raw, layout=layout, scale=1.0)
                run()
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(5):
                    run()
                b.record()
                torch.cuda.synchronize()
                ms = a.elapsed_time(b) / 5
                nbytes = 2 * nsamp + (0 if what == "none" else 16 * C * npart)
                print("C=%5d %-8s %-9s %8.3f ms  %7.1f GB/s  %9.0f Msamples/s" % (C, lname, what, ms, nbytes / ms / 1e6, nsamp / ms / 1e3))
            eng.close()
        del out, det
    ctx.close()


if __name__ == "__main__":
    main()
