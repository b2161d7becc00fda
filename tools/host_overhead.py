"""Host time per LoadToFold.process_block call (enqueue only) against the GPU time of the block.
usage: python tools/host_overhead.py [parts_per_block] [max_parts] [workload]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dspsr_amd import pipeline
import bench
ppb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
mp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
wl = bench.WORKLOADS[sys.argv[3] if len(sys.argv) > 3 else "target"]
info = pipeline.InputInfo(centre_frequency=wl["freq"], bandwidth=wl["bw"], nchan=wl["in_nchan"], npol=2, ndim=wl["ndim"], tsamp_us=wl["tsamp_us"],
                          machine=wl["machine"])
cfg = pipeline.Config(nchan=wl["nchan"], dispersion_measure=wl["dm"], nbin=wl["nbin"], folding_period=0.0893, freq_res=wl["freq_res"],
                      parts_per_block=ppb, max_parts=mp)
lt = pipeline.LoadToFold(cfg, info, device=0, stream=torch.cuda.current_stream().cuda_stream, subband=0 if wl["in_nchan"] > 1 else None)
raw = (torch.randn(lt.block_bytes(), device="cuda") * 24).round().clamp(-128, 127).to(torch.int8)
for _ in range(3):
    lt.process_block(raw)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
host = []
for _ in range(n):
    a = time.perf_counter()
    lt.process_block(raw)
    host.append(time.perf_counter() - a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("parts/block %d: host enqueue %.3f ms per block (min %.3f), total %.3f ms per block, %.1f us per part" %
      (ppb, 1e3 * sum(host) / n, 1e3 * min(host), 1e3 * (t2 - t0) / n, 1e6 * (t2 - t0) / n / ppb))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    lt.process_block(raw)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
