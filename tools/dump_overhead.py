"""Host time of LoadToFold.finish_subint (one rank) and of its pieces.  usage: python tools/dump_overhead.py [workload]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dspsr_amd import pipeline
import bench
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg4"]
info = pipeline.InputInfo(centre_frequency=wl["freq"], bandwidth=wl["bw"], nchan=wl["in_nchan"], npol=2, ndim=wl["ndim"], tsamp_us=wl["tsamp_us"],
                          machine=wl["machine"])
cfg = pipeline.Config(nchan=wl["nchan"], dispersion_measure=wl["dm"], nbin=wl["nbin"], folding_period=0.0893, freq_res=wl["freq_res"],
                      parts_per_block=64, max_parts=64)
lt = pipeline.LoadToFold(cfg, info, device=0, stream=torch.cuda.current_stream().cuda_stream, subband=0 if wl["in_nchan"] > 1 else None)
raw = (torch.randn(lt.block_bytes(), device="cuda") * 24).round().clamp(-128, 127).to(torch.int8)
for _ in range(3):
    lt.process_block(raw)
    lt.finish_subint(None, 0, 1, None, replicas=wl["in_nchan"] == 1, check_hits=False)
    lt.subints.clear()
torch.cuda.synchronize()
def t(f, n=20):
    torch.cuda.synchronize()
    a = time.perf_counter()
    for _ in range(n):
        f()
    b = time.perf_counter()
    torch.cuda.synchronize()
    c = time.perf_counter()
    return 1e6 * (b - a) / n, 1e6 * (c - a) / n
def dump():
    lt.finish_subint(None, 0, 1, None, replicas=wl["in_nchan"] == 1, check_hits=False)
    lt.subints.clear()
print("finish_subint         host %.1f us, with sync %.1f us" % t(dump))
print("profiles_tensor       host %.1f us, with sync %.1f us" % t(lambda: lt.profiles_tensor()))
p = lt.profiles_tensor()
print("clone                 host %.1f us, with sync %.1f us" % t(lambda: p.clone()))
print("fold.zero             host %.1f us, with sync %.1f us" % t(lambda: lt.fold.zero()))
