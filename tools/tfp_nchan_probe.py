import sys, time, torch
sys.path.insert(0, '.')
import dspsr_amd
from dspsr_amd import pipeline
for nchan in (512, 1024, 2048, 4096, 8192):
    nparts = 32768 * 4096 // nchan
    info = pipeline.InputInfo(centre_frequency=1382.0, bandwidth=-400.0, nchan=1, npol=2, ndim=1, tsamp_us=0.00125, machine="CASPSR")
    lf = pipeline.LoadToFil(pipeline.SearchConfig(nchan=nchan, tscrunch=16, nbit=8, parts_per_block=nparts), info, device=0,
                            stream=torch.cuda.current_stream().cuda_stream)
    nbytes = lf.block_bytes()
    raw = torch.randn(nbytes, device="cuda").mul_(24.0).round_().clamp_(-128, 127).to(torch.int8)
    for _ in range(3): lf.process_block(raw)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(10): lf.process_block(raw)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 10
    print("nchan %5d  %.3f ms/block  %.0f Msamples/s" % (nchan, dt * 1e3, nparts * 2 * nchan / dt / 1e6), flush=True)
    lf.close()
