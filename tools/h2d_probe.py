"""H2D rate of one pinned block (the size bench.py's PCIe-inclusive leg copies) as 1, 2 or 4 concurrent hipMemcpyAsync pieces on
separate streams -- does a second SDMA engine help?   python tools/h2d_probe.py [MB]"""
import sys, time
import torch
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 856
n = mb << 20
host = torch.empty(n, dtype=torch.int8).pin_memory()
dev = torch.empty(n, dtype=torch.int8, device="cuda")
for k in (1, 2, 4, 8):
    streams = [torch.cuda.Stream() for _ in range(k)]
    piece = n // k
    def run():
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                dev[i * piece:(i + 1) * piece].copy_(host[i * piece:(i + 1) * piece], non_blocking=True)
    run(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); run(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print("%d piece(s): %.2f ms  %.1f GB/s" % (k, best * 1e3, n / best / 1e9), flush=True)
