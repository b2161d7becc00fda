"""Streaming ceiling of the device as torch's own kernels see it: fill (write), sum (read), copy (read + write)."""
import torch, time
n = 1 << 30            # 4 GiB of float32
x = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
y = torch.empty_like(x)
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3
tc = t(lambda: y.copy_(x)); print("copy  4 GiB: %.3f ms  %.2f TB/s (read + write)" % (tc * 1e3, 2 * 4 * n / tc / 1e12))
tf = t(lambda: y.fill_(1.0)); print("fill  4 GiB: %.3f ms  %.2f TB/s (write)" % (tf * 1e3, 4 * n / tf / 1e12))
ts = t(lambda: x.sum()); print("sum   4 GiB: %.3f ms  %.2f TB/s (read)" % (ts * 1e3, 4 * n / ts / 1e12))
ta = t(lambda: torch.add(x, 1.0, out=y)); print("add   4 GiB: %.3f ms  %.2f TB/s (read + write)" % (ta * 1e3, 2 * 4 * n / ta / 1e12))
