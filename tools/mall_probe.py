"""Does a working set that fits the 256 MB Infinity Cache stream faster than one that does not?  torch elementwise
add (read x, write y) and in-place scale (read + write x) over buffers of growing size, repeated back to back."""
import torch
def t(f, reps):
    f(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3
for mb in (16, 32, 48, 64, 96, 128, 192, 256, 512, 1024, 4096):
    n = mb * (1 << 20) // 4
    x = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    y = torch.empty_like(x)
    reps = max(10, 20000 // mb)
    ta = t(lambda: torch.add(x, 1.0, out=y), reps)
    ti = t(lambda: x.mul_(1.0001), reps)
    tr = t(lambda: torch.ge(x, 100.0, out=y.view(torch.bool)[:n]) if False else x.max(), reps)
    print("%5d MB per buffer: add(x->y) %.2f TB/s   mul_(x) %.2f TB/s   max(x) read-only %.2f TB/s" %
          (mb, 2 * 4 * n / ta / 1e12, 2 * 4 * n / ti / 1e12, 4 * n / tr / 1e12))
    del x, y
