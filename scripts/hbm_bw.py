"""practical HBM bandwidth on the box: reduction (read only) and copy (read + write) of buffers of several sizes"""
import torch
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e-3
for mb in (103, 205, 411, 1024, 4096):
    n = mb * (1 << 20) // 4
    x = torch.randn(n, device="cuda"); y = torch.empty_like(x)
    s = t(lambda: x.sum()); c = t(lambda: y.copy_(x))
    print(f"{mb} MB: read {mb * 1.048576e-3 / s / 1e3:.2f} TB/s ({s * 1e6:.1f} us)   copy {2 * mb * 1.048576e-3 / c / 1e3:.2f} TB/s ({c * 1e6:.1f} us)", flush=True)
    del x, y
