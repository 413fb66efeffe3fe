import torch, time
x = torch.empty(1 << 30, dtype=torch.uint8, device="cuda").random_(0, 120)
xi = x.view(torch.int32)
y = torch.empty_like(xi)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ms = t(lambda: y.copy_(xi)); print("copy 1GiB int32: %.3f ms  %.0f GB/s (r+w)" % (ms, 2 * (1 << 30) / ms / 1e6))
ms = t(lambda: xi.sum()); print("sum int32 1GiB: %.3f ms  %.0f GB/s (read)" % (ms, (1 << 30) / ms / 1e6))
xf = x.view(torch.float32)
ms = t(lambda: xf.sum()); print("sum f32 1GiB: %.3f ms  %.0f GB/s (read)" % (ms, (1 << 30) / ms / 1e6))
ms = t(lambda: torch.max(xi)); print("max int32 1GiB: %.3f ms  %.0f GB/s (read)" % (ms, (1 << 30) / ms / 1e6))
