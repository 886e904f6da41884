import torch, time
x = torch.empty(28_500_000, dtype=torch.float64, device='cuda').normal_()
b = torch.empty(28_500_000, dtype=torch.uint8, device='cuda')
def t(f, n=20):
    f(); torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e3
us=t(lambda: x.sum()); print('sum f64 228MB: %.1f us  %.2f TB/s'%(us, 228e6/us/1e6))
y=torch.empty_like(x); us=t(lambda: y.copy_(x)); print('copy f64 228MB r+w: %.1f us %.2f TB/s (r+w)'%(us, 456e6/us/1e6))
us=t(lambda: b.zero_()); print('zero u8 28.5MB: %.1f us %.2f TB/s'%(us, 28.5e6/us/1e6))
us=t(lambda: y.zero_()); print('zero f64 228MB: %.1f us %.2f TB/s'%(us, 228e6/us/1e6))
us=t(lambda: torch.gt(x, 0, out=b.view(torch.bool))); print('gt f64->bool: %.1f us %.2f TB/s'%(us, (228e6+28.5e6)/us/1e6))
