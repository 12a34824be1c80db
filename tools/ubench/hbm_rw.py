import torch
x = torch.empty(204800, 768, device="cuda"); y = torch.randn(204800, 768, device="cuda")
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
us = t(lambda: x.zero_()); print(f"zero_ 629 MB: {us:.1f} us  {x.numel()*4/us/1e6:.2f} TB/s write")
us = t(lambda: x.copy_(y)); print(f"copy_ 629 MB: {us:.1f} us  {2*x.numel()*4/us/1e6:.2f} TB/s r+w")
us = t(lambda: y.sum()); print(f"sum 629 MB: {us:.1f} us  {x.numel()*4/us/1e6:.2f} TB/s read")
