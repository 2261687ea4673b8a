import torch
def t(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for rows in (163840, 163840 * 4, 163840 * 16):
    a = torch.randn(rows, 32, device="cuda"); b = torch.empty_like(a)
    us = t(lambda: b.copy_(a))
    print(f"copy {rows}x32 fp32: {us:.2f} us  {2 * a.numel() * 4 / us / 1e3:.0f} GB/s")
    us = t(lambda: torch.mul(a, 2.0, out=b))
    print(f"mul  {rows}x32 fp32: {us:.2f} us  {2 * a.numel() * 4 / us / 1e3:.0f} GB/s")
    us = t(lambda: b.zero_())
    print(f"zero {rows}x32 fp32: {us:.2f} us  {a.numel() * 4 / us / 1e3:.0f} GB/s")
