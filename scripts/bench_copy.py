import torch
dev = torch.device("cuda:0")
for n in (50176 * 256, 12544 * 512, 50176 * 64):
    x = torch.randn(n, device=dev).bfloat16(); y = torch.empty_like(x)
    for _ in range(5): y.copy_(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"copy {n*2/1e6:.1f} MB: {us:.1f} us  -> {2*n*2/us/1e6:.2f} TB/s (read+write)")
