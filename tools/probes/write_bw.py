#!/usr/bin/env python
"""Write / copy bandwidth of the box (torch fill_ / copy_ on the layer1 output size, 822 MB): the roof the fused layer1
kernels' output stream is priced against."""
import torch
dev = torch.device("cuda", 0)
n = 512 * 56 * 56 * 256
y = torch.empty(n, dtype=torch.bfloat16, device=dev)
x = torch.ones(n, dtype=torch.bfloat16, device=dev)
def t(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
us = t(lambda: y.fill_(1.0)); print("fill  822 MB: %.1f us  %.2f TB/s written" % (us, n * 2 / us * 1e-6))
us = t(lambda: y.zero_()); print("zero  822 MB: %.1f us  %.2f TB/s written" % (us, n * 2 / us * 1e-6))
us = t(lambda: y.copy_(x)); print("copy  822 MB: %.1f us  %.2f TB/s read + %.2f TB/s written" % (us, n * 2 / us * 1e-6, n * 2 / us * 1e-6))
us = t(lambda: torch.sum(x)); print("sum   822 MB: %.1f us  %.2f TB/s read" % (us, n * 2 / us * 1e-6))
