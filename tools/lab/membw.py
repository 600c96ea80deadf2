#!/usr/bin/env python3
"""What plain streaming kernels reach on this box for K1's traffic shape (1 B read + 4 B written per
pixel), next to an int32 copy and a write-only fill.  GPU box only."""
import torch
n = 1536 * 480 * 752
img = torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda")
sc = torch.empty(n, dtype=torch.int32, device="cuda")
sc2 = torch.empty_like(sc)


def t(f, reps=10):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


ms = t(lambda: sc.copy_(img))
print(f"u8 -> int32 convert (K1's traffic, 5 B/px): {ms:.3f} ms = {5*n/ms/1e6:.0f} GB/s")
ms = t(lambda: sc.zero_())
print(f"int32 fill (4 B/px written): {ms:.3f} ms = {4*n/ms/1e6:.0f} GB/s")
ms = t(lambda: sc2.copy_(sc))
print(f"int32 copy (8 B/px): {ms:.3f} ms = {8*n/ms/1e6:.0f} GB/s")
ms = t(lambda: torch.max(sc))
print(f"int32 max-reduce (4 B/px read): {ms:.3f} ms = {4*n/ms/1e6:.0f} GB/s")
