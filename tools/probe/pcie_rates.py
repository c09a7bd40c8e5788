#!/usr/bin/env python3
"""Probe (GPU box): what the host link gives - pinned H2D, D2H, and both at once (two streams)."""
import time, torch
n = 1 << 30
h_in = torch.empty(n, dtype=torch.uint8).pin_memory(); h_out = torch.empty(n, dtype=torch.uint8).pin_memory()
d_a = torch.empty(n, dtype=torch.uint8, device="cuda"); d_b = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(f, reps=4):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
def h2d():
    with torch.cuda.stream(s1): d_a.copy_(h_in, non_blocking=True)
def d2h():
    with torch.cuda.stream(s2): h_out.copy_(d_b, non_blocking=True)
def both(): h2d(); d2h()
def chunks(k=16):
    m = n // k
    for i in range(k):
        with torch.cuda.stream(s1): d_a[i*m:(i+1)*m].copy_(h_in[i*m:(i+1)*m], non_blocking=True)
        with torch.cuda.stream(s2): h_out[i*m:(i+1)*m].copy_(d_b[i*m:(i+1)*m], non_blocking=True)
g = n / 2**30
print("H2D %.1f GiB/s, D2H %.1f GiB/s, both at once %.1f + %.1f GiB/s, both in 64 MiB pieces %.1f + %.1f GiB/s" % (
    g / run(h2d), g / run(d2h), g / run(both), g / run(both), g / run(chunks), g / run(chunks)))
