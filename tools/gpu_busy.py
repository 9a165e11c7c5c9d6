"""Keeps cuda:0 busy for N seconds (large matmuls back to back): to see what a short-lived process gains from a device
that is out of its idle clocks.  usage: gpu_busy.py SECONDS"""
import sys, time, torch
t_end = time.time() + float(sys.argv[1])
a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
while time.time() < t_end:
    for _ in range(20):
        a = (a @ a).clamp_(-1, 1)
    torch.cuda.synchronize()
