#!/usr/bin/env python3
"""Fused Adam(amsgrad) on the c4 weight matrix (2 M x 200): TGCN_ADAM_NT=0 / 1, one process each."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
import pytextgcn_amd as pkg
p = torch.nn.Parameter(torch.randn(2_000_000, 200, device="cuda:0"))
p.grad = torch.randn_like(p)
opt = pkg.optim.Adam([p], lr=0.05, amsgrad=True)
for _ in range(3): opt.step()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
torch.cuda.synchronize(); ev[0].record()
for _ in range(20): opt.step()
ev[1].record(); torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / 20
print(f"  {ms:.3f} ms per step, {36 * p.numel() / ms / 1e6:.0f} GB/s")
''' % ROOT
for label, env in [("plain loads/stores", {"TGCN_ADAM_NT": "0"}), ("non-temporal state + gradient", {"TGCN_ADAM_NT": "1"})]:
    print(label, flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env={**os.environ, **env}, check=True)
