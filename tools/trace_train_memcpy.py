#!/usr/bin/env python3
"""Memcpy / memset activity of one eager training step (torch.profiler device rows), and which operators issue them."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import argparse, torch
import bench
args = argparse.Namespace(L=3, K=16, compute="bf16", batch=512, init="synthetic", init_mode="rank0")
dev = torch.device("cuda", 0)
train_step, _, _, _ = bench.train_objects(args, 0, 1, dev, "0")
for _ in range(3):
    train_step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    train_step()
    torch.cuda.synchronize()
rows = [r for r in prof.key_averages() if any(s in r.key.lower() for s in ("memcpy", "memset", "copybuffer", "fillbuffer"))]
for r in rows:
    print(f"{r.count:5d} x {r.device_time_total / max(r.count, 1):8.1f} us  {r.key}")
# the CPU operators that own the memcpy launches
par = collections.Counter()
for e in prof.events():
    if "memcpy" in e.name.lower() and e.cpu_parent is not None:
        p = e.cpu_parent
        chain = []
        while p is not None and len(chain) < 4:
            chain.append(p.name)
            p = p.cpu_parent
        par[" <- ".join(chain)] += 1
for k, n in par.most_common(15):
    print(f"{n:5d}  {k[:200]}")
