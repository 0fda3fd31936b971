#!/usr/bin/env python3
"""Where the peak memory of one training step (Glow L3 K16, B = 512, bf16) is reached: allocated bytes after the forward pass and the
running peak at the start / end of every level's backward node."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import argparse, torch
import bench
from normalizing_flow import training as T
args = argparse.Namespace(L=3, K=16, compute="bf16", batch=512, init="synthetic", init_mode="rank0")
dev = torch.device("cuda", 0)
train_step, fwd_bwd, opt, _ = bench.train_objects(args, 0, 1, dev, "0")
for _ in range(2):
    train_step()
torch.cuda.synchronize()
G = lambda b: f"{b / 2**30:6.2f} GiB"
orig = T.LevelFusedFn.backward


def wrapped(ctx, *g):
    torch.cuda.synchronize()
    h = ctx.saved_tensors[1].shape[2]
    print(f"  level {h}x{h} backward starts: allocated {G(torch.cuda.memory_allocated())}, peak so far {G(torch.cuda.max_memory_allocated())}")
    torch.cuda.reset_peak_memory_stats()
    out = orig(ctx, *g)
    torch.cuda.synchronize()
    print(f"  level {h}x{h} backward ends:   allocated {G(torch.cuda.memory_allocated())}, peak inside {G(torch.cuda.max_memory_allocated())}")
    return out


T.LevelFusedFn.backward = staticmethod(wrapped)
torch.cuda.reset_peak_memory_stats()
print(f"before the step: allocated {G(torch.cuda.memory_allocated())}")
train_step()
torch.cuda.synchronize()
print(f"after the step: allocated {G(torch.cuda.memory_allocated())}")
