#!/usr/bin/env python3
"""Which torch operators of one eager training step end in device-to-device memcpys (rocprofv3 shows ~360 __amd_rocclr_copyBuffer per
step): torch.profiler with stacks, grouped by the Python frame that issued them."""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    train_step()
    torch.cuda.synchronize()
ev = prof.events()
cnt = collections.Counter()
shapes = collections.defaultdict(set)
for e in ev:
    if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::fill_", "aten::zero_", "aten::zeros", "aten::empty_like"):
        st = [s for s in (e.stack or []) if ("training.py" in s or "nfdp_hip" in s or "bench.py" in s or "glow.py" in s or "utils.py" in s or "prior.py" in s)
              and "torch/" not in s]
        key = (e.name, " <- ".join(x.split("/")[-1][:60] for x in st[:2]) if st else "?")
        cnt[key] += 1
        shapes[key].add(str(e.input_shapes)[:80])
for (name, where), n in cnt.most_common(40):
    print(f"{n:5d}  {name:18s} {where[-110:]}   {sorted(shapes[(name, where)])[:2]}")
print()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
