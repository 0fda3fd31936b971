#!/usr/bin/env python3
"""Which capture shape ends in the host segfault inside hipStreamEndCapture (round 4: two Glow L5 passes on two streams in one graph)?
Each variant runs in a child process; prints ok / the child's exit status.  usage: python3 tools/probe_nested_capture.py [variant]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]

VARIANTS = ["torch_nested1", "torch_nested2", "torch_refork", "torch_nested2_alloc", "glow_on_fork1", "glow_twice_same_stream", "glow_two_forks_noside",
            "glow_two_forks_fwd_only", "glow_two_forks", "glow_two_forks_prewarmed"]


def child(v):
    import torch
    dev = "cuda"
    cap = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    a = torch.zeros(1 << 16, device=dev)
    b = torch.zeros(1 << 16, device=dev)

    def nested(x, alloc=False):
        main = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            y = (x * 2.0) if alloc else x.add_(1.0)
        x.mul_(1.5)
        main.wait_stream(side)
        if alloc:
            y.record_stream(main)
            x.add_(y)
        return x

    if v.startswith("torch"):
        torch.cuda.synchronize()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap), torch.cuda.graph(g, stream=cap):
            if v == "torch_refork":
                nested(a); nested(a)
            else:
                forks = [torch.cuda.Stream() for _ in range(1 if v == "torch_nested1" else 2)]
                for s, t in zip(forks, (a, b)):
                    s.wait_stream(cap)
                    with torch.cuda.stream(s):
                        nested(t, alloc=v.endswith("alloc"))
                for s in forks:
                    cap.wait_stream(s)
        g.replay(); torch.cuda.synchronize()
        print("ok", float(a.sum()), float(b.sum()))
        return
    import nfdp_hip as K
    from normalizing_flow import Glow, glow as G
    from tests.golden.synth import synth_tensor, DEEP, images_5bit
    import numpy as np
    with torch.no_grad():
        flow = Glow(in_channel=3, L=5, K=2, compute="bf16")
        flow.load_state_dict({k: torch.from_numpy(np.array(synth_tensor(k, s.shape, 9, DEEP))) for k, s in flow.state_dict().items()})
        flow.to(dev)
        xs = [K.to_nhwc(torch.from_numpy(images_5bit(f"two{i}", (2, 3, 64, 64), 21 + i)[0]).to(dev)).contiguous() for i in range(2)]
        z = lambda: torch.zeros(2, dtype=torch.float64, device=dev)
        for x in xs:
            zs = flow.transform_nhwc(x, z(), None); flow.invert_nhwc(zs)
        if v == "glow_two_forks_noside":
            G.LOGDETS_ON_SIDE_STREAM = False
        forks = [torch.cuda.Stream(), torch.cuda.Stream()]
        if v == "glow_two_forks_prewarmed":                      # eager passes on the very streams used under capture
            for s, x in zip(forks, xs):
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    flow.invert_nhwc(flow.transform_nhwc(x, z(), None))
        torch.cuda.synchronize()
        ldjs = [z(), z()]
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap), torch.cuda.graph(g, stream=cap):
            if v == "glow_twice_same_stream":
                for i in range(2):
                    flow.invert_nhwc(flow.transform_nhwc(xs[i], ldjs[i], None))
            else:
                use = forks[:1] if v == "glow_on_fork1" else forks
                for i, s in enumerate(use):
                    s.wait_stream(cap)
                    with torch.cuda.stream(s):
                        zs = flow.transform_nhwc(xs[i], ldjs[i], None)
                        if v != "glow_two_forks_fwd_only":
                            flow.invert_nhwc(zs)
                for s in use:
                    cap.wait_stream(s)
        g.replay(); torch.cuda.synchronize()
        print("ok", [float(l.sum()) for l in ldjs])


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        for v in (sys.argv[1:] or VARIANTS):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", v], capture_output=True, text=True, timeout=300)
            last = (r.stdout.strip().splitlines() or [""])[-1]
            err = [l for l in r.stderr.splitlines() if "Error" in l or "error" in l or "Segmentation" in l][:2]
            print(f"{v:32s} rc={r.returncode:4d}  {last}  {' | '.join(err)}", flush=True)
