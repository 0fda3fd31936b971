#!/usr/bin/env python3
"""What makes hipStreamEndCapture die when a training step is captured after eager steps?  (DESIGN.md "capture abort")

Round 2 recorded a host segfault in hipStreamEndCapture when the diffusion-prior training step was captured after eager steps on the
legacy default stream, and blamed autograd's AccumulateGrad nodes without showing it.  This probe runs each combination in a CHILD
process (a segfault then ends the child only), with faulthandler on so the Python frame of the fault is printed:

    model      : diffusion (DiffusionPrior p_losses, 3 MNIST parts)  |  glow (L3 K4 training step)
    warm-up on : the legacy default stream  |  a side stream (torch's recipe)
    grads      : direct_grads True (backward launches add into the flat buffer, no AccumulateGrad)  |  False (AccumulateGrad nodes)
    opt        : FusedAdam  |  torch.optim.Adam (direct_grads is then moot)

usage: python tools/probe_capture_abort.py            (parent: runs every case, prints a table, writes gpurun_out/probe_capture_abort.json)
       python tools/probe_capture_abort.py CASE       (child)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [(m, w, g) for m in ("diffusion", "glow") for w in ("default", "side") for g in ("direct", "accumulate", "torchadam")]
# the configuration round 2's crash was seen in (tools/bench_diffusion_train.py: K = 4, dim 64, T = 1000, B = 64), default-stream warm-up
CASES += [("diffusion_big", "default", g) for g in ("direct", "accumulate", "torchadam")]
if os.environ.get("NFDP_PROBE_ONLY"):
    CASES = [c for c in CASES if c[0] == os.environ["NFDP_PROBE_ONLY"]]


def child(case):
    import faulthandler
    faulthandler.enable()
    sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
    import torch
    model, warm, grads = case.split(":")
    from nfdp_hip.optim import FusedAdam
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    B = 16
    big = model == "diffusion_big"
    if big:
        model, B = "diffusion", 64
    if model == "diffusion":
        from normalizing_flow import NFBackbone, initialize_with_zeros
        from diffusion_prior import DiffusionPrior, IdentityFormater
        nf = NFBackbone(model_dir="", in_channel=1, L=3, K=4 if big else 2, learn_prior_mean_logs=True, freeze_flow=True).to(dev)
        fmt = IdentityFormater(3, 1, 32)
        shapes = fmt.get_input_shapes()
        prior = DiffusionPrior(latent_formater=fmt, unet_kwargs={"dim": 64 if big else 32, "dim_mults": (1, 2), "resnet_block_groups": 8},
                               diffusion_kwargs={"timesteps": 1000 if big else 100, "beta_schedule": "cosine", "loss_type": "l2"},
                               image_sizes=[int(s[1]) for s in shapes], latent_channels=[int(s[0]) for s in shapes]).to(dev)
        params = list(prior.parameters(only_trainable=True))
        x = (torch.rand(B, 1, 32, 32) - 0.5).to(dev)
        with torch.no_grad():
            latents, _ = nf.transform(x, initialize_with_zeros(1, B, dev))
        prior.train()
        loss_fn = lambda: sum(prior(latents))
    else:
        from normalizing_flow import Glow, GaussianPrior, calculate_loss
        flow, gp = Glow(in_channel=1, L=3, K=2, compute="bf16").to(dev), GaussianPrior(16).to(dev)
        params = list(flow.parameters()) + list(gp.parameters())
        x = (torch.rand(B, 1, 32, 32) - 0.5).to(dev)

        def loss_fn():
            ldj = torch.zeros(B, dtype=torch.float64, device=dev)
            logp = torch.zeros(B, dtype=torch.float64, device=dev)
            zs, ldj, logp = flow.transform(x, ldj, logp)
            return calculate_loss(ldj + logp + gp.compute_log_prob(zs[-1]), 32.0, 3072.0)
    if grads == "torchadam":
        opt = torch.optim.Adam(params, lr=1e-4, capturable=True)
    else:
        opt = FusedAdam(params, lr=1e-4, direct_grads=(grads == "direct"))

    def step():
        loss = loss_fn()
        opt.zero_grad(set_to_none=False) if grads == "torchadam" else opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    if warm == "default":
        for _ in range(3):
            step()
    else:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    print(f"CHILD {case}: warm-up done, capturing", flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss = step()
    print(f"CHILD {case}: capture ended", flush=True)
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    print(f"CHILD {case}: OK loss {float(loss.detach()):.4f}", flush=True)


def main():
    if len(sys.argv) > 1:
        return child(sys.argv[1])
    out = {}
    for m, w, g in CASES:
        case = f"{m}:{w}:{g}"
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), case], capture_output=True, text=True, timeout=240)
            rc, txt = r.returncode, (r.stdout + r.stderr)
        except subprocess.TimeoutExpired as e:
            rc, txt = -999, (e.stdout or "") + (e.stderr or "") if isinstance(e.stdout, str) else "timeout"
        ok = "OK loss" in txt
        tail = [ln for ln in txt.strip().splitlines() if ln.strip()][-14:]
        out[case] = {"rc": rc, "ok": ok, "tail": tail}
        print(f"{case:32s} rc={rc:5d} {'OK' if ok else 'FAILED'}", flush=True)
        if not ok:
            print("    " + "\n    ".join(tail), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "probe_capture_abort.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
