"""Shared helpers for the parity tests."""
import json
import os

import numpy as np
import torch

from tests.golden.synth import synth_tensor

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    g = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: g[k] for k in g.files}


def meta_of(g):
    return json.loads(str(g["meta"]))


def synth_sd(spec, seed, prefix="", profile=None):
    """{key: torch tensor} for a [(key, shape)] spec, using the shared synthesis rule (profile: a golden's meta["profile"] name)."""
    from tests.golden.synth import GOLDEN, DEEP
    prof = {None: GOLDEN, "GOLDEN": GOLDEN, "DEEP": DEEP}[profile]
    return {k: torch.from_numpy(np.array(synth_tensor(prefix + k, s, seed, prof))) for k, s in spec}


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def relerr(a, b):
    """max |a-b| / max|b|: error relative to the scale of the tensor."""
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def relerr_elem(a, b):
    """max_i |a_i - b_i| / max(|b_i|, rms(b)): ELEMENT-wise relative error, with the tensor's rms as the floor below which an
    element counts as 'near zero' (an fp32 result has no relative accuracy at a cancellation).  For a scalar this is |a-b|/|b|.
    Stricter than `relerr` by max|b| / rms(b) (3-5x for latents and images)."""
    a = torch.as_tensor(a).detach().double().cpu().reshape(-1)
    b = torch.as_tensor(b).detach().double().cpu().reshape(-1)
    if b.numel() == 0:
        return 0.0
    floor = b.pow(2).mean().sqrt().clamp_min(1e-30)
    return float(((a - b).abs() / torch.maximum(b.abs(), floor)).max())


_METRICS = {}


def record(name, value):
    """Keep a measured parity figure; tests/conftest.py writes them to gpurun_out/parity_metrics.json at the end of the session."""
    _METRICS[name] = float(value)


def assert_close(a, b, tol, what=""):
    """BASELINE.json's 'within 1e-4 relative' read element-wise (relerr_elem); the tensor-scale figure is reported beside it."""
    e, es = relerr_elem(a, b), relerr(a, b)
    assert e <= tol, f"{what}: element-wise rel err {e:.3e} (tensor-scale {es:.3e}) > {tol:.1e}"
