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


def synth_sd(spec, seed, prefix=""):
    """{key: torch tensor} for a [(key, shape)] spec, using the shared synthesis rule."""
    return {k: torch.from_numpy(np.array(synth_tensor(prefix + k, s, seed))) for k, s in spec}


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def relerr(a, b):
    """max |a-b| / max|b| — the 'relative' of BASELINE.json's 1e-4 gate (scale of the tensor)."""
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def assert_close(a, b, tol, what=""):
    e = relerr(a, b)
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol:.1e}"
