"""Deterministic synthetic weights for parity tests (test data generator, not reference code).

Golden files store only inputs/outputs; the (large) weights are regenerated from (key, shape, seed)
by this rule, both when tools/make_goldens.py drives the reference and when the tests drive the
oracle / the HIP path.  Only legacy np.random.RandomState (bit-stable stream) and elementwise
IEEE ops are used — no BLAS/LAPACK — so the arrays are bit-identical on any host.
"""
import zlib

import numpy as np

ZERO_CONV_SUFFIXES = ("net.4.weight", "split.conv.weight", "_GaussianPrior__conv.weight")


def _rs(key: str, seed: int) -> np.random.RandomState:
    return np.random.RandomState((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


GOLDEN = {"invconv_noise": 0.3, "zero_gain": 0.5, "flow_scale_mean": 0.0}
# Deep stacks (K=16): near-orthogonal 1x1 weights, gentler coupling scales and an ActNorm gain that offsets the
# coupling's contraction (s ~ sigmoid(2) = 0.88 on half the channels), so that 48 flows stay norm-preserving in
# BOTH directions like a trained Glow — fp32 round trips and temperature sampling stay meaningful.
DEEP = {"invconv_noise": 0.05, "zero_gain": 0.2, "flow_scale_mean": 0.064}


def synth_tensor(key: str, shape, seed: int, profile: dict = GOLDEN) -> np.ndarray:
    """'Trained-like' value for one state-dict entry (see SURVEY.md §8(d) weights (b)).  The golden files were
    generated with the default GOLDEN profile; do not change it."""
    shape = tuple(int(s) for s in shape)
    rs = _rs(key, seed)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "is_initialized":
        return np.array(1, dtype=np.uint8)
    n = int(np.prod(shape)) if shape else 1
    g = rs.randn(n).reshape(shape)
    if leaf == "logs":
        v = 0.1 * g
    elif leaf == "scale":
        v = 0.05 * g
        if key.endswith(".actnorm.scale") and "_Conv2dActNorm" not in key:
            v = v + profile["flow_scale_mean"]
    elif leaf == "g":
        v = 1.0 + 0.05 * g
    elif leaf == "bias":
        v = 0.05 * g
    elif leaf == "weight":
        if len(shape) == 1:                       # GroupNorm gain
            v = 1.0 + 0.05 * g
        elif "invconv2d" in key:                  # signed permutation + noise: invertible, no LAPACK
            c = shape[0]
            perm = rs.permutation(c)
            sign = np.where(rs.rand(c) < 0.5, -1.0, 1.0)
            q = np.zeros((c, c))
            q[np.arange(c), perm] = sign
            v = (q + profile["invconv_noise"] * g.reshape(c, c) / np.sqrt(c)).reshape(shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            if key.endswith(ZERO_CONV_SUFFIXES):
                std = profile["zero_gain"] / np.sqrt(fan_in)
            elif key.endswith("net.2._Conv2dActNorm__conv.weight"):
                std = np.sqrt(2.0 / fan_in)
            else:
                std = 1.0 / np.sqrt(fan_in)
            v = std * g
    elif leaf == "weights":                       # RandomOrLearnedSinusoidalPosEmb frequencies (cycles per timestep)
        v = 0.01 * g
    else:
        raise KeyError(f"no synth rule for {key}")
    return v.astype(np.float32)


def synth_state(spec, seed: int) -> dict:
    """spec: iterable of (key, shape) -> {key: np.ndarray}."""
    return {k: synth_tensor(k, s, seed) for k, s in spec}


def synth_input(name: str, shape, seed: int, kind: str = "normal") -> np.ndarray:
    rs = _rs("input:" + name, seed)
    n = int(np.prod(shape))
    if kind == "normal":
        return rs.randn(n).reshape(shape).astype(np.float32)
    if kind == "uniform":
        return rs.rand(n).reshape(shape).astype(np.float32)
    if kind == "u8":
        return rs.randint(0, 256, size=n).reshape(shape).astype(np.uint8)
    raise ValueError(kind)


def images_5bit(name: str, shape, seed: int, n_bits: int = 5):
    """Synthetic preprocessed + dequantised batch as in trainer.py:152-155; returns (x, u8, noise)."""
    u8 = synth_input(name + ":u8", shape, seed, "u8")
    noise = synth_input(name + ":dq", shape, seed, "uniform")
    n_bins = np.float32(2.0 ** n_bits)
    x = np.floor(u8.astype(np.float32) / np.float32(2 ** (8 - n_bits))) / n_bins - np.float32(0.5)
    return (x + noise / n_bins).astype(np.float32), u8, noise
