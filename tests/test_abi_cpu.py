"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/nfdp.h
declares, argument validation works without a GPU, host classes keep the reference's state-dict keys, and the
product path refuses to run without the GPU (no silent fallback)."""
import ctypes
import json
import os
import re

import pytest
import torch

import nfdp_hip
from tests.util import load_golden, meta_of

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    protos = nfdp_hip.parse_header()
    assert len(protos) >= 30
    raw = ctypes.CDLL(nfdp_hip.LIB_PATH)
    text = open(os.path.join(ROOT, "include", "nfdp.h")).read()
    declared = set(re.findall(r"\b(nfdp_\w+)\s*\(", re.sub(r"/\*.*?\*/", "", text, flags=re.S)))
    assert declared == set(protos)
    for name in declared:
        assert hasattr(raw, name), name
    assert nfdp_hip.lib().nfdp_version() >= 100


def test_argument_validation_without_gpu():
    l = nfdp_hip.lib()
    assert l.nfdp_squeeze_nhwc_f32(None, 3, None, 1, 3, 4, 4, None) == -1          # NFDP_EINVAL
    assert b"null" in l.nfdp_last_error()
    buf = (ctypes.c_float * 64)()
    p = ctypes.addressof(buf)
    assert l.nfdp_squeeze_nhwc_f32(p, 3, p, 1, 3, 5, 4, None) == -2                # odd H -> NFDP_ESHAPE
    assert l.nfdp_coupling_fwd_f32(p, p, 1, 4, 3, None, None, None) == -2           # odd C
    assert l.nfdp_conv2d_nhwc(p, 0, 4, p, p, 0, 4, 1, 2, 2, 4, 4, 2, 2, None, None, 0.0, 0, 0, None) == -3  # even kernel
    assert l.nfdp_preprocess_f32(p, None, p, 4, 9, None) == -1
    assert l.nfdp_conv_packed_bytes(512, 6, 3, 3, 1) == 512 * 64 * 2
    assert l.nfdp_conv_packed_bytes(12, 512, 3, 3, 0) == 16 * 4608 * 4
    assert l.nfdp_actnorm_init_ws_bytes(512) > 0
    # empty batches are a no-op, not an error
    assert l.nfdp_squeeze_nhwc_f32(p, 3, p, 0, 3, 4, 4, None) == 0
    assert l.nfdp_coupling_fwd_f32(p, p, 0, 4, 4, None, None, None) == 0


@pytest.mark.parametrize("tag", ["M", "N", "D"])
def test_state_dict_keys_match_reference(tag):
    from normalizing_flow import Glow, GaussianPrior
    m = meta_of(load_golden(f"glow_{tag}.npz"))
    flow = Glow(in_channel=m["C"], L=m["L"], K=m["K"])
    prior = GaussianPrior(2 ** (m["L"] + 1) * m["C"])
    assert [(k, list(v.shape)) for k, v in flow.state_dict().items()] == [(k, list(s)) for k, s in m["flow_spec"]]
    assert [(k, list(v.shape)) for k, v in prior.state_dict().items()] == [(k, list(s)) for k, s in m["prior_spec"]]
    assert flow.state_dict()["blocks.0.flows.0.actnorm.is_initialized"].dtype == torch.uint8
    assert (flow.L, flow.K, flow.in_channel) == (m["L"], m["K"], m["C"])


def test_unet_state_dict_keys_match_reference():
    Unet = pytest.importorskip("diffusion_prior.gaussian_diffusion").Unet
    dm = meta_of(load_golden("diffusion.npz"))
    for tag, m in dm.items():
        u = Unet(dim=64, dim_mults=tuple(m["mults"]), channels=m["ch"], resnet_block_groups=m["groups"])
        assert [(k, list(v.shape)) for k, v in u.state_dict().items()] == [(k, list(s)) for k, s in m["spec"]], tag


def test_output_shapes_and_errors():
    from normalizing_flow import calculate_output_shapes, init_optimizer
    assert calculate_output_shapes(3, 3, 32) == [(6, 16, 16), (12, 8, 8), (48, 4, 4)]
    with pytest.raises(ValueError):
        calculate_output_shapes(3, 3, 30)
    with pytest.raises(ValueError):
        init_optimizer("sgd", [], 1e-3)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_path_fails_loudly_without_gpu():
    from normalizing_flow import Glow
    flow = Glow(in_channel=1, L=2, K=1)
    x = torch.zeros(1, 1, 8, 8)
    with torch.no_grad(), pytest.raises(nfdp_hip.NfdpError):
        flow.transform(x, torch.zeros(1, dtype=torch.float64), torch.zeros(1, dtype=torch.float64))


def test_no_oracle_import_in_product():
    pkg = os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(d, f)).read()
                assert "oracle" not in src.lower() or f == "build.py", f"{f} mentions the oracle"
