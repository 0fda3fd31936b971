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
    # round 4 entries: host-side predicates and argument checks of the tiled-level kernels
    assert [c for c in range(2, 200, 2) if l.nfdp_tap_coupling_head_supported(c)] == [4, 8, 12, 16, 24, 32, 48]
    assert l.nfdp_tap_coupling_head_chunks(32, 64 * 64, 12) == 25 and l.nfdp_tap_coupling_head_chunks(512, 16, 48) == 1
    assert l.nfdp_tap_coupling_head_ws_bytes(32, 64 * 64, 12) == 32 * 25 * 8 and l.nfdp_tap_coupling_head_ws_bytes(4, 16, 96) == 0
    # kernel forms are per-call arguments (`form`): nothing to set in the library
    assert [l.nfdp_coupling_mlp_image_kind(c, n, 0, 0) for c, n in ((12, 131072), (12, 16384), (4, 100), (24, 131072), (48, 8192))] == [2, 1, 2, 1, 1]
    assert l.nfdp_coupling_mlp_image_kind(12, 131072, 1, 4) == 1 and l.nfdp_coupling_mlp_image_kind(12, 131072, 1, 0) == 2   # NFDP_MLP_NO_K16
    assert l.nfdp_coupling_mlp_image_kind(12, 131072, 0, 2) == 1 and l.nfdp_coupling_mlp_image_kind(12, 1024, 0, 1) == 2     # rows always / never
    assert l.nfdp_coupling_mlp_gather_bf16(p, 6, 1, 4, 4, p, p, 12, 3, None) == -1                              # form 3 does not exist
    assert l.nfdp_coupling_mlp_gather_k16(p, 12, 1, 4, 4, p, None, p, 24, None) == -3                         # C = 24 has no 16x16x32 image
    assert l.nfdp_coupling_fused_packed_bytes(12) > 0
    assert l.nfdp_coupling_mlp_gather_bf16x3(p, 6, 1, 4, 4, p, None, p, 12, 0, None) == -1                        # no low-piece image
    assert l.nfdp_coupling_mlp_gather_bf16x3(p, 48, 1, 4, 4, p, p, p, 96, 0, None) == -3                         # C = 96: NFDP_EUNSUPPORTED
    assert l.nfdp_coupling_mlp_gather_bf16(p + 4, 6, 1, 4, 4, p, p, 12, 0, None) == -2                           # x not 8-byte aligned
    assert l.nfdp_tap_coupling_head_f32(p, p, 1, 4, 4, 20, p, p, 3.0, None, None, None, 0, p, p, p, None, None, 0, None) == -3
    assert l.nfdp_ldj_finish_level_f64(None, 4, None, None, 2, 1, None) == -1
    assert l.nfdp_ldj_finish_level_f64(p, 0, ctypes.byref(ctypes.c_void_p(p)), ctypes.byref(ctypes.c_void_p(None)), 1, 1, None) == 0
    # the split launches of the wide levels: workspace sizes (H1 tiles + 4 partial tiles per pixel tile + a copy of the state), refusals
    assert l.nfdp_level_split_ws_bytes(32, 8, 8, 96) == 32 * (64 * 1024 + 4 * 64 * 96 * 4) + 32 * 64 * 96 * 4
    assert l.nfdp_level_split_ws_bytes(5, 4, 4, 192) == 3 * (32 * 1024 + 4 * 32 * 192 * 4) + 5 * 16 * 192 * 4
    assert l.nfdp_level_split_ws_bytes(32, 4, 4, 48) == 0 and l.nfdp_level_split_ws_bytes(0, 8, 8, 96) == 0
    assert l.nfdp_coupling_fused_ws_bf16(p, p, 1, 4, 4, 48, None, None, 0, p, 1 << 20, None) == -3              # not a wide level
    assert l.nfdp_coupling_fused_ws_bf16(p, p, 1, 8, 8, 96, None, None, 0, p, 16, None) == -2                   # workspace too small
    assert l.nfdp_coupling_fused_ws_bf16(p, p, 1, 8, 8, 96, None, None, 0, None, 1 << 20, None) == -1           # no workspace
    # packed fp32 3x3 weights with Cin % 32 == 0 carry their split image behind the matrix (12 KiB per (64-row tile, 32-k stage))
    assert l.nfdp_conv_packed_bytes(64, 64, 3, 3, 0) == 64 * 576 * 4 + 1 * 18 * 12288
    assert l.nfdp_conv_packed_bytes(100, 96, 3, 3, 0) == 128 * 896 * 4 + 2 * 28 * 12288
    assert l.nfdp_conv_packed_bytes(64, 64, 3, 3, 1) == 64 * 576 * 2 and l.nfdp_conv_packed_bytes(64, 40, 3, 3, 0) == 64 * 384 * 4
    # a ResnetBlock's first GroupNorm carried into its second conv: the producer's tile (a part of an image), the consumer's geometry
    assert l.nfdp_conv2d_stats_tile(64, 16, 16, 64, 64, 3, 8) == 64 and l.nfdp_conv2d_stats_tile(64, 8, 8, 192, 128, 3, 8) == 32
    assert l.nfdp_conv2d_stats_tile(4, 16, 16, 64, 64, 3, 8) == 16 and l.nfdp_conv2d_stats_tile(64, 4, 4, 64, 64, 3, 8) == 16
    assert l.nfdp_conv2d_stats_tile(1100, 4, 4, 64, 64, 3, 8) == 0         # a tile of several images: the one-launch Block's case
    assert l.nfdp_conv2d_stats_tile(64, 16, 16, 64, 96, 3, 8) == 0 and l.nfdp_conv2d_stats_tile(64, 16, 16, 64, 64, 3, 32) == 0
    assert l.nfdp_conv2d_gnin_supported(64, 16, 16, 64, 64, 8, 0) == 1 and l.nfdp_conv2d_gnin_supported(64, 8, 8, 128, 128, 8, 0) == 1
    assert l.nfdp_conv2d_gnin_supported(64, 12, 12, 64, 64, 8, 0) == 0     # H*W not a power of two
    assert l.nfdp_conv2d_gnin_supported(64, 16, 16, 160, 64, 8, 0) == 0    # 20 channels per group / no split image beyond 128 inputs
    assert l.nfdp_conv2d_gnin_supported(64, 16, 16, 64, 64, 8, 8) == 0 and l.nfdp_conv2d_gnin_supported(64, 4, 4, 64, 64, 8, 8) == 1
    assert l.nfdp_conv2d_stats_nhwc_f32(p, 64, 64, None, 0, 0, p, p, 64, 64, 16, 16, 64, 3, None, 8, None, 0, None) == -1
    assert l.nfdp_conv2d_stats_nhwc_f32(p, 64, 64, None, 0, 0, p, p, 64, 64, 12, 12, 96, 3, None, 8, p, 0, None) == -3
    assert l.nfdp_conv2d_gnin_nhwc_f32(p, 64, p, 48, 8, p, p, None, 0, 1e-5, p, p, 64, 64, 16, 16, 64, 64, None, 0, None, None, None, 0,
                                       1e-5, None, 0, 0, None, None) == -2  # in_tile does not divide the image
    assert l.nfdp_groupnorm_silu_sums_supported(64, 256, 64, 8, 64) == 1 and l.nfdp_groupnorm_silu_sums_supported(32, 4096, 64, 16, 64) == 1
    assert l.nfdp_groupnorm_silu_sums_supported(64, 256, 192, 8, 64) == 0 and l.nfdp_groupnorm_silu_sums_supported(64, 256, 64, 8, 48) == 0
    assert l.nfdp_groupnorm_silu_sums_f32(p, p, None, 64, 64, 256, 64, 8, p, p, None, 0, 1e-5, None, None) == -1
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


def test_tensor_of_another_device_is_refused(monkeypatch):
    """Every launch goes to the CURRENT device's stream: a pointer of a tensor that lives on another GPU must raise (a rank
    that never called torch.cuda.set_device would otherwise queue kernels on device 0 against memory of device N)."""
    class Dev:
        def __init__(self, i): self.index, self.type = i, "cuda"
        def __str__(self): return f"cuda:{self.index}"

    class FakeTensor:
        is_cuda = True
        def __init__(self, i): self.device = Dev(i)
        def numel(self): return 4
        def data_ptr(self): return 0x1000

    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    assert nfdp_hip._ptr(FakeTensor(0)) == 0x1000
    with pytest.raises(nfdp_hip.NfdpError, match="current device is cuda:0"):
        nfdp_hip._ptr(FakeTensor(1))
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 1)
    assert nfdp_hip._ptr(FakeTensor(1)) == 0x1000


def test_default_device_makes_the_ranks_gpu_current(monkeypatch):
    """Under a launcher (LOCAL_RANK set) building a module selects the rank's GPU as the current device, like the reference's
    process-wide "cuda" but per process; without one it is the current device."""
    from normalizing_flow import base
    calls = []
    state = {"cur": 0}
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: state["cur"])
    monkeypatch.setattr(torch.cuda, "set_device", lambda i: (calls.append(i), state.update(cur=i)))
    monkeypatch.setenv("LOCAL_RANK", "3")
    assert base.default_device() == torch.device("cuda", 3) and calls == [3]
    assert base.default_device() == torch.device("cuda", 3) and calls == [3]       # already current: not set again
    monkeypatch.delenv("LOCAL_RANK")
    state["cur"] = 5
    assert base.default_device() == torch.device("cuda", 5) and calls == [3]


def test_fused_adam_state_dict_is_torch_adam_layout():
    """FusedAdam.state_dict()/load_state_dict() interchange with torch.optim.Adam's (the reference checkpoints hold
    optim.state_dict(), prior.py:113, and resume with optimizer.load_state_dict, trainer.py:115)."""
    from nfdp_hip.optim import FusedAdam
    torch.manual_seed(0)
    mk = lambda: [torch.nn.Parameter(torch.randn(4, 3, 1, 1)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(2, 2))]
    pt, pf = mk(), mk()
    ref = torch.optim.Adam(pt, lr=3e-4)
    assert ref.state_dict()["state"] == {} and FusedAdam(mk()).state_dict()["state"] == {}
    for _ in range(3):
        for p in pt:
            p.grad = torch.randn_like(p)
        ref.step()
    sd = ref.state_dict()
    opt = FusedAdam(clipped=pf[:2], unclipped=pf[2:], lr=1e-3)
    opt.load_state_dict(sd)
    assert opt.lr == 3e-4 and opt.step_count == 3
    back = opt.state_dict()
    assert set(back) == {"state", "param_groups"} and set(back["state"]) == {0, 1, 2}
    for i in range(3):
        assert float(back["state"][i]["step"]) == 3.0
        for k in ("exp_avg", "exp_avg_sq"):
            assert back["state"][i][k].shape == sd["state"][i][k].shape
            assert torch.equal(back["state"][i][k], sd["state"][i][k])
    g = back["param_groups"][0]
    assert g["params"] == [0, 1, 2] and g["lr"] == 3e-4 and tuple(g["betas"]) == (0.9, 0.999) and g["weight_decay"] == 0.0
    fresh = torch.optim.Adam(mk(), lr=1.0)
    fresh.load_state_dict(back)                                       # torch accepts what FusedAdam wrote
    assert fresh.state_dict()["param_groups"][0]["lr"] == 3e-4
    assert torch.equal(fresh.state_dict()["state"][1]["exp_avg"], sd["state"][1]["exp_avg"])
    # the reference's resume loop (trainer.py:119-120) sets the learning rate through param_groups
    for group in opt.param_groups:
        group["lr"] = 5e-5
    assert opt.lr == 5e-5
    # defaults of the optimizers init_optimizer builds (utils.py:131-134): Adam 0, AdamW 0.01
    assert FusedAdam(mk()).weight_decay == 0.0 and FusedAdam(mk(), adamw=True).weight_decay == 0.01
    # legacy flat layout still loads
    opt.load_state_dict({"step": 7, "exp_avg": torch.zeros(opt.n), "exp_avg_sq": torch.ones(opt.n), "lr": 1e-3})
    assert opt.step_count == 7 and opt.lr == 1e-3
    with pytest.raises(ValueError):
        opt.load_state_dict(torch.optim.Adam(mk()[:2]).state_dict())
    # what one flat plain-Adam launch cannot represent is refused, not loaded as something else
    pa = mk()
    ams = torch.optim.Adam(pa, amsgrad=True)
    with pytest.raises(ValueError, match="amsgrad"):
        opt.load_state_dict(ams.state_dict())
    with pytest.raises(ValueError, match="decoupled_weight_decay"):
        opt.load_state_dict(torch.optim.AdamW(mk()).state_dict())
    two = torch.optim.Adam([{"params": pa[:2], "lr": 1e-3}, {"params": pa[2:], "lr": 1e-4}])
    with pytest.raises(ValueError, match="differ in lr"):
        opt.load_state_dict(two.state_dict())
    FusedAdam(mk(), adamw=True).load_state_dict(torch.optim.AdamW(mk()).state_dict())
    # param-group writes: the four hyper-parameters reach the optimizer, anything else is read-only
    grp = opt.param_groups[0]
    grp["weight_decay"], grp["betas"], grp["eps"] = 0.02, (0.8, 0.9), 1e-7
    assert opt.weight_decay == 0.02 and opt.betas == (0.8, 0.9) and opt.eps == 1e-7
    with pytest.raises(KeyError):
        grp["amsgrad"] = True


def test_actnorm_init_mirror_follows_the_buffer():
    """The host mirror of ActNorm.is_initialized is keyed on the buffer (identity, storage, version): fill_/copy_, load_state_dict
    and re-assigning the buffer are all seen without a reset call."""
    from normalizing_flow.transforms import ActNorm
    a = ActNorm(4)
    assert a.initialized() is False
    a.is_initialized.fill_(1)
    assert a.initialized() is True
    a.is_initialized = torch.tensor(0, dtype=torch.uint8)
    assert a.initialized() is False
    sd = a.state_dict()
    sd["is_initialized"] = torch.tensor(1, dtype=torch.uint8)
    a.load_state_dict(sd)
    assert a.initialized() is True
    a.is_initialized.data.zero_()                                     # invisible to version counters: documented reset
    a.reset_init_cache()
    assert a.initialized() is False


def test_module_apply_bumps_the_parameter_epoch():
    from normalizing_flow.transforms import ActNorm
    e0 = nfdp_hip.param_epoch()
    ActNorm(4).float()
    assert nfdp_hip.param_epoch() > e0


def test_shipped_library_has_no_measurement_switches():
    """The default build reads no environment variable and carries no NFDP_* switch name (A/B environment reads and the timing-only
    NFDP_ABLATE_* / *_STAMPS code paths compile only with -DNFDP_DIAG: tools/build_variant.sh), and the process-wide mode setters of
    earlier rounds are gone from the ABI (modes are per-call arguments: `compute`, `form`, `variant`)."""
    import nfdp_hip.build as B
    blob = open(B.LIB, "rb").read()
    assert b"NFDP_" not in blob
    l = nfdp_hip.lib()
    for gone in ("nfdp_set_f32_products", "nfdp_get_f32_products", "nfdp_set_mlp_rows", "nfdp_set_mlp_k16"):
        assert not hasattr(l, gone), gone
    hdr = open(os.path.join(ROOT, "include", "nfdp.h")).read()
    assert "nfdp_set_" not in hdr
