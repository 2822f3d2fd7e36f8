"""GPU parity of the composed path: backbone forward and the samplers, HIP vs the golden
fixtures produced by the reference (tests/golden) and vs the CPU oracle."""
import numpy as np
import pytest
import torch

import fdbm_amd
from fdbm_amd.arch import Spec, VARIANTS
from fdbm_amd.backbone import HipNCSNpp
from fdbm_amd.weights import fill_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MINI64 = dict(nf=64, ch_mult=(1, 1, 2, 2, 2, 2, 2), num_res_blocks=2, attn_resolutions=(16,))


def T(a):
    return torch.from_numpy(np.asarray(a))


_CACHE = {}


def net(name, dtype=torch.float32, fused=None):
    key = (name, dtype, fused)
    if key not in _CACHE:
        hp = MINI64 if name == "mini64" else VARIANTS[name]
        _CACHE[key] = HipNCSNpp(dtype=dtype, device=DEV, fused=fused, **hp)
    return _CACHE[key]


@pytest.mark.parametrize("name,fix", [("mini64", "backbone_mini64"), ("ncsnpp_v2_5M", "backbone_v2_5M")])
def test_backbone_fp32_vs_reference(golden, name, fix):
    g = golden(fix)
    out = net(name)(T(g["x"]).to(DEV), T(g["y"]).to(DEV), T(g["t"]).to(DEV)).cpu()
    ref = T(g["out"])
    assert out.shape == ref.shape
    err = (out - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err < 5e-5 * max(scale, 1.0), (err, scale)      # fp32 MFMA vs oneDNN summation order
    assert torch.all(out[:, :, 256] == 0)                  # Nyquist row re-appended as zeros


@pytest.mark.parametrize("name,fix", [("mini64", "backbone_mini64"), ("ncsnpp_v2_5M", "backbone_v2_5M")])
def test_backbone_fp32_fused_mode(golden, name, fix):
    """The fused program (GroupNorm statistics from conv epilogues via fp32 atomics, GroupNorm+SiLU
    inside the consuming conv, Combine in the epilogue) in f32 storage: same function, looser
    tolerance because the statistics are summed in run-to-run order in fp32."""
    g = golden(fix)
    m = net(name, torch.float32, fused=True)
    out = m(T(g["x"]).to(DEV), T(g["y"]).to(DEV), T(g["t"]).to(DEV)).cpu()
    ref = T(g["out"])
    err = (out - ref).abs().max().item()
    assert err < 2e-4 * max(ref.abs().max().item(), 1.0), err
    prog = m.program(*[int(v) for v in (g["x"].shape[0], g["x"].shape[2], g["x"].shape[3])])
    assert prog.n_slots > 0


@pytest.mark.parametrize("name,fix", [("mini64", "backbone_mini64"), ("ncsnpp_v2_5M", "backbone_v2_5M")])
def test_backbone_bf16_close(golden, name, fix):
    g = golden(fix)
    out = net(name, torch.bfloat16)(T(g["x"]).to(DEV), T(g["y"]).to(DEV), T(g["t"]).to(DEV)).cpu()
    ref = T(g["out"])
    rel = ((out - ref).abs().pow(2).sum() / ref.abs().pow(2).sum()).sqrt().item()
    assert rel < 3e-2, rel                                  # bf16 storage, fp32 accumulate


SAMPLERS = [
    ("sb_bb_ode_ei_N5", dict(path="sb", noise_schedule="bb", N=5, sampler_type="ode_ei"), {}),
    ("fm_ot_ode_ei_N5", dict(path="fm", noise_schedule="ot", N=5, sampler_type="ode_ei"), {}),
    ("sb_bb_sde_ei_N5", dict(path="sb", noise_schedule="bb", N=5, sampler_type="sde_ei"), {}),
    ("sb_ve_sde_ei_N4", dict(path="sb", noise_schedule="ve", N=4, sampler_type="sde_ei"), {}),
    ("sb_bb_pc_N4", dict(path="sb", noise_schedule="bb", N=4, sampler_type="pc"),
     dict(predictor_name="euler_maruyama", corrector_name="ald", corrector_steps=1, snr=0.5, denoise=False)),
    ("sb_vp_pc_N3", dict(path="sb", noise_schedule="vp", N=3, sampler_type="pc"),
     dict(predictor_name="euler_maruyama", corrector_name="langevin", corrector_steps=1, snr=0.3, denoise=True)),
]


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("tag,bkw,skw", SAMPLERS, ids=[s[0] for s in SAMPLERS])
def test_samplers_vs_reference(golden, tag, bkw, skw, use_graph):
    """Identical (noisy_spec, seed, N): <= 1e-4 max-abs on the final complex spectrogram."""
    g = golden("samplers")
    y = T(g["y"]).to(DEV)
    br = fdbm_amd.Bridge(**bkw)
    gen = torch.Generator().manual_seed(1234)               # the reference ran torch.manual_seed(1234) on CPU
    out = br.sampler(net("ncsnpp_v2_5M"), y, generator=gen, use_graph=use_graph, **skw).cpu()
    err = (out - T(g[tag])).abs().max().item()
    assert err < 1e-4, (tag, err)


def test_sampler_batched_and_graph_equals_eager(golden):
    g = golden("samplers")
    y = T(g["mini64_y"]).to(DEV)
    br = fdbm_amd.Bridge("sb", N=3, sampler_type="ode_ei")
    m = net("mini64")
    a = br.sampler(m, y, generator=torch.Generator().manual_seed(99), use_graph=False)
    b = br.sampler(m, y, generator=torch.Generator().manual_seed(99), use_graph=True)
    c = br.sampler(m, y, generator=torch.Generator().manual_seed(99), use_graph=True)   # replay
    assert torch.equal(torch.view_as_real(a), torch.view_as_real(b))
    assert torch.equal(torch.view_as_real(b), torch.view_as_real(c))
    assert (a.cpu() - T(g["mini64_sb_bb_ode_ei_N3"])).abs().max() < 1e-4


def test_sampler_ode_int(golden):
    g = golden("samplers")
    br = fdbm_amd.Bridge("sb", N=5, sampler_type="ode_int")
    out = br.sampler(net("ncsnpp_v2_5M"), T(g["y"]).to(DEV), generator=torch.Generator().manual_seed(1234),
                     rtol=1e-2, atol=1e-2).cpu()
    assert (out - T(g["sb_bb_ode_int"])).abs().max() < 2e-3   # adaptive steps amplify fp noise


def test_full_size_ncsnpp_v2_vs_reference(golden):
    """BASELINE configs[1] geometry: [1,1,257,256], ncsnpp_v2 (65.6 M), N=30 ode_ei, fp32 parity mode."""
    g = golden("full_ncsnpp_v2")
    m = net("ncsnpp_v2")
    out = m(T(g["x"]).to(DEV), T(g["y"]).to(DEV), T(g["t"]).to(DEV)).cpu()
    ref = T(g["fwd"])
    err = (out - ref).abs().max().item()
    assert err < 5e-5 * max(ref.abs().max().item(), 1.0), err
    y = T(g["y"]).to(DEV)
    for key, bkw in (("sb_bb_ode_ei_N30", dict(path="sb", noise_schedule="bb")),
                     ("fm_ot_ode_ei_N30", dict(path="fm", noise_schedule="ot"))):
        br = fdbm_amd.Bridge(N=30, sampler_type="ode_ei", **bkw)
        out = br.sampler(m, y, generator=torch.Generator().manual_seed(4321)).cpu()
        err = (out - T(g[key])).abs().max().item()
        assert err < 1e-4, (key, err)


def test_fail_loudly_on_cpu_tensors():
    m = net("mini64")
    x = torch.zeros(1, 1, 257, 64, dtype=torch.complex64)
    with pytest.raises(RuntimeError):
        m(x, x, torch.ones(1))
