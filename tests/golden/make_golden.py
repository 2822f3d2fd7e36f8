#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE.

Run in the build container only (needs /root/reference; the GPU box never has
it):   python tests/golden/make_golden.py [--full]

The reference is imported as-is from /root/reference with inert stub modules for
libraries its hot path never calls but its files import at module top
(pesq, pystoi; pytorch_lightning and torchaudio for data_module) [SURVEY.md 8(c)].
Weights come from the product's deterministic filler (fdbm_amd.weights) written
into the reference modules' state_dict by key.  Inputs are seeded; noise is drawn
with torch.manual_seed on CPU in the reference's call order.

Outputs are DATA only (inputs + expected outputs as .npz); no reference source is
copied.  Fixture metadata records torch version / thread count.
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

# ---- inert stubs for metric / IO / trainer libraries ----------------------------
for name in ("pesq", "pystoi"):
    m = types.ModuleType(name)
    setattr(m, name if name == "pesq" else "stoi", lambda *a, **k: None)
    sys.modules[name] = m
pl = types.ModuleType("pytorch_lightning")
pl.LightningDataModule = object
pl.LightningModule = object
sys.modules["pytorch_lightning"] = pl
ta = types.ModuleType("torchaudio")
ta.load = lambda *a, **k: None
sys.modules["torchaudio"] = ta

import fdbm_amd  # noqa: E402
from fdbm_amd.arch import Spec, VARIANTS  # noqa: E402
from fdbm_amd.weights import fill_state_dict  # noqa: E402

torch.set_num_threads(8)
META = dict(torch=torch.__version__, threads=torch.get_num_threads(),
            mkldnn=bool(torch.backends.mkldnn.is_available()))


def save(name, **arrs):
    arrs = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    arrs["_meta"] = np.array(repr(META))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"wrote {name}.npz ({os.path.getsize(path) / 1024:.0f} KiB)")


def synth_wave(n, seed):
    g = np.random.Generator(np.random.Philox(seed))
    t = np.arange(n) / 16000.0
    sig = sum(np.sin(2 * np.pi * f0 * t * (1 + 0.02 * np.sin(2 * np.pi * 3 * t))) / (i + 1)
              for i, f0 in enumerate((180.0, 360.0, 540.0, 1200.0, 2400.0)))
    sig = sig * (0.5 + 0.5 * np.sin(2 * np.pi * 2.0 * t) ** 2) + 0.3 * g.standard_normal(n)
    sig = sig / np.max(np.abs(sig))
    return sig.astype(np.float32)


def synth_spec(B, T, seed):
    """complex64 [B,1,257,T] 'noisy spectrogram' with spectrogram-like magnitudes."""
    g = np.random.Generator(np.random.Philox(seed + 1000))
    mag = 0.15 * np.abs(g.standard_normal((B, 1, 257, T))) ** 1.5
    ph = g.uniform(-np.pi, np.pi, (B, 1, 257, T))
    return torch.from_numpy((mag * np.exp(1j * ph)).astype(np.complex64))


# ---- 1. front-end ------------------------------------------------------------
def gen_frontend():
    from fdbm.data_module import SpecsDataModule
    from fdbm.util.other import pad_spec
    for tag, kw in (("512", dict(n_fft=512, hop_length=256, window="sqrthann")),
                    ("510", dict(n_fft=510, hop_length=128, window="hann"))):
        dm = SpecsDataModule(base_dir="x", **kw)
        y = torch.from_numpy(synth_wave(16000 + 37, 7))[None]
        S = dm.stft(y)
        Sc = dm.spec_fwd(S)
        Y = Sc[None]
        out = dict(wave=y, stft=S, spec_fwd=Sc,
                   pad_zero=pad_spec(Y, "zero_pad"), pad_reflect=pad_spec(Y, "reflection"),
                   spec_back=dm.spec_back(Sc), istft=dm.istft(dm.spec_back(Sc), y.shape[-1]))
        dm.transform_type = "log"
        out["spec_fwd_log"] = dm.spec_fwd(S)
        out["spec_back_log"] = dm.spec_back(out["spec_fwd_log"])
        save(f"frontend_{tag}", **out)


# ---- 2. coefficients -----------------------------------------------------------
def gen_coeffs():
    from fdbm.bridge import Bridge
    out = {}
    for path, sched in (("sb", "bb"), ("sb", "ve"), ("sb", "vp"), ("sb", "gmax"), ("fm", "ot")):
        for N in (5, 30, 100):
            br = Bridge(path, N=N, noise_schedule=sched)
            ts = torch.linspace(br.start_time, br.end_time, N + 1)
            key = f"{path}_{sched}_N{N}"
            out[key + "_ts"] = ts
            ode = [torch.stack(br.path.sampling_param_ode_ei(ts[i], ts[i - 1], 1, "cpu")).flatten()
                   for i in range(1, N + 1)]
            out[key + "_ode_ei"] = torch.stack(ode)
            pp = [torch.stack(br.path.path_param(ts[i] * torch.ones(1))).flatten() for i in range(N + 1)]
            out[key + "_path_param"] = torch.stack(pp)
            if path == "sb":
                sde = [torch.stack(br.path.sampling_param_sde_ei(ts[i], ts[i - 1], 1, "cpu")).flatten()
                       for i in range(1, N + 1)]
                out[key + "_sde_ei"] = torch.stack(sde)
                # drift weights via the linear response of path.sde / path.ode at B=1
                one = torch.ones(1, 1, 1, 1)
                zero = torch.zeros(1, 1, 1, 1)
                rows = []
                for i in range(N):
                    tv = ts[i] * torch.ones(1)
                    wx = br.path.sde(tv, one, zero, zero)[0].flatten()
                    ws = br.path.sde(tv, zero, one, zero)[0].flatten()
                    wy = br.path.sde(tv, zero, zero, one)[0].flatten()
                    gd = torch.as_tensor(br.path.sde(tv, one, zero, zero)[1]).flatten()
                    ox = br.path.ode(tv, one, zero, zero).flatten()
                    os_ = br.path.ode(tv, zero, one, zero).flatten()
                    oy = br.path.ode(tv, zero, zero, one).flatten()
                    rows.append(torch.cat([wx, ws, wy, gd, ox, os_, oy]))
                out[key + "_sde_ode_w"] = torch.stack(rows)
    save("coeffs", **out)


# ---- 3. layer-level ops ----------------------------------------------------------
def gen_ops():
    from fdbm.backbones.ncsnpp_utils import layerspp, up_or_down_sampling as uds
    import torch.nn as nn
    g = torch.Generator().manual_seed(11)
    out = {}
    x = torch.randn(2, 8, 12, 20, generator=g)
    out["resample_x"] = x
    out["upsample"] = uds.upsample_2d(x, (1, 3, 3, 1), factor=2)
    out["downsample"] = uds.downsample_2d(x, (1, 3, 3, 1), factor=2)
    # raw upfirdn2d native with an asymmetric kernel / odd pads (the native-op boundary)
    from fdbm.backbones.ncsnpp_utils.op.upfirdn2d import upfirdn2d_native
    k = torch.tensor([[1., 2., 0.5], [0.25, -1., 3.], [2., 0.75, -0.5]])
    out["ufd_kernel"] = k
    out["ufd_up2_down1"] = upfirdn2d_native(x, k, 2, 2, 1, 1, 2, 1, 2, 1)
    out["ufd_up1_down2"] = upfirdn2d_native(x, k, 1, 1, 2, 2, 1, 1, 1, 1)
    out["ufd_up2_down3_negpad"] = upfirdn2d_native(x, k, 2, 2, 3, 3, -1, 2, -1, 2)

    act = nn.SiLU()
    temb = torch.randn(2, 64, generator=g)
    out["temb"] = temb
    cases = dict(plain=dict(in_ch=32, out_ch=32), widen=dict(in_ch=32, out_ch=64),
                 up=dict(in_ch=32, up=True), down=dict(in_ch=32, down=True),
                 cat=dict(in_ch=96, out_ch=32))
    for name, kw in cases.items():
        blk = layerspp.ResnetBlockBigGANpp(act=act, temb_dim=64, dropout=0., fir=True,
                                           fir_kernel=[1, 3, 3, 1], init_scale=0., skip_rescale=True, **kw)
        sd = blk.state_dict()
        fill = fill_state_dict({f"all_modules.9.{k}": tuple(v.shape) for k, v in sd.items()}, seed=3)
        blk.load_state_dict({k: torch.from_numpy(fill[f"all_modules.9.{k}"]) for k in sd})
        xin = torch.randn(2, kw["in_ch"], 8, 16, generator=g)
        with torch.no_grad():
            out[f"res_{name}_x"] = xin
            out[f"res_{name}_y"] = blk(xin, temb)
    attn = layerspp.AttnBlockpp(channels=32, skip_rescale=True, init_scale=0.)
    sd = attn.state_dict()
    fill = fill_state_dict({f"all_modules.9.{k}": tuple(v.shape) for k, v in sd.items()}, seed=3)
    attn.load_state_dict({k: torch.from_numpy(fill[f"all_modules.9.{k}"]) for k in sd})
    xin = torch.randn(2, 32, 16, 4, generator=g)
    with torch.no_grad():
        out["attn_x"] = xin
        out["attn_y"] = attn(xin)
    comb = layerspp.Combine(4, 32, method="sum")
    sd = comb.state_dict()
    fill = fill_state_dict({f"all_modules.9.{k}": tuple(v.shape) for k, v in sd.items()}, seed=3)
    comb.load_state_dict({k: torch.from_numpy(fill[f"all_modules.9.{k}"]) for k in sd})
    pin, hin = torch.randn(2, 4, 8, 16, generator=g), torch.randn(2, 32, 8, 16, generator=g)
    with torch.no_grad():
        out["comb_p"], out["comb_h"], out["comb_y"] = pin, hin, comb(pin, hin)
    save("ops", **out)


# ---- 4/5. backbone + samplers -------------------------------------------------------
NETS = {
    # name: (reference class name, ctor kwargs, product hyper-parameters)
    "mini64": ("NCSNpp_v2", dict(nf=64, ch_mult=(1, 1, 2, 2, 2, 2, 2), num_res_blocks=2, attn_resolutions=(16,))),
    "v2_5M": ("NCSNpp_v2_5M", None),
}


def build_ref_net(name, seed=0):
    import fdbm.backbones as bb
    cls_name, kw = NETS[name]
    if kw is None:
        net = getattr(bb, cls_name)()
        hp = VARIANTS["ncsnpp_" + name]
    else:
        net = getattr(bb, cls_name)(**kw)
        hp = kw
    spec = Spec(**hp)
    shapes = spec.param_shapes()
    sd = net.state_dict()
    assert set(sd) == set(shapes), (set(sd) ^ set(shapes))
    for k in sd:
        assert tuple(sd[k].shape) == tuple(shapes[k]), k
    fill = fill_state_dict(shapes, seed=seed)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in fill.items()})
    return net.eval(), hp


def gen_backbone():
    for name, B, T in (("mini64", 2, 64), ("v2_5M", 1, 64)):
        net, hp = build_ref_net(name)
        x, y = synth_spec(B, T, 1), synth_spec(B, T, 2)
        t = torch.tensor([0.7, 0.031][:B])
        with torch.no_grad():
            out = net(x, y, t)
        save(f"backbone_{name}", x=x, y=y, t=t, out=out)


def gen_samplers():
    from fdbm.bridge import Bridge
    net, hp = build_ref_net("v2_5M")
    y = synth_spec(1, 64, 5)
    out = dict(y=y)
    runs = [("sb_bb_ode_ei_N5", dict(path="sb", noise_schedule="bb", N=5, sampler_type="ode_ei"), {}),
            ("fm_ot_ode_ei_N5", dict(path="fm", noise_schedule="ot", N=5, sampler_type="ode_ei"), {}),
            ("sb_bb_sde_ei_N5", dict(path="sb", noise_schedule="bb", N=5, sampler_type="sde_ei"), {}),
            ("sb_ve_sde_ei_N4", dict(path="sb", noise_schedule="ve", N=4, sampler_type="sde_ei"), {}),
            ("sb_bb_pc_N4", dict(path="sb", noise_schedule="bb", N=4, sampler_type="pc"),
             dict(predictor_name="euler_maruyama", corrector_name="ald", corrector_steps=1, snr=0.5, denoise=False)),
            ("sb_vp_pc_N3", dict(path="sb", noise_schedule="vp", N=3, sampler_type="pc"),
             dict(predictor_name="euler_maruyama", corrector_name="langevin", corrector_steps=1, snr=0.3, denoise=True)),
            ("sb_bb_ode_int", dict(path="sb", noise_schedule="bb", N=5, sampler_type="ode_int"),
             dict(rtol=1e-2, atol=1e-2)),
            ("fm_ot_ode_int", dict(path="fm", noise_schedule="ot", N=5, sampler_type="ode_int"),
             dict(rtol=1e-2, atol=1e-2)),
            ]
    for tag, bkw, skw in runs:
        br = Bridge(**bkw)
        torch.manual_seed(1234)
        with torch.no_grad():
            out[tag] = br.sampler(net, y, **skw)
        print(" sampler", tag, "done")
    # batched ode_ei on the mini net (B=2)
    net2, _ = build_ref_net("mini64")
    y2 = synth_spec(2, 64, 6)
    br = Bridge("sb", N=3, sampler_type="ode_ei")
    torch.manual_seed(99)
    with torch.no_grad():
        out["mini64_y"] = y2
        out["mini64_sb_bb_ode_ei_N3"] = br.sampler(net2, y2)
    # oracle self-noise: same run on 1 thread
    torch.set_num_threads(1)
    br = Bridge("sb", N=5, sampler_type="ode_ei")
    torch.manual_seed(1234)
    with torch.no_grad():
        out["sb_bb_ode_ei_N5_1thread"] = br.sampler(net, y)
    torch.set_num_threads(8)
    save("samplers", **out)


def gen_full(n_steps):
    """Full-size ncsnpp_v2 at [1,1,257,256]: one forward and the N-step ode_ei result."""
    import fdbm.backbones as bb
    from fdbm.bridge import Bridge
    import time
    spec = Spec(**VARIANTS["ncsnpp_v2"])
    net = bb.NCSNpp_v2()
    fill = fill_state_dict(spec.param_shapes(), seed=0)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in fill.items()})
    net.eval()
    y = synth_spec(1, 256, 21)
    x = synth_spec(1, 256, 22)
    t0 = time.time()
    with torch.no_grad():
        fwd = net(x, y, torch.tensor([0.5]))
    print(f" full forward {time.time() - t0:.1f}s")
    out = dict(x=x, y=y, t=torch.tensor([0.5]), fwd=fwd)
    for path, sched in (("sb", "bb"), ("fm", "ot")):
        br = Bridge(path, N=n_steps, noise_schedule=sched, sampler_type="ode_ei")
        torch.manual_seed(4321)
        t0 = time.time()
        with torch.no_grad():
            out[f"{path}_{sched}_ode_ei_N{n_steps}"] = br.sampler(net, y)
        print(f" full {path} N={n_steps} {time.time() - t0:.1f}s")
    save("full_ncsnpp_v2", **out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    ap.add_argument("--full", action="store_true", help="also the full-size ncsnpp_v2 N=30 run (minutes)")
    ap.add_argument("--full_steps", type=int, default=30)
    a = ap.parse_args()
    todo = a.only or ["frontend", "coeffs", "ops", "backbone", "samplers"]
    with torch.no_grad():
        for name in todo:
            globals()["gen_" + name]()
    if a.full:
        gen_full(a.full_steps)
