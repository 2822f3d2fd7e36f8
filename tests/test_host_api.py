"""Host-side logic on CPU: registries, bridge coefficient tables (bit-exact with the reference's,
tests/golden/coeffs.npz), sampler drivers with a plain callable model (vs the oracle), error
behaviour, architecture bookkeeping, and that the C-ABI library loads and exports every symbol the
header declares (no compute calls here: no GPU)."""
import ctypes
import math
import os
import re

import numpy as np
import pytest
import torch

import fdbm_amd
from fdbm_amd import hip
from fdbm_amd.arch import Spec, VARIANTS
from fdbm_amd.program import count_macs
from fdbm_amd.registry import Registry
from oracle import sampler as osamp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_registry_behaviour():
    r = Registry("Thing")

    @r.register("a")
    class A:
        pass

    assert r.get_by_name("a") is A and r.get_all_names() == ["a"]
    with pytest.warns(UserWarning):
        @r.register("a")
        class A2:
            pass
    assert r.get_by_name("a") is A2
    with pytest.raises(ValueError):
        r.get_by_name("missing")
    assert set(fdbm_amd.BridgeRegistry.get_all_names()) == {"sb", "fm"}
    from fdbm_amd.tfgridnet import VARIANTS as TFG_VARIANTS
    assert set(fdbm_amd.BackboneRegistry.get_all_names()) == set(VARIANTS) | set(TFG_VARIANTS)
    assert {"euler_maruyama", "none"} <= set(fdbm_amd.PredictorRegistry.get_all_names())
    assert {"langevin", "ald", "none"} <= set(fdbm_amd.CorrectorRegistry.get_all_names())


@pytest.mark.parametrize("path,sched", [("sb", "bb"), ("sb", "ve"), ("sb", "vp"), ("sb", "gmax"), ("fm", "ot")])
def test_coefficient_tables_bit_exact(golden, path, sched):
    g = golden("coeffs")
    for N in (5, 30, 100):
        br = fdbm_amd.Bridge(path, N=N, noise_schedule=sched)
        key = f"{path}_{sched}_N{N}"
        assert np.array_equal(br.time_grid(N + 1).numpy(), g[key + "_ts"])
        tab, t_model = br.ei_weight_table("ode", 1)
        assert np.array_equal(tab[:, :, 0].numpy(), g[key + "_ode_ei"]), key
        assert np.array_equal(t_model.numpy(), g[key + "_ts"][:-1])
        if path == "sb":
            tab, _ = br.ei_weight_table("sde", 1)
            ref = g[key + "_sde_ei"].copy()
            ref[-1, 2] = 0.0                                  # last step: w_z = 0 (bridge.py:105-106)
            assert np.array_equal(tab[:, :, 0].numpy(), ref), key
            ts = torch.from_numpy(g[key + "_ts"])
            w = torch.stack([torch.cat(list(br.path.sde_weights(ts[i] * torch.ones(1))) +
                                       list(br.path.ode_weights(ts[i] * torch.ones(1)))) for i in range(N)])
            assert np.array_equal(w.numpy(), g[key + "_sde_ode_w"]), key
        pp = torch.stack([torch.stack(br.path.path_param(torch.from_numpy(g[key + "_ts"])[i] * torch.ones(1))).flatten()
                          for i in range(N + 1)])
        assert np.array_equal(pp.numpy(), g[key + "_path_param"])


def _toy_model(xt, y, t):
    # any callable (xt, y, t) -> s: a cheap smooth map so the drivers can be checked on CPU
    tt = t.to(xt.device)[:, None, None, None]
    return 0.6 * y + 0.3 * xt * torch.cos(tt) + 0.05 * torch.roll(xt, 1, dims=-1)


@pytest.mark.parametrize("kind,path,pkw", [("ode_ei", "sb", dict(noise_schedule="bb")), ("ode_ei", "fm", {}),
                                           ("sde_ei", "sb", dict(noise_schedule="ve")),
                                           ("pc", "sb", dict(noise_schedule="bb"))])
def test_sampler_drivers_match_oracle_on_cpu(kind, path, pkw):
    g = torch.Generator().manual_seed(3)
    y = torch.view_as_complex(torch.randn(2 if kind != "pc" else 1, 1, 9, 8, 2, generator=g))
    br = fdbm_amd.Bridge(path, N=6, sampler_type=kind, **pkw)
    skw = dict(predictor_name="euler_maruyama", corrector_name="ald", snr=0.4, denoise=False) if kind == "pc" else {}
    out = br.sampler(_toy_model, y, generator=torch.Generator().manual_seed(5), **skw)
    smp = osamp.Sampler(path, N=6, **pkw)
    okw = dict(corrector="ald", snr=0.4, denoise=False) if kind == "pc" else {}
    ref = getattr(smp, kind)(_toy_model, y, torch.Generator().manual_seed(5), **okw)
    assert (out - ref).abs().max() < 1e-5
    # injected noise wins over the generator; default noise is drawn on the tensor's device
    z0 = torch.zeros_like(y)
    a = br.sampler(_toy_model, y, prior_noise=z0, step_noise=lambda i: z0, **skw)
    b = br.sampler(_toy_model, y, prior_noise=z0, step_noise=[z0] * 64, **skw)
    assert torch.equal(torch.view_as_real(a), torch.view_as_real(b))


def test_reference_defects_mirrored():
    y = torch.zeros(1, 1, 5, 4, dtype=torch.complex64)
    br = fdbm_amd.Bridge("sb", N=3, sampler_type="pc")
    with pytest.raises(ValueError):                    # default predictor 'reverse_diffusion' is unregistered
        br.sampler(_toy_model, y)
    br = fdbm_amd.Bridge("fm", N=3, sampler_type="sde_ei")
    with pytest.raises(NotImplementedError):           # the fm path defines no SDE
        br.sampler(_toy_model, y)
    br = fdbm_amd.Bridge("sb", N=3, sampler_type="nope")
    with pytest.raises(ValueError):
        br.sampler(_toy_model, y)
    assert fdbm_amd.Bridge("sb", T=0.5).path.T == 1.0  # T is dropped by the sb path, as in the reference
    x = br.prior_sampling(torch.ones(2, 1, 3, 3, dtype=torch.complex64))
    assert torch.equal(torch.view_as_real(x), torch.view_as_real(torch.ones(2, 1, 3, 3, dtype=torch.complex64)))


def test_architecture_bookkeeping():
    counts = {"ncsnpp_v2": 65_590_822, "ncsnpp_v2_5M": 5_216_762, "ncsnpp_v2_16M": 16_241_190, "ncsnpp_v2_37M": 36_926_630}
    for name, kw in VARIANTS.items():
        assert Spec(**kw).num_params() == counts[name]
    spec = Spec(**VARIANTS["ncsnpp_v2"])
    assert len(spec.mods) == 77 and len(spec.param_shapes()) == 647
    assert count_macs(spec, 256, 256) == 266_073_636_864
    assert count_macs(spec, 256, 512) > 2 * count_macs(spec, 256, 256) - 10**9


def test_complex_randn_matches_torch():
    a = fdbm_amd.complex_randn((2, 1, 5, 7), torch.Generator().manual_seed(11))
    torch.manual_seed(11)
    b = torch.randn_like(torch.zeros(2, 1, 5, 7, dtype=torch.complex64))
    assert torch.equal(torch.view_as_real(a), torch.view_as_real(b))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(REPO, "include", "fdbm_hip.h")).read()
    declared = set(re.findall(r"\b(fdbm_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"fdbm_conv_seg", "fdbm_conv_args", "fdbm_op"}
    assert len(declared) >= 25
    L = ctypes.CDLL(hip.LIB_PATH)
    missing = [n for n in sorted(declared) if not hasattr(L, n)]
    assert not missing, missing
    assert set(hip.EXPORTS) <= declared | {"fdbm_last_error"}
    assert hip.lib().fdbm_version() >= 1


def test_no_cpu_fallback():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        fdbm_amd.BackboneRegistry.get_by_name("ncsnpp_v2_5M")()
    from fdbm_amd.frontend import SpecFrontend
    with pytest.raises(RuntimeError):
        SpecFrontend()


# ---- checkpoints (the reference's data format in front of the hot path) -------------------------------
def test_param_order_matches_reference_fixture():
    """tests/golden/param_order.json = named_parameters() order of the reference backbones
    (tests/golden/make_param_order.py): torch_ema's shadow_params follow it."""
    import json, os
    from fdbm_amd.arch import Spec, VARIANTS
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "param_order.json")))
    for name, keys in ref.items():
        assert Spec(**VARIANTS[name]).param_order() == keys, name


def test_lightning_checkpoint_import(tmp_path):
    import torch
    from fdbm_amd.arch import Spec, VARIANTS
    from fdbm_amd.checkpoint import load_lightning_checkpoint
    from fdbm_amd.weights import fill_state_dict
    name = "ncsnpp_v2_5M"
    spec = Spec(**VARIANTS[name])
    raw = {k: torch.from_numpy(v) for k, v in fill_state_dict(spec.param_shapes(), seed=1).items()}
    ema = {k: torch.from_numpy(v) for k, v in fill_state_dict(spec.param_shapes(), seed=2).items()}
    ckpt = {
        "state_dict": {"dnn." + k: v for k, v in raw.items()},
        "hyper_parameters": {"backbone": name, "bridge": "sb", "noise_schedule": "bb", "n_fft": 512, "hop_length": 256},
        "ema": {"decay": 0.999, "num_updates": 10, "shadow_params": [ema[k] for k in spec.param_order()], "collected_params": None},
    }
    path = tmp_path / "epoch=1.ckpt"
    torch.save(ckpt, path)
    hp, state = load_lightning_checkpoint(str(path))
    assert hp["backbone"] == name and hp["bridge"] == "sb"
    assert set(state) == set(raw) and all(torch.equal(state[k], ema[k]) for k in raw)          # EMA weights by default
    _, state_raw = load_lightning_checkpoint(str(path), use_ema=False)
    assert all(torch.equal(state_raw[k], raw[k]) for k in raw)
    bad = dict(ckpt, ema={"shadow_params": ckpt["ema"]["shadow_params"][:-2]})
    with pytest.raises(ValueError):
        load_lightning_checkpoint(bad)
    with pytest.raises(KeyError):
        load_lightning_checkpoint({"hyper_parameters": {}})
    # the other torch_ema layout: shadow copies of the TRAINABLE parameters only (the fixed Fourier frequencies
    # all_modules.0.W, requires_grad=False, are then taken from the state_dict) - same rule for both backbone families
    trainable = dict(ckpt, ema={"shadow_params": [ema[k] for k in spec.param_order() if k != "all_modules.0.W"]})
    _, st = load_lightning_checkpoint(trainable)
    assert torch.equal(st["all_modules.0.W"], raw["all_modules.0.W"])
    assert all(torch.equal(st[k], ema[k]) for k in raw if k != "all_modules.0.W")
    swapped = dict(ckpt, ema={"shadow_params": list(reversed(ckpt["ema"]["shadow_params"]))})
    with pytest.raises(ValueError):                    # right count, wrong order: caught by the shape check
        load_lightning_checkpoint(swapped)


def test_lightning_checkpoint_import_tfgridnet_both_ema_layouts():
    """TF-GridNet checkpoints under the same rule: shadow_params of all parameters (incl. the frozen
    get_time_emb.W) or of the trainable ones only."""
    import torch
    from fdbm_amd import tfgridnet as tg
    from fdbm_amd.checkpoint import load_lightning_checkpoint
    name = "tfgridnet_4l32c80"
    shapes = tg.param_shapes(**tg.VARIANTS[name])
    raw = {k: torch.from_numpy(np.asarray(v)) for k, v in tg.fill_state(shapes, seed=5).items()}
    ema = {k: torch.from_numpy(np.asarray(v)) for k, v in tg.fill_state(shapes, seed=6).items()}
    base = {"state_dict": {"dnn." + k: v for k, v in raw.items()}, "hyper_parameters": {"backbone": name}}
    _, st = load_lightning_checkpoint(dict(base, ema={"shadow_params": [ema[k] for k in shapes]}))
    assert all(torch.equal(st[k], ema[k]) for k in shapes)
    _, st = load_lightning_checkpoint(dict(base, ema={"shadow_params": [ema[k] for k in shapes if k != "get_time_emb.W"]}))
    assert torch.equal(st["get_time_emb.W"], raw["get_time_emb.W"])
    assert all(torch.equal(st[k], ema[k]) for k in shapes if k != "get_time_emb.W")
    with pytest.raises(ValueError):
        load_lightning_checkpoint(dict(base, ema={"shadow_params": [ema[k] for k in list(shapes)[:-3]]}))


def test_infer_driver_file_side(tmp_path):
    """fdbm_amd.infer without a GPU: file discovery order, WAV decoding of the PCM formats, resampling length,
    output paths (infer_folder.py:58-65, 124-131)."""
    import argparse
    from scipy.io import wavfile
    from fdbm_amd import infer
    (tmp_path / "d" / "e").mkdir(parents=True)
    x16 = (np.arange(800) % 100 * 300 - 15000).astype(np.int16)
    wavfile.write(tmp_path / "d" / "b.wav", 8000, x16)
    wavfile.write(tmp_path / "d" / "e" / "a.wav", 16000, np.stack([x16, -x16], 1))             # stereo
    wavfile.write(tmp_path / "d" / "c.wav", 16000, (x16 / 32768.0).astype(np.float32))
    files = infer.get_audio_files(str(tmp_path / "d"))
    assert [os.path.relpath(f, tmp_path / "d") for f in files] == ["b.wav", "c.wav", os.path.join("e", "a.wav")]
    y, sr = infer.read_wav(files[0])
    assert sr == 8000 and y.shape == (1, 800) and y.dtype == np.float32 and abs(y[0, 0] + 15000 / 32768) < 1e-6
    assert infer.resample_to(y, sr).shape == (1, 1600)
    y2, _ = infer.read_wav(files[2])
    assert y2.shape == (2, 800) and np.allclose(y2[0], -y2[1])
    yf, _ = infer.read_wav(files[1])
    assert np.allclose(yf[0], y[0], atol=1e-6)
    args = argparse.Namespace(test_dir=str(tmp_path / "d"), enhanced_dir="/out", keep_structure=True)
    assert infer.output_path(files[2], args) == os.path.join("/out", "e", "a.wav")
    args.keep_structure = False
    assert infer.output_path(files[2], args) == os.path.join("/out", "a.wav")
    ns = infer.build_parser().parse_args(["--test_dir", "a", "--enhanced_dir", "b", "--ckpt", "c", "-D", "0", "1",
                                          "--sampler_kwargs", "{'corrector': 'ald'}"])
    assert ns.device == ["0", "1"] and ns.N == 30 and ns.sampler_type == "ode_ei" and ns.sampler_kwargs == {"corrector": "ald"}


def test_infer_driver_config_files(tmp_path):
    """-C config.yaml: ${} interpolation, booleans as flags, config wins over the command line
    (infer_folder.py:41-55 appends the config to sys.argv), unknown keys ignored."""
    from fdbm_amd import infer
    cfg = tmp_path / "config_infer_folder.yaml"
    cfg.write_text("test_dir: /data/noisy\nenhanced_dir: ${log_dir}/enhanced/${checkpoint}_sampler=${sampler_type}_N=${N}\n"
                   "version: v7\ncheckpoint: last\nexp_dir: ./logs\nlog_dir: ${exp_dir}/${version}\n"
                   "ckpt: ${log_dir}/checkpoints/${checkpoint}.ckpt\nN: 5\nsampler_type: ode_ei\n"
                   "sampler_kwargs:\n  corrector_name: ald\n  snr: 0.5\nkeep_structure: true\nnothing: null\n")
    c = infer.load_config(str(cfg))
    assert c["ckpt"] == "./logs/v7/checkpoints/last.ckpt"
    assert c["enhanced_dir"] == "./logs/v7/enhanced/last_sampler=ode_ei_N=5"
    args = infer.parse_args(["-C", str(cfg), "--N", "30", "--ckpt", "x.ckpt", "-D", "0", "1"])
    assert args.N == 5 and args.ckpt == "./logs/v7/checkpoints/last.ckpt"      # the config wins
    assert args.keep_structure and args.sampler_kwargs == {"corrector_name": "ald", "snr": 0.5}
    assert args.test_dir == "/data/noisy" and args.device == ["0", "1"]
    single = infer.parse_args(["--ckpt", "m.ckpt", "--noisy_file", "a.wav"])
    assert single.noisy_file == "a.wav" and single.output_file is None and single.N == 30
    with pytest.raises(SystemExit):
        infer.parse_args(["--ckpt", "m.ckpt"])


def test_rk45_restatement_matches_scipy():
    """fdbm_amd/odeint.py restates SciPy's RK45 (tableau, initial step, error norm, step control) so that the state can
    stay on the device; on CPU tensors it must take the same steps as scipy.integrate.solve_ivp on the same problem."""
    import numpy as np
    import torch
    from scipy import integrate
    from fdbm_amd.odeint import rk45
    rng = np.random.default_rng(3)
    n = 257
    lam = (-0.5 + 2.0j) * (1.0 + rng.random(n))
    drive = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    y0 = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)

    def f_np(t, y):
        return lam * y + drive * np.cos(3.0 * t) + 0.1 * np.roll(y, 1) ** 2

    lam_t, drive_t = torch.from_numpy(lam), torch.from_numpy(drive)

    def f_t(t, y):
        return lam_t * y + drive_t * math.cos(3.0 * t) + 0.1 * torch.roll(y, 1) ** 2

    for (t0, t1, rtol, atol) in ((1.0, 0.03, 1e-5, 1e-5), (0.0, 2.0, 1e-3, 1e-6), (1.0, 1e-4, 1e-7, 1e-9)):
        sol = integrate.solve_ivp(f_np, (t0, t1), y0, rtol=rtol, atol=atol, method="RK45")
        y, st = rk45(f_t, t0, t1, torch.from_numpy(y0), rtol=rtol, atol=atol)
        assert sol.status == 0
        assert st["nfev"] == sol.nfev and st["steps"] == sol.t.size - 1, (st, sol.nfev, sol.t.size)
        err = np.abs(y.numpy() - sol.y[:, -1]).max()
        assert err < 1e-9 * max(1.0, np.abs(sol.y[:, -1]).max()), err


def test_shard_by_length_balances_and_covers():
    """Length-balanced sharding of the batched folder driver (fdbm_amd.dist.shard_by_length): every index once, shards in
    descending length order, totals within one longest utterance of each other, deterministic."""
    from fdbm_amd.dist import shard_by_length
    rng = np.random.default_rng(0)
    lengths = rng.integers(8000, 160000, size=203).tolist()
    for w in (1, 2, 3, 8):
        sh = shard_by_length(lengths, w)
        assert sorted(i for s_ in sh for i in s_) == list(range(len(lengths)))
        tot = [sum(lengths[i] for i in s_) for s_ in sh]
        assert max(tot) - min(tot) <= max(lengths)
        for s_ in sh:
            ls = [lengths[i] for i in s_]
            assert ls == sorted(ls, reverse=True)
        assert sh == shard_by_length(lengths, w)
    assert shard_by_length([], 4) == [[], [], [], []]


def _flac_cases():
    rng = np.random.default_rng(11)
    t = np.arange(9000)
    tone = (6000 * np.sin(2 * np.pi * 220 * t / 16000) + 500 * rng.standard_normal(t.size)).astype(np.int64)
    stereo = np.stack([tone, (0.6 * tone + 300 * rng.standard_normal(t.size)).astype(np.int64)])
    fixed = [(4096, ("fixed", o, p, e), "indep") for o, p, e in ((0, 0, False), (1, 2, False), (2, 3, True))] + \
            [(1152, ("fixed", 3, 1, False), "indep"), (192, ("fixed", 4, 0, False), "indep"), (300, "verbatim", "indep")]
    yield "mono16_fixed", tone[None], 16000, 16, fixed, {}
    yield "mono16_lpc_variable", tone[None], 16000, 16, [(4096, ("lpc", 8, 12, 9, 2), "indep"), (4096, ("lpc", 1, 5, 3, 0), "indep"),
                                                          (808, ("lpc", 32, 14, 10, 0), "indep")], dict(variable=True, rate_in_header="hz")
    for st in ("ls", "rs", "ms"):
        yield f"stereo16_{st}", stereo, 44100, 16, [(4096, ("fixed", 2, 2, False), st), (4096, ("lpc", 4, 10, 8, 1), st),
                                                     (808, "verbatim", st)], {}
    big = (tone * 200 + rng.integers(-100, 100, tone.size))                       # 24-bit range
    yield "mono24_streaminfo_params", big[None], 48000, 24, [(2304, ("fixed", 2, 0, False), "indep")] * 3 + [(2088, ("lpc", 6, 15, 12, 0), "indep")], \
        dict(rate_in_header="none", bps_in_header=False)
    yield "mono8_khz", (tone // 64)[None], 32000, 8, [(576, ("fixed", 1, 0, False), "indep")] * 15 + [(360, "verbatim", "indep")], dict(rate_in_header="khz")
    const = np.concatenate([np.full(256, 77), np.zeros(256, dtype=np.int64), (tone[:512] // 8) * 8])      # constant blocks, wasted bits
    yield "mono16_constant_wasted", const[None], 22050, 16, [(256, "constant", "indep"), (256, "constant", "indep"),
                                                              (512, ("fixed", 2, 1, False), "indep")], dict(rate_in_header="tens")


@pytest.mark.parametrize("case", list(_flac_cases()), ids=lambda c: c[0])
def test_flac_decoder(case):
    """fdbm_amd/flac.py against the test encoder (tests/flac_encode.py, written from the same specification): every
    subframe type, fixed orders 0-4, LPC orders 1-32, Rice partitions incl. escaped ones, wasted bits, the three stereo
    decorrelations, fixed and variable blocking, every way a frame header can carry block size / sample rate / sample size -
    bit-exact samples, both CRCs and the MD5 signature verified by the decoder."""
    from fdbm_amd import flac
    from flac_encode import encode
    name, x, rate, bps, blocks, kw = case
    data = encode(x, rate, bps, blocks, **kw)
    y, r, b = flac.decode(data)
    assert r == rate and b == bps and y.shape == x.shape and np.array_equal(y, x), name
    # integrity checks bite: a flipped payload bit is caught by the frame CRC-16, a flipped header bit by the CRC-8
    bad = bytearray(data)
    bad[-5] ^= 0x10
    with pytest.raises(flac.FlacError):
        flac.decode(bytes(bad))
    first_frame = data.index(b"\xff\xf8" if not kw.get("variable") else b"\xff\xf9", 4 + 4 + 34 + 12)
    bad = bytearray(data)
    bad[first_frame + 2] ^= 0x01
    with pytest.raises(flac.FlacError):
        flac.decode(bytes(bad))


def test_flac_in_the_folder_driver(tmp_path):
    """infer.read_audio takes .flac next to .wav (infer_folder.py:58-65,94)."""
    from fdbm_amd import infer
    from flac_encode import encode
    rng = np.random.default_rng(3)
    x = (8000 * np.sin(np.arange(5000) / 9.0) + 100 * rng.standard_normal(5000)).astype(np.int64)
    p = tmp_path / "a.flac"
    p.write_bytes(encode(x[None], 16000, 16, [(4096, ("fixed", 2, 2, False), "indep"), (904, ("lpc", 3, 9, 7, 0), "indep")]))
    y, sr = infer.read_audio(str(p))
    assert sr == 16000 and y.shape == (1, 5000) and y.dtype == np.float32
    assert np.array_equal(y[0], (x / 32768.0).astype(np.float32))
    assert str(p) in infer.get_audio_files(str(tmp_path))


def test_philox_restatement_known_answers():
    """oracle/rng.py (the numpy restatement the device generator is tested against) reproduces Random123's published
    known-answer vectors for philox4x32-10, and its normals have the stated distribution."""
    from oracle import rng
    for counter, key, out in rng.KAT:
        assert tuple(int(v) for v in rng.philox4x32_10(*counter, *key)) == out
    z = rng.complex_normal(200000, 1, 12345)
    assert abs(z.real.mean()) < 5e-3 and abs(z.real.var() - 0.5) < 5e-3 and abs(z.imag.var() - 0.5) < 5e-3
    assert np.array_equal(rng.complex_normal(16, 1, 12345, first=100), z[100:116])       # element-addressable


def test_ode_int_with_oracle_network_is_the_reference(golden):
    """`Bridge.ode_sampler_int` (the SciPy route, host tensors) driven by the CPU oracle network reproduces the REFERENCE's
    own run of fdbm/bridge.py:115-140 (fixture ode_int_5M: ncsnpp_v2_5M, contractive filler, sb/bb, rtol = atol = 1e-3)
    bit for bit, with the same number of network evaluations: the flow, the prior and the solver call are the reference's.
    (What the GPU test can hold for this ill-conditioned integration is then a matter of the network's rounding alone.)"""
    import torch
    import fdbm_amd
    from fdbm_amd.arch import Spec, VARIANTS
    from fdbm_amd.weights import fill_state_dict
    from oracle import ncsnpp as onet
    g = golden("ode_int_5M")
    hp = VARIANTS["ncsnpp_v2_5M"]
    model = onet.Model(fill_state_dict(Spec(**hp).param_shapes(), seed=0, profile="contractive"), hp)
    y = torch.from_numpy(g["y"])
    torch.set_num_threads(8)
    br = fdbm_amd.Bridge("sb", N=5, sampler_type="ode_int")
    out = br.sampler(model, y, generator=torch.Generator().manual_seed(11), rtol=1e-3, atol=1e-3)
    assert br.last_ode_stats["nfev"] == int(g["sb_nfev"])
    assert torch.equal(torch.view_as_real(out), torch.view_as_real(torch.from_numpy(g["sb_ode_int"])))
