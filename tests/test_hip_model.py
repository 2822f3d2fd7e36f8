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


def net(name, dtype=torch.float32, fused=None, split=False):
    key = (name, dtype, fused, split)
    if key not in _CACHE:
        hp = MINI64 if name == "mini64" else VARIANTS[name]
        _CACHE[key] = HipNCSNpp(dtype=dtype, device=DEV, fused=fused, split=split, **hp)
    return _CACHE[key]


@pytest.fixture(params=["f32", "f32s"])
def split(request):
    """f32 = exact f32 MFMA; f32s = f32 tensors, split-precision matrix products (fdbm_conv_args.mma_mode 1): same bars."""
    return request.param == "f32s"


@pytest.mark.parametrize("name,fix", [("mini64", "backbone_mini64"), ("ncsnpp_v2_5M", "backbone_v2_5M")])
def test_backbone_fp32_vs_reference(golden, name, fix, split):
    g = golden(fix)
    out = net(name, split=split)(T(g["x"]).to(DEV), T(g["y"]).to(DEV), T(g["t"]).to(DEV)).cpu()
    ref = T(g["out"])
    assert out.shape == ref.shape
    err = (out - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err < 4e-5, (err, scale)      # ABSOLUTE (|s| up to 10): fp32 MFMA vs oneDNN summation order, measured 1.1-1.5e-5
    assert torch.all(out[:, :, 256] == 0)                  # Nyquist row re-appended as zeros


@pytest.mark.parametrize("name,fix", [("mini64", "backbone_mini64"), ("ncsnpp_v2_5M", "backbone_v2_5M")])
def test_backbone_fp32_unfused_program(golden, name, fix):
    """fused=False: explicit GroupNorm statistics / normalise / Combine passes (the comparison program).  Same
    function and the same bar as the default program, whose statistics come from producer epilogues in fp64."""
    g = golden(fix)
    m = net(name, torch.float32, fused=False)
    out = m(T(g["x"]).to(DEV), T(g["y"]).to(DEV), T(g["t"]).to(DEV)).cpu()
    ref = T(g["out"])
    err = (out - ref).abs().max().item()
    assert err < 4e-5, err
    dims = [int(v) for v in (g["x"].shape[0], g["x"].shape[2], g["x"].shape[3])]
    assert m.program(*dims).n_slots == 0 and net(name).program(*dims).n_slots > 0
    assert m.program(*dims).n_ops > net(name).program(*dims).n_ops


@pytest.mark.parametrize("name,fix", [("mini64", "backbone_mini64"), ("ncsnpp_v2_5M", "backbone_v2_5M")])
def test_backbone_bf16_close(golden, name, fix):
    g = golden(fix)
    out = net(name, torch.bfloat16)(T(g["x"]).to(DEV), T(g["y"]).to(DEV), T(g["t"]).to(DEV)).cpu()
    ref = T(g["out"])
    rel = ((out - ref).abs().pow(2).sum() / ref.abs().pow(2).sum()).sqrt().item()
    assert rel < 3e-2, rel                                  # bf16 storage, fp32 accumulate


@pytest.mark.parametrize("name,Tn,B", [("ncsnpp_v2_16M", 64, 2), ("ncsnpp_v2_37M", 64, 1), ("ncsnpp_v2_5M", 320, 1),
                                       ("ncsnpp_v2", 128, 1), ("ncsnpp_v2", 320, 1)])
def test_registered_variants_bf16_vs_fp32(name, Tn, B):
    """Every registered backbone, also on widths whose deeper levels do not tile by 16 (T = 320: the maps
    fall back to the tap-outer kernel and to explicit statistics passes): the bf16 program against the
    un-fused f32 program of the same weights."""
    g = torch.Generator().manual_seed(7)
    x = torch.view_as_complex(torch.randn(B, 1, 257, Tn, 2, generator=g)).to(DEV)
    y = torch.view_as_complex(torch.randn(B, 1, 257, Tn, 2, generator=g)).to(DEV)
    t = torch.full((B,), 0.37)
    ref = net(name, torch.float32, fused=False)(x, y, t.to(DEV)).cpu()
    out = net(name, torch.bfloat16)(x, y, t.to(DEV)).cpu()
    assert torch.isfinite(out.real).all() and torch.isfinite(out.imag).all()
    rel = ((out - ref).abs().pow(2).sum() / ref.abs().pow(2).sum()).sqrt().item()
    assert rel < 4e-2, (name, Tn, rel)


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


# Direct tolerance vs the reference's fp32 result, per case.  1e-4 (the north-star bar) wherever two
# fp32 evaluations can agree that well; the two ODE cases carry the reference's OWN fp32 error
# (distance to the fp64 arbiter: 7.3e-4 and 1.3e-4, tests/golden/reference_spread.json): the sb
# first step computes 3999.45*y - 3998.65*y in fp32, so a 1e-7 difference in the network output
# flips roundings on a 1.2e-4 .. 9.8e-4 grid (hard part 2).
DIRECT_TOL = {"sb_bb_ode_ei_N5": 1.5e-3, "fm_ot_ode_ei_N5": 2e-4, "sb_bb_sde_ei_N5": 1e-4, "sb_ve_sde_ei_N4": 1e-4,
              "sb_bb_pc_N4": 3e-4,
              # Langevin corrector at t = 1 (sb): sigma = 0 makes the score exactly 0, so the reference's step size
              # is (snr*|z|/1e-8)^2 and its output is ~8e9: mirrored, compared RELATIVE to that magnitude
              "sb_vp_pc_N3": None}       # pc: 2N network evaluations + N*2 noise injections


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("tag,bkw,skw", SAMPLERS, ids=[s[0] for s in SAMPLERS])
def test_samplers_vs_reference(golden, tag, bkw, skw, use_graph, split):
    """Identical (noisy_spec, seed, N) -> final complex spectrogram.
    (1) direct: max-abs vs the reference within DIRECT_TOL;
    (2) arbiter: the HIP result is no farther from the fp64 trajectory than the reference is
        (x1.25 + 2e-5), i.e. it is as accurate an fp32 evaluation as the reference itself."""
    g = golden("samplers")
    y = T(g["y"]).to(DEV)
    br = fdbm_amd.Bridge(**bkw)
    gen = torch.Generator().manual_seed(1234)               # the reference ran torch.manual_seed(1234) on CPU
    out = br.sampler(net("ncsnpp_v2_5M", split=split), y, generator=gen, use_graph=use_graph, **skw).cpu()
    ref = T(g[tag])
    err = (out - ref).abs().max().item()
    tol = DIRECT_TOL[tag] if DIRECT_TOL[tag] is not None else 1e-5 * ref.abs().max().item()
    assert err < tol, (tag, err)
    arb = golden("fp64_arbiter")
    if tag in arb:
        a = T(arb[tag])
        e_hip, e_ref = (out - a).abs().max().item(), (ref - a).abs().max().item()
        assert e_hip < 1.25 * e_ref + 2e-5, (tag, e_hip, e_ref)


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
    # sb, N=3: the first step multiplies y by +-6666 in fp32 (rounding grid up to 2e-3), see DIRECT_TOL
    assert (a.cpu() - T(g["mini64_sb_bb_ode_ei_N3"])).abs().max() < 4e-3


def test_time_embedding_rows_hoisted_out_of_the_step_loop(golden):
    """Program.dense_table (the sampler graphs' batched time-embedding pass) gives, bit for bit, the rows a forward
    computes for itself, and run_body() on those rows equals run()."""
    g = golden("backbone_mini64")
    x, y, t = T(g["x"]).to(DEV), T(g["y"]).to(DEV), T(g["t"]).to(DEV)
    m = net("mini64")
    B, _, F, Tn = x.shape
    prog = m.program(B, F, Tn)
    ref = m(x, y, t).clone()                                  # full forward: leaves its Dense_0 rows in prog.dense_out
    R = m.dense_rows
    dense_out = prog.dense_out.view(torch.float32)            # (pool buffers are raw bytes)
    rows_fwd = dense_out[: B * R].clone().reshape(B, R)
    ts = torch.cat([t.float() * 0.5, t.float(), t.float() * 0.25]).cpu()      # 3 "steps" of B times, the forward's in the middle
    tab = prog.dense_table(fdbm_amd.hip.log_time(ts).to(DEV).contiguous())       # log t on the host, like HipNCSNpp.forward
    assert tab.shape == (3 * B, R)
    assert torch.equal(tab[B:2 * B], rows_fwd)
    assert not torch.equal(tab[:B], rows_fwd)
    # another step's rows in place, then the body alone: differs; the right rows: the full forward again, bit for bit
    prog.x_in.copy_(x); prog.y_in.copy_(y)
    dense_out[: B * R].copy_(tab[:B].reshape(-1))
    prog.run_body()
    assert not torch.equal(torch.view_as_real(prog.s_out), torch.view_as_real(ref))
    dense_out[: B * R].copy_(tab[B:2 * B].reshape(-1))
    prog.run_body()
    torch.cuda.synchronize()
    assert torch.equal(torch.view_as_real(prog.s_out), torch.view_as_real(ref))


def test_batch_rows_are_independent():
    """Samples of a batch never mix (GroupNorm and attention are per sample): evaluating [a, b] gives, row by
    row, what evaluating [a] and [b] gives - up to rounding, because the kernel / tile choice depends on the
    batch size.  Full-size map (257 x 256), f32 mode."""
    g = torch.Generator().manual_seed(3)
    x = torch.view_as_complex(torch.randn(2, 1, 257, 256, 2, generator=g)).to(DEV)
    y = torch.view_as_complex(torch.randn(2, 1, 257, 256, 2, generator=g)).to(DEV)
    t = torch.tensor([0.8, 0.3])
    m = net("ncsnpp_v2_5M")
    both = m(x, y, t.to(DEV)).cpu()
    for i in range(2):
        one = m(x[i:i + 1], y[i:i + 1], t[i:i + 1].to(DEV)).cpu()
        assert (both[i:i + 1] - one).abs().max() < 5e-5 * max(1.0, one.abs().max().item())


def test_graph_replay_survives_host_copies(golden):
    """Regression: replays of the sampler graph, with a synchronous device-to-host copy between them (what a caller
    that fetches the spectrogram, or a host-side gather, does), are BIT-IDENTICAL.  Round 1's graph zeroed the
    GroupNorm-statistics arena with a hipMemsetAsync node; such a node zeroes on the first replay only (ROCm 7.2:
    tools/memset_node_repro.py), every later result was NaN.  The arena is zeroed by a kernel of the library now, the
    statistics accumulate in fp64, so the bf16 mode is reproducible to the bit."""
    g = golden("samplers")
    y = T(g["mini64_y"]).to(DEV)
    br = fdbm_amd.Bridge("fm", N=3, sampler_type="ode_ei")
    m = net("mini64", torch.bfloat16)                    # fused mode: statistics through the arena
    outs = []
    for _ in range(3):
        x = br.sampler(m, y, generator=torch.Generator().manual_seed(5), use_graph=True)
        outs.append(torch.view_as_real(x.contiguous()).cpu())          # the host copy
    assert all(torch.isfinite(o).all() for o in outs)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


def test_infer_folder_driver(tmp_path):
    """The caller of the hot path (fdbm_amd.infer, mirror of infer_folder.py) end to end: synthetic Lightning
    checkpoint with EMA weights, WAV files of two formats / rates in nested directories, --keep_structure."""
    import argparse
    from scipy.io import wavfile
    from fdbm_amd import infer
    name = "ncsnpp_v2_5M"
    spec = Spec(**VARIANTS[name])
    raw = {k: T(v) for k, v in fill_state_dict(spec.param_shapes(), seed=1).items()}
    ema = {k: T(v) for k, v in fill_state_dict(spec.param_shapes(), seed=0).items()}
    ckpt = tmp_path / "model.ckpt"
    torch.save({"state_dict": {"dnn." + k: v for k, v in raw.items()},
                "hyper_parameters": dict(backbone=name, bridge="sb", noise_schedule="bb", n_fft=512, hop_length=256,
                                         window="sqrthann", spec_factor=0.15, spec_abs_exponent=0.5, normalize="noisy"),
                "ema": {"shadow_params": [ema[k] for k in spec.param_order()]}}, ckpt)
    rng = np.random.default_rng(0)
    src = tmp_path / "noisy"
    (src / "spk1").mkdir(parents=True)
    a = (0.3 * rng.standard_normal(16000)).astype(np.float32)                       # 1 s, 16 kHz float
    b = (8000 * rng.standard_normal(6000)).clip(-32768, 32767).astype(np.int16)     # 0.75 s, 8 kHz int16 -> resampled
    wavfile.write(src / "a.wav", 16000, a)
    wavfile.write(src / "spk1" / "b.wav", 8000, b)
    out = tmp_path / "enhanced"
    args = argparse.Namespace(device=["0"], test_dir=str(src), enhanced_dir=str(out), ckpt=str(ckpt), sampler_type="ode_ei",
                              sampler_kwargs=None, N=3, keep_structure=True, fp32=False)
    assert infer.enhance_folder(args) == 2
    sr, ea = wavfile.read(out / "a.wav")
    assert sr == 16000 and ea.shape == (16000,) and np.isfinite(ea).all() and np.abs(ea).max() <= 1.0
    sr, eb = wavfile.read(out / "spk1" / "b.wav")
    assert sr == 16000 and eb.shape == (12000,) and np.isfinite(eb).all()
    # the driver is the documented pipeline and nothing else (and the bf16 path is reproducible)
    enh = infer.Enhancer(str(ckpt), device=DEV, N=3)
    assert np.array_equal(enh(a[None])[0], ea)


@pytest.mark.parametrize("mode", ["single", "folder"])
def test_driver_reproduces_reference_on_bundled_clip(golden, tmp_path, mode):
    """Driver parity THROUGH the driver (BASELINE configs[0]): the bundled audio_samples/Sample1_Noisy.wav
    (its PCM is in the fixture) written as a WAV file, a Lightning-format checkpoint of ncsnpp_v2_5M, then
    `infer.enhance_single` (infer_single.py:53-101: N = 5, sb/bb ode_ei, 0.5 clip rule) - the file it writes must be the
    reference's x_hat to 2e-4 of full scale.  `folder`: the same file through `infer.enhance_folder` (infer_folder.py:
    94-121, 0.95 clip rule): identical up to the clip constant (x_hat 0.95 / 0.5 when the rule fired, else equal)."""
    import argparse
    from scipy.io import wavfile
    from fdbm_amd import infer
    g = golden("config0_sample1")
    name = "ncsnpp_v2_5M"
    spec = Spec(**VARIANTS[name])
    sd = {k: T(v) for k, v in fill_state_dict(spec.param_shapes(), seed=0).items()}
    ckpt = tmp_path / "model.ckpt"
    torch.save({"state_dict": {"dnn." + k: v for k, v in sd.items()},
                "hyper_parameters": dict(backbone=name, bridge="sb", noise_schedule="bb", n_fft=512, hop_length=256,
                                         window="sqrthann", spec_factor=0.15, spec_abs_exponent=0.5, normalize="noisy")}, ckpt)
    src = tmp_path / "noisy"
    src.mkdir()
    wavfile.write(src / "Sample1_Noisy.wav", 16000, g["pcm"].astype(np.int16))
    ref = np.asarray(g["x_hat"]).reshape(-1)
    if mode == "single":
        out = tmp_path / "out.wav"
        args = argparse.Namespace(device=["0"], noisy_file=str(src / "Sample1_Noisy.wav"), output_file=str(out), ckpt=str(ckpt),
                                  sampler_type="ode_ei", sampler_kwargs=None, N=5, fp32=True)
        assert infer.enhance_single(args) == str(out)
    else:
        out = tmp_path / "enhanced" / "Sample1_Noisy.wav"
        args = argparse.Namespace(device=["0"], test_dir=str(src), enhanced_dir=str(tmp_path / "enhanced"), ckpt=str(ckpt),
                                  sampler_type="ode_ei", sampler_kwargs=None, N=5, keep_structure=False, fp32=True, batch=1)
        assert infer.enhance_folder(args) == 1
        if np.abs(ref).max() == 0.5:                       # the reference's 0.5 rule fired: the folder driver scales to 0.95
            ref = ref / 0.5 * 0.95
    sr, x = wavfile.read(out)
    assert sr == 16000 and x.shape == ref.shape
    err = float(np.abs(x.astype(np.float64) - ref).max())
    assert err <= 2e-4, (mode, err, float(np.abs(ref).max()))


def test_infer_folder_batched(tmp_path):
    """BASELINE configs[3] on the driver side: `--batch B` buckets the files by padded length and runs the sampler on
    batches of rows; every file must come out as the one-file-at-a-time path gives it (same prior noise; the kernels a
    layer runs on - and with them the rounding - depend on the batch size, and three evaluations of the random-weight
    net amplify that: 3e-4 of full scale, measured 4e-5), the batches must have been formed by length, and the folder
    driver writes the same set of files."""
    import argparse
    from scipy.io import wavfile
    from fdbm_amd import infer
    name = "ncsnpp_v2_5M"
    spec = Spec(**VARIANTS[name])
    sd = {k: T(v) for k, v in fill_state_dict(spec.param_shapes(), seed=0).items()}
    ckpt = tmp_path / "model.ckpt"
    torch.save({"state_dict": {"dnn." + k: v for k, v in sd.items()},
                "hyper_parameters": dict(backbone=name, bridge="sb", noise_schedule="bb", n_fft=512, hop_length=256,
                                         window="sqrthann", spec_factor=0.15, spec_abs_exponent=0.5, normalize="noisy")}, ckpt)
    rng = np.random.default_rng(1)
    lengths = [16000, 16100, 15000, 40000, 40500, 33000, 16384]
    waves = [(0.2 * rng.standard_normal(n)).astype(np.float32) for n in lengths]
    stereo = (0.2 * rng.standard_normal((2, 16000))).astype(np.float32)
    enh = infer.Enhancer(str(ckpt), device=DEV, N=3, dtype=torch.float32)

    def noise(fi, c, shape):
        g = torch.Generator().manual_seed(1000 + 10 * fi + c)
        return torch.view_as_complex(torch.randn(tuple(shape) + (2,), generator=g)).to(DEV)

    many = enh.enhance_many(waves + [stereo], batch=3, prior_noise_fn=noise)
    shapes = list(enh.batch_shapes)
    assert len(many) == len(waves) + 1 and many[-1].shape == (2, 16000)
    assert all(b <= 3 for b, _ in shapes) and [f for _, f in shapes] == sorted((f for _, f in shapes), reverse=True)
    assert len(shapes) < len(waves) + 2                      # fewer sampler calls than rows: rows were batched
    for fi, w in enumerate(waves + [stereo]):
        w2 = w if w.ndim == 2 else w[None]
        enh.sampler_kwargs = dict(prior_noise=torch.stack([noise(fi, c, enh_row_shape(enh, w2.shape[-1])) for c in range(w2.shape[0])], 0))
        one = enh(w2)
        assert one.shape == many[fi].shape
        assert np.abs(one - many[fi]).max() <= 3e-4 * max(1.0, np.abs(one).max()), (fi, np.abs(one - many[fi]).max())
    enh.sampler_kwargs = {}
    # the folder driver in batched mode
    src = tmp_path / "noisy"
    src.mkdir()
    for i, w in enumerate(waves):
        wavfile.write(src / f"f{i}.wav", 16000, w)
    out = tmp_path / "enhanced"
    args = argparse.Namespace(device=["0"], test_dir=str(src), enhanced_dir=str(out), ckpt=str(ckpt), sampler_type="ode_ei",
                              sampler_kwargs=None, N=3, keep_structure=False, fp32=True, batch=4)
    assert infer.enhance_folder(args) == len(waves)
    for i, n in enumerate(lengths):
        sr, e = wavfile.read(out / f"f{i}.wav")
        assert sr == 16000 and e.shape == (n,) and np.isfinite(e).all()
    # a file the front-end refuses (shorter than n_fft / 2: reflect padding impossible) costs ITSELF, not its window of
    # 4 x batch files: the window is redone one file at a time, as the reference's per-file try / except does
    wavfile.write(src / "zz_too_short.wav", 16000, np.zeros(100, np.float32))
    out2 = tmp_path / "enhanced2"
    args.enhanced_dir = str(out2)
    assert infer.enhance_folder(args) == len(waves)
    assert not (out2 / "zz_too_short.wav").exists() and all((out2 / f"f{i}.wav").exists() for i in range(len(waves)))


def test_infer_driver_with_tfgridnet_checkpoint(tmp_path):
    """The drivers take whatever backbone the checkpoint names (infer_folder.py:74-77): a TF-GridNet checkpoint (no
    spectrogram padding for this family, infer_folder.py:83-88) runs end to end, one file and batched."""
    from fdbm_amd import infer
    from fdbm_amd import tfgridnet as tg
    name = "tfgridnet_4l32c80"
    sd = {k: T(v) for k, v in tg.fill_state(tg.param_shapes(**tg.VARIANTS[name])).items()}
    ckpt = tmp_path / "tfg.ckpt"
    raw = {k: T(v) for k, v in tg.fill_state(tg.param_shapes(**tg.VARIANTS[name]), seed=5).items()}
    raw["get_time_emb.W"] = sd["get_time_emb.W"]                    # (fixed, not tracked by the EMA)
    torch.save({"state_dict": {"dnn." + k: v for k, v in raw.items()},
                "hyper_parameters": dict(backbone=name, bridge="fm", noise_schedule="ot", n_fft=512, hop_length=256,
                                         window="sqrthann", spec_factor=0.15, spec_abs_exponent=0.5, normalize="noisy"),
                "ema": {"shadow_params": [sd[k] for k in sd if k != "get_time_emb.W"]}}, ckpt)
    from fdbm_amd.checkpoint import load_lightning_checkpoint
    _, st = load_lightning_checkpoint(str(ckpt))
    assert all(torch.equal(st[k], sd[k]) for k in sd)               # the EMA weights are the ones evaluated
    rng = np.random.default_rng(2)
    waves = [(0.2 * rng.standard_normal(n)).astype(np.float32) for n in (6000, 6000, 9000)]
    enh = infer.Enhancer(str(ckpt), device=DEV, N=2)
    assert enh.pad_mode is None
    one = enh(waves[0][None])
    assert one.shape == (1, 6000) and np.isfinite(one).all()
    many = enh.enhance_many(waves, batch=2)
    assert [m.shape for m in many] == [(1, 6000), (1, 6000), (1, 9000)] and all(np.isfinite(m).all() for m in many)
    assert enh.batch_shapes[0][0] == 1 and enh.batch_shapes[1][0] == 2          # the 9000-sample file alone, then the two 6000s


def enh_row_shape(enh, n_samples):
    """[1, F, Tpad] of one row's padded spectrogram for a waveform of n_samples (as spec_forward_padded gives it)."""
    y = torch.zeros(1, n_samples, device=DEV)
    return tuple(enh.fe.spec_forward_padded(y, enh.pad_mode).shape[1:])


def _toy_model(xt, y, t):
    tt = t.to(xt.device)[:, None, None, None]
    return 0.6 * y + 0.3 * xt * torch.cos(tt) + 0.05 * torch.roll(xt, 1, dims=-1)


def test_sampler_ode_int(golden):
    """SciPy RK45 over the flattened state (host loop, as in the reference).
    (1) plumbing: with a smooth callable model the device path (host<->device hops, HIP flow
        kernel) reproduces the CPU oracle;
    (2) with the HIP backbone it runs and stays finite.  Values are not compared there: adaptive
        step accept/reject decisions on the random-weight network turn 1e-5 differences into O(1)
        ones (the CPU oracle on another CPU already lands 1.5e-2 from the reference's result), and for sb the flow at t = 1 has weights of +-3.3e7 that cancel in
        fp32."""
    from oracle import sampler as osamp
    g = golden("samplers")
    y = T(g["y"])
    for path in ("fm", "sb"):
        br = fdbm_amd.Bridge(path, N=5, sampler_type="ode_int")
        out = br.sampler(_toy_model, y.to(DEV), generator=torch.Generator().manual_seed(5), rtol=1e-3, atol=1e-3).cpu()
        ref = osamp.Sampler(path, N=5).ode_int(_toy_model, y, torch.Generator().manual_seed(5), rtol=1e-3, atol=1e-3)
        assert (out - ref).abs().max() < 1e-4 * max(1.0, ref.abs().max().item()), path
        out = br.sampler(net("ncsnpp_v2_5M"), y.to(DEV), generator=torch.Generator().manual_seed(1234), rtol=1e-2, atol=1e-2)
        assert out.shape == y.shape and torch.isfinite(torch.view_as_real(out)).all()


def test_full_size_ncsnpp_v2_vs_reference(golden):
    """BASELINE configs[1] geometry: [1,1,257,256], ncsnpp_v2 (65.6 M), N=30 ode_ei, fp32 parity mode.

    One backbone evaluation: <= 5e-5 max-abs vs the reference (|s| up to 6.7) and at least as close
    to the fp64 arbiter as the reference.  After N=30 evaluations the random-weight network has
    amplified fp32 rounding noise ~30x: the reference run on another CPU lands 3.6e-3 (sb) /
    7.6e-4 (fm) from the reference run that made the golden, and the reference is itself 1.9e-3 /
    1.2e-4 from the fp64 trajectory (tests/golden/reference_spread.json).  So the N=30 check is
    bounded by that spread, not by 1e-4, and the arbiter check carries the weight."""
    g = golden("full_ncsnpp_v2")
    arb = golden("fp64_arbiter")
    m = net("ncsnpp_v2")
    out = m(T(g["x"]).to(DEV), T(g["y"]).to(DEV), T(g["t"]).to(DEV)).cpu()
    ref = T(g["fwd"])
    err = (out - ref).abs().max().item()
    assert err < 5e-5, err                  # absolute; measured 1.5e-5 at |s| <= 6.7
    a = T(arb["full_fwd"])
    assert (out - a).abs().max().item() < 1.5 * (ref - a).abs().max().item() + 2e-6
    y = T(g["y"]).to(DEV)
    for key, bkw, tol in (("sb_bb_ode_ei_N30", dict(path="sb", noise_schedule="bb"), 6e-3),
                          ("fm_ot_ode_ei_N30", dict(path="fm", noise_schedule="ot"), 1.5e-3)):
        br = fdbm_amd.Bridge(N=30, sampler_type="ode_ei", **bkw)
        out = br.sampler(m, y, generator=torch.Generator().manual_seed(4321)).cpu()
        ref = T(g[key])
        err = (out - ref).abs().max().item()
        assert err < tol, (key, err)
        a = T(arb["full_" + key])
        rms = lambda d: d.abs().pow(2).mean().sqrt().item()
        # rms distance to the exact trajectory stays at the 1e-4 level (|x| up to 27)
        assert rms(out - a) < 3e-4, (key, rms(out - a), rms(ref - a))


def test_fail_loudly_on_cpu_tensors():
    m = net("mini64")
    x = torch.zeros(1, 1, 257, 64, dtype=torch.complex64)
    with pytest.raises(RuntimeError):
        m(x, x, torch.ones(1))


@pytest.mark.parametrize("sched", ["ve", "vp", "gmax"])
def test_pc_sampler_other_schedules_batch2(golden, sched):
    """SURVEY 8(f4), second half: the predictor-corrector sampler under the ve / vp / gmax schedules of the Schroedinger
    bridge at B > 1 - per-sample [B] weights (the reference's own broadcast is only right for B = 1, bridge.py:287-306).
    The captured graph, the eager loop and the CPU oracle (oracle/sampler.py with the oracle backbone) agree on a
    two-row batch with the same host-drawn noise."""
    from oracle import sampler as osamp
    from oracle import ncsnpp as onet
    g = golden("samplers")
    y1 = T(g["mini64_y"])
    y = torch.cat([y1, 0.8 * torch.roll(y1, 2, dims=-1)], 0)
    hp = MINI64
    sd = fill_state_dict(Spec(**hp).param_shapes(), seed=0)
    cpu_net = onet.Model(sd, hp)
    br = fdbm_amd.Bridge("sb", N=3, noise_schedule=sched, sampler_type="pc")
    kw = dict(predictor_name="euler_maruyama", corrector_name="ald", corrector_steps=1, snr=0.4, denoise=True)
    outs = [br.sampler(net("mini64"), y.to(DEV), generator=torch.Generator().manual_seed(77), use_graph=ug, **kw).cpu()
            for ug in (True, False)]
    ref = osamp.Sampler("sb", N=3, noise_schedule=sched).pc(cpu_net, y, torch.Generator().manual_seed(77), corrector="ald",
                                                             snr=0.4, corrector_steps=1, denoise=True)
    scale = max(1.0, ref.abs().max().item())
    assert (outs[0] - outs[1]).abs().max().item() < 2e-4 * scale, sched
    assert (outs[1] - ref).abs().max().item() < 5e-4 * scale, (sched, (outs[1] - ref).abs().max().item())


def test_samplers_with_device_generated_noise(golden):
    """device_seed: the stochastic samplers draw their noise with the library's counter-based generator ON the device
    (draw 0 = prior, 1 + i = the i-th step draw), inside the kernel that consumes it when the sampler runs as a graph.
    Graph == eager (which materialises the same draws with fdbm_randn_complex) bit for bit; the same seed reproduces,
    another seed does not; the result equals injecting those draws by hand."""
    g = golden("samplers")
    y = T(g["y"]).to(DEV)
    m = net("ncsnpp_v2_5M")
    skw = dict(predictor_name="euler_maruyama", corrector_name="ald", corrector_steps=1, snr=0.5, denoise=False)
    lkw = dict(skw, corrector_name="langevin")
    for st, kw, n in (("sde_ei", {}, 4), ("pc", skw, 3), ("pc", lkw, 2)):
        br = fdbm_amd.Bridge("sb", N=n, noise_schedule="bb", sampler_type=st)
        a = br.sampler(m, y, device_seed=77, use_graph=True, **kw)
        b = br.sampler(m, y, device_seed=77, use_graph=False, **kw)
        assert torch.isfinite(torch.view_as_real(a)).all()
        if kw.get("corrector_name") == "langevin":
            # (eager takes the norms through torch reductions, the graph through fdbm_langevin_step: same draws, fp32-close)
            assert (a - b).abs().max().item() <= 1e-4 * max(1.0, b.abs().max().item()), st
        else:
            assert torch.equal(torch.view_as_real(a), torch.view_as_real(b)), st
        assert torch.equal(torch.view_as_real(a), torch.view_as_real(br.sampler(m, y, device_seed=77, **kw)))
        assert not torch.equal(torch.view_as_real(a), torch.view_as_real(br.sampler(m, y, device_seed=78, **kw)))
        # by hand: the draws as tensors
        from fdbm_amd.bridge import rng_state
        from fdbm_amd import hip
        stw = rng_state(77, DEV)

        def draw(i):
            z = torch.empty_like(y)
            hip.call("fdbm_randn_complex", hip.ptr(z), z.numel(), hip.ptr(stw), 1 + i)
            return z
        c = br.sampler(m, y, prior_noise=torch.zeros_like(y), step_noise=draw, **kw)
        if kw.get("corrector_name") == "langevin":
            assert (a - c).abs().max().item() <= 1e-4 * max(1.0, c.abs().max().item())
        else:
            assert torch.equal(torch.view_as_real(a), torch.view_as_real(c)), st
