#!/usr/bin/env python3
"""Benchmark of the reverse-sampling hot path (BASELINE.json metric: real-time factor at
N=30 steps on 16 kHz / 4 s clips).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: one rank per GPU over RCCL.  Under torch.distributed.run - WORLD_SIZE set - this process is one rank; started
   bare with --gpus N > 1 it launches the N ranks itself, before touching any GPU, and relays rank 0's JSON line)
  python bench.py --gpus 8 --clips 2000 --batch 64      BASELINE configs[3]: a list of clips sharded over the ranks

A "step" is one pass of the hot path over one batch of synthetic clips already resident
in HBM:  normalise -> STFT + compression + time padding -> N=30 ODE-EI sampler (30 NCSN++
evaluations + 30 fused state updates, one HIP graph) -> inverse compression + iSTFT ->
renormalise [-> RCCL all-gather of the enhanced spectrograms when N > 1].
Workload at every N (weak scaling): BASELINE.json configs[1] per rank - one synthetic
4 s clip, ncsnpp_v2 (65.6 M parameters, deterministic synthetic weights), bridge sb/bb,
bf16 storage + fp32 accumulate.  --batch 64 gives configs[2];
--batch 16 --N 100 --dtype f16 --sampler sde_ei | pc gives configs[4] as the timed workload (the default run times both in `extras`).

Rank 0 prints ONE JSON line.  `roofline` prices the convolution kernel that carries most of the
algorithmic flops (conv_ring_kernel<R=16>, MFMA-bound: 57 % of them at batch 1, 97 % at batch 64) from HIP-event timings of its launches
in one eager forward, lists every convolution kernel family the same way (`families`) and carries
the HBM bytes per launch measured offline with rocprofv3 PMC passes (profiles/r0*/..pmc_traffic..json);
`cpu_baseline` times the CPU oracle (a port, oracle/) on a bounded sample on the host cores.
FDBM_BENCH_BACKEND=gloo rehearses the N > 1 control flow on a box with fewer GPUs than ranks.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SR = 16000
CLIP_SECONDS = 4.0
BF16_DENSE_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
F32_MFMA_PEAK_TFLOPS = 157.3


def synth_clips(B, seed, device):
    """Speech-like band-limited harmonic signal + white noise at 0..10 dB SNR, peak-normalised."""
    n = int(SR * CLIP_SECONDS)
    out = np.empty((B, n), np.float32)
    for b in range(B):
        g = np.random.Generator(np.random.Philox(seed * 100003 + b))
        t = np.arange(n) / SR
        f0 = g.uniform(90, 250)
        sig = sum(np.sin(2 * np.pi * f0 * (k + 1) * t + g.uniform(0, 6.28)) / (k + 1) for k in range(12))
        env = 0.5 + 0.5 * np.sin(2 * np.pi * g.uniform(1.5, 4.0) * t) ** 2
        sig = sig * env
        snr_db = g.uniform(0, 10)
        noise = g.standard_normal(n)
        noise *= np.sqrt(np.mean(sig ** 2) / (np.mean(noise ** 2) * 10 ** (snr_db / 10)))
        x = sig + noise
        out[b] = (x / np.max(np.abs(x))).astype(np.float32)
    return torch.from_numpy(out).to(device)


class HotPath:
    """waveforms in HBM -> enhanced waveforms in HBM, everything through libfdbm_hip.so."""

    def __init__(self, device, dtype, n_steps, batch, backbone="ncsnpp_v2", bridge="sb", schedule="bb", sampler="ode_ei", split=False):
        import fdbm_amd
        from fdbm_amd.frontend import SpecFrontend, pad_mode_for
        self.dev = device
        self.B = batch
        self.net = fdbm_amd.BackboneRegistry.get_by_name(backbone)(dtype=dtype, device=device, **(dict(split=True) if split else {}))
        self.fe = SpecFrontend(n_fft=512, hop_length=256, window="sqrthann", device=device)   # config.yaml:35-38
        self.bridge = fdbm_amd.Bridge(bridge, N=n_steps, sampler_type=sampler, noise_schedule=schedule)
        self.pad_mode = pad_mode_for(backbone)
        self.gen = torch.Generator().manual_seed(0)
        self.kw = {}
        self.calls = 0
        self.stochastic = sampler != "ode_ei"
        if sampler == "pc":
            # BASELINE configs[4]: pc = euler_maruyama + ald, 1 corrector step, snr 0.5 (the reference's defaults, bridge.py:142-166)
            self.kw.update(predictor_name="euler_maruyama", corrector_name="ald", corrector_steps=1, snr=0.5, denoise=True)

    def enhance(self, wave):
        """infer_folder.py:102-121 per batch.  Returns (enhanced waveform, enhanced spectrogram)."""
        nf = self.fe.norm_factor(wave)                                    # max |y| per clip (fdbm_wave_norm_factor)
        Y = self.fe.spec_forward_padded(wave, self.pad_mode, norm=nf)     # y / nf fused into the STFT launch
        if self.stochastic:
            # fresh Gaussian noise for every step of every call, generated INSIDE the timed region by the kernels that consume
            # it (the library's counter-based generator, seed = call index): no host generator, no uploads (bridge.py:47,108)
            self.calls += 1
            X = self.bridge.sampler(self.net, Y, device_seed=self.calls, **self.kw)
        else:
            X = self.bridge.sampler(self.net, Y, generator=self.gen)
        x_hat = self.fe.to_audio(X[:, 0], wave.shape[-1], norm=nf, clip=0.95)   # * nf and the 0.95 clip rule fused
        return x_hat, X


class StubHotPath:
    """FDBM_BENCH_STUB=1: a host-only stand-in for HotPath (x_hat = the input, X = a constant spectrogram per clip) so that
    the multi-rank control flow - launcher, world-size assertion, sharding, gather rounds, barrier / max-over-ranks timing,
    teardown order - is covered by a CPU test under gloo (tests/test_bench_ranks.py).  Never used for a measurement."""

    def __init__(self, batch):
        self.B = batch
        self.net = None

    def enhance(self, wave):
        X = torch.full((wave.shape[0], 1, 257, 256), 1.0 + 0.5j, dtype=torch.complex64) * wave[:, :1, None, None].abs().max()
        return wave, X


def time_conv_launches(net, B, F, T, reps=3):
    """HIP-event time of every conv_igemm launch of one forward (eager, same stream)."""
    from fdbm_amd import hip
    prog = net.program(B, F, T)
    prog.run()
    torch.cuda.synchronize()
    conv_ids = [i for i, op in enumerate(prog.ops) if op[0] == hip.OP_CONV]
    best = [float("inf")] * len(conv_ids)
    fwd_ms = float("inf")
    # A SHORT launch is timed as 4 back-to-back repetitions between one pair of events, divided by 4: a single
    # 30-40 us launch between two events reads 2-5 us long (event / dispatch overhead) against rocprofv3's kernel
    # duration.  Long launches (batch 64: hundreds of us) are timed one by one - repeated, they would find their
    # operands in the Infinity Cache and read 8 % short.  (The repetitions accumulate into the launch's statistics
    # buffers; nothing downstream is used, and every forward re-zeroes them.)
    inner = [1] * len(conv_ids)
    for rep in range(reps + 1):
        evs, launched = [], []
        lo = 0
        for k, i in enumerate(conv_ids):
            if i > lo:
                prog.run_range(lo, i)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _r in range(inner[k]):
                prog.run_range(i, i + 1)
            b.record()
            launched.append(hip.lib().fdbm_conv_last_kind())
            evs.append((a, b))
            lo = i + 1
        prog.run_range(lo, prog.n_ops)
        torch.cuda.synchronize()
        if rep == 0:                # sizing pass
            inner = [4 if a.elapsed_time(b) < 0.15 else 1 for (a, b) in evs]
            continue
        for k, (a, b) in enumerate(evs):
            best[k] = min(best[k], a.elapsed_time(b) / inner[k])
    for _ in range(reps):           # one plain eager forward (also leaves the statistics buffers consistent)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); prog.run(); e1.record()
        torch.cuda.synchronize()
        fwd_ms = min(fwd_ms, e0.elapsed_time(e1))
    # algorithmic flops of each conv launch (all segments), whole batch; which kernel runs it
    flops, kinds = [], []
    for ca in prog.keep_conv:
        k = sum(ca.seg[s].cin * ca.seg[s].taps for s in range(ca.nseg))
        flops.append(2.0 * ca.B * ca.H * ca.W * ca.Cout * k)
    kinds = launched            # kernel family of each launch as the library reports it (fdbm_conv_last_kind)
    return best, flops, kinds, fwd_ms, prog


def end_to_end_error_n30(dev, dtype16):
    """The modes' error on the FINAL spectrogram after N = 30 (the quantity the north-star's 1e-4 is about): the contractive
    fixture (tests/golden/contractive_ncsnpp_v2.npz - the reference's own N = 30 sb/bb result for this noisy spectrogram and
    these weights) evaluated in the 16-bit throughput mode, in the f32 mode and in the split-precision mode.
    Asserted in tests/test_hip_parity.py::test_modes_end_to_end_error_n30."""
    import numpy as np
    import fdbm_amd
    from fdbm_amd.arch import Spec, VARIANTS
    from fdbm_amd.backbone import HipNCSNpp
    from fdbm_amd.weights import fill_state_dict
    fix = os.path.join(ROOT, "tests", "golden", "contractive_ncsnpp_v2.npz")
    if not os.path.exists(fix):
        return {}
    g = np.load(fix)
    y = torch.from_numpy(g["y"]).to(dev)
    ref = torch.from_numpy(g["sb_bb_ode_ei_N30"])
    sd = {k: torch.from_numpy(np.asarray(v))
          for k, v in fill_state_dict(Spec(**VARIANTS["ncsnpp_v2"]).param_shapes(), seed=0, profile="contractive").items()}
    br = fdbm_amd.Bridge("sb", N=30, noise_schedule="bb", sampler_type="ode_ei")
    if dtype16 not in (torch.bfloat16, torch.float16):
        dtype16 = torch.bfloat16
    tag16 = "bf16" if dtype16 == torch.bfloat16 else "f16"
    xs = {}
    for tag, kw in ((tag16, dict(dtype=dtype16)), ("f32", dict(dtype=torch.float32)), ("f32s", dict(dtype=torch.float32, split=True))):
        m = HipNCSNpp(device=dev, state=sd, **VARIANTS["ncsnpp_v2"], **kw)
        xs[tag] = br.sampler(m, y, generator=torch.Generator().manual_seed(4321)).cpu()
        del m
    out = {f"{tag}_final_max_abs_vs_reference_n30": float((x - ref).abs().max()) for tag, x in xs.items()}
    out[f"{tag16}_final_max_abs_vs_f32_n30"] = float((xs[tag16] - xs["f32"]).abs().max())
    out["final_spectrogram_max_abs"] = float(ref.abs().max())
    return out


def cpu_baseline(n_forwards, n_steps):
    """Times the CPU oracle (oracle/, a port of the reference path) on the host cores."""
    from fdbm_amd.arch import Spec, VARIANTS
    from fdbm_amd.weights import fill_state_dict
    from oracle import ncsnpp as onet, frontend as ofe
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))          # the GPU box's CPU share for one GPU is 16 cores
    torch.set_num_threads(cores)
    hp = VARIANTS["ncsnpp_v2"]
    model = onet.Model(fill_state_dict(Spec(**hp).param_shapes(), seed=0), hp)
    wave = synth_clips(1, 0, "cpu")
    t0 = time.time()
    Y = ofe.pad_spec(ofe.spec_fwd(ofe.stft(wave / wave.abs().max()))[:, None], "reflection")
    t_front = time.time() - t0
    xt = Y.clone()
    t0 = time.time()
    for i in range(n_forwards):
        s = model(xt, Y, torch.tensor([1.0 - 0.03 * i]))
        xt = 0.9 * xt + 0.1 * s
    t_fwd = (time.time() - t0) / n_forwards
    t0 = time.time()
    ofe.istft(ofe.spec_back(xt[:, 0]), wave.shape[-1])
    t_back = time.time() - t0
    total = t_front + n_steps * t_fwd + t_back
    return dict(value=CLIP_SECONDS / total, unit="x real-time (audio-s/wall-s)", cores=cores, kind="port",
                sample=f"{n_forwards} of the {n_steps} backbone evaluations of one 4 s clip (fp32 torch CPU ops, "
                       f"{cores} threads, {t_fwd:.2f} s each) + full front-end/back-end; sampler extrapolated x{n_steps}/{n_forwards}")


def launch_ranks(n, argv):
    """--gpus n > 1 without a launcher: start the n ranks (one per GPU, RCCL) as children of THIS process, which has
    not touched a GPU (never re-exec a process that has), relay rank 0's JSON line, fail if any rank fails."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if proc.returncode != 0 or line is None:
        sys.stderr.write(proc.stdout)
        sys.exit(proc.returncode or 1)
    print(line)
    sys.exit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1, help="clips per rank per step (1 = configs[1], 64 = configs[2])")
    ap.add_argument("--clips", type=int, default=0,
                    help="configs[3]: a list of this many synthetic clips, strided over the ranks, enhanced in batches of "
                         "--batch; a step = the whole list once (strong scaling)")
    ap.add_argument("--N", type=int, default=30, help="sampler steps")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32", "f32s"],
                    help="bf16 / f16: 16-bit storage, f32 accumulation (throughput modes); f32: f32 storage, exact f32 MFMA; "
                         "f32s: f32 storage, split-precision matrix products (three f16 MFMAs over 22-bit (hi, lo) operand pairs) - "
                         "the parity mode that runs on the 16-bit matrix pipe")
    ap.add_argument("--sampler", default="ode_ei", choices=["ode_ei", "sde_ei", "pc"],
                    help="ode_ei = configs[1..3]; sde_ei / pc with --batch 16 --N 100 --dtype f16 = configs[4]")
    ap.add_argument("--backbone", default="ncsnpp_v2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-forwards", type=int, default=2)
    ap.add_argument("--no-extras", action="store_true", help="skip the fp32-parity-mode and batch-64 side measurements")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus, sys.argv[1:])

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # FDBM_BENCH_BACKEND=gloo: rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks
    # (ranks share devices, the gather goes through host memory); the real runs use RCCL
    backend = os.environ.get("FDBM_BENCH_BACKEND", "nccl")
    stub = os.environ.get("FDBM_BENCH_STUB", "") == "1"        # host-only rehearsal of the rank control flow (CPU tests)
    if stub and backend != "gloo":
        raise SystemExit("bench.py: FDBM_BENCH_STUB=1 is a gloo rehearsal (set FDBM_BENCH_BACKEND=gloo); it measures nothing")
    if backend == "gloo" and not stub:
        local_rank = local_rank % max(1, torch.cuda.device_count())
    comm_world = 1
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend)
        comm_world = dist.get_world_size()
        assert comm_world == args.gpus, (comm_world, args.gpus)
    if stub:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device(f"cuda:{local_rank}")
    sync = (lambda: None) if stub else torch.cuda.synchronize
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32, "f32s": torch.float32}[args.dtype]
    split = args.dtype == "f32s"
    evals_per_step = 2 * args.N if args.sampler == "pc" else args.N        # pc: predictor + 1 corrector evaluation

    from fdbm_amd import dist as fdist
    hp = StubHotPath(args.batch) if stub else HotPath(dev, dtype, args.N, args.batch, backbone=args.backbone, sampler=args.sampler, split=split)
    if args.clips:
        # configs[3]: the list is sharded by index (rank r takes clips r, r + W, ...: fdbm_amd.dist.shard_indices), every
        # rank walks its shard in batches of --batch (a ragged last batch is padded with its first clip and trimmed)
        mine = fdist.shard_indices(args.clips, rank, world)
        waves = [synth_clips(1, 5000 + i, dev) for i in mine]              # resident in HBM before timing
        n_batches = (len(mine) + args.batch - 1) // args.batch
        batches = []
        for bi in range(n_batches):
            part = waves[bi * args.batch:(bi + 1) * args.batch]
            real = len(part)
            part = part + [part[0]] * (args.batch - real)
            batches.append((torch.cat(part, 0), real))
        # every rank runs the same number of gather rounds (a rank whose shard is a batch shorter sends an empty one)
        rounds = (len(fdist.shard_indices(args.clips, 0, world)) + args.batch - 1) // args.batch
    else:
        batches = [(synth_clips(args.batch, 1000 + rank, dev), args.batch)]   # resident in HBM before timing
        rounds = 1
    wave = batches[0][0] if batches else None

    gather_stats = {"bytes": 0, "seconds": 0.0, "calls": 0}

    def step():
        x_hat = torch.zeros(0, device=dev)
        X = torch.zeros(0, 1, 257, 256, dtype=torch.complex64, device=dev)     # (a rank whose shard is empty sends empty batches)
        for r in range(rounds):
            if r < len(batches):
                w, real = batches[r]
                x_hat, X = hp.enhance(w)
                X = X[:real]
            else:
                X = X[:0]
            if os.environ.get("FDBM_BENCH_DEBUG"):
                torch.cuda.synchronize()
                print(f"[dbg] rank {rank}: X finite {bool(torch.isfinite(torch.view_as_real(X)).all())}; "
                      f"x_hat finite {bool(torch.isfinite(x_hat).all())}", file=sys.stderr, flush=True)
            if world > 1:
                # enhanced spectrograms only, to rank 0, over RCCL / xGMI (gloo rehearsal: through host memory)
                sync()
                tg = time.perf_counter()
                fdist.gather_spectrograms(X if backend == "nccl" else X.cpu(), dst=0)
                sync()
                gather_stats["seconds"] += time.perf_counter() - tg
                gather_stats["bytes"] += X.numel() * 8
                gather_stats["calls"] += 1
        return x_hat

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    for _ in range(args.warmup):
        step()
    gather_stats.update(bytes=0, seconds=0.0, calls=0)
    sync(); barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync()
    own_elapsed = time.perf_counter() - t0          # this rank's own work, before it waits for the others
    barrier()
    elapsed = time.perf_counter() - t0
    per_rank = None
    if world > 1:
        import torch.distributed as dist
        cdev = dev if backend == "nccl" else "cpu"
        tt = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        # self-diagnosis for the scaling curve: every rank's own time, gather time and gathered bytes (rank 0 reports them)
        mine = torch.tensor([own_elapsed, gather_stats["seconds"], float(gather_stats["bytes"]), float(gather_stats["calls"])],
                            device=cdev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [dict(rank=r, elapsed_s=float(v[0]), gather_s=float(v[1]), gather_bytes_sent=int(v[2]), gather_calls=int(v[3]))
                    for r, v in enumerate(allr)]
    assert torch.isfinite(out).all(), f"rank {rank}: {int((~torch.isfinite(out)).sum())} non-finite samples of {out.numel()}"

    n_clips_step = args.clips if args.clips else world * args.batch
    audio_s = args.steps * n_clips_step * CLIP_SECONDS
    rtf = audio_s / elapsed
    if world > 1:
        # every rank leaves the process group HERE, together: rank 0's roofline measurements below are single-GPU work and
        # would otherwise tear the communicator down minutes after its peers have exited
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    if stub:
        print(json.dumps({"metric": "stub (FDBM_BENCH_STUB=1: rank control flow only, nothing measured)", "value": None, "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "rccl_world_size": comm_world, "backend": backend,
                          "clips_per_step": n_clips_step, "per_rank": per_rank}))
        return

    F, T = 257, 256
    flops_fwd = hp.net.flops_per_forward(256, T)
    result = {
        "metric": f"real-time factor (audio-sec/wall-sec) @ N={args.N} steps, 16 kHz 4 s clips",
        "value": rtf, "unit": "x real-time (audio-s / wall-s)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "strong" if args.clips else "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": (f"BASELINE configs[3]: {args.clips} synthetic 4 s 16 kHz clips sharded by index over the ranks, batches of {args.batch}, "
                                if args.clips else
                                f"BASELINE configs[{4 if args.sampler != 'ode_ei' else 1 if args.batch == 1 else 2}]: {args.batch} synthetic 4 s 16 kHz clip(s) per rank per step, ") +
                               f"{args.backbone} (deterministic synthetic weights), bridge sb/bb, {args.sampler} N={args.N}, "
                               "STFT 512/256 sqrt-Hann -> [257 x 256] complex spectrogram",
                   "batch_per_rank": args.batch, "sampler_steps": args.N, "rccl_world_size": comm_world,
                   "parallelism": f"dp{world} (utterance sharding, gather of spectrograms)"},
        "whole_step_tflops": n_clips_step * evals_per_step * flops_fwd / (elapsed / args.steps) / 1e12,
    }
    if per_rank is not None:
        result["per_rank"] = per_rank          # own time / gather time / gathered bytes of every rank over the timed steps

    # ---- roofline of the dominant kernel ---------------------------------------------------
    # conv_patch_kernel (3x3 convs of the large feature maps) carries most of the algorithmic
    # flops; its achieved rate = its launches' algorithmic flops / their HIP-event durations.
    # all_conv = the same over every convolution launch (both kernels), i.e. incl. the
    # latency-bound small-map layers.
    times, flops, kinds, fwd_ms, prog = time_conv_launches(hp.net, args.batch, F, T)
    # f16 = the bf16 dense MFMA peak; f32s computes every product as three f16 MFMAs: its ALGORITHMIC flops are priced against
    # a third of that peak (833 TFLOP/s effective)
    peak = F32_MFMA_PEAK_TFLOPS if args.dtype == "f32" else BF16_DENSE_PEAK_TFLOPS / 3 if args.dtype == "f32s" else BF16_DENSE_PEAK_TFLOPS
    dom = 3 if any(k == 3 for k in kinds) else 1
    sel = [i for i, k in enumerate(kinds) if k == dom] or list(range(len(times)))
    t_dom = sum(times[i] for i in sel) * 1e-3
    f_dom = sum(flops[i] for i in sel)
    achieved = f_dom / t_dom / 1e12
    traffic = None
    fam_name = {0: "conv_igemm_kernel", 1: "conv_patch_kernel", 2: "conv_tap_kernel", 3: "conv_ring_kernel<R=16>",
                4: "conv_ring_kernel<R=8>", 5: "conv_head_kernel", 6: "conv_small_kernel", 7: "conv_mid_kernel"}
    pmc_tab, pmc_src = {}, None
    for rnd in ("r03", "r02"):          # the newest committed PMC pass (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over this command)
        cand = os.path.join(ROOT, "profiles", rnd, f"pmc_traffic_b{args.batch}_bf16.json")
        if args.dtype == "bf16" and os.path.exists(cand):
            try:
                pmc_tab, pmc_src = json.load(open(cand)), f"profiles/{rnd}/pmc_traffic_b{args.batch}_bf16.json (offline rocprofv3 --pmc pass, not measured by this run)"
                break
            except Exception:
                pmc_tab, pmc_src = {}, None
    traffic = pmc_tab.get(fam_name[dom], {}).get("hbm_bytes_per_launch_corrected")
    t_conv = sum(times) * 1e-3
    families = {}
    for kind, name in fam_name.items():
        ids = [i for i, k in enumerate(kinds) if k == kind]
        if ids:
            tt, ff = sum(times[i] for i in ids) * 1e-3, sum(flops[i] for i in ids)
            families[name] = {"launches_per_forward": len(ids), "ms_per_forward": 1e3 * tt,
                              "achieved": ff / tt / 1e12, "frac": ff / tt / 1e12 / peak,
                              "share_of_forward_flops": ff / sum(flops), "share_of_conv_time": tt / t_conv,
                              "traffic_over_algorithmic": pmc_tab.get(name, {}).get("traffic_over_algorithmic")}
    # what bounds the timed workload: the family with the largest share of the convolution TIME (at batch 1 that is not the
    # family with the most flops)
    bt_name = max(families, key=lambda k: families[k]["share_of_conv_time"])
    bt = families[bt_name]
    result["roofline"] = {
        "bound": "mfma", "kernel": fam_name[dom] if any(k == dom for k in kinds) else "conv_igemm_kernel",
        "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
        "traffic_source": pmc_src,
        "selected_by": "share of the forward's algorithmic flops",
        "by_time": {"kernel": bt_name, "selected_by": "share of the forward's convolution time", "share_of_conv_time": bt["share_of_conv_time"],
                    "achieved": bt["achieved"], "frac": bt["frac"], "ms_per_forward": bt["ms_per_forward"],
                    "launches_per_forward": bt["launches_per_forward"], "traffic_over_algorithmic": bt["traffic_over_algorithmic"],
                    "traffic_source": pmc_src},
        "launches_per_forward": len(sel), "avg_launch_us": 1e6 * t_dom / len(sel),
        "algorithmic_gflop_per_launch_avg": f_dom / len(sel) / 1e9,
        "share_of_forward_flops": f_dom / sum(flops),
        # every convolution kernel family of the forward (HIP-event time of its launches, eager replay):
        # conv_patch carries most of the flops, the latency-bound small-map kernel most of the batch-1 time
        "families": families,
        "all_conv": {"achieved": sum(flops) / t_conv / 1e12, "frac": sum(flops) / t_conv / 1e12 / peak,
                     "launches_per_forward": len(times), "ms_per_forward": 1e3 * t_conv},
        "forward_ms_eager": fwd_ms, "forward_ms_in_graph": 1e3 * elapsed / args.steps / evals_per_step,
    }

    if not args.no_extras and world == 1:        # side measurements belong to the single-GPU line
        extras = {}
        try:
            hp32 = HotPath(dev, torch.float32, args.N, 1, backbone=args.backbone)
            w1 = wave[:1]
            hp32.enhance(w1); torch.cuda.synchronize()
            t1 = time.perf_counter(); hp32.enhance(w1); hp32.enhance(w1); torch.cuda.synchronize()
            extras["fp32_parity_mode_rtf_b1"] = 2 * CLIP_SECONDS / (time.perf_counter() - t1)
            del hp32
            if args.dtype != "f32s":
                # the parity mode that runs on the 16-bit matrix pipe (f32 tensors, three f16 MFMAs per product): its own bench
                # line is `--dtype f32s`; here 5 timed clips beside the headline
                hps = HotPath(dev, torch.float32, args.N, 1, backbone=args.backbone, split=True)
                hps.enhance(w1); torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(5):
                    hps.enhance(w1)
                torch.cuda.synchronize()
                extras["f32s_parity_mode_rtf_b1"] = 5 * CLIP_SECONDS / (time.perf_counter() - t1)
                del hps
            if args.backbone == "ncsnpp_v2":
                extras.update(end_to_end_error_n30(dev, dtype if args.dtype != "f32s" else torch.bfloat16))
            if args.batch == 1 and not args.clips:
                hp64 = HotPath(dev, dtype, args.N, 64, backbone=args.backbone)
                w64 = synth_clips(64, 77, dev)
                hp64.enhance(w64); torch.cuda.synchronize()
                t1 = time.perf_counter(); hp64.enhance(w64); torch.cuda.synchronize()
                dt = time.perf_counter() - t1
                extras["batch64_rtf"] = 64 * CLIP_SECONDS / dt
                extras["batch64_whole_step_tflops"] = 64 * args.N * flops_fwd / dt / 1e12
                del hp64
            if args.batch == 1 and not args.clips:
                # BASELINE configs[4]: stochastic bridge sampler N=100 and predictor-corrector N=100 (euler_maruyama +
                # ald, 1 corrector step, snr 0.5), batch 16, fp16 storage - each one HIP graph; resident inputs
                hp16 = HotPath(dev, torch.float16, 100, 16, backbone=args.backbone)
                w16 = synth_clips(16, 99, dev)
                nf = hp16.fe.norm_factor(w16)
                Y16 = hp16.fe.spec_forward_padded(w16, hp16.pad_mode, norm=nf)
                import fdbm_amd
                for tag, st, kw, evals in (("sde_ei", "sde_ei", {}, 100),
                                           ("pc", "pc", dict(predictor_name="euler_maruyama", corrector_name="ald",
                                                             corrector_steps=1, snr=0.5, denoise=True), 200)):
                    br = fdbm_amd.Bridge("sb", N=100, sampler_type=st, noise_schedule="bb")
                    # every step's noise is drawn inside the timed call, on the device, by the consuming kernels (device_seed)
                    br.sampler(hp16.net, Y16, device_seed=1, **kw); torch.cuda.synchronize()
                    t1 = time.perf_counter(); br.sampler(hp16.net, Y16, device_seed=2, **kw); torch.cuda.synchronize()
                    dt = time.perf_counter() - t1
                    extras[f"config4_{tag}_n100_b16_f16_rtf"] = 16 * CLIP_SECONDS / dt
                    extras[f"config4_{tag}_n100_b16_f16_tflops"] = 16 * evals * flops_fwd / dt / 1e12
                del hp16
            if args.batch == 1 and args.backbone == "ncsnpp_v2" and args.dtype == "bf16":
                # the reference's other backbone family on the same clip: TF-GridNet (f32, eager sampler, fm/ot N = 30)
                for name in ("tfgridnet_5l32c100", "tfgridnet_4l32c80"):
                    hpt = HotPath(dev, torch.float32, args.N, 1, backbone=name, bridge="fm", schedule="ot")
                    w1 = wave[:1]
                    hpt.enhance(w1); torch.cuda.synchronize()
                    t1 = time.perf_counter(); hpt.enhance(w1); torch.cuda.synchronize()
                    extras[f"{name}_f32_rtf_b1"] = CLIP_SECONDS / (time.perf_counter() - t1)
                    del hpt
        except Exception as e:       # side measurements must never kill the headline line
            extras["error"] = repr(e)
        result["extras"] = extras

    if not args.no_cpu_baseline and world == 1:
        result["cpu_baseline"] = cpu_baseline(args.cpu_forwards, args.N)
    print(json.dumps(result))


if __name__ == "__main__":
    main()
