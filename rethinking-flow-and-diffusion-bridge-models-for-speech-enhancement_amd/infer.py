"""Enhancement drivers - the callers of the hot path (mirror of the reference's infer_folder.py / infer_single.py).

  python -m fdbm_amd.infer --ckpt model.ckpt --test_dir noisy/ --enhanced_dir out/ [--N 30]
         [--sampler_type ode_ei] [--sampler_kwargs "{...}"] [--keep_structure] [-D 0 1 2 ...] [-C config.yaml]
         [--batch 64]
  python -m fdbm_amd.infer --ckpt model.ckpt --noisy_file a.wav [--output_file a_enhanced.wav] ...

`-C file.yaml` reads the reference's config_infer_*.yaml files: `${key}` interpolation, `true` -> flag, null
skipped, and - as in the reference, which appends them to sys.argv (infer_folder.py:41-55) - config values win
over flags given on the command line.  The single-file mode clips to 0.5 instead of 0.95 (infer_single.py:99).

Same arguments and per-file procedure as infer_folder.py:69-148: load, resample to 16 kHz, normalise
("noisy": peak, "std": standard deviation), STFT + compression + time padding, `bridge.sampler`,
inverse transform, renormalise, clip to 0.95 if the peak exceeds 1, write.  One worker process per
listed GPU over contiguous chunks of the sorted file list (infer_folder.py:150-230), no communication.

`--batch B` (B > 1) is the batched form of the same thing (BASELINE configs[3]: 2 000 clips, batch 64, 8 GPUs): the
GPUs get length-balanced shards of the file list (fdbm_amd.dist.shard_by_length), each worker buckets its files by
padded spectrogram length and runs the sampler on up to B rows at a time (Enhancer.enhance_many); every file still gets
its own normalisation factor, STFT, padding, inverse transform and clip rule, so its result is the one-at-a-time result.

Differences, all on the audio-file side (the image has neither soundfile / torchaudio nor librosa): WAV is read and
written with scipy.io.wavfile (PCM 8/16/24/32-bit and float; output float32 WAV, lossless, where soundfile's default for
float arrays would be 16-bit PCM), FLAC is read by a decoder written from the format specification (fdbm_amd/flac.py:
every subframe type, both CRCs and the MD5 signature verified; checked against a test encoder, no libFLAC here), and
resampling uses scipy.signal.resample_poly (polyphase Kaiser FIR) instead of librosa's soxr - only for inputs that are
not 16 kHz already.
"""
import argparse
import ast
import glob
import os
from os.path import basename, dirname, join

import numpy as np
import torch

TARGET_SR = 16000


def get_audio_files(test_dir):
    """infer_folder.py:58-65 (same order: top-level first, then recursive; duplicates removed, order kept)."""
    files = []
    for pat in ("*.wav", join("**", "*.wav"), "*.flac", join("**", "*.flac")):
        files += sorted(glob.glob(join(test_dir, pat), recursive=True))
    seen, out = set(), []
    for f in files:
        if f not in seen:
            seen.add(f)
            out.append(f)
    return out


def read_wav(path):
    """-> (float32 mono-or-multichannel array [C, L] in [-1, 1], sample rate)."""
    from scipy.io import wavfile
    sr, x = wavfile.read(path)
    if x.dtype == np.int16:
        x = x.astype(np.float32) / 32768.0
    elif x.dtype == np.int32:
        x = x.astype(np.float32) / 2147483648.0
    elif x.dtype == np.uint8:
        x = (x.astype(np.float32) - 128.0) / 128.0
    else:
        x = x.astype(np.float32)
    x = x[None, :] if x.ndim == 1 else x.T
    return np.ascontiguousarray(x), int(sr)


def read_audio(path):
    """wav or flac (infer_folder.py:58-65,94 reads both through soundfile) -> (float32 [C, L], sample rate)."""
    if path.lower().endswith(".flac"):
        from .flac import read_flac
        return read_flac(path)
    return read_wav(path)


def resample_to(x, sr, target=TARGET_SR):
    if sr == target:
        return x
    from math import gcd
    from scipy.signal import resample_poly
    g = gcd(int(sr), int(target))
    return resample_poly(x, target // g, sr // g, axis=-1).astype(np.float32)


def write_wav(path, x, sr=TARGET_SR):
    from scipy.io import wavfile
    x = np.asarray(x, dtype=np.float32)
    wavfile.write(path, sr, x.T if x.ndim == 2 else x)


class Enhancer:
    """Checkpoint -> callable(waveform [C, L] float32 at 16 kHz) -> enhanced waveform, on one GPU."""

    def __init__(self, ckpt, device="cuda:0", N=30, sampler_type="ode_ei", sampler_kwargs=None,
                 dtype=torch.bfloat16, use_ema=True):
        from . import Bridge
        from .checkpoint import backbone_from_checkpoint
        from .frontend import SpecFrontend, pad_mode_for
        self.device = torch.device(device)
        self.net, hp = backbone_from_checkpoint(ckpt, dtype=dtype, device=device, use_ema=use_ema)
        self.hp = hp
        name = hp.get("backbone", "ncsnpp_v2")
        self.pad_mode = pad_mode_for(name)
        self.normalize = hp.get("normalize", "noisy")
        # like BridgeModel (model.py:47-52) every scalar hyper-parameter is offered to the bridge, which keeps the
        # ones it knows; N and sampler_type come from the command line (infer_folder.py:75-76)
        bkw = {k: v for k, v in hp.items() if isinstance(v, (int, float, str, bool))
               and k not in ("bridge", "backbone", "N", "sampler_type")}
        self.bridge = Bridge(hp.get("bridge", "sb"), N=N, sampler_type=sampler_type, **bkw)
        self.fe = SpecFrontend(n_fft=hp.get("n_fft", 510), hop_length=hp.get("hop_length", 128),
                               window=hp.get("window", "hann"), spec_factor=hp.get("spec_factor", 0.15),
                               spec_abs_exponent=hp.get("spec_abs_exponent", 0.5),
                               transform_type=hp.get("transform_type", "exponent"), device=device)
        self.sampler_kwargs = dict(sampler_kwargs or {})

    @torch.no_grad()
    def __call__(self, y, clip=0.95):
        y = torch.as_tensor(y, dtype=torch.float32)
        if y.dim() == 1:
            y = y[None]
        T_orig = y.shape[-1]
        y = y.to(self.device)
        self.fe.normalize = self.normalize
        # one factor for the whole file (infer_folder.py:102-107 reduce over every channel); the channels ride as the
        # batch.  Mono: the kernels' per-row factor IS the file's; more channels: the rows' maximum ("noisy") or the
        # file's standard deviation ("std") broadcast to every row
        nf = self.fe.norm_factor(y)
        if y.shape[0] > 1:
            nf = (nf.max() if self.normalize != "std" else y.std()).expand(y.shape[0]).contiguous()
        Y = self.fe.spec_forward_padded(y, self.pad_mode, norm=nf)        # [C,1,F,Tpad]
        sample = self.bridge.sampler(self.net, Y, **self.sampler_kwargs)
        if y.shape[0] == 1:
            x_hat = self.fe.to_audio(sample[:, 0], T_orig, norm=nf, clip=clip)       # renormalise + clip rule fused
        else:
            x_hat = self.fe.to_audio(sample[:, 0], T_orig, norm=nf)
            peak = x_hat.abs().max()                                                  # over all channels (infer_folder.py:119-121)
            if peak > 1.0:
                x_hat = x_hat / peak * clip
        return x_hat.cpu().numpy()


    @torch.no_grad()
    def enhance_many(self, waves, batch=8, clip=0.95, prior_noise_fn=None):
        """Batched form of __call__ for a list of waveforms ([L] or [C, L] float32 at 16 kHz) -> list of [C, L] arrays.

        Each file keeps the reference's per-file procedure (its own normalisation factor, STFT, time padding, inverse
        transform with its own length, clip rule): only the sampler runs batched.  Rows (one per channel of a file)
        are BUCKETED by padded frame count - a batch has one spectrogram shape - and buckets are processed longest
        first in batches of up to `batch` rows, so a mixed-length folder costs one program / HIP graph per distinct
        (rows, frames) instead of one per file.  A row's result does not depend on its batch mates (GroupNorm is per
        sample: tests/test_hip_parity.py::test_batch64_rows_match_batch1).  prior_noise_fn(file_index, channel, row_shape)
        -> complex tensor [1, F, Tpad]: the prior draw of one row (tests); default: the device generator, like the reference."""
        self.fe.normalize = self.normalize
        rows = []                                   # (frames, file index, channel, Y [1,F,Tpad], T_orig, nf scalar tensor)
        chans = []
        for fi, y in enumerate(waves):
            y = torch.as_tensor(y, dtype=torch.float32)
            if y.dim() == 1:
                y = y[None]
            chans.append(y.shape[0])
            y = y.to(self.device)
            nf = self.fe.norm_factor(y)
            if y.shape[0] > 1:
                nf = (nf.max() if self.normalize != "std" else y.std()).expand(y.shape[0]).contiguous()
            Y = self.fe.spec_forward_padded(y, self.pad_mode, norm=nf)            # [C,1,F,Tpad]
            for c in range(y.shape[0]):
                rows.append((int(Y.shape[-1]), fi, c, Y[c], int(y.shape[-1]), nf[c:c + 1]))
        order = sorted(range(len(rows)), key=lambda i: (-rows[i][0], rows[i][1], rows[i][2]))
        out = [[None] * c for c in chans]
        self.batch_shapes = []                      # (rows, frames) of every sampler call (for tests / logs)
        i = 0
        while i < len(order):
            frames = rows[order[i]][0]
            j = i
            while j < len(order) and j - i < batch and rows[order[j]][0] == frames:
                j += 1
            idx = order[i:j]
            Y = torch.stack([rows[k][3] for k in idx], 0)                          # [b,1,F,Tpad]
            kw = dict(self.sampler_kwargs)
            if prior_noise_fn is not None:
                kw["prior_noise"] = torch.stack([prior_noise_fn(rows[k][1], rows[k][2], tuple(rows[k][3].shape)) for k in idx], 0)
            sample = self.bridge.sampler(self.net, Y, **kw)
            self.batch_shapes.append((len(idx), frames))
            for r, k in enumerate(idx):
                _, fi, c, _, T_orig, nf = rows[k]
                mono = chans[fi] == 1
                x = self.fe.to_audio(sample[r:r + 1, 0], T_orig, norm=nf, clip=clip if mono else 0.0)
                out[fi][c] = x[0]
            i = j
        res = []
        for fi, per_c in enumerate(out):
            x_hat = torch.stack(per_c, 0)
            if len(per_c) > 1:
                peak = x_hat.abs().max()                                           # over all channels (infer_folder.py:119-121)
                if peak > 1.0:
                    x_hat = x_hat / peak * clip
            res.append(x_hat.cpu().numpy())
        return res


def output_path(noisy_file, args):
    if args.keep_structure:
        return join(args.enhanced_dir, os.path.relpath(noisy_file, args.test_dir))
    return join(args.enhanced_dir, basename(noisy_file))


def enhance_files(gpu_id, file_list, args, counter=None):
    """Worker: one GPU, its chunk of the file list (infer_folder.py:68-148).  Returns #files written."""
    enh = Enhancer(args.ckpt, device=f"cuda:{gpu_id}", N=args.N, sampler_type=args.sampler_type,
                   sampler_kwargs=args.sampler_kwargs, dtype=torch.float32 if args.fp32 else torch.bfloat16)
    done = 0
    for noisy_file in file_list:
        try:
            y, sr = read_audio(noisy_file)
            x_hat = enh(resample_to(y, sr))
            out = output_path(noisy_file, args)
            os.makedirs(dirname(out) or ".", exist_ok=True)
            write_wav(out, x_hat[0] if x_hat.shape[0] == 1 else x_hat)
            done += 1
        except Exception as e:      # like the reference: report, count, go on
            print(f"\nError processing {noisy_file} on GPU {gpu_id}: {e}")
        if counter is not None:
            with counter.get_lock():
                counter.value += 1
    return done


def wav_num_samples(path):
    """Length of a WAV file in samples at 16 kHz, from its header / size alone (for length-balanced sharding)."""
    try:
        import wave
        with wave.open(path, "rb") as w:
            return int(w.getnframes() * TARGET_SR / max(1, w.getframerate()))
    except Exception:                      # float WAVs and the like: the file size orders them well enough
        return os.path.getsize(path) // 2


def enhance_files_batched(gpu_id, file_list, args, counter=None):
    """Worker of the batched mode (--batch B > 1): its shard of the file list in windows of 4 B files, each window
    bucketed by padded length and enhanced B rows at a time (Enhancer.enhance_many).  Returns #files written."""
    enh = Enhancer(args.ckpt, device=f"cuda:{gpu_id}", N=args.N, sampler_type=args.sampler_type,
                   sampler_kwargs=args.sampler_kwargs, dtype=torch.float32 if args.fp32 else torch.bfloat16)
    done = 0
    win = max(1, 4 * args.batch)
    for w0 in range(0, len(file_list), win):
        names, waves = [], []
        for noisy_file in file_list[w0:w0 + win]:
            try:
                y, sr = read_audio(noisy_file)
                waves.append(resample_to(y, sr))
                names.append(noisy_file)
            except Exception as e:
                print(f"\nError processing {noisy_file} on GPU {gpu_id}: {e}")
        try:
            outs = enh.enhance_many(waves, batch=args.batch) if waves else []
        except Exception as e:
            # one bad row (e.g. a clip shorter than n_fft / 2) must not cost its whole window: like the reference and the
            # one-file-at-a-time path, only the offending file is skipped - the window is redone file by file
            print(f"\nA window of {len(waves)} files failed on GPU {gpu_id} ({e}); retrying its files one at a time")
            good_names, outs = [], []
            for noisy_file, w in zip(names, waves):
                try:
                    outs.append(enh(w))
                    good_names.append(noisy_file)
                except Exception as e1:
                    print(f"\nError processing {noisy_file} on GPU {gpu_id}: {e1}")
            names = good_names
        for noisy_file, x_hat in zip(names, outs):
            out = output_path(noisy_file, args)
            os.makedirs(dirname(out) or ".", exist_ok=True)
            write_wav(out, x_hat[0] if x_hat.shape[0] == 1 else x_hat)
            done += 1
        if counter is not None:
            with counter.get_lock():
                counter.value += len(file_list[w0:w0 + win])
    return done


def enhance_folder(args):
    from .dist import split_list, shard_by_length
    files = get_audio_files(args.test_dir)
    if not files:
        print(f"No audio files found in {args.test_dir}")
        return 0
    print(f"Found {len(files)} audio files")
    os.makedirs(args.enhanced_dir, exist_ok=True)
    gpus = [int(d) for d in args.device]
    print(f"Using {len(gpus)} GPU(s): {','.join(map(str, gpus))}")
    batched = getattr(args, "batch", 1) > 1
    worker = enhance_files_batched if batched else enhance_files
    if batched:
        # length-balanced shards (every GPU gets the same amount of audio, its files in descending length order so
        # that batch mates have the same padded length); the reference's contiguous chunks otherwise
        shards = shard_by_length([wav_num_samples(f) for f in files], len(gpus))
        chunks = [[files[i] for i in sh] for sh in shards]
    else:
        chunks = split_list(files, len(gpus))
    if len(gpus) == 1:
        n = worker(gpus[0], chunks[0], args)
    else:
        import torch.multiprocessing as mp
        ctx = mp.get_context("spawn")
        counter = ctx.Value("i", 0)
        procs = []
        for gpu, chunk in zip(gpus, chunks):
            if chunk:
                p = ctx.Process(target=worker, args=(gpu, chunk, args, counter))
                p.start()
                procs.append(p)
        for p in procs:
            p.join()
        n = counter.value
    print(f"Enhancement completed! Results saved to {args.enhanced_dir}")
    return n


def enhance_single(args):
    """infer_single.py:60-106: one file, default output next to the input with an _enhanced suffix, clip to 0.5."""
    enh = Enhancer(args.ckpt, device=f"cuda:{int(args.device[0])}", N=args.N, sampler_type=args.sampler_type,
                   sampler_kwargs=args.sampler_kwargs, dtype=torch.float32 if args.fp32 else torch.bfloat16)
    y, sr = read_audio(args.noisy_file)
    x_hat = enh(resample_to(y, sr), clip=0.5)
    out = args.output_file
    if not out:
        root, ext = os.path.splitext(args.noisy_file)
        out = root + "_enhanced" + (ext or ".wav")
    os.makedirs(dirname(out) or ".", exist_ok=True)
    write_wav(out, x_hat[0] if x_hat.shape[0] == 1 else x_hat)
    print(f"Enhanced audio saved to {out}")
    return out


def load_config(path):
    """YAML with OmegaConf-style `${key}` interpolation (top-level keys, resolved recursively)."""
    import re
    import yaml
    with open(path) as f:
        cfg = yaml.safe_load(f) or {}

    def resolve(v, depth=0):
        if isinstance(v, str):
            if depth > 20:
                raise ValueError(f"cyclic interpolation in {path}")
            return re.sub(r"\$\{([^}]+)\}", lambda m: str(resolve(cfg[m.group(1)], depth + 1)), v)
        return v

    return {k: resolve(v) for k, v in cfg.items()}


def config_argv(cfg):
    """The flags the reference would append to sys.argv for this config (infer_folder.py:41-55)."""
    out = []
    for k, v in cfg.items():
        if v is None:
            continue
        if isinstance(v, bool):
            if v:
                out.append(f"--{k}")
        else:
            out += [f"--{k}", str(v)]
    return out


def build_parser():
    p = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    p.add_argument("-C", "--config", default=None, type=str, help="config_infer_folder.yaml / config_infer_single.yaml")
    p.add_argument("-D", "--device", default=["0"], nargs="+", help="GPU indices, e.g. 0 1 2 3")
    p.add_argument("--test_dir", type=str, default=None)
    p.add_argument("--enhanced_dir", type=str, default=None)
    p.add_argument("--noisy_file", type=str, default=None)
    p.add_argument("--output_file", type=str, default=None)
    p.add_argument("--ckpt", type=str, required=True)
    p.add_argument("--sampler_type", type=str, default="ode_ei")
    p.add_argument("--sampler_kwargs", type=ast.literal_eval, default=None)
    p.add_argument("--N", type=int, default=30)
    p.add_argument("--keep_structure", action="store_true")
    p.add_argument("--fp32", action="store_true", help="f32 parity mode instead of bf16 storage")
    p.add_argument("--batch", type=int, default=1,
                   help="rows per sampler call: > 1 buckets the files by padded length, balances the GPUs by audio length "
                        "and runs the sampler batched (BASELINE configs[3]); 1 = one file at a time, like the reference")
    return p


def parse_args(argv):
    pre = argparse.ArgumentParser(add_help=False)
    pre.add_argument("-C", "--config", default=None)
    known, _ = pre.parse_known_args(argv)
    if known.config:
        argv = list(argv) + config_argv(load_config(known.config))      # appended: config wins, like the reference
    args, _ = build_parser().parse_known_args(argv)                       # unknown config keys (version, exp_dir ...) ignored
    if not args.noisy_file and not (args.test_dir and args.enhanced_dir):
        raise SystemExit("give --noisy_file, or --test_dir and --enhanced_dir")
    return args


def main(argv=None):
    import sys
    args = parse_args(sys.argv[1:] if argv is None else argv)
    return enhance_single(args) if args.noisy_file else enhance_folder(args)


if __name__ == "__main__":
    main()
