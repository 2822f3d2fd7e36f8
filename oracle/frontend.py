"""Oracle: STFT front-end, spectral compression, time padding (CPU, torch).

Restates, without torch.stft/istft, what the reference gets from them:
  * get_window            fdbm/data_module.py:13-19
  * SpecsDataModule.stft  fdbm/data_module.py:223-225  (kwargs :201-210)
  * SpecsDataModule.istft fdbm/data_module.py:227-229
  * spec_fwd / spec_back  fdbm/data_module.py:173-199
  * pad_spec              fdbm/util/other.py:76-90
Closed forms per SURVEY.md 8(a2): centred, reflect-padded frames, one-sided rfft;
inverse = windowed irfft frames, overlap-add, divide by the window-square
envelope, trim n_fft/2 and cut to `length`.
"""
import torch


def get_window(window_type, n):
    w = torch.hann_window(n, periodic=True)
    if window_type == "sqrthann":
        return torch.sqrt(w)
    if window_type == "hann":
        return w
    raise NotImplementedError(window_type)


def stft(sig, n_fft=512, hop=256, window="sqrthann"):
    """sig [..., L] float32 -> complex64 [..., n_fft//2+1, 1 + L//hop]."""
    w = get_window(window, n_fft)
    lead = sig.shape[:-1]
    x = sig.reshape(-1, 1, sig.shape[-1])
    x = torch.nn.functional.pad(x, (n_fft // 2, n_fft // 2), mode="reflect")[:, 0]
    frames = x.unfold(-1, n_fft, hop)                  # [N, frames, n_fft]
    spec = torch.fft.rfft(frames * w, dim=-1)          # [N, frames, bins]
    return spec.transpose(-1, -2).reshape(*lead, n_fft // 2 + 1, -1)


def istft(spec, length, n_fft=512, hop=256, window="sqrthann"):
    """complex64 [..., bins, frames] -> float32 [..., length]."""
    w = get_window(window, n_fft)
    lead = spec.shape[:-2]
    s = spec.reshape(-1, spec.shape[-2], spec.shape[-1]).transpose(-1, -2)   # [N, frames, bins]
    frames = torch.fft.irfft(s, n=n_fft, dim=-1) * w                          # [N, frames, n_fft]
    n_frames = frames.shape[1]
    total = n_fft + hop * (n_frames - 1)
    out = torch.zeros(frames.shape[0], total, dtype=frames.dtype)
    env = torch.zeros(total, dtype=frames.dtype)
    wsq = w * w
    for k in range(n_frames):
        out[:, k * hop:k * hop + n_fft] += frames[:, k]
        env[k * hop:k * hop + n_fft] += wsq
    start = n_fft // 2
    out = out[:, start:start + length] / env[start:start + length]
    return out.reshape(*lead, length)


def spec_fwd(spec, transform_type="exponent", spec_factor=0.15, spec_abs_exponent=0.5):
    if transform_type == "exponent":
        if spec_abs_exponent != 1:
            spec = spec.abs() ** spec_abs_exponent * torch.exp(1j * spec.angle())
        return spec * spec_factor
    if transform_type == "log":
        return torch.log(1 + spec.abs()) * torch.exp(1j * spec.angle()) * spec_factor
    if transform_type == "none":
        return spec
    raise NotImplementedError(transform_type)


def spec_back(spec, transform_type="exponent", spec_factor=0.15, spec_abs_exponent=0.5):
    if transform_type == "exponent":
        spec = spec / spec_factor
        if spec_abs_exponent != 1:
            spec = spec.abs() ** (1 / spec_abs_exponent) * torch.exp(1j * spec.angle())
        return spec
    if transform_type == "log":
        spec = spec / spec_factor
        return (torch.exp(spec.abs()) - 1) * torch.exp(1j * spec.angle())
    if transform_type == "none":
        return spec
    raise NotImplementedError(transform_type)


def pad_spec(Y, mode="zero_pad"):
    """Right-pad the time axis of [B,1,F,T] to a multiple of 64."""
    T = Y.shape[3]
    n = (64 - T % 64) % 64
    if n == 0:
        return Y
    if mode == "zero_pad":
        return torch.cat([Y, torch.zeros_like(Y[..., :n])], dim=3)
    if mode == "reflection":
        idx = torch.arange(T - 2, T - 2 - n, -1)
        return torch.cat([Y, Y[..., idx]], dim=3)
    if mode == "replication":
        return torch.cat([Y, Y[..., -1:].expand(*Y.shape[:3], n)], dim=3)
    raise NotImplementedError(mode)


def pad_mode_for(backbone_name):
    """infer_folder.py:83-88,111-112: reflection iff the name is exactly 'ncsnpp_v2'."""
    if backbone_name == "ncsnpp_v2":
        return "reflection"
    if backbone_name.startswith("ncsnpp"):
        return "zero_pad"
    return None


def norm_factor(y, normalize="noisy"):
    """infer_folder.py:102-105 / infer_single.py:79-83: max |y| ('noisy') or y.std() ('std') over the whole file."""
    if normalize == "noisy":
        return y.abs().max()
    if normalize == "std":
        return y.std()
    raise ValueError(normalize)


def renormalize(x_hat, nf, clip=0.95):
    """infer_folder.py:118-121 (clip 0.95) / infer_single.py:97-99 (clip 0.5)."""
    x_hat = x_hat * nf
    if x_hat.abs().max() > 1.0:
        x_hat = x_hat / x_hat.abs().max() * clip
    return x_hat
