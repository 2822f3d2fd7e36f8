"""STFT front-end on the device: the drop-in for the reference's SpecsDataModule
stft / istft / spec_fwd / spec_back (fdbm/data_module.py:173-229), pad_spec
(fdbm/util/other.py:76-90) and BridgeModel's pass-throughs _stft / _istft /
_forward_transform / _backward_transform / to_audio (fdbm/model.py:376-389).

All arithmetic is in libfdbm_hip.so (csrc/frontend.hip); the drivers use the fused
forms: waveform -> padded compressed spectrogram in one launch, spectrogram -> waveform
in two.
"""
import torch

from . import hip

_TRANSFORM = {"none": 0, "exponent": 1, "log": 2}
_PAD = {"zero_pad": 0, "reflection": 1, "replication": 2}


def pad_mode_for(backbone_name):
    """infer_folder.py:83-88,111-112: 'reflection' iff the backbone is exactly 'ncsnpp_v2',
    'zero_pad' for other ncsnpp*, no padding otherwise."""
    if backbone_name == "ncsnpp_v2":
        return "reflection"
    if backbone_name.startswith("ncsnpp"):
        return "zero_pad"
    return None


class SpecFrontend:
    def __init__(self, n_fft=510, hop_length=128, window="hann", spec_factor=0.15,
                 spec_abs_exponent=0.5, transform_type="exponent", normalize="noisy",
                 device=None, **ignored):
        if not torch.cuda.is_available():
            raise RuntimeError("SpecFrontend needs a HIP device (no CPU fallback)")
        hip.lib()
        self.n_fft, self.hop_length = n_fft, hop_length
        self.spec_factor, self.spec_abs_exponent = spec_factor, spec_abs_exponent
        self.transform_type, self.normalize = transform_type, normalize
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        w = torch.hann_window(n_fft, periodic=True)          # get_window, data_module.py:13-19
        if window == "sqrthann":
            w = torch.sqrt(w)
        elif window != "hann":
            raise NotImplementedError(f"Window type {window} not implemented!")
        self.window = w.to(self.device)

    @property
    def bins(self):
        return self.n_fft // 2 + 1

    def _tcode(self, fused):
        t = _TRANSFORM[self.transform_type] if fused else 0
        if t == 1 and self.spec_abs_exponent == 1:
            # reference: factor only, magnitudes untouched (data_module.py:175-180)
            return 1
        return t

    def _stft_call(self, sig, Tpad, pad_mode, fused, norm=None):
        lead = sig.shape[:-1]
        L = sig.shape[-1]
        x = sig.reshape(-1, L).to(self.device, torch.float32).contiguous()
        B = x.shape[0]
        frames = 1 + L // self.hop_length
        Tpad = Tpad or frames
        out = torch.empty(B, self.bins, Tpad, dtype=torch.complex64, device=self.device)
        if norm is None:
            hip.call("fdbm_stft", hip.ptr(out), hip.ptr(x), hip.ptr(self.window), B, L, self.n_fft,
                     self.hop_length, frames, Tpad, pad_mode, self._tcode(fused), self.spec_factor,
                     self.spec_abs_exponent)
        else:
            assert norm.numel() == B and norm.dtype == torch.float32 and norm.is_cuda
            hip.call("fdbm_stft_norm", hip.ptr(out), hip.ptr(x), hip.ptr(self.window), hip.ptr(norm), B, L, self.n_fft,
                     self.hop_length, frames, Tpad, pad_mode, self._tcode(fused), self.spec_factor,
                     self.spec_abs_exponent)
        return out.reshape(*lead, self.bins, Tpad)

    def norm_factor(self, sig):
        """Per-clip normalisation factor of the drivers (infer_folder.py:102-107): max |y| for normalize == "noisy",
        torch.std(y) for "std".  -> f32 [B] on the device."""
        x = sig.reshape(-1, sig.shape[-1]).to(self.device, torch.float32).contiguous()
        nf = torch.empty(x.shape[0], dtype=torch.float32, device=self.device)
        mode = {"noisy": 0, "std": 1}[self.normalize]
        hip.call("fdbm_wave_norm_factor", hip.ptr(nf), hip.ptr(x), x.shape[0], x.shape[1], mode)
        return nf

    # ---- reference-named pieces -------------------------------------------------------
    def stft(self, sig):
        return self._stft_call(sig, None, 0, fused=False)

    def istft(self, spec, length=None):
        return self._istft_call(spec, length, fused=False)

    def _transform(self, spec, inverse):
        spec = spec.to(self.device).contiguous()
        out = torch.empty_like(spec)
        hip.call("fdbm_spec_transform", hip.ptr(out), hip.ptr(spec), spec.numel(), _TRANSFORM[self.transform_type],
                 self.spec_factor, self.spec_abs_exponent, int(inverse))
        return out

    def spec_fwd(self, spec):
        return self._transform(spec, False)

    def spec_back(self, spec):
        return self._transform(spec, True)

    def pad_spec(self, Y, mode="zero_pad"):
        T = Y.shape[-1]
        n = (64 - T % 64) % 64
        Y = Y.to(self.device).contiguous()
        out = torch.empty(*Y.shape[:-1], T + n, dtype=Y.dtype, device=self.device)
        hip.call("fdbm_pad_spec", hip.ptr(out), hip.ptr(Y), Y.numel() // T, T, T + n, _PAD[mode])
        return out

    # ---- fused forms used by the drivers ------------------------------------------------
    def spec_forward_padded(self, sig, pad_mode="reflection", norm=None):
        """[B, L] (or [L]) waveform -> complex64 [B,1,bins,Tpad]: [y / norm +] stft + spec_fwd + pad_spec
        (infer_folder.py:106-112) in one launch.  pad_mode None: no padding; norm: f32 [B] from norm_factor()."""
        sig2 = sig.reshape(-1, sig.shape[-1])
        frames = 1 + sig2.shape[-1] // self.hop_length
        Tpad = frames + ((64 - frames % 64) % 64 if pad_mode else 0)
        Y = self._stft_call(sig2, Tpad, _PAD.get(pad_mode, 0), fused=True, norm=norm)
        return Y[:, None]

    def _istft_call(self, spec, length, fused, norm=None, clip=0.0):
        spec = spec.to(self.device)
        lead = spec.shape[:-2]
        bins, Tp = spec.shape[-2], spec.shape[-1]
        assert bins == self.bins, (bins, self.bins)
        s = spec.reshape(-1, bins, Tp).contiguous()
        B = s.shape[0]
        if length is None:
            length = self.hop_length * (Tp - 1)
        ws = torch.empty(B, Tp, self.n_fft, device=self.device)
        out = torch.empty(B, length, device=self.device)
        if norm is None:
            hip.call("fdbm_istft", hip.ptr(out), hip.ptr(s), hip.ptr(self.window), hip.ptr(ws), B, length,
                     self.n_fft, self.hop_length, Tp, Tp, self._tcode(fused), self.spec_factor,
                     self.spec_abs_exponent)
        else:
            assert norm.numel() == B and norm.dtype == torch.float32 and norm.is_cuda
            peak = torch.empty(B, dtype=torch.float32, device=self.device)
            hip.call("fdbm_istft_renorm", hip.ptr(out), hip.ptr(s), hip.ptr(self.window), hip.ptr(ws), hip.ptr(norm),
                     hip.ptr(peak), float(clip), B, length, self.n_fft, self.hop_length, Tp, Tp, self._tcode(fused),
                     self.spec_factor, self.spec_abs_exponent)
        return out.reshape(*lead, length)

    def to_audio(self, spec, length=None, norm=None, clip=0.0):
        """spec_back + istft (model.py:376-377) [+ x_hat * norm and the drivers' clip rule: if max |x_hat| > 1 then
        x_hat / max |x_hat| * clip (0.95 infer_folder.py:121, 0.5 infer_single.py:99); norm: f32 [B]]."""
        return self._istft_call(spec, length, fused=True, norm=norm, clip=clip)
