"""Reading the reference's checkpoints (the data format on the input side of the hot path).

A Lightning checkpoint of `BridgeModel` (fdbm/model.py) is a `torch.save`d dict with
  * `state_dict`        : `dnn.<backbone key>` tensors (the backbone is the only parameterised sub-module),
  * `hyper_parameters`  : the constructor arguments (`backbone`, `bridge`, `noise_schedule`, STFT settings ...),
  * `ema`               : torch_ema's state (`on_save_checkpoint`, model.py:143-144) whose `shadow_params` list
                          follows `parameters()` order - the weights inference uses (`eval()` copies them
                          over the raw ones, model.py:146-160).
Nothing here needs pytorch_lightning or torch_ema: the file is plain tensors, lists and dicts.
"""
import torch

from .arch import Spec, VARIANTS


# Parameters the reference creates with requires_grad=False: the fixed Fourier frequencies of the time embedding
# (GaussianFourierProjection.W, layerspp.py:34) - `all_modules.0.W` in NCSN++, `get_time_emb.W` in TF-GridNet.
FROZEN_KEYS = ("all_modules.0.W", "get_time_emb.W")


def ema_state_dict(order, shapes, shadow_params, base=None):
    """torch_ema `shadow_params` (a list in `parameters()` order) -> {backbone key: tensor}, ONE rule for every backbone
    family.  The reference builds `ExponentialMovingAverage(self.parameters())` over whatever backbone it holds
    (model.py:131-133); torch_ema releases differ in whether parameters with requires_grad=False get a shadow copy, and
    the repository pins no version, so both layouts are accepted and told apart by their LENGTH:
      * len == all parameters            -> `order` as it stands (unfiltered torch_ema),
      * len == trainable parameters only -> `order` without FROZEN_KEYS; those tensors come from `base` (the raw
                                            state_dict - they never change during training).
    Every tensor's shape is checked against `shapes` either way."""
    order = list(order)
    trainable = [k for k in order if k not in FROZEN_KEYS]
    if len(shadow_params) == len(order):
        keys, out = order, {}
    elif len(shadow_params) == len(trainable) and len(trainable) != len(order):
        keys = trainable
        frozen = [k for k in order if k in FROZEN_KEYS]
        if base is None or any(k not in base for k in frozen):
            raise ValueError(f"EMA holds the trainable parameters only; the frozen ones ({frozen}) must come from the state_dict")
        out = {k: base[k] for k in frozen}
    else:
        raise ValueError(f"EMA holds {len(shadow_params)} tensors; the backbone has {len(order)} parameters "
                         f"({len(trainable)} of them trainable)")
    for k, v in zip(keys, shadow_params):
        if tuple(v.shape) != tuple(shapes[k]):
            raise ValueError(f"EMA tensor for {k}: shape {tuple(v.shape)} != expected {tuple(shapes[k])}")
        out[k] = v
    return out


def load_lightning_checkpoint(ckpt, use_ema=True):
    """ckpt: path or already-loaded dict -> (hyper_parameters dict, {backbone key: tensor}).

    use_ema=True returns the EMA weights when the checkpoint carries them (what the reference evaluates
    with), the raw `state_dict` weights otherwise (the reference's `no_ema` path)."""
    if not isinstance(ckpt, dict):
        ckpt = torch.load(ckpt, map_location="cpu", weights_only=False)
    if "state_dict" not in ckpt:
        raise KeyError("not a Lightning checkpoint: no 'state_dict'")
    hp = dict(ckpt.get("hyper_parameters", {}))
    state = {k[4:]: v for k, v in ckpt["state_dict"].items() if k.startswith("dnn.")}
    if not state:
        raise KeyError("checkpoint state_dict has no 'dnn.*' tensors")
    name = hp.get("backbone", "ncsnpp_v2")
    from .tfgridnet import VARIANTS as TFG_VARIANTS, param_shapes as tfg_param_shapes
    if name not in VARIANTS and name not in TFG_VARIANTS:
        raise KeyError(f"backbone {name!r} is not one of {sorted(VARIANTS) + sorted(TFG_VARIANTS)}")
    ema = ckpt.get("ema") if use_ema else None
    if ema is not None and ema.get("shadow_params") is not None:
        if name in TFG_VARIANTS:
            shapes = tfg_param_shapes(**TFG_VARIANTS[name])         # parameters() order = state-dict order (no buffers)
            order = list(shapes)
        else:
            spec = Spec(**VARIANTS[name])
            shapes, order = spec.param_shapes(), spec.param_order()
        ema_state = ema_state_dict(order, shapes, ema["shadow_params"], base=state)
        state = dict(state, **ema_state)
    return hp, state


def backbone_from_checkpoint(ckpt, dtype=torch.bfloat16, device="cuda:0", use_ema=True):
    """-> (backbone, hyper_parameters): the registered HIP backbone named by the checkpoint, weights loaded."""
    from .registry import BackboneRegistry
    hp, state = load_lightning_checkpoint(ckpt, use_ema=use_ema)
    net = BackboneRegistry.get_by_name(hp.get("backbone", "ncsnpp_v2"))(state=state, dtype=dtype, device=device)
    return net, hp
