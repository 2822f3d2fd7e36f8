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


def ema_state_dict(spec, shadow_params):
    """torch_ema `shadow_params` (list in `parameters()` order) -> {backbone key: tensor}."""
    order = spec.param_order()
    shapes = spec.param_shapes()
    if len(shadow_params) != len(order):
        raise ValueError(f"EMA holds {len(shadow_params)} tensors, the backbone has {len(order)} parameters")
    out = {}
    for k, v in zip(order, shadow_params):
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
            # TF-GridNet: parameters() order = state-dict order (no buffers); the EMA tracks the trainable ones, i.e. all but
            # the fixed Fourier frequencies get_time_emb.W (requires_grad=False, layerspp.py:34), which stay as stored
            shapes = tfg_param_shapes(**TFG_VARIANTS[name])
            order = [k for k in shapes if k != "get_time_emb.W"]
            sp = ema["shadow_params"]
            if len(sp) != len(order):
                raise ValueError(f"EMA holds {len(sp)} tensors, the backbone has {len(order)} trainable parameters")
            for k, v in zip(order, sp):
                if tuple(v.shape) != tuple(shapes[k]):
                    raise ValueError(f"EMA tensor for {k}: shape {tuple(v.shape)} != expected {tuple(shapes[k])}")
                state[k] = v
        else:
            state = ema_state_dict(Spec(**VARIANTS[name]), ema["shadow_params"])
    return hp, state


def backbone_from_checkpoint(ckpt, dtype=torch.bfloat16, device="cuda:0", use_ema=True):
    """-> (backbone, hyper_parameters): the registered HIP backbone named by the checkpoint, weights loaded."""
    from .registry import BackboneRegistry
    hp, state = load_lightning_checkpoint(ckpt, use_ema=use_ema)
    net = BackboneRegistry.get_by_name(hp.get("backbone", "ncsnpp_v2"))(state=state, dtype=dtype, device=device)
    return net, hp
