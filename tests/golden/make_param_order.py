#!/usr/bin/env python3
"""Fixture: the order of `parameters()` of the reference backbones (what torch_ema's `shadow_params`
list follows in a Lightning checkpoint of BridgeModel, fdbm/model.py:56,143-144).  Run in the build
container (imports /root/reference, same inert stubs as make_golden.py); writes param_order.json."""
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, "/root/reference")
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

from fdbm.backbones import BackboneRegistry  # noqa: E402

out = {}
for name in ("ncsnpp_v2", "ncsnpp_v2_5M", "ncsnpp_v2_16M", "ncsnpp_v2_37M"):
    net = BackboneRegistry.get_by_name(name)()
    keys = [k for k, _ in net.named_parameters()]
    assert keys == list(net.state_dict().keys()), "buffers present: parameters() and state_dict() orders differ"
    out[name] = keys
json.dump(out, open(os.path.join(HERE, "param_order.json"), "w"))
print({k: len(v) for k, v in out.items()})
