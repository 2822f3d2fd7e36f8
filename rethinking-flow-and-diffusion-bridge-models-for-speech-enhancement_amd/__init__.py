"""fdbm_amd: MI355X-native reverse-sampling path for flow / diffusion-bridge
speech enhancement (drop-in for the reference's fdbm.bridge + fdbm.backbones
inference path).  Import as ``fdbm_amd`` (see ../fdbm_amd.py).
"""
from .registry import (Registry, BridgeRegistry, BackboneRegistry,  # noqa: F401
                       PredictorRegistry, CorrectorRegistry)
from .bridge import Bridge, NoiseSource, complex_randn  # noqa: F401
from . import backbone as _backbone  # noqa: F401  (registers ncsnpp_v2 / _5M / _16M / _37M)
from . import tfgridnet as _tfgridnet  # noqa: F401  (registers tfgridnet_5l32c100 / tfgridnet_4l32c80)

__all__ = ["Registry", "BridgeRegistry", "BackboneRegistry", "PredictorRegistry",
           "CorrectorRegistry", "Bridge", "NoiseSource", "complex_randn"]
