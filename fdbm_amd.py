"""Import alias for the product package.

The package directory carries the (hyphenated, hence not `import`-able) name the
build contract prescribes:
``rethinking-flow-and-diffusion-bridge-models-for-speech-enhancement_amd/``.
This one-file loader registers that directory as the importable package
``fdbm_amd`` (same short name the reference uses, ``fdbm``, plus ``_amd``), so
``import fdbm_amd.bridge`` resolves to files inside the hyphenated directory.
"""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(
    os.path.dirname(os.path.abspath(__file__)),
    "rethinking-flow-and-diffusion-bridge-models-for-speech-enhancement_amd",
)
_spec = importlib.util.spec_from_file_location(
    "fdbm_amd",
    os.path.join(_PKG_DIR, "__init__.py"),
    submodule_search_locations=[_PKG_DIR],
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["fdbm_amd"] = _mod
_spec.loader.exec_module(_mod)
