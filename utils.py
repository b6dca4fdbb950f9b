"""Reference import path `from utils import ...` (ref:utils.py) -> HIP-backed host mirror."""
import importlib as _il
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
_m = _il.import_module("llm-speech-summarization_amd.utils")
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("_")})
