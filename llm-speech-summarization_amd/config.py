"""YAML config loader: accepts the reference's config schema verbatim (ref:config/llama3_hubert.yaml:1-47;
the reference loads it with OmegaConf, ref:inference.py:158) and exposes it with attribute access."""
from __future__ import annotations

import re

import yaml

# PyYAML (YAML 1.1) reads `5e-5` as a string; OmegaConf, which the reference uses, reads it as a float
# (ref:config/llama3_hubert.yaml:30 `lr: 5e-5`).  Resolve such scalars the OmegaConf way.
_FLOAT_RE = re.compile(r"^[+-]?(\d+\.?\d*|\.\d+)[eE][+-]?\d+$")


class AttrDict(dict):
    """dict with attribute access, nested, like the OmegaConf nodes the reference passes around."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(o):
    if isinstance(o, dict):
        return AttrDict({k: _wrap(v) for k, v in o.items()})
    if isinstance(o, list):
        return [_wrap(v) for v in o]
    if isinstance(o, str) and _FLOAT_RE.match(o):
        return float(o)
    return o


def load_config(path: str) -> AttrDict:
    with open(path) as f:
        return _wrap(yaml.safe_load(f))


def from_dict(d: dict) -> AttrDict:
    return _wrap(d)
