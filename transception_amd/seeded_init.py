"""Deterministic, name-seeded weights for MSTransception.

The reference checkpoint (189 MB fp32) cannot be committed and there is no
network, so every place that needs "the same weights on both sides" -- the
golden-fixture generator (which loads them into the imported reference with
``load_state_dict(strict=True)``), the CPU oracle, the HIP model, the bench --
derives them from the parameter *name*: each ``state_dict`` key gets its own
counter-based Philox stream keyed by a hash of its canonical name.  numpy's
Philox bit-stream is stable across numpy versions, unlike ``torch.manual_seed``.

Value ranges follow the initialisation that is actually in effect in the
reference (PyTorch defaults ~ U(-1/sqrt(fan_in), +))
so activations stay O(1) through the 54 residual blocks; norm scales/shifts and
BatchNorm running statistics are randomised (not 1/0) so that parity tests
exercise them.

Shared modules (``cpe``/``crpe`` appear under several ``state_dict`` aliases,
reference ``networks/MSTr.py:920-921,927,964-975``) must receive identical
values under every alias, hence the *canonical* name from the manifest.
"""
from __future__ import annotations

import gzip
import hashlib
import json
import os
from typing import Dict, Iterable, Tuple

import numpy as np

MANIFEST_PATH = os.path.join(os.path.dirname(__file__), "state_dict_manifest.json.gz")


def _stream(name: str, salt: int) -> np.random.Generator:
    digest = hashlib.sha256(f"{salt}:{name}".encode()).digest()
    key = int.from_bytes(digest[:8], "little")
    return np.random.Generator(np.random.Philox(key=key))


def seeded_array(name: str, shape: Tuple[int, ...], salt: int = 0) -> np.ndarray:
    """Values for one state_dict entry, chosen by the trailing field of its name."""
    leaf = name.rsplit(".", 1)[-1]
    shape = tuple(int(s) for s in shape)
    g = _stream(name, salt)
    if leaf == "num_batches_tracked":
        return np.zeros(shape, dtype=np.int64)
    if leaf == "running_mean":
        return g.uniform(-0.2, 0.2, size=shape).astype(np.float32)
    if leaf == "running_var":
        return g.uniform(0.6, 1.4, size=shape).astype(np.float32)
    if leaf == "weight" and len(shape) == 1:          # LayerNorm / BatchNorm scale
        return g.uniform(0.75, 1.25, size=shape).astype(np.float32)
    if leaf == "bias":
        return g.uniform(-0.1, 0.1, size=shape).astype(np.float32)
    if leaf == "weight":
        fan_in = int(np.prod(shape[1:]))
        bound = 1.0 / np.sqrt(max(fan_in, 1))
        return g.uniform(-bound, bound, size=shape).astype(np.float32)
    if leaf == "gamma":                               # CAM_Module's residual scale (a zero-initialised scalar in the reference: a nonzero value exercises the path)
        return g.uniform(0.3, 0.7, size=shape).astype(np.float32)
    raise ValueError(f"seeded_init: do not know how to fill '{name}'")


def load_manifest(path: str = MANIFEST_PATH):
    """[(key, shape, canonical_key)] in the reference's state_dict order."""
    with gzip.open(path, "rt") as f:
        doc = json.load(f)
    return [(e["key"], tuple(e["shape"]), e["canonical"]) for e in doc["entries"]]


def seeded_state_dict_numpy(entries: Iterable[Tuple[str, Tuple[int, ...], str]] | None = None,
                            salt: int = 0) -> Dict[str, np.ndarray]:
    entries = load_manifest() if entries is None else entries
    cache: Dict[str, np.ndarray] = {}
    out: Dict[str, np.ndarray] = {}
    for key, shape, canonical in entries:
        if canonical not in cache:
            cache[canonical] = seeded_array(canonical, shape, salt)
        out[key] = cache[canonical]
    return out


def seeded_state_dict(entries=None, salt: int = 0):
    """Same, as torch tensors (aliases share storage, like the reference's state_dict)."""
    import torch

    arrays = seeded_state_dict_numpy(entries, salt)
    seen: Dict[int, "torch.Tensor"] = {}
    out = {}
    for k, a in arrays.items():
        if id(a) not in seen:
            seen[id(a)] = torch.from_numpy(a.copy())
        out[k] = seen[id(a)]
    return out


def seeded_input(batch: int, in_ch: int = 1, size: int = 224, seed: int = 7) -> np.ndarray:
    """Synthetic Synapse-like slice, already normalised as trainer.py:89-92 does: (x-0.5)/0.5."""
    g = _stream(f"input:{batch}:{in_ch}:{size}", seed)
    x = g.uniform(0.0, 1.0, size=(batch, in_ch, size, size)).astype(np.float32)
    return (x - 0.5) / 0.5


def seeded_labels(batch: int, num_classes: int = 9, size: int = 224, seed: int = 7) -> np.ndarray:
    g = _stream(f"label:{batch}:{num_classes}:{size}", seed)
    return g.integers(0, num_classes, size=(batch, size, size), dtype=np.int64)


def seeded_tensor(tag: str, shape: Tuple[int, ...], scale: float = 1.0, seed: int = 11) -> np.ndarray:
    """Generic seeded N(0, scale) array for per-kernel parity inputs / upstream gradients."""
    g = _stream(f"tensor:{tag}", seed)
    return (g.standard_normal(size=tuple(int(s) for s in shape)) * scale).astype(np.float32)


def schema_entries(module) -> list:
    """[(key, shape, canonical_key)] of a torch module's state_dict, in its order: the canonical name of an entry is the first key
    that shares its storage (shared modules appear under several aliases).  For the default MSTransception this reproduces the
    committed manifest; the ablation variants (concat / have_bridge / br_ch_att_list) derive their schema this way on both sides --
    the fixture generator from the imported reference, the tests from the build's module -- and the fixture pins the result."""
    first = {}
    out = []
    for k, t in module.state_dict().items():
        ptr = (t.data_ptr(), tuple(t.shape), str(t.dtype)) if t.numel() else (id(t),)
        first.setdefault(ptr, k)
        out.append((k, tuple(int(v) for v in t.shape), first[ptr]))
    return out


def schema_digest(entries) -> str:
    """sha256 over 'key|shape|canonical' lines: what a fixture stores to pin a variant's state_dict schema."""
    h = hashlib.sha256()
    for k, shp, c in entries:
        h.update(f"{k}|{','.join(str(v) for v in shp)}|{c}\n".encode())
    return h.hexdigest()
