"""Import recipe for the upstream reference.

Only usable in the build container where /root/reference exists; nothing under
tests/ that runs on the GPU box may import this module.  It is used by the
fixture generators (make_manifest.py, make_golden.py) and nothing else.
"""
import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    if "torchvision" not in sys.modules:
        _stub("torchvision", models=_stub("torchvision.models"), transforms=_stub("torchvision.transforms"))
    if "torchinfo" not in sys.modules:
        _stub("torchinfo", summary=lambda *a, **k: None)
    if "medpy" not in sys.modules:
        _stub("medpy", metric=_stub("medpy.metric"))
    if "SimpleITK" not in sys.modules:
        _stub("SimpleITK")
    # networks/MSTr.py:1276 calls .cuda() on the default path
    torch.Tensor.cuda = lambda self, *a, **k: self
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from networks.MSTr import MSTransception  # noqa: E402
    from utils import DiceLoss  # noqa: E402
    return MSTransception, DiceLoss
