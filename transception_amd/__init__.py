"""transception_amd -- MI355X-native TransCeption (MSTransception) forward/backward path.

The package never imports the CPU oracle and has no CPU fallback: all arithmetic is HIP kernels in
libtransception_hip.so reached through the C ABI declared in include/transception_hip.h.
"""
from .model import MSTransception, TransCeption  # noqa: F401

__all__ = ["MSTransception", "TransCeption"]
