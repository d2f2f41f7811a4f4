"""pyopenvino_amd: MI355X (gfx950) back end for the pyopenvino per-layer ``compute()`` hot path.

``inference_engine`` mirrors the reference's IECore / IENetwork / Executable_Network; ``op_plugins``
holds one module per IR layer type exposing ``compute(node, inputs, kernel_type, debug)`` whose numeric
body is a hand-written HIP kernel reached through the C ABI of ``libpvhip.so`` (``include/pvhip.h``).
"""
from .inference_engine import IECore, IENetwork, Executable_Network  # noqa: F401

__all__ = ['IECore', 'IENetwork', 'Executable_Network']
