"""wurblpt_amd -- MI355X-native path-tracing core behind WurblPT's mcpt() boundary.

  include/wurblpt_hip.h      the C ABI (drop-in boundary)
  include/wurblpt/*.hpp      host-side C++ API mirroring the reference's classes
  wurblpt_amd/csrc/          HIP kernels for gfx950 and the C ABI implementation
  wurblpt_amd/host/          C entry points over the host C++ API (scene factory)
  wurblpt_amd/device.py      Python binding of the C ABI (torch supplies device memory)
"""
from . import _abi, host  # noqa: F401
