"""bridges_hip: Python binding of libbridges_hip.so (hand-written gfx950 kernels for the
assembly_gym stability / rasteriser hot path and the successor-DQN target ops)."""
from .abi import BridgesHipError, lib, require_gpu  # noqa: F401
