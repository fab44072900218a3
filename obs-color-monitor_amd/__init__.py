"""obs-color-monitor_amd -- MI355X-native scope accumulation for obs-color-monitor.

Layout:
  csrc/          hand-written gfx950 HIP kernels (scope_kernels.hip) + the C-ABI (scope_api.hip)
  host/          C mirror of the plugin's scope sources calling the C-ABI (drop-in boundary)
  scope_hip.py   ctypes front end of libscope_hip.so for tests / bench

The directory name carries the reference's hyphen; `import obs_color_monitor_amd` (shim at the
repo root) resolves to this package.
"""
from .scope_hip import *  # noqa: F401,F403
from .scope_hip import FramePool, ScopeContext, ScopeError, ScopePipe, load_library, DEFAULT_LIB, ABI_SYMBOLS  # noqa: F401
