"""MI355X (gfx950) render path for the scene vocabulary of crates/core (reference: joeferner/rust-raytracer).

The product is `libcaustic_hip.so` (C ABI in include/caustic_hip.h, sources in csrc/);
`bindings` wraps it with ctypes and `scenes` restates the reference's scene definitions.
The directory name contains a hyphen, so import it with
`importlib.import_module("rust-raytracer_amd")`.
"""
from . import bindings, distributed, scenes  # noqa: F401
from .bindings import CausticError, load_gpu, gpu_device_count, write_png  # noqa: F401
