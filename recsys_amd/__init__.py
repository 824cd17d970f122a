"""recsys_amd — MI355X-native embedding-lookup + interaction hot path.

Drop-in modules with the reference's constructor kwargs, parameter names and
``forward(inputs) -> [B, N]`` / ``loss(inputs)`` contract live in ``recsys_amd.models``; they run
on hand-written HIP kernels behind the C ABI of ``include/recsys_hip.h`` (``librecsys_hip.so``).
Importing the package loads that library and raises if it has not been built.
"""
from . import _lib  # noqa: F401  (fails loudly when librecsys_hip.so is missing)

__all__ = ["_lib"]
