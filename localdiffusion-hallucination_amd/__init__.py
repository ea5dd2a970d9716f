"""MI355X-native local-diffusion sampling hot path.

Host-side mirror of the reference's ``Unet`` / ``GaussianDiffusion`` sampling surface
(/root/reference/ddpm.py:286-451, 496-1125) over hand-written gfx950 HIP kernels that are reached
through the C-ABI library ``csrc/liblocaldiff_hip.so`` (declared in ``include/localdiff_hip.h``).

Importing the package does not load the HIP library: pure-host helpers (``rng``, ``weights``,
``schedule``) work anywhere.  ``Unet`` / ``GaussianDiffusion`` load it on first use and raise
``RuntimeError`` if it is missing -- there is no CPU fallback on the product path.
"""
from . import rng, schedule, weights  # noqa: F401
from .weights import UnetConfig  # noqa: F401

__all__ = ["rng", "schedule", "weights", "UnetConfig", "Unet", "GaussianDiffusion"]


def __getattr__(name):
    if name == "Unet":
        from .unet import Unet
        return Unet
    if name == "GaussianDiffusion":
        from .diffusion import GaussianDiffusion
        return GaussianDiffusion
    raise AttributeError(name)
