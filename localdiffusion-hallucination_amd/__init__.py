"""MI355X-native local-diffusion sampling hot path.

Host-side mirror of the reference's ``Unet`` / ``GaussianDiffusion`` sampling surface
(/root/reference/ddpm.py:286-451, 496-1125) over hand-written gfx950 HIP kernels that are reached
through the C-ABI library ``csrc/liblocaldiff_hip.so`` (declared in ``include/localdiff_hip.h``).

Importing the package does not load the HIP library: pure-host helpers (``rng``, ``weights``,
``schedule``) work anywhere.  ``Unet`` / ``GaussianDiffusion`` load it on first use and raise
``RuntimeError`` if it is missing -- there is no CPU fallback on the product path.
"""
import os as _os

# Runtime setting for HIP-graph replay (the default sampling regime replays one graph per reverse step and sub-batch):
# with ROCm 7.2's "graph packet capture" the replayed kernel nodes carry ~0.4 us more each on the GPU side -- cfg3's step
# 1.585 -> 1.570 ms with it off, a 64^2 x 4-patch step 0.818 -> 0.773 ms, no workload slower (DESIGN finding 47); the
# host then needs 300 us instead of 36 us per replay, still below the step.  Read by the HIP runtime when it
# initialises, i.e. it only takes effect if this package is imported before the first GPU call of the process; an
# explicit value in the environment wins.
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

from . import rng, schedule, weights  # noqa: F401,E402
from .weights import UnetConfig  # noqa: F401,E402

__all__ = ["rng", "schedule", "weights", "UnetConfig", "Unet", "GaussianDiffusion"]


def __getattr__(name):
    if name == "Unet":
        from .unet import Unet
        return Unet
    if name == "GaussianDiffusion":
        from .diffusion import GaussianDiffusion
        return GaussianDiffusion
    raise AttributeError(name)
