"""vstnet_amd — MI355X-native (gfx950) inference hot path of CAP-VSTNet (RevResNet + cWCT).

``RevResNet`` / ``cWCT`` mirror the reference's ``models.RevResNet.RevResNet`` / ``models.cWCT.cWCT``;
all device work runs in hand-written HIP kernels behind the C ABI of include/vstnet.h.
"""
from .revresnet import RevResNet  # noqa: F401
from .cwct import cWCT  # noqa: F401
from ._lib import VstError, build, lib  # noqa: F401

__all__ = ["RevResNet", "cWCT", "VstError", "build", "lib"]
