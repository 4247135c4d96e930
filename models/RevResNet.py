"""Drop-in for the reference's models/RevResNet.py: same import path, class name and call surface."""
from vstnet_amd.revresnet import RevResNet, residual_block, channel_reduction  # noqa: F401
