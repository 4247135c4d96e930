"""Drop-in for the reference's models/cWCT.py: same import path, class name and call surface."""
from vstnet_amd.cwct import cWCT  # noqa: F401
