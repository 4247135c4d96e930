"""Build a VARIANT of the library for same-box A/B timing (never the shipped one: __graft_entry__.build() passes no -D flag).

    python tools/ab_build.py vstnet_amd/libvstnet_hip_ab.so -DVST_NO_RANGE_CHECK=1 [-D...]

Run a benchmark against it with VSTNET_HIP_LIB=<path> (vstnet_amd/_lib.py honours the override).
"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from vstnet_amd import _lib  # noqa: E402

if __name__ == "__main__":
    out, flags = sys.argv[1], sys.argv[2:]
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17",
           "-Xarch_host", "-fvisibility=hidden", "-I", os.path.join(REPO, "include"), "-o", out] + flags + _lib.SOURCES
    print(" ".join(cmd))
    subprocess.run(cmd, check=True)
