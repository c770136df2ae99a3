"""Developer A/B runs only: point the package's ctypes loader at a variant library
(`make -C watersurfacerendering_amd/csrc variant NAME=x DEFS=...` -> libocean_hip_x.so) named by the
environment variable OCEAN_HIP_LIB.  Imported FIRST by the scripts under tools/; the shipped package
itself reads no such variable (watersurfacerendering_amd/_abi.py loads the in-tree library only)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from watersurfacerendering_amd import _abi  # noqa: E402

_p = os.environ.get("OCEAN_HIP_LIB")
if _p:
    _abi.LIB_PATH = os.path.abspath(_p)
