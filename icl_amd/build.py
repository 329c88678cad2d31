"""Build libicl_hip.so (gfx950) in-tree with hipcc.  `python -m icl_amd.build`."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libicl_hip.so")


def _sources():
    out = [os.path.join(CSRC, "icl_hip.hip"), os.path.join(CSRC, "icl_abi.inc"),
           os.path.join(CSRC, "device_env_hip.h"), os.path.join(HERE, "..", "include", "icl_hip.h")]
    kd = os.path.join(CSRC, "kernels")
    out += [os.path.join(kd, f) for f in sorted(os.listdir(kd)) if f.endswith(".h")]
    return out


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in _sources())


# the compile flags of the shipped library (tests/test_host_logic.py scans the device assembly built with exactly these)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17"]


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, *FLAGS, "-shared", "-fPIC", "-I", CSRC, "-o", LIB, os.path.join(CSRC, "icl_hip.hip")]
    if verbose:
        print("[icl_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
