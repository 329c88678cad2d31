"""Build libicl_hip.so (gfx950) in-tree with hipcc.  `python -m icl_amd.build`."""
from __future__ import annotations

import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libicl_hip.so")

# the compile flags of the shipped library (tests/test_host_logic.py scans the device assembly built with exactly these)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17"]
# translation units: (source, extra flags).  The second one holds the loader-wave forward convolution, compiled with hipcc's SLP
# vectoriser off (csrc/icl_hip_noslp.hip says why; profiles/r6_pk_add_ab.txt has the measurement)
UNITS = [("icl_hip.hip", []), ("icl_hip_noslp.hip", ["-fno-slp-vectorize"])]


def _sources():
    out = [os.path.join(CSRC, u) for u, _ in UNITS] + [os.path.join(CSRC, "icl_abi.inc"),
           os.path.join(CSRC, "device_env_hip.h"), os.path.join(HERE, "..", "include", "icl_hip.h"), os.path.abspath(__file__)]
    kd = os.path.join(CSRC, "kernels")
    out += [os.path.join(kd, f) for f in sorted(os.listdir(kd)) if f.endswith(".h")]
    return out


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in _sources())


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory(prefix="icl_build_") as tmp:
        procs, objs = [], []
        for src, extra in UNITS:
            obj = os.path.join(tmp, src.replace(".hip", ".o"))
            cmd = [hipcc, *FLAGS, *extra, "-fPIC", "-c", "-I", CSRC, "-o", obj, os.path.join(CSRC, src)]
            if verbose:
                print("[icl_amd.build]", " ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))      # the units compile side by side
            objs.append(obj)
        for cmd, p in procs:
            if p.wait() != 0:
                raise subprocess.CalledProcessError(p.returncode, cmd)
        link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        if verbose:
            print("[icl_amd.build]", " ".join(link), flush=True)
        subprocess.check_call(link)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
