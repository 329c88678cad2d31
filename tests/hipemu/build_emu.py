"""TEST ONLY: build the CPU-fiber emulation of libicl_hip (same C ABI, host pointers)."""
import fcntl
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "_build", "libicl_emu.so")


def build_emu() -> str:
    src = [os.path.join(HERE, "icl_emu.cpp"), os.path.join(HERE, "hipemu.h"),
           os.path.join(ROOT, "include", "icl_hip.h"), os.path.join(ROOT, "icl_amd", "csrc", "icl_abi.inc")]
    kd = os.path.join(ROOT, "icl_amd", "csrc", "kernels")
    src += [os.path.join(kd, f) for f in os.listdir(kd)]
    def fresh():
        return os.path.exists(OUT) and all(os.path.getmtime(s) <= os.path.getmtime(OUT) for s in src)

    if fresh():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT + ".lock", "w") as lock:      # parallel test workers (pytest -n): one of them builds, the others wait
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not fresh():
            tmp = f"{OUT}.{os.getpid()}.tmp"
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread", "-Wno-psabi",
                                   "-I", os.path.join(ROOT, "icl_amd", "csrc"), "-o", tmp, src[0]])
            os.replace(tmp, OUT)
    return OUT
