"""Build libaqe_hip.so (hipcc, gfx950 only) in-tree: approximatequeryengine_amd/lib/libaqe_hip.so.

    python -m approximatequeryengine_amd.build [--force]

hipcc cross-compiles without a GPU.  The library links only the HIP runtime (no torch, no pybind11).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
LIB = PKG / "lib" / "libaqe_hip.so"
SOURCES = [CSRC / "capi.hip", CSRC / "table.hip", CSRC / "plans.hip", CSRC / "kernels.hip", CSRC / "persist.hip", CSRC / "lean.hip", CSRC / "grouped.hip", CSRC / "sort.hip", CSRC / "comm.hip", CSRC / "mailbox.hip", CSRC / "planner.cpp"]
HEADERS = [CSRC / "host.hpp", CSRC / "kernels.hpp", CSRC / "device_common.hpp", CSRC / "planner.hpp", ROOT / "include" / "aqe_hip.h"]
ARCH = "gfx950"


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found: libaqe_hip.so cannot be built (ROCm toolchain required)")
    return exe


def is_stale() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any(p.stat().st_mtime > t for p in SOURCES + HEADERS + [Path(__file__)])


def build_native(force: bool = False, verbose: bool = False) -> Path:
    """Compile the HIP extension if missing or older than its sources; returns the .so path."""
    if not force and not is_stale():
        return LIB
    LIB.parent.mkdir(parents=True, exist_ok=True)
    tmp = LIB.with_suffix(".so.tmp%d" % os.getpid())
    cmd = [hipcc(), "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-shared", "-fvisibility=hidden",
           "-Wall", "-Wno-unused-function", "-fno-fast-math", "-ffp-contract=off", "-pthread", "-ldl",
           "-I", str(ROOT / "include"), "-o", str(tmp)] + [str(s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
