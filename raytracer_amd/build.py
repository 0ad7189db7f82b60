"""Builds librt_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "librt_hip.so")
SOURCES = ["rt_api.cpp", "rt_kernels.hip"]
HEADERS = ["rt_device.h", "filter_asm.inc", os.path.join("..", "..", "include", "rt_hip.h")]

# -ffp-contract=off and IEEE divide/sqrt are part of the numerical contract of the
# library (bit-level agreement with a strict-FP32 evaluation), not tuning knobs.
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
    # hipcc 7.2's SLP vectoriser crashes the gfx950 backend on the trace kernel
    # (RenameIndependentSubregs) and costs VGPRs elsewhere; packed math is written by hand
    # where it is wanted.
    "-fno-slp-vectorize",
    "-Wall", "-Wno-unused-function",
]


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(p) > t for p in deps)


def build_library(force: bool = False, verbose: bool = False, extra_flags=()) -> str:
    if not force and not needs_build():
        return LIB
    cmd = [hipcc(), *FLAGS, *extra_flags, "-x", "hip", *[os.path.join(CSRC, s) for s in SOURCES],
           "-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB


TOOLS_DIR = os.path.join(os.path.dirname(HERE), "tools")


def build_tools(force: bool = False) -> None:
    """Standalone gfx950 programs under tools/ (microbenchmarks)."""
    for src in ("valu_bench.hip", "valu_issue_bench.hip", "pk_loop_bench.hip", "filter_loop_bench.hip", "prio_bench.hip"):
        path = os.path.join(TOOLS_DIR, src)
        exe = path[: -len(".hip")]
        if not os.path.exists(path):
            continue
        if not force and os.path.exists(exe) and os.path.getmtime(exe) >= os.path.getmtime(path):
            continue
        subprocess.run([hipcc(), "--offload-arch=gfx950", "-O3", "-ffp-contract=off",
                        "-fno-slp-vectorize", "-Wno-unused-result", path, "-o", exe], check=True)


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
