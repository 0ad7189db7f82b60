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
# What each source includes (rt_kernels.hip takes minutes to compile: it must not be rebuilt
# for a change of the public header, which it does not see).
HEADERS = {
    "rt_api.cpp": ["rt_device.h", os.path.join("..", "..", "include", "rt_hip.h")],
    "rt_kernels.hip": ["rt_device.h", "filter_asm.inc"],
}

# -ffp-contract=off and IEEE divide/sqrt are part of the numerical contract of the
# library (bit-level agreement with a strict-FP32 evaluation), not tuning knobs.
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
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


OBJ_DIR = os.path.join(CSRC, "build")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(p) > t for p in deps)


def _deps(source: str):
    return [os.path.join(CSRC, f) for f in [source] + HEADERS[source]] + [os.path.abspath(__file__)]


def needs_build() -> bool:
    return any(_stale(LIB, _deps(s)) for s in SOURCES)


def build_library(force: bool = False, verbose: bool = False, extra_flags=(), out: str = LIB) -> str:
    """One object per source (the kernels take minutes, the host side seconds), then the link.
    ``extra_flags`` / ``out`` build a variant next to the in-tree library (its objects are not kept)."""
    variant = bool(extra_flags) or out != LIB
    if not force and not variant and not needs_build():
        return LIB
    obj_dir = OBJ_DIR + ("_variant" if variant else "")
    os.makedirs(obj_dir, exist_ok=True)
    objs = []
    for src in SOURCES:
        obj = os.path.join(obj_dir, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if not force and not variant and not _stale(obj, _deps(src)):
            continue
        cmd = [hipcc(), *FLAGS, *extra_flags, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True, cwd=CSRC)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return out


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
                        "-fno-slp-vectorize", "-Wno-unused-result", "-I", CSRC, path, "-o", exe], check=True)


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
