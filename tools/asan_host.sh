#!/bin/bash
# Host-side AddressSanitizer run of the library (scene packing, the lights' tables, the grid, the frame
# pipeline's host code): tools/asan_host.sh build   (here, on the CPU: hipcc cross-compiles)
#                        tools/asan_host.sh run     (on the GPU box, after build)
# The device code is not instrumented (-Xarch_host); the C consumer of tests/ is the host program.
set -e
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
V=$REPO/variants
CLANG=/opt/rocm/lib/llvm/bin/clang
case "$1" in
build)
    mkdir -p "$V"
    cd "$REPO"
    python3 - <<PY
from raytracer_amd.build import build_library
print(build_library(extra_flags=["-Xarch_host", "-fsanitize=address", "-g", "-fno-omit-frame-pointer"], out="$V/librt_hip_asan.so"))
PY
    "$CLANG" -std=c99 -O1 -g -fsanitize=address -I "$REPO/include" "$REPO/tests/abi_consumer.c" -o "$V/abi_consumer_asan" \
        -L "$V" -lrt_hip_asan -Wl,-rpath,"\$ORIGIN" -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib
    ;;
run)
    W=/tmp/asan_host
    rm -rf "$W" && mkdir -p "$W"
    cd "$REPO"
    # scenes and expected frames from the ordinary library
    python3 - "$W" <<'PY'
import sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from raytracer_amd import Camera, RenderContext, default_scene, scene_1k, scene_64k
from test_abi_consumer import write_scene
out = sys.argv[1]
for name, scene, w, h, depth in (("default", default_scene, 96, 96, 5), ("1k", scene_1k, 320, 180, 4), ("64k", scene_64k, 192, 108, 8),
                                 ("1k_blocks", scene_1k, 2048, 1609, 3)):   # (3.3 M pixels: row blocks, strided copies, helper threads)
    fs = scene().flatten()
    with RenderContext([0]) as c:
        c.set_scene(fs)
        frame = c.render(Camera(), w, h, depth)
    write_scene(os.path.join(out, name + ".bin"), fs)
    frame.astype(np.int32).tofile(os.path.join(out, name + ".raw"))
    print(name, w, h, depth)
PY
    for cfg in "default 96 96 5" "1k 320 180 4" "64k 192 108 8" "1k_blocks 2048 1609 3"; do
        set -- $cfg
        echo "== $1"
        ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 timeout -k 10 300 "$V/abi_consumer_asan" "$W/$1.bin" "$2" "$3" "$4" "$W/$1.raw"
    done
    ;;
*)
    echo "usage: $0 build|run"; exit 2;;
esac
