#!/bin/bash
# Kernel timeline of the last frame as the library runs it (two streams): tools/timeline.sh <cfg> <tag> [launches]
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
W=/tmp/tl_$2
rm -rf "$W"
timeout -k 10 200 rocprofv3 --kernel-trace --output-format rocpd -d "$W" -- \
    python3 "$REPO/tools/one_frame.py" "$1" 4 > "$REPO/gpurun_out/tl_$2.log" 2>&1 || exit 1
python3 "$REPO/tools/level_times.py" "$(find "$W" -name '*.db' | head -1)" "${3:-20}" > "$REPO/gpurun_out/tl_$2.txt"
cat "$REPO/gpurun_out/tl_$2.txt"
