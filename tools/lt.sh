#!/bin/bash
# per-launch times of one frame with a given library: tools/lt.sh <lib or ""> <cfg> <tag>
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
LIB=$1
[ -n "$LIB" ] && LIB=$REPO/$LIB
cd /tmp && export TMPDIR=/tmp
W=/tmp/lt_$3
rm -rf "$W"
RT_HIP_LIBRARY=$LIB RT_HIP_LEVEL_KERNELS=1 timeout -k 10 200 rocprofv3 --kernel-trace --output-format rocpd -d "$W" -- \
    python3 "$REPO/tools/one_frame.py" "$2" 3 > "$REPO/gpurun_out/lt_$3.log" 2>&1 || exit 1
python3 "$REPO/tools/level_times.py" "$(find "$W" -name '*.db' | head -1)" 24 > "$REPO/gpurun_out/lt_$3.txt"
echo "== $3"; cat "$REPO/gpurun_out/lt_$3.txt"
