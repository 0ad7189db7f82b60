#!/bin/bash
# Per-launch times of one frame, every launch alone on the device (RT_HIP_LEVEL_KERNELS=1), with a given
# build of the library: tools/lt.sh <library path relative to the repo, or ""> <c3|c5> <tag> [launches]
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
LIB=$1
[ -n "$LIB" ] && LIB=$REPO/$LIB
cd /tmp && export TMPDIR=/tmp
W=/tmp/lt_$3
rm -rf "$W"
RT_HIP_LIBRARY=$LIB RT_HIP_LEVEL_KERNELS=1 timeout -k 10 200 rocprofv3 --kernel-trace --output-format rocpd -d "$W" -- \
    python3 "$REPO/tools/one_frame.py" "$2" 3 > "$REPO/gpurun_out/lt_$3.log" 2>&1 || exit 1
DB=$(find "$W" -name '*.db' | head -1)
python3 "$REPO/tools/level_times.py" "$DB" "${4:-24}" > "$REPO/gpurun_out/lt_$3.txt"
echo "== $3"; cat "$REPO/gpurun_out/lt_$3.txt"
