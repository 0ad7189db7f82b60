#!/bin/bash
# One-stream per-launch times of a benchmark configuration (every launch alone on the device) and the
# records per level.  usage: tools/one_stream_times.sh [c3|c5] ; results under gpurun_out/
set -o pipefail
CFG=${1:-c3}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
W=/tmp/one_stream_$CFG
rm -rf "$W"
RT_HIP_LEVEL_KERNELS=1 timeout -k 10 200 rocprofv3 --kernel-trace --output-format rocpd -d "$W" -- \
    python3 "$REPO/tools/one_frame.py" "$CFG" 3 > "$REPO/gpurun_out/one_stream_$CFG.log" 2>&1 || exit 1
python3 "$REPO/tools/level_times.py" "$(find "$W" -name '*.db' | head -1)" 40 > "$REPO/gpurun_out/${CFG}_level_times_one_stream.txt" &&
timeout -k 10 200 python3 "$REPO/tools/level_costs.py" "$CFG" 4 > "$REPO/gpurun_out/${CFG}_level_costs.txt"
