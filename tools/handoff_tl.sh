#!/bin/bash
# Timeline (kernels + copies) of the last rt_render call: tools/handoff_tl.sh <cfg> <tag>   (RT_HIP_* knobs from the environment)
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
W=/tmp/htl_$2
rm -rf "$W"
timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format rocpd -d "$W" -- \
    python3 "$REPO/tools/render_frames.py" "$1" 6 > "$REPO/gpurun_out/htl_$2.log" 2>&1 || exit 1
python3 "$REPO/tools/handoff_timeline.py" "$(find "$W" -name '*.db' | head -1)" > "$REPO/gpurun_out/htl_$2.txt"
