#!/bin/bash
# One counter pass over a few frames, summed per kernel: tools/pmc_one.sh <cfg> <tag> <counters...>
CFG=$1; TAG=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
W=/tmp/pmc_$TAG
rm -rf "$W"
RT_HIP_LEVEL_KERNELS=1 timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$W" -- \
    python3 "$REPO/tools/one_frame.py" "$CFG" 2 > "$REPO/gpurun_out/pmc_$TAG.log" 2>&1 || exit 1
python3 "$REPO/tools/pmc_summary.py" "$(find "$W" -name '*counter_collection.csv' | head -1)" > "$REPO/gpurun_out/pmc_$TAG.txt"
cat "$REPO/gpurun_out/pmc_$TAG.txt"
