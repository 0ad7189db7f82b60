#!/bin/bash
# Collects the round's profiles on the GPU box: kernel trace of bench.py, counter passes of a few
# plain frames, for both benchmark configurations.  usage: tools/profile_round.sh <tag>
# Results: gpurun_out/prof_<tag>/*.txt|json (copy what is to be kept into profiles/).
set -o pipefail
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
W=/tmp/prof_work_$TAG
rm -rf "$W"

trace() {   # name, config, steps
    timeout -k 10 240 rocprofv3 --kernel-trace --output-format rocpd -d "$W/$1" -- \
        python3 "$REPO/bench.py" --config "$2" --steps "$3" --warmup 3 --no-cpu-baseline > "$OUT/$1_bench_under_trace.json" 2> "$OUT/$1_trace.log" || return 1
    local db
    db=$(find "$W/$1" -name "*.db" | head -1)
    {
        echo "# rocprofv3 --kernel-trace -- python3 bench.py --config $2 --steps $3 --warmup 3 --no-cpu-baseline (one MI355X)"
        python3 "$REPO/tools/rocprof_summary.py" "$db"
        echo
        echo "# the timed loop's frames (warm-up + steps), tools/stage_profile.py span:"
        python3 "$REPO/tools/stage_profile.py" span "$db" $(( $3 + 3 ))
    } > "$OUT/$1_kernel_stats.txt"
}

pmc() {     # name, config, frames, label, counters...
    local name=$1 cfg=$2 frames=$3 label=$4
    shift 4
    timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d "$W/${name}_$label" -- \
        python3 "$REPO/tools/one_frame.py" "$cfg" "$frames" > "$OUT/${name}_pmc_$label.log" 2>&1 || return 1
    local csv
    csv=$(find "$W/${name}_$label" -name "*counter_collection.csv" | head -1)
    cp "$csv" "$W/${name}_$label.csv"
    {
        echo "== pass $label: rocprofv3 --pmc $* -- python3 tools/one_frame.py $cfg $frames"
        python3 "$REPO/tools/pmc_summary.py" "$csv"
        echo
    } >> "$OUT/${name}_pmc_counters.txt"
}

one() {     # name, config, steps, frames
    rm -f "$OUT/$1_pmc_counters.txt"
    trace "$1" "$2" "$3" &&
    pmc "$1" "$2" "$4" sq SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES &&
    pmc "$1" "$2" "$4" sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM &&
    pmc "$1" "$2" "$4" fetch FETCH_SIZE &&
    pmc "$1" "$2" "$4" write WRITE_SIZE &&
    python3 "$REPO/tools/stage_profile.py" pmc "$OUT/$1_trace_pmc.json" sq="$W/$1_sq.csv" fetch="$W/$1_fetch.csv" write="$W/$1_write.csv" > /dev/null
}

one "${TAG}_c3" c3 20 4 && one "${TAG}_c5" c5 10 2 && echo "profiles in $OUT"
