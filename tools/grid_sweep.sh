#!/bin/bash
# Sweep of the spheres' grid resolution (RT_HIP_GRID_DENSITY = cells per sphere): frame time, cells
# visited and exact tests per frame.  usage: tools/grid_sweep.sh [c3|c5] [densities...]
CFG=${1:-c3}
shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for d in "${@:-0.5 1 2 4 8}"; do
    RT_HIP_GRID_DENSITY=$d timeout -k 10 120 python3 "$REPO/bench.py" --config "$CFG" --no-cpu-baseline --steps 100 > /tmp/gs.json 2> /tmp/gs.err || exit 1
    python3 - "$d" <<'PY'
import json, sys
d = json.load(open("/tmp/gs.json"))
r = d["roofline"]
print(f"density {sys.argv[1]:>5}: {d['ms_per_step']:.4f} ms/frame, stage {r['kernel_ms']:.4f} ms, cells+filter {r.get('filter_ray_tests')}, "
      f"reflection candidates {r.get('reflection_candidates')}")
PY
done
