#!/bin/bash
# Frame times of the two forms of the levels' work (RT_HIP_PIPELINE) over frame shares and frames in flight.
for P in levels fused; do
  export RT_HIP_PIPELINE=$P
  echo "== pipeline $P"
  timeout -k 10 200 python tools/config_sweep.py --configs pipe 2>&1 | grep "config" | grep -v "4 contexts\|2 contexts in turn, device share 1"
done
