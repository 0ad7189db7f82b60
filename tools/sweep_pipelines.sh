#!/bin/bash
# Frame times over frame shares and frames in flight: the library's own choice of the form of the levels'
# work, then each form forced (RT_HIP_PIPELINE).
echo "== pipeline chosen by the library"
timeout -k 10 300 python tools/config_sweep.py --configs pipe --frames 10 2>&1 | grep "config"
for P in levels fused; do
  export RT_HIP_PIPELINE=$P
  echo "== pipeline $P"
  timeout -k 10 300 python tools/config_sweep.py --configs pipe --frames 10 2>&1 | grep "config" | grep -v "6 contexts\|8 contexts"
done
