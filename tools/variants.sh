#!/bin/bash
for lib in libmclhip.so libmclhip_w4.so libmclhip_w5.so; do
 for a in 0 16; do
  echo -n "$lib ablate=$a  "
  MCLHIP_LIBNAME=$lib MCLHIP_ABLATE=$a timeout -k 10 120 python bench.py --steps 5 --warmup 1 --cpu-particles 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel_ms']['raycast'])"
 done
done
