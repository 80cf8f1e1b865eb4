#!/bin/bash
for a in 0 32 64 72 1; do
  echo -n "ablate=$a  "
  MCLHIP_ABLATE=$a timeout -k 10 120 python bench.py --steps 5 --warmup 1 --cpu-particles 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel_ms']['raycast'])"
done
