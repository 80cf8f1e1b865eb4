#!/bin/bash
for nb in 768 1536 3072 8192 32768; do
  echo -n "blocks=$nb  "
  MCLHIP_RAY_BLOCKS=$nb timeout -k 10 120 python bench.py --steps 5 --warmup 1 --cpu-particles 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel_ms']['raycast'])"
done
