#!/bin/bash
# The other lines of the results table: bench.py variants (no CPU baseline) and the augmented-MCL timing.
# usage: tools/results_table.sh <tag>   -> gpurun_out/<tag>/*.json
out=gpurun_out/${1:-table}
mkdir -p $out
run() { name=$1; shift; timeout -k 10 400 python bench.py --cpu-particles 0 "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; exit 1; }; echo "$name $(python -c "import json,sys; d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1]); print('%.3f ms  %.3e evals/s  ray %.3f ms' % (d['ms_per_step'], d['value'], d['kernel_ms']['raycast']), d.get('exchange'))")"; }
run tracking --regime tracking
run f64 --dtype f64
run rehearse --rehearse-collectives
run rehearse_host --rehearse-collectives --host-collectives
run config2_share --particles 8388608 --beams 360 --map 16384 --steps 5 --warmup 2
timeout -k 10 300 python tools/amcl_bench.py > $out/amcl.log 2>&1 || { echo "amcl failed"; tail -3 $out/amcl.log; exit 1; }
tail -1 $out/amcl.log | cut -c1-400
