#!/bin/bash
# Benchmarks every slam_amd/libmclhip_v_*.so variant build (kernel time of the sector query), three rounds; extra
# bench.py arguments in $BENCH_ARGS.
for round in 1 2 3; do
for f in slam_amd/libmclhip_v_*.so; do
  n=$(basename $f)
  MCLHIP_LIBNAME=$n timeout -k 10 300 python bench.py --steps 30 --cpu-particles 0 $BENCH_ARGS > gpurun_out/v.json 2> gpurun_out/v.err || exit 1
  python -c "import json; d=json.load(open('gpurun_out/v.json')); print('$n', round(d['kernel_ms']['raycast'],4), round(d['ms_per_step'],4))"
done
done
