#!/bin/bash
# The bench lines kept under profiles/: the default run and its variants, one JSON line each into gpurun_out/bl/.
set -e
mkdir -p gpurun_out/bl
run() { name=$1; shift; timeout -k 10 170 python bench.py "$@" > gpurun_out/bl/$name.json 2> gpurun_out/bl/$name.err; echo "$name done"; }
run bench
run tracking --regime tracking --cpu-particles 0
run f64 --dtype f64 --cpu-particles 0
run 360 --beams 360 --cpu-particles 0
run rehearse --rehearse-collectives --cpu-particles 0
run rehearse_host --rehearse-collectives --host-collectives --cpu-particles 0
run config2 --particles 8388608 --beams 360 --map 16384 --cpu-particles 0
