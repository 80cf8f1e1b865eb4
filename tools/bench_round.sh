o=gpurun_out/bench_r03; mkdir -p $o
timeout -k 10 200 python bench.py > $o/bench.json 2> $o/bench.err &&
timeout -k 10 200 python bench.py --regime tracking --cpu-particles 0 > $o/bench_tracking.json 2>> $o/bench.err &&
timeout -k 10 200 python bench.py --dtype f64 --cpu-particles 0 > $o/bench_f64.json 2>> $o/bench.err &&
timeout -k 10 200 python bench.py --beams 360 --cpu-particles 0 > $o/bench_360.json 2>> $o/bench.err &&
timeout -k 10 300 python bench.py --particles 8388608 --beams 360 --map 16384 --cpu-particles 0 --steps 10 --warmup 2 > $o/bench_config2_share.json 2>> $o/bench.err &&
timeout -k 10 200 python bench.py --rehearse-collectives --cpu-particles 0 > $o/bench_rehearse.json 2>> $o/bench.err &&
timeout -k 10 200 python bench.py --rehearse-collectives --host-collectives --cpu-particles 0 > $o/bench_rehearse_host.json 2>> $o/bench.err &&
timeout -k 10 200 python tools/amcl_bench.py > $o/amcl_bench.json 2>> $o/bench.err &&
GRID_CPU=0 timeout -k 10 200 python tools/grid_bench.py > $o/grid_bench.json 2>> $o/bench.err &&
FS_CPU=0 timeout -k 10 200 python tools/fastslam_bench.py > $o/fastslam_bench.json 2>> $o/bench.err
echo rc=$?
for f in $o/*.json; do echo $f; tail -c 300 $f; echo; done
