#!/bin/bash
# Collects the round's rocprofv3 evidence into gpurun_out/prof_<tag>/ (tools/profile_collect.sh then writes the
# summaries that are committed under profiles/).  One counter pass per run; kernel trace and counters never together.
# usage: tools/profile_round.sh r03 [mcl|config2|grid|fastslam ...]
tag=${1:-r03}; shift
what=${@:-mcl config2 grid fastslam}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in $what; do
  case $w in
  mcl)
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/mcl_trace -o bench -- python bench.py --steps 20 --warmup 3 --cpu-particles 0 > $out/mcl_bench_under_profiler.json 2> $out/mcl_bench.err || exit 1
    MCL_CELLS=1 bash tools/pmc_mix.sh prof_$tag/mcl tools/prof_step.py || exit 1
    # the counting build of the sector query: batches fetched against cells within z_max
    MCL_COUNTS_JSON=$out/mcl_counts.json timeout -k 10 200 python tools/stamps.py > $out/mcl_counts.log 2>&1 || exit 1 ;;
  config2)   # the per-GPU share of the eight-GPU configuration: 2^23 particles x 360 beams on the 16384^2 map
    export MCL_M=8388608 MCL_B=360 MCL_MAP=16384 MCL_STEPS=2
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/config2_trace -o t -- python tools/prof_step.py > $out/config2_under_profiler.log 2>&1 || exit 1
    MCL_CELLS=1 bash tools/pmc_mix.sh prof_$tag/config2 tools/prof_step.py || exit 1
    unset MCL_M MCL_B MCL_MAP MCL_STEPS ;;
  grid)
    export GRID_CPU=0
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/grid_trace -o t -- python tools/grid_bench.py > $out/grid_bench_under_profiler.log 2>&1 || exit 1
    bash tools/pmc_mix.sh prof_$tag/grid tools/grid_bench.py || exit 1 ;;
  fastslam)
    export FS_CPU=0 FS_STEPS=5
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/fs_trace -o t -- python tools/fastslam_bench.py > $out/fs_bench_under_profiler.log 2>&1 || exit 1
    bash tools/pmc_mix.sh prof_$tag/fastslam tools/fastslam_bench.py || exit 1 ;;
  esac
done
ls $out
