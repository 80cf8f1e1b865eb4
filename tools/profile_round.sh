#!/bin/bash
# Collects the round's rocprofv3 evidence into gpurun_out/prof_rNN/ (copy the summaries into profiles/).
# usage: tools/profile_round.sh r01
set -e
tag=${1:-r01}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python bench.py --steps 20 --warmup 3 --cpu-particles 0 > $out/bench_under_profiler.json 2> $out/bench.err
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o p -- python tools/prof_step.py > $out/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o p -- python tools/prof_step.py > $out/pmc_write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_l2 -o p -- python tools/prof_step.py > $out/pmc_l2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $out/pmc_sq -o p -- python tools/prof_step.py > $out/pmc_sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $out/pmc_grbm -o p -- python tools/prof_step.py > $out/pmc_grbm.log 2>&1
ls -R $out | head -40
