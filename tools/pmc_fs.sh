#!/bin/bash
# rocprofv3 counter passes over the FastSLAM timing script (instruction mix, busy cycles, HBM bytes).
set -e
out=gpurun_out/${1:-fs}
mkdir -p $out
export FS_CPU=0 FS_STEPS=${FS_STEPS:-5}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $out/pmc_sq -o p -- python tools/fastslam_bench.py > $out/pmc_sq.log 2>&1
timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o p -- python tools/fastslam_bench.py > $out/pmc_fetch.log 2>&1
timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o p -- python tools/fastslam_bench.py > $out/pmc_write.log 2>&1
timeout -s KILL 170 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python tools/fastslam_bench.py > $out/trace.log 2>&1
