#!/bin/bash
# One rocprofv3 counter pass (instruction mix and busy cycles) over tools/prof_step.py.
set -e
out=gpurun_out/${1:-sq}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $out/pmc_sq -o p -- python tools/prof_step.py > $out/pmc_sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT --output-format csv -d $out/pmc_sq2 -o p -- python tools/prof_step.py > $out/pmc_sq2.log 2>&1
