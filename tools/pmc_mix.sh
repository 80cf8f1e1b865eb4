#!/bin/bash
# rocprofv3 counter passes over tools/prof_step.py (three steps of the benchmark workload): the vector-instruction
# mix by type, busy / wait cycles, LDS and memory traffic.  One pass per run; summaries via tools/pmc_summary.py.
# usage: tools/pmc_mix.sh <tag> [script.py]     -> gpurun_out/<tag>/pmc_*/   (default script: tools/prof_step.py)
out=gpurun_out/${1:-mix}
script=${2:-tools/prof_step.py}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {   # name counters...
  local name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$name -o p -- python $script > $out/pmc_$name.log 2>&1 || { echo "pass $name failed"; tail -5 $out/pmc_$name.log; exit 1; }
}
run f64 SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT
run f32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM
run busy SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS
run lds SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
run fetch FETCH_SIZE
run write WRITE_SIZE
run l2 TCC_HIT_sum TCC_MISS_sum
# lanes switched on per vector instruction (a build of the profiler that does not know the counter fails this pass only)
timeout -s KILL 150 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VALU --output-format csv -d $out/pmc_lanes -o p -- python $script > $out/pmc_lanes.log 2>&1 || echo "pass lanes failed (skipped)"
echo passes done
