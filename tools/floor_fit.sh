#!/bin/bash
# Vector instructions of the sector-query kernel as a function of the work: SQ_INSTS_VALU per launch at two map
# occupancies (cells in range per particle) and two beam counts; tools/floor_fit.py fits
# instructions per particle = a + b * cells in range + c * beams.     usage: tools/floor_fit.sh <tag>
out=gpurun_out/${1:-fit}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MCL_CELLS=1
for occ in 0.04 0.02; do
  for b in 180 90; do
    export MCL_OCC=$occ MCL_B=$b
    timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d $out/fit_${occ}_${b} -o p -- python tools/prof_step.py > $out/fit_${occ}_${b}.log 2>&1 || { echo "pass $occ $b failed"; tail -5 $out/fit_${occ}_${b}.log; exit 1; }
  done
done
echo fit passes done
