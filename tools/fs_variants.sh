#!/bin/bash
# Times tools/fastslam_bench.py with every slam_amd/libmclhip_v_*.so (tools/build_variants.sh with VARIANT_UNIT=fastslam)
# and with the production library.     usage: tools/fs_variants.sh  -> gpurun_out/fsvar/*.log
mkdir -p gpurun_out/fsvar
export FS_CPU=0 FS_STEPS=${FS_STEPS:-6}
for lib in libmclhip.so $(cd slam_amd && ls libmclhip_v_*.so 2>/dev/null); do
  MCLHIP_LIBNAME=$lib timeout -k 10 200 python tools/fastslam_bench.py > gpurun_out/fsvar/$lib.log 2>&1 || { echo "$lib failed"; tail -3 gpurun_out/fsvar/$lib.log; exit 1; }
  echo "$lib $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/fsvar/$lib.log) $(grep '^step' gpurun_out/fsvar/$lib.log | awk '{printf "%s ", $3}')"
done
