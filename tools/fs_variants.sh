#!/bin/bash
# Kernel times of tools/fastslam_bench.py with every slam_amd/libmclhip_v_*.so variant build.
export FS_CPU=0 FS_STEPS=${FS_STEPS:-6}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for f in slam_amd/libmclhip_v_*.so; do
  n=$(basename $f .so)
  MCLHIP_LIBNAME=$n.so timeout -s KILL 170 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fsv_$n -o t -- python tools/fastslam_bench.py > gpurun_out/fsv_$n.log 2>&1 || exit 1
  echo "$n $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/fsv_$n.log)"
  python - "$n" <<'PY'
import csv, sys
for r in csv.DictReader(open("gpurun_out/fsv_%s/t_kernel_stats.csv" % sys.argv[1])):
    if "fs_" in r["Name"] and float(r["AverageNs"]) > 1e5:
        print("   %-24s avg %.3f max %.3f ms" % (r["Name"][5:28], float(r["AverageNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
PY
done
