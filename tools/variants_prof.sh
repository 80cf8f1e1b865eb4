#!/bin/bash
# Per-kernel times (rocprofv3 --kernel-trace --stats) of the sector-query kernels for every slam_amd/libmclhip_v_*.so
# variant build; extra bench.py arguments in $BENCH_ARGS.  Run on the GPU box from the repository root.
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
for f in $root/slam_amd/libmclhip_v_*.so; do
  n=$(basename $f)
  rm -rf /tmp/vp_$n
  MCLHIP_LIBNAME=$n timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vp_$n -o t -- python3 $root/bench.py --steps 30 --cpu-particles 0 $BENCH_ARGS > /tmp/vp_$n.json 2> /tmp/vp_$n.err || { echo "$n failed"; tail -3 /tmp/vp_$n.err; exit 1; }
  echo "== $n"
  python3 - "$n" <<'PY'
import csv, glob, sys
n = sys.argv[1]
for path in glob.glob('/tmp/vp_%s/**/*kernel_stats.csv' % n, recursive=True):
    for row in csv.DictReader(open(path)):
        if 'raycast_kernel' in row['Name'] and 'true, false, false, false' not in row['Name'][:60]:
            print('  %-100s calls %4s avg %8.1f us' % (row['Name'][:100], row['Calls'], float(row['AverageNs']) / 1e3))
PY
done
