#!/bin/bash
# Builds slam_amd/libmclhip_v_<name>.so for each "name:flags" argument: one translation unit (VARIANT_UNIT, default
# raycast) recompiled with the given -D flags, everything else shared with the production objects.  For tools/variants.sh.
# usage: tools/build_variants.sh "g8:-DMCL_GROUP=8" "w5:-DMCL_RAY_WAVES=5"
#        VARIANT_UNIT=fastslam tools/build_variants.sh "b1w4:-DFS_BATCH=1 -DFS_WAVES=4"
set -e
cd "$(dirname "$0")/../slam_amd/csrc"
make -j8 > /dev/null
rm -f ../libmclhip_v_*.so
CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
unit=${VARIANT_UNIT:-raycast}
others=$(for u in api map_motion raycast weights grid fastslam; do if [ $u != $unit ]; then echo $u.o; fi; done)
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc $CXXFLAGS $flags -c $unit.hip -o /tmp/${unit}_v_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmclhip_v_$name.so $others /tmp/${unit}_v_$name.o ) &
done
wait
ls -la ../libmclhip_v_*.so
