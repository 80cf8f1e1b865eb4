#!/bin/bash
# Turns what tools/profile_round.sh left under gpurun_out/prof_<tag>/ into the summaries committed under profiles/.
# Runs in the build container (needs hipcc for the kernels' assembly, no GPU).   usage: tools/profile_collect.sh r03
set -e
tag=${1:-r03}
src=gpurun_out/prof_$tag
cd "$(dirname "$0")/.."
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -S --cuda-device-only"
/opt/rocm/bin/hipcc $FLAGS slam_amd/csrc/raycast.hip -o /tmp/raycast_$tag.s 2>/dev/null
RAYSRC="slam_amd/csrc/raycast.hip slam_amd/csrc/mclhip_internal.h slam_amd/csrc/device_rng.h slam_amd/csrc/libm_bits.h"
cells_of() { grep -ho "cells_in_range [0-9.]*" $1/pmc_f64.log 2>/dev/null | head -1 | cut -d' ' -f2; }
if [ -d $src/mcl ]; then
  cp $src/mcl_trace/bench_kernel_stats.csv profiles/${tag}_bench_kernel_stats.csv
  cp $src/mcl_bench_under_profiler.json profiles/${tag}_bench_under_profiler.json
  python tools/pmc_summary.py $src/mcl profiles/${tag}_pmc_per_kernel.json > /dev/null
  cells=$(cells_of $src/mcl); cells=${cells:-132.459}
  counts=""; [ -f $src/mcl_counts.json ] && counts="--counts $src/mcl_counts.json"
  python tools/valu_roofline.py profiles/${tag}_pmc_per_kernel.json profiles/r02_valu_issue_cost.json /tmp/raycast_$tag.s \
      raycast_kernelIfLb0ELb0ELb0ELi4ELb0ELb1ELb1ELb0ELi192ELi5E profiles/${tag}_valu_roofline.json \
      "raycast_kernel<float, false, false, false, 4, false, true, true, false, 192, 5>" \
      --sources $RAYSRC --workload '{"particles": 1048576, "beams": 180, "map": 4096, "dtype": "f32", "regime": "global"}' \
      --work 1048576,$cells,180 $counts
fi
if [ -d $src/config2 ]; then
  cp $src/config2_trace/t_kernel_stats.csv profiles/${tag}_config2_share_kernel_stats.csv
  python tools/pmc_summary.py $src/config2 profiles/${tag}_config2_share_pmc_per_kernel.json > /dev/null
  cells=$(cells_of $src/config2); cells=${cells:-132}
  python tools/valu_roofline.py profiles/${tag}_config2_share_pmc_per_kernel.json profiles/r02_valu_issue_cost.json /tmp/raycast_$tag.s \
      raycast_kernelIfLb0ELb0ELb0ELi8ELb0ELb1ELb1ELb0ELi384ELi10E profiles/${tag}_valu_roofline_config2_share.json \
      "raycast_kernel<float, false, false, false, 8, false, true, true, false, 384, 10>" \
      --sources $RAYSRC --workload '{"particles": 8388608, "beams": 360, "map": 16384, "dtype": "f32", "regime": "global"}' \
      --work 8388608,$cells,360
fi
if [ -d $src/grid ]; then
  cp $src/grid_trace/t_kernel_stats.csv profiles/${tag}_grid_kernel_stats.csv
  python tools/pmc_summary.py $src/grid profiles/${tag}_grid_pmc_per_kernel.json > /dev/null
  python tools/valu_roofline.py profiles/${tag}_grid_pmc_per_kernel.json profiles/r02_valu_issue_cost.json /tmp/raycast_$tag.s \
      raycast_kernelIdLb0ELb0ELb1ELi16ELb0ELb0ELb0ELb0ELi0ELin1E profiles/${tag}_grid_valu_roofline.json "raycast_kernel<double, false, false, true, 16" \
      --sources $RAYSRC
fi
if [ -d $src/fastslam ]; then
  cp $src/fs_trace/t_kernel_stats.csv profiles/${tag}_fastslam_kernel_stats.csv
  python tools/pmc_summary.py $src/fastslam profiles/${tag}_fastslam_pmc_per_kernel.json > /dev/null
  /opt/rocm/bin/hipcc $FLAGS slam_amd/csrc/fastslam.hip -o /tmp/fastslam_$tag.s 2>/dev/null
  python tools/valu_roofline.py profiles/${tag}_fastslam_pmc_per_kernel.json profiles/r02_valu_issue_cost.json /tmp/fastslam_$tag.s \
      fs_associate_kernel profiles/${tag}_fastslam_valu_roofline.json "fs_associate_kernel" \
      --sources slam_amd/csrc/fastslam.hip slam_amd/csrc/mclhip_internal.h slam_amd/csrc/device_rng.h
fi
ls -la profiles | grep $tag
