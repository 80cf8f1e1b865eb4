#!/bin/bash
# Turns what tools/profile_round.sh left under gpurun_out/prof_<tag>/ into the summaries committed under profiles/.
# Runs in the build container (needs hipcc for the kernels' assembly, no GPU).   usage: tools/profile_collect.sh r02
set -e
tag=${1:-r02}
src=gpurun_out/prof_$tag
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -S --cuda-device-only slam_amd/csrc/raycast.hip -o /tmp/raycast_$tag.s 2>/dev/null
if [ -d $src/mcl ]; then
  cp $src/mcl_trace/bench_kernel_stats.csv profiles/${tag}_bench_kernel_stats.csv
  cp $src/mcl_bench_under_profiler.json profiles/${tag}_bench_under_profiler.json
  python tools/pmc_summary.py $src/mcl profiles/${tag}_pmc_per_kernel.json > /dev/null
  python tools/valu_roofline.py profiles/${tag}_pmc_per_kernel.json profiles/r02_valu_issue_cost.json /tmp/raycast_$tag.s \
      raycast_kernelIfLb0ELb0ELb0ELi4ELb0ELb1ELb1ELb0ELi192ELi5E profiles/${tag}_valu_roofline.json "raycast_kernel<float, false, false, false, 4, false, true, true, false, 192, 5>"
fi
if [ -d $src/grid ]; then
  cp $src/grid_trace/t_kernel_stats.csv profiles/${tag}_grid_kernel_stats.csv
  python tools/pmc_summary.py $src/grid profiles/${tag}_grid_pmc_per_kernel.json > /dev/null
  python tools/valu_roofline.py profiles/${tag}_grid_pmc_per_kernel.json profiles/r02_valu_issue_cost.json /tmp/raycast_$tag.s \
      raycast_kernelIdLb0ELb0ELb1ELi16ELb0ELb0E profiles/${tag}_grid_valu_roofline.json "raycast_kernel<double, false, false, true, 16"
fi
if [ -d $src/fastslam ]; then
  cp $src/fs_trace/t_kernel_stats.csv profiles/${tag}_fastslam_kernel_stats.csv
  python tools/pmc_summary.py $src/fastslam profiles/${tag}_fastslam_pmc_per_kernel.json > /dev/null
fi
ls -la profiles | grep $tag
