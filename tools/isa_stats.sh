#!/bin/bash
# Static facts of one instantiation of the sector-query kernel from the compiler's assembly: registers, scratch,
# vector instructions and the scalar-register spill traffic (v_readlane / v_writelane) of the whole kernel.
#   tools/isa_stats.sh [mangled-name fragment, default the float32-mixture instantiation] [extra hipcc flags]
set -e
cd "$(dirname "$0")/../slam_amd/csrc"
frag=${1:-raycast_kernelIfLb0ELb0ELb0ELi4ELb0ELb1E}
shift || true
out=/tmp/isa
mkdir -p $out
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -S --cuda-device-only "$@" raycast.hip -o $out/raycast.s 2>/dev/null
start=$(grep -n "^_ZN3mcl14${frag}.*:" $out/raycast.s | head -1 | cut -d: -f1)
end=$(awk -v s=$start 'NR>s && /s_endpgm/{print NR; exit}' $out/raycast.s)
sed -n "${start},${end}p" $out/raycast.s > $out/kernel.s
echo "lines $start-$end -> $out/kernel.s"
grep -A12 "\.name: *_ZN3mcl14${frag}" $out/raycast.s | grep "vgpr_count\|sgpr_count\|private_segment_fixed_size\|sgpr_spill\|vgpr_spill"
echo "valu: $(grep -c '^\s*v_' $out/kernel.s)  readlane: $(grep -c 'v_readlane' $out/kernel.s)  writelane: $(grep -c 'v_writelane' $out/kernel.s)  scratch: $(grep -c 'scratch_' $out/kernel.s)  ds: $(grep -c '^\s*ds_' $out/kernel.s)"
loop=$(grep -n "This Loop Header: Depth=1" $out/kernel.s | head -1 | cut -d: -f1)
echo "inside the particle loop (from line $loop): valu $(tail -n +$loop $out/kernel.s | grep -c '^\s*v_')  readlane $(tail -n +$loop $out/kernel.s | grep -c 'v_readlane')  writelane $(tail -n +$loop $out/kernel.s | grep -c 'v_writelane')"
