#!/bin/bash
# Register / spill / code-size figures of the cluster_kernel instantiations in the built objects (bepuphysics2_amd/csrc/build/*.o).
# usage: tools/kernel_regs.sh [unit ...]   (default: every bepu_cluster_* unit)
B=$(dirname "$0")/../bepuphysics2_amd/csrc/build
BUNDLER=/opt/rocm/lib/llvm/bin/clang-offload-bundler
READELF=/opt/rocm/lib/llvm/bin/llvm-readelf
units=("$@"); [ ${#units[@]} -eq 0 ] && units=($(cd $B && ls bepu_cluster_*.o | sed 's/\.o$//'))
for u in "${units[@]}"; do
    co=$(mktemp /tmp/regs.XXXXXX.co)
    fat=$(mktemp /tmp/regs.XXXXXX.fat)
    objcopy -O binary --only-section=.hip_fatbin $B/$u.o $fat && $BUNDLER --unbundle --type=o --input=$fat --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$co 2>/dev/null || { echo "$u: cannot unbundle"; rm -f $fat; continue; }
    rm -f $fat
    $READELF --notes $co | awk -v unit=$u '
        /\.name:/ { name=$2 } /\.private_segment_fixed_size:/ { scratch=$2 } /\.sgpr_count:/ { sgpr=$2 } /\.vgpr_count:/ { vgpr=$2 }
        /\.vgpr_spill_count:/ { if (name ~ /cluster_kernel/) { t = (name ~ /ELb1ELb[01]ELb[01]ELb[01]EE/ ) ? "trace" : "plain"; printf "%-28s %-5s vgpr %3d spilled %4d scratch %5d B sgpr %3d\n", unit, t, vgpr, $2, scratch, sgpr } }'
    $READELF -sW $co | awk -v unit=$u '$4=="FUNC" && $8 ~ /cluster_kernel/ && $8 !~ /\.kd$/ { printf "%-28s code %d bytes\n", unit, $3 }' | sort -u | head -2
    rm -f $co
done
