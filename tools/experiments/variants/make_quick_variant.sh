#!/bin/bash
# A variant library for same-box A/Bs of COMPILER FLAGS on the three cluster units the bench scenes run (hot_1024: headline, hot_512s: pile, hot_768s: crowd): those
# three are recompiled with the extra flags, everything else is linked from the objects of the last regular build. CPU only, about a minute.
# Usage: make_quick_variant.sh <name> <flag> [<flag> ...]   ->  tools/experiments/variants/libbepuhip_<name>.so   (BEPUHIP_LIB=<that> python tools/ab_scene.py ...)
set -e
NAME=$1; shift
HERE=$(cd $(dirname $0) && pwd); REPO=$(cd $HERE/../../.. && pwd); SRC=$REPO/bepuphysics2_amd/csrc
T=${TMPDIR:-/tmp}/bepu_quick_$NAME; rm -rf $T; mkdir -p $T
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Xarch_device -fno-slp-vectorize -fPIC -Wno-unused-result -Wno-unused-value -Wno-array-bounds"
for u in ${UNITS:-bepu_cluster_hot_1024 bepu_cluster_hot_512s bepu_cluster_hot_768s}; do ( cd $SRC && hipcc $FLAGS "$@" -c -o $T/$u.o $u.hip ) & done; wait
OBJS=""; for o in $SRC/build/*.o; do b=$(basename $o); if [ -f $T/$b ]; then OBJS="$OBJS $T/$b"; else OBJS="$OBJS $o"; fi; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $HERE/libbepuhip_$NAME.so $OBJS
echo "built $HERE/libbepuhip_$NAME.so"
