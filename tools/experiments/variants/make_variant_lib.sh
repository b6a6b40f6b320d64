#!/bin/bash
# Builds a variant of libbepuhip.so from a copy of the tree with extra compiler defines (experiments that are compiled in or out by a macro), for same-box A/B runs
# (BEPUHIP_LIB=<variant> python tools/ab_scene.py ...). CPU only. Usage: make_variant_lib.sh <name> -DMACRO [-DMACRO ...]    -> tools/experiments/variants/libbepuhip_<name>.so
set -e
NAME=$1; shift
HERE=$(cd $(dirname $0) && pwd); REPO=$(cd $HERE/../../.. && pwd)
T=${TMPDIR:-/tmp}/bepu_variant_$NAME; rm -rf $T; mkdir -p $T/bepuphysics2_amd $T/include
cp -r $REPO/bepuphysics2_amd/csrc $T/bepuphysics2_amd/csrc; rm -rf $T/bepuphysics2_amd/csrc/build $T/bepuphysics2_amd/csrc/*.so
cp $REPO/include/bepuhip.h $T/include/; cp $REPO/bepuphysics2_amd/build.py $T/bepuphysics2_amd/; touch $T/bepuphysics2_amd/__init__.py
( cd $T && python -c "
import sys
from bepuphysics2_amd import build
build.HIP_FLAGS += sys.argv[1:]; build.HIP_COMPILE_FLAGS += sys.argv[1:]
build.build_hip()" "$@" )
cp $T/bepuphysics2_amd/csrc/libbepuhip.so $HERE/libbepuhip_$NAME.so
echo "built $HERE/libbepuhip_$NAME.so"
