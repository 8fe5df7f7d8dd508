#!/bin/bash
# variant_build.sh NAME [XFLAGS...]: developer (NS = 2 only) build of the current tree with extra
# -D flags into rays_amd/lib/librays_hip_exp_NAME.so; tools/variant_run.sh times every such library.
name=$1; shift
cd "$(dirname "$0")/../rays_amd/csrc" && make FAST=1 -j8 BUILD=build_exp_$name OUT=../lib/librays_hip_exp_$name.so XFLAGS="$*" ${SCHED:+SCHED=$SCHED} 2>&1 | grep -E "error|warning: v|Error" ; ls -la ../lib/librays_hip_exp_$name.so
