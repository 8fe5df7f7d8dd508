#!/bin/bash
# exp_build.sh NAME [XFLAGS...]: FAST build of the current tree into rays_amd/lib/librays_hip_exp_NAME.so
name=$1; shift
cd /root/repo/rays_amd/csrc && make FAST=1 -j8 BUILD=build_exp_$name OUT=../lib/librays_hip_exp_$name.so XFLAGS="$*" 2>&1 | grep -E "error|warning: v|Error" ; ls -la ../lib/librays_hip_exp_$name.so
