#!/bin/bash
# Builds a copy of libmpcx.so with one source recompiled under extra flags (A/B experiments):
#   tools/build_variant.sh <name> <source.hip> <extra flags...>   ->  build/exp/libmpcx_<name>.so
set -e
cd "$(dirname "$0")/../mpc-rs_amd/csrc"
name=$1; src=$2; shift 2
mkdir -p ../../build/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -I../../include "$@" -c "$src" -o ../../build/exp/${src%.hip}_$name.o
objs=""
for o in $(sed -n 's/^SRCS *:= *//p' Makefile | sed 's/\.hip//g'); do
  if [ "$o.hip" == "$src" ]; then objs="$objs ../../build/exp/${o}_$name.o"; else objs="$objs $o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/exp/libmpcx_$name.so $objs -ldl
echo built build/exp/libmpcx_$name.so
