#!/bin/bash
# build libslm_hip.so with ONE source recompiled under extra -D flags into tools/probes/tmp_libs/<name>.so
# usage: tools/build_variant.sh <name> <source.hip> [-DFLAG ...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/scalellm_amd/csrc
name=$1; src=$2; shift 2
mkdir -p $R/tools/probes/tmp_libs /tmp/var_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fno-gpu-rdc -I$R/include -I$C "$@" -c $C/$src -o /tmp/var_$name/$src.o
objs=$(ls $C/build/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/probes/tmp_libs/$name.so $objs /tmp/var_$name/$src.o
ls -la $R/tools/probes/tmp_libs/$name.so
