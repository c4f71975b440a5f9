#!/bin/bash
# build libslm_hip.so with attn_tile.hip compiled under extra -D flags into tools/probes/tmp_libs/<name>.so
# usage: tools/build_tile_variant.sh <name> [-DFLAG ...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/scalellm_amd/csrc
name=$1; shift
mkdir -p $R/tools/probes/tmp_libs /tmp/tilevar_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fno-gpu-rdc -I$R/include -I$C "$@" -c $C/attn_tile.hip -o /tmp/tilevar_$name/attn_tile.hip.o
objs=$(ls $C/build/*.o | grep -v attn_tile.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/probes/tmp_libs/$name.so $objs /tmp/tilevar_$name/attn_tile.hip.o
ls -la $R/tools/probes/tmp_libs/$name.so
