#!/bin/bash
# usage: tools/envvariant.sh <name> [extra hipcc flags for wg_env.hip]  -> windgym_amd/variants/lib_<name>.so
# (only wg_env.hip is recompiled; the other objects come from tools/fastbuild.sh's cache of the default build)
set -e
cd "$(dirname "$0")/.."
mkdir -p windgym_amd/variants
n=$1; shift
O=/tmp/obj; tag=$(echo "" | md5sum | cut -c1-8)
for s in wg_flow wg_kernels wg_api wg_mann wg_steady; do [ -f $O/${s}_$tag.o ] || { echo "run tools/fastbuild.sh first"; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value "$@" -c windgym_amd/csrc/wg_env.hip -o $O/wg_env_var_$n.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o windgym_amd/variants/lib_$n.so $O/wg_flow_$tag.o $O/wg_env_var_$n.o $O/wg_kernels_$tag.o $O/wg_api_$tag.o $O/wg_mann_$tag.o $O/wg_steady_$tag.o -lhipfft
echo windgym_amd/variants/lib_$n.so
