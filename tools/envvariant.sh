#!/bin/bash
# usage: tools/envvariant.sh <name> [extra hipcc flags]  -> windgym_amd/variants/lib_<name>.so
# (only ONE translation unit is recompiled with the extra flags — wg_env.hip, or the file named by WG_VARIANT_TU, e.g.
# WG_VARIANT_TU=wg_envb — the other objects come from tools/fastbuild.sh's cache of the default build)
set -e
cd "$(dirname "$0")/.."
mkdir -p windgym_amd/variants
n=$1; shift
tu=${WG_VARIANT_TU:-wg_env}
O=/tmp/obj; tag=$(echo "" | md5sum | cut -c1-8)
objs=""
for s in wg_flow wg_env wg_envb wg_kernels wg_api wg_mann wg_steady; do
  [ -f $O/${s}_$tag.o ] || { echo "run tools/fastbuild.sh first"; exit 1; }
  if [ "$s" = "$tu" ]; then objs="$objs $O/${tu}_var_$n.o"; else objs="$objs $O/${s}_$tag.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value "$@" -c windgym_amd/csrc/$tu.hip -o $O/${tu}_var_$n.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o windgym_amd/variants/lib_$n.so $objs -lhipfft
echo windgym_amd/variants/lib_$n.so
