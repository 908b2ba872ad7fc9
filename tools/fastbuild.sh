#!/bin/bash
# usage: tools/fastbuild.sh [out.so] [extra hipcc flags]  — per-file objects cached under /tmp/obj (only changed sources recompile;
# wg_flow.hip alone takes a minute), linked into the in-tree library (or `out.so`).  Same flags as windgym_amd/build.py.
set -e
cd "$(dirname "$0")/.."
OUT=${1:-windgym_amd/libwindgym_hip.so}; shift || true
C=windgym_amd/csrc; O=/tmp/obj; mkdir -p $O
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value $@"
tag=$(echo "$@" | md5sum | cut -c1-8)
pids=()
for s in wg_flow wg_env wg_envb wg_kernels wg_api wg_mann wg_steady; do
  o=$O/${s}_$tag.o
  if [ ! -f $o ] || [ -n "$(find $C include -newer $o \( -name "$s.hip" -o -name '*.h' -o -name '*.inc' \) | head -1)" ]; then
    /opt/rocm/bin/hipcc $F -c $C/$s.hip -o $o & pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $O/wg_flow_$tag.o $O/wg_env_$tag.o $O/wg_envb_$tag.o $O/wg_kernels_$tag.o $O/wg_api_$tag.o $O/wg_mann_$tag.o $O/wg_steady_$tag.o -lhipfft
echo $OUT
