#!/bin/bash
# usage: tools/build_variant.sh <name> [extra hipcc flags]   -> windgym_amd/variants/lib_<name>.so  (A/B builds; WG_LIB selects one)
set -e
cd "$(dirname "$0")/.."
mkdir -p windgym_amd/variants
n=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" -Wno-unused-result -Wno-unused-value \
  -o windgym_amd/variants/lib_$n.so windgym_amd/csrc/wg_flow.hip windgym_amd/csrc/wg_env.hip windgym_amd/csrc/wg_envb.hip windgym_amd/csrc/wg_kernels.hip windgym_amd/csrc/wg_api.hip windgym_amd/csrc/wg_mann.hip windgym_amd/csrc/wg_steady.hip -lhipfft
echo windgym_amd/variants/lib_$n.so
