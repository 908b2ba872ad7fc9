#!/bin/bash
# usage (on the GPU box): tools/ab.sh "<bench args>" name1 name2 ...   -> one summary line per variant, interleaved twice
args=$1; shift
for rep in 1 2; do
  for n in "$@"; do
    if [ "$n" = "tree" ]; then lib=""; else lib=$PWD/windgym_amd/variants/lib_$n.so; fi
    WG_DEBUG_HOOKS=1 WG_LIB=$lib python3 bench.py --no-cpu $args 2>/dev/null | python tools/benchline.py $n
  done
done
