# usage (GPU box): bash tools/ablate_traffic.sh name1 name2 ...  (variants built with tools/build_variant.sh; "tree" = in-tree lib)
# -> per variant: bench line (k_flow us) and FETCH_SIZE / WRITE_SIZE per k_flow launch (KiB)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp WG_NOCHECK=1 WG_DEBUG_HOOKS=1
for n in "$@"; do
  if [ "$n" = "tree" ]; then lib=""; else lib=$PWD/windgym_amd/variants/lib_$n.so; fi
  export WG_LIB=$lib
  python3 bench.py --no-cpu --reps 3 $BARGS 2>/dev/null | python tools/benchline.py $n
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/abl; rocprofv3 --pmc $c -d /tmp/abl -o p -- python3 bench.py --steps 60 --warmup 10 --reps 1 --preroll 300 --no-cpu $BARGS > /dev/null 2>&1
    python3 - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob('/tmp/abl/**/p_results.db', recursive=True)[0])
v = [r[0] for r in db.execute("select value from counters_collection where kernel_name like '%k_flow%' order by dispatch_id")]
v = v[-60:]
print("   $n $c per launch (KiB): %.0f" % (sum(v) / len(v)))
PY
  done
done
