# usage (GPU box): bash tools/pmc_variant.sh "<counters>" name1 name2 ...  -> per-launch averages for k_flow
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp WG_NOCHECK=1 WG_DEBUG_HOOKS=1
C=$1; shift
for n in "$@"; do
  if [ "$n" = "tree" ]; then lib=""; else lib=$PWD/windgym_amd/variants/lib_$n.so; fi
  rm -rf /tmp/pv; WG_LIB=$lib rocprofv3 --pmc $C -d /tmp/pv -o p -- python3 bench.py --steps 60 --warmup 10 --reps 1 --preroll 300 --no-cpu > /dev/null 2>&1
  python3 - <<PY
import sqlite3, glob, collections
db = sqlite3.connect(glob.glob('/tmp/pv/**/p_results.db', recursive=True)[0])
acc = collections.defaultdict(list)
for k, c, v in db.execute("select kernel_name, counter_name, value from counters_collection order by dispatch_id"):
    if 'k_flow' in k: acc[c].append(v)
print("$n", {c: round(sum(v[-60:]) / 60) for c, v in sorted(acc.items())})
PY
done
