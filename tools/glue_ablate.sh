# usage (GPU box): bash tools/glue_ablate.sh name1 name2 ...  -> per variant: k_glue / k_flow avg us from the rocprofv3 kernel trace
# (variants = windgym_amd/variants/lib_<name>.so built with -DWG_GLUE_ABLATE=<n>, "tree" = in-tree library)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for n in "$@"; do
  if [ "$n" = "tree" ]; then lib=""; else lib=$PWD/windgym_amd/variants/lib_$n.so; fi
  rm -rf /tmp/ga_$n
  WG_DEBUG_HOOKS=1 WG_LIB=$lib rocprofv3 --kernel-trace --stats -d /tmp/ga_$n -o t -- python bench.py --steps 100 --warmup 10 --reps 1 --preroll 300 --no-cpu $BARGS > /tmp/ga_$n.log 2>&1
  python - <<PY
import sqlite3, collections
db = sqlite3.connect('/tmp/ga_$n/t_results.db'); cur = db.cursor()
rows = list(cur.execute("select name, start, end from kernels order by start"))
fl = [(e-s)/1e3 for nm,s,e in rows if 'k_flow' in nm][-100:]
gl = [(e-s)/1e3 for nm,s,e in rows if 'k_glue' in nm][-100:]
# gap between the end of a k_flow launch and the start of the k_glue after it, and from the k_glue to the next k_flow
ks = [(nm,s,e) for nm,s,e in rows if 'k_flow' in nm or 'k_glue' in nm][-200:]
g1 = [(ks[i+1][1]-ks[i][2])/1e3 for i in range(len(ks)-1) if 'k_flow' in ks[i][0] and 'k_glue' in ks[i+1][0]]
g2 = [(ks[i+1][1]-ks[i][2])/1e3 for i in range(len(ks)-1) if 'k_glue' in ks[i][0] and 'k_flow' in ks[i+1][0]]
fs = [s for nm,s,e in rows if 'k_flow' in nm][-100:]
oth = collections.Counter(nm.split('(')[0][:40] for nm,s,e in rows if s >= fs[0] and 'k_flow' not in nm and 'k_glue' not in nm)
print("$n k_glue %.2f us  k_flow %.2f us  gap flow->glue %.2f  gap glue->flow %.2f  period %.2f us  other kernels in window: %s" % (sum(gl)/len(gl), sum(fl)/len(fl), sum(g1)/max(len(g1),1), sum(g2)/max(len(g2),1), (fs[-1]-fs[0])/1e3/(len(fs)-1), dict(oth)))
PY
done
