# usage (GPU box): bash tools/pmc_quick.sh <tag> [bench args...] -> gpurun_out/pmcq_<tag>.txt : trace + instruction / wait counters of the flow and glue kernels
cd $GRAFT_REPO_ROOT
TAG=$1; shift
export TMPDIR=/tmp
OUT=/tmp/pmcq_$TAG
mkdir -p $OUT gpurun_out
BENCH="python bench.py --steps 100 --warmup 10 --reps 1 --preroll 300 --no-cpu $@"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $BENCH > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $OUT/pmc1 -o p -- $BENCH > $OUT/bench_pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM -d $OUT/pmc2 -o p -- $BENCH > $OUT/bench_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc3 -o p -- $BENCH > $OUT/bench_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc4 -o p -- $BENCH > $OUT/bench_pmc4.log 2>&1
python - > gpurun_out/pmcq_$TAG.txt <<PY
import sqlite3, collections, glob
out = "$OUT"
db = sqlite3.connect(out + '/trace/t_results.db'); cur = db.cursor()
for r in cur.execute("select * from top_kernels limit 4"): print(r)
rows = list(cur.execute("select name, start, end from kernels order by start"))
for key in ('k_flow', 'k_glue', 'k_step'):
    fl = [(e-s)/1e3 for n,s,e in rows if key in n]
    if fl: print("# %s last 100 launches: avg_us %.2f min %.2f max %.2f" % (key, sum(fl[-100:])/len(fl[-100:]), min(fl[-100:]), max(fl[-100:])))
st = [s for n,s,e in rows if 'k_flow' in n or 'k_step' in n][-100:]
if len(st) > 1: print("# step period (start to start of the flow kernel), last 100: %.2f us" % ((st[-1]-st[0])/1e3/(len(st)-1)))
for n in sorted(glob.glob(out + '/pmc*/p_results.db')):
    db = sqlite3.connect(n); cur = db.cursor()
    acc = collections.defaultdict(list)
    for r in cur.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection order by dispatch_id"):
        k = 'flow' if ('k_flow' in r[0] or 'k_step' in r[0]) else 'glue' if 'k_glue' in r[0] else None
        if k: acc[(k, r[1])].append(r[2])
    for k,v in sorted(acc.items()):
        v = v[-100:]; print(k[0], k[1], round(sum(v)/len(v),1))
PY
rm -rf $OUT
cat gpurun_out/pmcq_$TAG.txt; tail -1 /dev/null
