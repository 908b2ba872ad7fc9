# usage (GPU box): bash tools/profile_kflow.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/{summary.txt,traffic.json}
# (kernel trace and every --pmc set in separate runs; the raw databases are deleted, only the summaries travel back)
cd $GRAFT_REPO_ROOT
TAG=$1; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
BENCH="python bench.py --steps 100 --warmup 10 --reps 1 --preroll 300 --no-cpu $@"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $BENCH > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $OUT/pmc1 -o p -- $BENCH > $OUT/bench_pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM -d $OUT/pmc2 -o p -- $BENCH > $OUT/bench_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc3 -o p -- $BENCH > $OUT/bench_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc4 -o p -- $BENCH > $OUT/bench_pmc4.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d $OUT/pmc5 -o p -- $BENCH > $OUT/bench_pmc5.log 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_32B_sum -d $OUT/pmc6 -o p -- $BENCH > $OUT/bench_pmc6.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCC_REQ_sum -d $OUT/pmc7 -o p -- $BENCH > $OUT/bench_pmc7.log 2>&1
rocprofv3 --pmc TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum SQ_INST_CYCLES_VMEM -d $OUT/pmc8 -o p -- $BENCH > $OUT/bench_pmc8.log 2>&1
python - > $OUT/summary.txt <<PY
import sqlite3, collections, glob
out = "$OUT"
vals = {}
db = sqlite3.connect(out + '/trace/t_results.db'); cur = db.cursor()
print("# rocprofv3 --kernel-trace --stats : top kernels (name, calls, total_us, avg_us, pct)")
for r in cur.execute("select * from top_kernels limit 8"): print(r)
rows = list(cur.execute("select name, start, end from kernels order by start"))
fl = [(e-s)/1e3 for n,s,e in rows if 'k_flow' in n]
gl = [(e-s)/1e3 for n,s,e in rows if 'k_glue' in n]
print("# k_flow STEP-mode launches (last 100): avg_us %.2f min %.2f max %.2f ; k_glue avg_us %.2f" % (sum(fl[-100:])/100, min(fl[-100:]), max(fl[-100:]), sum(gl[-100:])/100))
print("# k_flow launches before the last 110 (reset / development, then the pre-roll steps; first 24 of %d) us:" % max(len(fl) - 110, 0), [round(x) for x in fl[:len(fl)-110][:24]])
print("# PMC counters, k_flow STEP-mode launches, average per launch (last 100 dispatches)")
for n in sorted(glob.glob(out + '/pmc*/p_results.db')):
    db = sqlite3.connect(n); cur = db.cursor()
    acc = collections.defaultdict(list)
    for r in cur.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection order by dispatch_id"):
        if 'k_flow' in r[0]: acc[r[1]].append(r[2])
    for k,v in sorted(acc.items()):
        v = v[-100:]; print(k, round(sum(v)/len(v),1)); vals[k] = sum(v)/len(v)
import json
if 'FETCH_SIZE' in vals and 'WRITE_SIZE' in vals:
    # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 B (MI355X_MICROARCH.md, HBM): x2
    hbm = (2 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024
    json.dump({"hbm_bytes_per_launch": hbm, "fetch_kib": vals['FETCH_SIZE'], "write_kib": vals['WRITE_SIZE'],
               "rdreq_x128B": vals.get('TCC_EA0_RDREQ_sum', 0) * 128, "kernel": "k_flow (STEP-mode launches, last 100)",
               "kflow_avg_us": sum(fl[-100:]) / 100, "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x 2 (gfx950 correction)"},
              open(out + '/traffic.json', 'w'), indent=1)
PY
rm -rf $OUT/trace $OUT/pmc*
cat $OUT/summary.txt; tail -1 $OUT/bench_trace.log | cut -c1-600
