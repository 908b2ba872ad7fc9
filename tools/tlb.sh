# usage (GPU box): bash tools/tlb.sh <tag> [bench args...] -> address-translation counters of k_flow
cd $GRAFT_REPO_ROOT
TAG=$1; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/tlb_$TAG
mkdir -p $OUT
BENCH="python bench.py --steps 50 --warmup 10 --reps 1 --preroll 300 --no-cpu $@"
rocprofv3 --list-avail 2>/dev/null | grep -o "[A-Z0-9_]*UTCL[A-Za-z0-9_]*" | sort -u | tr '\n' ' ' > $OUT.txt; echo >> $OUT.txt
i=0
for set in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_PERMISSION_MISS_sum" \
           "TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
           "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_LRU_INFLIGHT_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $OUT/pmc$i -o p -- $BENCH > $OUT/log$i.txt 2>&1
done
python - >> $OUT.txt <<PY
import sqlite3, collections, glob
for n in sorted(glob.glob("$OUT/pmc*/p_results.db")):
    cur = sqlite3.connect(n).cursor()
    acc = collections.defaultdict(list)
    for r in cur.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection order by dispatch_id"):
        if 'k_flow' in r[0]: acc[r[1]].append(r[2])
    for k, v in sorted(acc.items()):
        v = v[-50:]; print(k, round(sum(v) / len(v), 1))
PY
tail -3 $OUT/log*.txt | grep -i "error\|invalid" | head -5 >> $OUT.txt
rm -rf $OUT
cat $OUT.txt
