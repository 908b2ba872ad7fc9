# usage (GPU box): bash tools/instmix.sh <tag> [bench args...] -> gpurun_out/instmix_<tag>.txt : VALU instruction mix of k_flow / k_glue
cd $GRAFT_REPO_ROOT
TAG=$1; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/instmix_$TAG
mkdir -p $OUT
BENCH="python bench.py --steps 50 --warmup 10 --reps 1 --preroll 300 --no-cpu $@"
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_INSTS_VALU_[A-Z0-9_]*\|SQ_INSTS_[A-Z0-9_]*\|SQ_INST_LEVEL[A-Z0-9_]*\|SQ_VALU_[A-Z0-9_]*" | sort -u > $OUT/avail.txt
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $OUT/pmc$i -o p -- $BENCH > $OUT/log$i.txt 2>&1
done
python - > $OUT.txt <<PY
import sqlite3, collections, glob
for n in sorted(glob.glob("$OUT/pmc*/p_results.db")):
    cur = sqlite3.connect(n).cursor()
    acc = collections.defaultdict(list)
    for r in cur.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection order by dispatch_id"):
        k = 'k_flow' if 'k_flow' in r[0] else 'k_glue' if 'k_glue' in r[0] else None
        if k: acc[(k, r[1])].append(r[2])
    for k, v in sorted(acc.items()):
        v = v[-50:]; print(k[0], k[1], round(sum(v) / len(v), 1))
PY
cat $OUT/avail.txt | tr '\n' ' ' >> $OUT.txt
rm -rf $OUT
cat $OUT.txt
