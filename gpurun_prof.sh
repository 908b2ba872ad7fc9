# usage: bash gpurun_prof.sh <tag> [bench args...]
cd $GRAFT_REPO_ROOT
TAG=$1; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
BENCH="python bench.py --steps 100 --warmup 10 --no-cpu $@"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $BENCH > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $OUT/pmc1 -o p -- $BENCH > $OUT/bench_pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM -d $OUT/pmc2 -o p -- $BENCH > $OUT/bench_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc3 -o p -- $BENCH > $OUT/bench_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc4 -o p -- $BENCH > $OUT/bench_pmc4.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc5 -o p -- $BENCH > $OUT/bench_pmc5.log 2>&1
find $OUT -name "*.csv" | head -30
python - <<PY
import csv, glob, collections, os
out = "$OUT"
for f in sorted(glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)):
    print("==", f); print(open(f).read()[:3000])
for d in ("pmc1","pmc2","pmc3","pmc4","pmc5"):
    for f in glob.glob(out + f"/{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
        for k, v in acc.items():
            if "k_flow" in k or "k_glue" in k:
                print(d, k, {c: round(val / cnt[(k, c)], 1) for c, val in v.items()})
PY
