# usage (GPU box): bash tools/final_profiles.sh [round tag, default r06]  -> gpurun_out/: everything profiles/ needs of the CURRENT tree
# (rocprofv3 trace + PMC passes + traffic of every GPU config, all configs F = 2 / F = 1, batch scaling, the bench line in the
# driver's style).  Refuses to finish without the four per-config profiles of this tree (cfg3 / cfg5 were stale).
cd $GRAFT_REPO_ROOT
R=${1:-r06}
for c in cfg2 cfg3 cfg4 cfg5; do
  bash tools/profile_kflow.sh ${R}_$c --workload $c > /dev/null 2>&1
done
bash tools/all_configs.sh > /dev/null 2>&1
( for n in 256 512 1024 2048 4096 8192 32768; do python bench.py --workload cfg2 --envs $n --no-cpu 2>/dev/null | python tools/benchline.py "envs=$n"; done
  python bench.py --workload cfg2 --scaling strong --gpus 1 --envs 512 --no-cpu 2>/dev/null | python tools/benchline.py "strong: 4096 / 8 = 512 envs per rank"
  for n in 256 512 1024 2048 4096; do python bench.py --workload cfg4 --envs $n --no-cpu 2>/dev/null | python tools/benchline.py "cfg4 envs=$n"; done
  for n in 256 512 1024 2048; do python bench.py --workload cfg5 --envs $n --no-cpu 2>/dev/null | python tools/benchline.py "cfg5 envs=$n"; done ) > gpurun_out/batch_scaling.txt 2>&1
python bench.py > gpurun_out/bench_line.json 2>/dev/null
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_line_driver_style.json 2>/dev/null
missing=0
for c in cfg2 cfg3 cfg4 cfg5; do
  for f in summary.txt traffic.json; do
    [ -s gpurun_out/prof_${R}_$c/$f ] || { echo "MISSING gpurun_out/prof_${R}_$c/$f"; missing=1; }
  done
done
[ $missing = 0 ] || { echo "final_profiles.sh: incomplete — do not commit profiles/ from this run"; exit 1; }
tail -c 1500 gpurun_out/bench_line.json
