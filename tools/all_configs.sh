# usage (on the GPU box): bash tools/all_configs.sh  -> gpurun_out/all_configs.jsonl
# every GPU config of BASELINE.json, with the baseline farm on (F = 2) and off (F = 1), + cfg1 on one host core
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/all_configs.jsonl; : > $OUT
for wl in cfg2 cfg3 cfg4 cfg5; do
  for f in "" "--one-farm"; do
    python bench.py --workload $wl $f --steps 200 --warmup 20 --no-cpu 2>/dev/null | tail -1 >> $OUT
  done
done
python tools/cpu_cfg1.py >> $OUT
cat $OUT | cut -c1-400
