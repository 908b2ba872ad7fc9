cd $GRAFT_REPO_ROOT
bash tools/profile_kflow.sh r05_cfg2 --workload cfg2 > /dev/null 2>&1
bash tools/profile_kflow.sh r05_cfg4 --workload cfg4 > /dev/null 2>&1
bash tools/all_configs.sh > /dev/null 2>&1
( for n in 256 512 1024 2048 4096 8192 32768; do python bench.py --workload cfg2 --envs $n --no-cpu 2>/dev/null | python tools/benchline.py "envs=$n"; done
  python bench.py --workload cfg2 --scaling strong --gpus 1 --envs 512 --no-cpu 2>/dev/null | python tools/benchline.py "strong: 4096 / 8 = 512 envs per rank"
  for n in 256 512 1024 2048 4096; do python bench.py --workload cfg4 --envs $n --no-cpu 2>/dev/null | python tools/benchline.py "cfg4 envs=$n"; done ) > gpurun_out/batch_scaling.txt 2>&1
python bench.py > gpurun_out/bench_line.json 2>/dev/null
tail -c 1500 gpurun_out/bench_line.json
