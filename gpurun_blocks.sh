cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for b in 64 128 256; do
  echo "== WG_FLOW_BLOCK=$b"; WG_FLOW_BLOCK=$b python bench.py --steps 300 --warmup 30 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['glue_kernel_ms'], d['roofline']['frac'])"
  WG_FLOW_BLOCK=$b python -m pytest tests -m gpu -x -q -k "physics or autoreset or partial" 2>&1 | tail -1
done
