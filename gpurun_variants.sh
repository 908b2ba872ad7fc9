set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for w in 4 5 6 8; do
  WG_HIPCC_FLAGS="-DWG_FLOW_WAVES=$w" python windgym_amd/build.py > /dev/null 2>&1
  echo "== WG_FLOW_WAVES=$w"; python bench.py --steps 300 --warmup 30 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['glue_kernel_ms'])"
done
