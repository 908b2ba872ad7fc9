cd $GRAFT_REPO_ROOT
for a in 0 1 2 3; do
  WG_HIPCC_FLAGS="-DWG_ABLATE=$a" python windgym_amd/build.py > /dev/null 2>&1
  echo "== WG_ABLATE=$a"; python bench.py --steps 300 --warmup 30 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['glue_kernel_ms'])"
done
