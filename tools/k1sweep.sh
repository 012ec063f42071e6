cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "512 384" "512 256" "768 256" "512 512" "256 768" "1024 128"; do
  set -- $cfg
  RSPT_K1_GRID=$1 RSPT_TILE=$2 timeout -k 10 100 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('grid=$1 tile=$2', d['value'], d['roofline']['kernel_ms']['preprocess'])" || exit 1
done
done
