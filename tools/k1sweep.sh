cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "256 352" "512 352" "1024 352" "256 256" "512 256" "256 176" "512 176" "256 192" ; do
  set -- $cfg
  echo "threads=$1 tile=$2"
  RSPT_K1_THREADS=$1 RSPT_TILE=$2 timeout -k 10 100 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms']['preprocess'])" || exit 1
done
