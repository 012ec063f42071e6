cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for A in 0 1048576 1064960 1081344 1114112 1097728 1163264; do
  RSPT_ABLATE=$A timeout -k 10 100 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate', $A, bin($A), 'preprocess', d['roofline']['kernel_ms']['preprocess'])" || exit 1
done
