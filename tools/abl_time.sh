# timing ablations of k_encode (the outputs are invalid): which phase costs what
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for A in 1048576 1 2 8 32 262144 524288 3 11; do
  RSPT_ABLATE=$A timeout -k 10 100 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']; print('ablate', $A, 'encode', k['hzr_encode'], 'hist', k['hzr_hist'])" || exit 1
done
