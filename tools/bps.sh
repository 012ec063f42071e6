cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for bps in 4 3 2; do
timeout -k 10 150 python bench.py --steps 10 --warmup 2 --no-cpu --bps $bps | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print($bps, d[\"value\"], d[\"ms_per_step\"], d[\"roofline\"][\"kernel_ms\"])" || exit 1
done
