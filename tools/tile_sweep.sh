# front-end tile length / grid sweep (diagnostic build: RSPT_TILE, RSPT_K1_GRID)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for CFG in "384 512" "256 512" "256 768" "256 1024" "128 1024" "128 1536" "192 1024" "320 512" "512 256"; do
  set -- $CFG
  RSPT_TILE=$1 RSPT_K1_GRID=$2 RSPT_HIP_LIB=$PWD/rspt_amd/librspt_hip_diag.so timeout -k 10 100 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']; print('TILE $1 GRID $2', d['ms_per_step'], d['verified'], ' '.join('%s=%.3f'%(a[:10],b) for a,b in k.items()))" || echo "TILE $1 GRID $2 failed"
done
