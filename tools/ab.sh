# A/B timing of two builds of librspt_hip.so inside one gpurun call (same box, alternating):
#   tools/ab.sh gpurun_out/base.so rspt_amd/librspt_hip.so [rounds]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A=$1; B=$2; R=${3:-3}
for i in $(seq 1 $R); do
  for L in $A $B; do
    RSPT_HIP_LIB=$PWD/$L timeout -k 10 100 python bench.py --steps 20 --warmup 3 --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']; print('$L', d['value'], d['ms_per_step'], ' '.join('%s=%.3f'%(a[:8],b) for a,b in k.items()))" || exit 1
  done
done
