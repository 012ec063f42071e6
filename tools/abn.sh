# timing of several builds of librspt_hip.so inside one gpurun call (same box, alternating):
#   tools/abn.sh rounds lib1.so lib2.so ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$1; shift
for i in $(seq 1 $R); do
  for L in "$@"; do
    RSPT_HIP_LIB=$PWD/$L timeout -k 10 100 python bench.py --steps 20 --warmup 3 --no-cpu ${ABN_ARGS:-} 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']; print('$L', d['value'], d['ms_per_step'], ' '.join('%s=%.3f'%(a[:8],b) for a,b in k.items()))" || exit 1
  done
done
