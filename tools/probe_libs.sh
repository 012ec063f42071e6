# timing probes: several diagnostic libraries x RSPT_PLANESEL settings in one call:  tools/probe_libs.sh "sel1 sel2" lib1.so lib2.so ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SELS=$1; shift
for PS in $SELS; do for L in "$@"; do
  RSPT_PLANESEL=$PS RSPT_HIP_LIB=$PWD/$L timeout -k 10 100 python bench.py --steps 10 --warmup 2 --no-cpu --no-verify 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']; print('PLANESEL $PS $L', d['ms_per_step'], ' '.join('%s=%.3f'%(a[:10],b) for a,b in k.items()))" || exit 1
done; done
