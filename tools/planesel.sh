# which planes cost what in the hzr kernels: diagnostic build (python -m rspt_amd.build --diag), RSPT_PLANESEL probes
# (bit 0/1: k_hist without plane 0 / planes >= 1; 2/3: k_encode; 4/5: k_tree; 8/9: stop behind k_hist / k_tree)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for PS in 0 256 257 258 512 528 544 4 8; do
  RSPT_PLANESEL=$PS RSPT_HIP_LIB=$PWD/rspt_amd/librspt_hip_diag.so timeout -k 10 100 python bench.py --steps 10 --warmup 2 --no-cpu --no-verify 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']; print('PLANESEL $PS', d['ms_per_step'], ' '.join('%s=%.3f'%(a[:10],b) for a,b in k.items()))" || exit 1
done
