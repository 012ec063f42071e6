cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for A in $ABL_LIST ; do
  rm -rf gpurun_out/abl_$A
  RSPT_ABLATE=$A timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/abl_$A -- python3 bench.py --steps 1 --warmup 1 --no-cpu > gpurun_out/abl_$A.log 2>&1 || exit 1
done
echo done
