# kernel timeline of a few steady-state steps of a bench line (rocprofv3 kernel trace): start / end of every kernel and what ran
# beside it -- the evidence for (or against) overlap between the batches in flight.   TL_ARGS: extra bench.py arguments
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/timeline${TL_TAG:-}
rm -rf $O && mkdir -p $O
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 bench.py --steps 8 --warmup 3 --no-cpu --no-verify ${TL_ARGS:-} > $O/log 2>&1 || exit 1
python3 tools/timeline.py $O/t > $O/timeline.txt
rm -rf $O/t
