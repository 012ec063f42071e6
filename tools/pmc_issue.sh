# instruction-issue counters of the hzr kernels: which pipe binds (vector / scalar / LDS)?  Separate passes, counters only.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc
rm -rf $O && mkdir -p $O
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $O/a -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-verify > $O/a.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/b -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-verify > $O/b.log 2>&1 || exit 1
python3 tools/pmc_summary.py $O/a k_encode_small "k_encode" k_hist k_tile_stream k_tree > $O/pmc_a.txt
python3 tools/pmc_summary.py $O/b k_encode_small "k_encode" k_hist k_tile_stream k_tree > $O/pmc_b.txt
cat $O/pmc_a.txt $O/pmc_b.txt
rm -rf $O/a $O/b
