# What binds the hzr kernels?  Wave-cycle accounting from the SQ counters (quad-cycle units, MI355X_MICROARCH.md):
#   SQ_WAIT_ANY + SQ_WAIT_INST_ANY + SQ_ACTIVE_INST_ANY ~ SQ_WAVE_CYCLES
# Three counter-only passes (8 SQ slots each); summary as fractions of wave-cycles per kernel.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${PMC_OUT:-pmcw}
rm -rf $O && mkdir -p $O
B="python3 bench.py --steps 3 --warmup 1 --no-cpu --no-verify ${PMC_ARGS:-}"   # PMC_ARGS="--op decompress" PMC_KERNELS="k_dec_block k_inv_rows ..." for another line
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d $O/a -- $B > $O/a.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/b -- $B > $O/b.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_LDS_ATOMIC SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAVES SQ_BUSY_CU_CYCLES --output-format csv -d $O/c -- $B > $O/c.log 2>&1 || exit 1
K="${PMC_KERNELS:-k_encode_small k_encode k_histlist k_hist k_tile_stream k_tree k_layout}"
python3 tools/pmc_summary.py $O/a $K > $O/pmc_wait_a.txt
python3 tools/pmc_summary.py $O/b $K > $O/pmc_wait_b.txt
python3 tools/pmc_summary.py $O/c $K > $O/pmc_wait_c.txt
cat $O/pmc_wait_a.txt $O/pmc_wait_b.txt $O/pmc_wait_c.txt
rm -rf $O/a $O/b $O/c
