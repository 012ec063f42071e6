# refresh the judged artifacts: kernel stats, HBM traffic counters (separate --pmc passes), issue counters, bench lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof
rm -rf $O && mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 20 --warmup 2 --no-cpu --no-verify > $O/stats.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-verify > $O/fetch.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-verify > $O/write.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/valu -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-verify > $O/valu.log 2>&1 || exit 1
python3 tools/hbm_traffic.py $O/fetch $O/write $O/hbm_traffic.json > /dev/null || exit 1
python3 tools/pmc_summary.py $O/valu k_encode_small k_encode k_hist k_tile_stream k_tree k_layout > $O/pmc_issue.txt || exit 1
cp $O/hbm_traffic.json profiles/hbm_traffic.json
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err || exit 1
cat $O/bench.json
timeout -k 10 200 python bench.py --no-cpu --workload c5 > $O/bench_c5.json 2>> $O/bench.err
timeout -k 10 200 python bench.py --no-cpu --op decompress > $O/bench_decompress.json 2>> $O/bench.err
timeout -k 10 200 python bench.py --no-cpu --packer hadamard --blocks 16 > $O/bench_hadamard.json 2>> $O/bench.err
timeout -k 10 200 python bench.py --no-cpu --packer dct --blocks 16 > $O/bench_dct.json 2>> $O/bench.err
timeout -k 10 200 python bench.py --no-cpu --big-endian > $O/bench_big_endian.json 2>> $O/bench.err
timeout -k 10 200 python bench.py --no-cpu --op decompress --big-endian > $O/bench_decompress_big_endian.json 2>> $O/bench.err
timeout -k 10 200 python bench.py --no-cpu --blocks 1 > $O/bench_one_block.json 2>> $O/bench.err
timeout -k 10 200 python bench.py --op prefilter --steps 5 --warmup 1 > $O/bench_prefilter.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --op prefilter --iir-mode shared --steps 2 --warmup 1 > $O/bench_prefilter_shared.json 2>> $O/bench.err
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dstats -- python3 bench.py --steps 20 --warmup 2 --no-cpu --no-verify --op decompress > $O/dstats.log 2>&1
find $O/dstats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/dec_kernel_stats.csv; rm -rf $O/dstats
timeout -k 10 100 python tools/host_api_rate.py > $O/host_api_rate.txt 2>> $O/bench.err
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
rm -rf $O/fetch $O/write $O/valu $O/stats
tail -3 $O/bench.err
