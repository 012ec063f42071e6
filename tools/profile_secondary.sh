# kernel stats of the secondary rows: decompress, hadamard, dct (16 blocks each for the transforms)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof2
rm -rf $O && mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dec -- python3 tools/decode_rate.py > $O/dec.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/had -- python3 bench.py --steps 10 --warmup 2 --no-cpu --packer hadamard --blocks 16 > $O/had.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dct -- python3 bench.py --steps 10 --warmup 2 --no-cpu --packer dct --blocks 16 > $O/dct.log 2>&1 || exit 1
for n in dec had dct; do find $O/$n -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${n}_kernel_stats.csv; done
tail -2 $O/dec.log; tail -1 $O/had.log | cut -c1-200; tail -1 $O/dct.log | cut -c1-200
rm -rf $O/dec $O/had $O/dct
