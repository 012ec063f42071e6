# gpurun_out/prof + gpurun_out/pmcw (tools/profile_round.sh, tools/pmc_wait.sh) -> profiles/rNN_*      usage: bash tools/copy_profiles.sh r03
R=${1:?round tag, e.g. r03}
P=gpurun_out/prof
for f in bench bench_c5 bench_decompress bench_hadamard bench_dct bench_big_endian bench_decompress_big_endian bench_one_block bench_prefilter bench_prefilter_shared hbm_traffic; do cp $P/$f.json profiles/${R}_$f.json; done
cp $P/hbm_traffic.json profiles/hbm_traffic.json
cp $P/kernel_stats.csv profiles/${R}_kernel_stats.csv
cp $P/dec_kernel_stats.csv profiles/${R}_dec_kernel_stats.csv
cp $P/pmc_issue.txt profiles/${R}_pmc_issue.txt
cp $P/host_api_rate.txt profiles/${R}_host_api_rate.txt
for x in a b c; do [ -f gpurun_out/pmcw/pmc_wait_$x.txt ] && cp gpurun_out/pmcw/pmc_wait_$x.txt profiles/${R}_pmc_wait_$x.txt; done; true
