cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for T in 256 512 1024; do
RSPT_FFT_THREADS=$T timeout -k 10 150 python bench.py --steps 10 --warmup 2 --no-cpu --packer dct --blocks 16 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print($T, d[\"value\"], d[\"ms_per_step\"], d[\"roofline\"][\"kernel_ms\"][\"preprocess\"])" || exit 1
done
