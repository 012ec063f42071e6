# the other BASELINE configs on one GPU (stage times per shape)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { timeout -k 10 150 python bench.py --steps 10 --warmup 2 --no-cpu "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']; print('$*', d['value'], d['ms_per_step'], ' '.join('%s=%.3f'%(a[:10],b) for a,b in k.items()))" || exit 1; }
run --nch 12 --ns 8192 --blocks 128
run --nch 12 --ns 8192 --blocks 1024
run --nch 12 --ns 34199 --blocks 256
run --packer hzr --blocks 32
run --packer hadamard --blocks 16
run --packer dct --blocks 16
