cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for NS in 65536 65600 65792 66560 61440; do
timeout -k 10 100 python bench.py --steps 10 --warmup 2 --no-cpu --nch 64 --blocks 64 --ns $NS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']; print('ns=$NS', d['value'], 'pre/1e6samples', round(k['preprocess']/($NS*64*64/1e6)*1000,3), k)"
done
done
