# the per-rank shard of BASELINE configs[4] at N = 8 / 4 / 2 / 1 GPUs, on one GPU: 128 / 256 / 512 / 1024 blocks of 12ch x 8192
# (strong-scaling forecast: DESIGN.md 6b)
mkdir -p gpurun_out/c5
for n in 128 256 512 1024; do
  timeout -k 10 200 python bench.py --no-cpu --workload c5 --blocks $n --steps 40 --warmup 5 > gpurun_out/c5/bench_c5_$n.json 2> gpurun_out/c5/err_$n.log || exit 1
done
python - <<'PY'
import json
for n in (128, 256, 512, 1024):
    for l in open("gpurun_out/c5/bench_c5_%d.json" % n):
        if l.startswith("{"):
            d = json.loads(l)
            print(n, d["ms_per_step"], d["verified"], d["roofline"]["kernel_ms"])
PY
