# the per-rank shard of BASELINE configs[4] at N = 8 / 4 / 2 / 1 GPUs, on one GPU: 128 / 256 / 512 / 1024 blocks of 12ch x 8192
# (strong-scaling forecast: DESIGN.md 6b), with one and three steps in flight per rank (--slots)
mkdir -p gpurun_out/c5
for s in 1 3; do
for n in 128 256 512 1024; do
  timeout -k 10 200 python bench.py --no-cpu --workload c5 --blocks $n --slots $s --steps 60 --warmup 6 > gpurun_out/c5/bench_c5_${n}_s$s.json 2> gpurun_out/c5/err_${n}_s$s.log || exit 1
done
done
python - <<'PY'
import json
for s in (1, 3):
    for n in (128, 256, 512, 1024):
        for l in open("gpurun_out/c5/bench_c5_%d_s%d.json" % (n, s)):
            if l.startswith("{"):
                d = json.loads(l)
                print(s, n, d["ms_per_step"], d["config"]["ms_per_step_one_in_flight"], d["verified"], d["roofline"]["kernel_ms"])
PY
