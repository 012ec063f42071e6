# the secondary lines with several batches in flight (bench.py --slots; one GPU): hadamard, dct, decompress -- and the headline, which does not gain
mkdir -p gpurun_out/slots
run() { name=$1; shift; timeout -k 10 300 python bench.py --no-cpu "$@" > gpurun_out/slots/$name.json 2> gpurun_out/slots/$name.err || exit 1; }
run bench_hadamard --packer hadamard --blocks 16 --steps 30
run bench_hadamard_slots2 --packer hadamard --blocks 16 --slots 2 --steps 30
run bench_hadamard_slots3 --packer hadamard --blocks 16 --slots 3 --steps 30
run bench_dct --packer dct --blocks 16 --steps 30
run bench_dct_slots3 --packer dct --blocks 16 --slots 3 --steps 30
run bench_decompress --op decompress --steps 20
run bench_decompress_slots2 --op decompress --slots 2 --steps 20
run bench_slots2 --slots 2 --steps 20
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/slots/*.json")):
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f.split("/")[-1], d["config"]["steps_in_flight"], d["ms_per_step"], d["config"]["ms_per_step_one_in_flight"], d["verified"])
PY
