export RSPT_HIP_LIB=rspt_amd/librspt_hip_diag.so
run() { python bench.py --no-cpu --steps 20 > gpurun_out/x.json 2>&1; python -c "
import json
for l in open('gpurun_out/x.json'):
    if l.startswith('{'):
        d=json.loads(l); print('$1', d['ms_per_step'], d['verified'], d['roofline']['kernel_ms'])
"; }
run base
RSPT_ABLATE=67108864 run direct_stores
run base_again
RSPT_ABLATE=67108864 run direct_stores_again
