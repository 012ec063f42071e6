#!/usr/bin/env python3
"""bench.py -- MSamples/s of signal_packer compress on MI355X (default: xdelta_hzr, 64ch x 65536 x int32).

    python bench.py --gpus N --steps K --warmup W [--workload c3|c5] [--packer ...]

One "step" = one pass of the hot path (rspt_hip_compress_batch_dev) over one batch of
synthetic blocks per GPU, inputs already resident in HBM.  Two distinct batches alternate in
the timed loop.  With --gpus N > 1 and no RANK in the environment this process only starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child (before any GPU
call) and relays rank 0's JSON line; under torch.distributed.run it is one rank per GPU.

  --workload c3  (default) weak scaling: every rank compresses its own `--blocks` blocks of
                 nch x ns (BASELINE configs[2] shape on the xdelta_hzr path); the ranks exchange
                 the sizes index every step (RCCL all-gather), the payload once after the timed steps
  --workload c5  strong scaling: 1024 blocks of 12ch x 8192 in total (BASELINE configs[4]), contiguous
                 shards per rank (shard.shard_range), container pack + RCCL gather to rank 0 INSIDE
                 the timed step

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline`
(HIP-event timed on the launch stream), `cpu_baseline` (the checker library timed on
this node's host cores, N=1 only) and `verified` (streams of both batches compared with
the oracle after the timed region).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
C5_TOTAL_BLOCKS = 1024


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=["c3", "c5"])
    ap.add_argument("--blocks", type=int, default=None, help="c3: blocks per GPU per step (default 64); c5: total blocks (default 1024)")
    ap.add_argument("--slots", type=int, default=1, help="steps in flight per rank (c5; on one GPU also c3 and decompress lines), each on its own handle (its own plane workspace, the "
                    "reference's one packer object per worker) and stream, taken round-robin; small shards leave most of the GPU idle inside "
                    "every kernel of a step (DESIGN 6b)")
    ap.add_argument("--nch", type=int, default=None)
    ap.add_argument("--ns", type=int, default=None)
    ap.add_argument("--nb", type=int, default=3)
    ap.add_argument("--bps", type=int, default=4, choices=[1, 2, 3, 4], help="bytes per sample of the input (the metric is quoted on 4)")
    ap.add_argument("--packer", default="xdelta_hzr", choices=["xdelta_hzr", "hzr", "hadamard", "dct"])
    ap.add_argument("--op", default="compress", choices=["compress", "decompress", "prefilter"],
                    help="decompress: the timed step decodes the streams of one batch (secondary line, c3 workload, one GPU); "
                         "prefilter: the IIR pre-filter stage in front of the packers (rspt_hip_iir_prefilter_batch_dev)")
    ap.add_argument("--iir-mode", default="per_channel", choices=["per_channel", "shared"],
                    help="prefilter: a fresh filter per channel, or the harness's one filter object whose state runs on from channel to channel")
    ap.add_argument("--big-endian", action="store_true", help="feed the samples most significant byte first (rspt_hip_set_byte_order)")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the RCCL exchange altogether")
    ap.add_argument("--gather-every-step", action="store_true",
                    help="c3, N>1: ship every step's streams to rank 0 (link-bound beyond 2-3 GPUs: see DESIGN.md)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle check of the produced streams")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    a = ap.parse_args(argv)
    if a.workload == "c5":
        a.nch = a.nch or 12
        a.ns = a.ns or 8192
        a.blocks = a.blocks or C5_TOTAL_BLOCKS
    else:
        a.nch = a.nch or 64
        a.ns = a.ns or 65536
        a.blocks = a.blocks or 64
    return a


def spawn_ranks(args):
    """--gpus N without a launcher: start the N ranks ourselves (a child process, no exec),
    relay rank 0's JSON line, fail loudly if the node cannot run N ranks."""
    import torch  # device_count() does not initialise the GPU

    have = torch.cuda.device_count()
    if have < args.gpus:
        sys.stderr.write("bench.py: --gpus %d requested but %d device(s) visible; refusing to report a smaller run\n" % (args.gpus, have))
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            sys.stderr.write(ln + "\n")
    if p.returncode != 0 or line is None:
        sys.stderr.write("bench.py: the %d-rank run failed (exit %d)\n" % (args.gpus, p.returncode))
        return p.returncode or 1
    print(line, flush=True)
    return 0


def cpu_info():
    model, phys = "unknown", set()
    try:
        pid = cid = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name") and model == "unknown":
                model = ln.split(":", 1)[1].strip()
            elif ln.startswith("physical id"):
                pid = ln.split(":", 1)[1].strip()
            elif ln.startswith("core id"):
                cid = ln.split(":", 1)[1].strip()
                phys.add((pid, cid))
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return model, len(phys) or (os.cpu_count() or 1), os.cpu_count() or 1, usable


def cpu_quota():
    """CPUs' worth of time the container may use (cgroup cpu.max / cfs quota), or None when unlimited: a box may show 256 logical
    CPUs and grant 16 -- 128 workers then share 16 cores' time and the "all cores" figure is a tenth of what the cores can do"""
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: (t.split()[0], t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()))):
        try:
            q, per = parse(open(path).read())
            if q not in ("max", "-1") and int(per) > 0 and int(q) > 0:
                return max(1, int(q) // int(per))
        except (OSError, ValueError, IndexError):
            continue
    return None


def _cpu_worker(job):
    """one PROCESS of the all-cores leg: its own address space (the reference allocates a verify buffer of the block's size in
    every compress() call, signal_packer_xdelta_hzr.cpp:59 -- threads of one process serialise on that mmap/page-fault traffic)"""
    packer, bps, nch, ns, nb, sample, start_at, stop_at, use_ref = job
    from oracle import oracle as orc_mod

    lib = orc_mod.Ref() if use_ref else orc_mod.Oracle()
    pk = lib.packer(packer, bps, nch, ns, nb)
    pk.compress(sample)  # first call pays page faults
    while time.time() < start_at:
        time.sleep(0.001)
    n = 0
    t0 = time.time()
    while time.time() < stop_at:
        pk.compress(sample)
        n += 1
    dt = time.time() - t0
    pk.close()
    return n, dt


def cpu_baseline(args, sample_native):
    """Time the checker on host cores: oracle/_ref (the compiled reference) when the prebuilt library travelled here, else our
    restatement.  Bounded sample.  One core first, then every physical core this process may use with one packer instance in
    one PROCESS per core (forked before anything touches the GPU)."""
    import multiprocessing as mp

    from oracle import oracle as orc_mod

    kind = "reference" if orc_mod.have_ref() else "port"
    lib = orc_mod.Ref() if kind == "reference" else orc_mod.Oracle()
    model, physical, logical, usable = cpu_info()
    quota = cpu_quota()
    nproc = max(1, min(usable, physical, quota or physical))  # every physical core this process may run on AND has the time for
    samples_per_block = args.nch * args.ns
    # 1 core
    pk = lib.packer(args.packer, args.bps, args.nch, args.ns, args.nb)
    pk.compress(sample_native)
    t0 = time.perf_counter()
    n1 = 0
    while time.perf_counter() - t0 < args.cpu_seconds / 3 or n1 < 2:
        pk.compress(sample_native)
        n1 += 1
    dt1 = time.perf_counter() - t0
    pk.close()
    # all cores
    span = args.cpu_seconds * 2 / 3
    start_at = time.time() + 2.0 + 0.01 * nproc  # (workers load the library and take their first call before the clock starts)
    job = (args.packer, args.bps, args.nch, args.ns, args.nb, sample_native, start_at, start_at + span, kind == "reference")
    with mp.get_context("fork").Pool(nproc) as pool:
        res = pool.map(_cpu_worker, [job] * nproc, chunksize=1)
    calls = sum(n for n, _ in res)
    dtn = max(dt for _, dt in res)
    v1 = n1 * samples_per_block / dt1 / 1e6
    vn = calls * samples_per_block / dtn / 1e6
    return {
        "value": round(vn, 2),
        "unit": "MSamples/s",
        "cores": nproc,
        "kind": kind,
        "value_1core": round(v1, 2),
        "scaling_efficiency": round(vn / (v1 * nproc), 3),
        "parallel": "one process per core the container has time for (min of physical cores, affinity, cgroup quota), one packer instance each",
        "cpu_model": model,
        "cpus": {"physical": physical, "logical": logical, "usable": usable, "cgroup_quota": quota},
        "sample": "%d + %d compress() calls of one %dch x %d x int%d synthetic block (%s), verify-decode included as in the reference"
        % (n1, calls, args.nch, args.ns, 8 * args.bps, args.packer),
    }


# the harness's band-pass (0.4-200 Hz Butterworth @ 2000 Sps, lib_rspt_test/rspt_test.cpp:123-125) and its history length (:129)
IIR_N = [1.00000000000, -3.14332095199, 3.70064088865, -1.97083923944, 0.41351972908]
IIR_D = [0.06722876941, 0.00000000000, -0.13445753881, 0.00000000000, 0.06722876941]
IIR_INIT = 2000
F64_DEP_NS = 2.0  # one dependent v_mul_f64 / v_add_f64 of a wave alone on its SIMD: 2.0 ns = the interval at which a lone wave issues at all (tools/issue_rate.hip, profiles/r03_issue_rate.txt)


def bench_prefilter(args):
    """Secondary line: the IIR pre-filter step of the reference's pipeline (rspt_test.cpp:116-136) on a device-resident batch.
    The bound is neither HBM nor MFMA but a serial fp64 recurrence: per sample one product and NC - 1 subtractions that each
    need the one before (filter_opt, iir_filter.cpp:79-104), NC = 5 here; the history initialisation (4 * 2000 calls of
    filter(), :106-110) is a chain of one product and 2 (NC - 1) - 1 sums per call.  `roofline.chain_floor_ms` is that chain
    at the measured latency of a dependent fp64 instruction; per-channel mode runs the channels of all blocks side by side,
    shared mode (bit-exact with the harness) chains the channels of a block behind one another."""
    import torch

    from rspt_amd import api, synth

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    api.lib()
    nch, ns, B, bps = args.nch, args.ns, args.blocks, args.bps
    pk = api.SignalPacker("xdelta_hzr", bps, nch, ns, args.nb)
    src = synth.synth_batch_native(B, nch, ns, first_block=0, bps=bps, ecg=True, device=dev)
    work = src.clone()
    per_channel = args.iir_mode == "per_channel"
    # check first (outside the timed region): blocks 0 and B-1 against the CPU restatement of the same mode
    pk.iir_prefilter_batch(work, IIR_N, IIR_D, IIR_INIT, per_channel=per_channel)
    torch.cuda.synchronize()
    verified = None
    if not args.no_verify:
        from oracle.oracle import Oracle

        orc = Oracle()
        verified = True
        for b in sorted({0, B - 1}):
            want = orc.iir_prefilter(src[b].cpu().numpy(), bps, nch, ns, IIR_N, IIR_D, IIR_INIT, shared_state=not per_channel)
            verified = verified and work[b].cpu().numpy().tobytes() == want
    for _ in range(args.warmup):
        pk.iir_prefilter_batch(work, IIR_N, IIR_D, IIR_INIT, per_channel=per_channel)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pk.iir_prefilter_batch(work, IIR_N, IIR_D, IIR_INIT, per_channel=per_channel)  # (in place: every step filters the previous step's output)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms = dt / args.steps * 1e3
    nc = len(IIR_N)
    # instructions the wave that holds a channel's filter state cannot hand to anyone else: per sample the NC - 1 feedback products and
    # NC - 1 subtractions of filter_opt; per history call of filter() NC - 1 products and 2 (NC - 1) sums (its feed-forward products are constants)
    ops = ns * 2 * (nc - 1) + 4 * IIR_INIT * 3 * (nc - 1)
    floor_ms = ops * F64_DEP_NS * 1e-6 * (1 if per_channel else nch)
    alg = 2 * B * pk.block_bytes  # read and written once
    res = {
        "metric": "MSamples/s prefilter (iir band-pass nc=5, %dch x %d int%d, %s)" % (nch, ns, 8 * bps, args.iir_mode),
        "value": round(B * nch * ns * args.steps / dt / 1e6, 1), "unit": "MSamples/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic (SURVEY 8d, ECG-like variant)",
        "verified": verified,
        "config": {"workload": "iir pre-filter of %d blocks of %dch x %d int%d, in place, device-resident; mode %s" % (B, nch, ns, 8 * bps, args.iir_mode),
                   "coefficients": "band-pass of rspt_test.cpp:123-125 (5 coefficients), init_history_values(first sample, %d)" % IIR_INIT},
        "roofline": {"bound": "hbm", "kernel": "k_iir", "achieved": round(alg / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": None,
                     "true_bound": "serial fp64 recurrence, not bandwidth",
                     "recurrence_instructions_per_channel": ops, "dependent_f64_instruction_ns": F64_DEP_NS,
                     "chain_floor_ms": round(floor_ms, 3), "ms_over_chain_floor": round(ms / floor_ms, 2)},
    }
    print(json.dumps(res), flush=True)
    pk.close()
    return 0 if verified is not False else 3


def kernels_sha():
    """fingerprint of the kernel sources: counters collected on other sources are stale"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "rspt_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def main():
    args = parse_args()
    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))

    if args.op == "prefilter":
        sys.exit(bench_prefilter(args))

    if args.slots > 1:
        # The HIP runtime multiplexes all streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that
        # share a queue run one after the other: with 2 slots (4 streams of the handles + torch's) the two steps in flight landed
        # on one queue and nothing overlapped (128-block shard: 0.163 ms per step; 0.115 with 8 queues -- profiles/r04_notes.md 3).
        # Read by the runtime when it initialises, i.e. before anything here touches the GPU.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(8, 2 * args.slots)))
    import torch

    from rspt_amd import api, shard, synth

    # the CPU leg runs FIRST (one GPU, no launcher): its worker processes are forked while this process has not touched the GPU;
    # the sample is block 0 of the first batch -- the generator is integer-only, so the CPU makes the same bytes as the device
    cpu_res = None
    if "RANK" not in os.environ and args.gpus == 1 and not args.no_cpu:
        cpu_res = cpu_baseline(args, synth.synth_native(args.nch, args.ns, 0, args.bps, device="cpu").numpy())

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d rank(s)\n" % (args.gpus, world))
        sys.exit(2)
    api.lib()  # load (never build: a stale library fails loudly) before any GPU call
    dist = None
    if "RANK" in os.environ:  # launched by torch.distributed.run (also with one rank: same code path)
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    nch, ns = args.nch, args.ns
    c5 = args.workload == "c5"
    if c5:
        first, B = shard.shard_range(args.blocks, rank, world)  # this rank's contiguous shard of the job's blocks
        total_blocks = args.blocks
    else:
        B = args.blocks
        first = rank * B
        total_blocks = world * B
    S = args.slots
    decomp_op = args.op == "decompress"
    if S < 1 or (S > 1 and not c5 and dist is not None):
        sys.stderr.write("bench.py: --slots with --workload c3 is a one-GPU line (secondary packers, decompress)\n")
        sys.exit(2)
    pks = [api.SignalPacker(args.packer, args.bps, nch, ns, args.nb, device=local_rank) for _ in range(S)]
    pk = pks[0]
    for q in pks:
        q.reserve(B)
    nbuf = 2 if decomp_op else 2 * S  # every slot alternates between two buffer sets (as its handle alternates between two workspace sets)
    # synthetic input, resident in HBM: TWO distinct batches alternate in the timed loop (the front end's behaviour depends on
    # what the previous call left in the plane workspace); every rank gets different blocks (SURVEY 8d generator)
    d_src = [synth.synth_batch_native(B, nch, ns, first_block=first + s * total_blocks, bps=args.bps, device=dev) for s in range(nbuf)]
    if args.big_endian:  # the same samples, bytes reversed; the streams are the little-endian ones (checked below against the oracle)
        d_le = d_src
        d_src = [x.view(B, -1, args.bps).flip(2).contiguous().view(B, -1) for x in d_le]
        for q in pks:
            q.set_byte_order(True)
    else:
        d_le = d_src
    dst_stride = (pk.max_compressed_size + 255) // 256 * 256
    d_dst = [torch.empty((B, dst_stride), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    d_sizes = [torch.empty(B, dtype=torch.int64, device=dev) for _ in range(nbuf)]
    stream = torch.cuda.current_stream(dev)
    # several steps in flight: each on its handle's own stream (two streams per handle exist anyway; further ones would share
    # hardware queues with them -- see GPU_MAX_HW_QUEUES above)
    slot_streams = [stream] if S == 1 else [torch.cuda.ExternalStream(q.stream_ptr, device=dev) for q in pks]

    # gather plan (N>1): each rank packs its streams into a container on the device
    # (rspt_hip_pack_batch_dev) and the containers go to rank 0 over RCCL: sizes by
    # all_gather, payload by send/recv (rspt_amd/shard.py).
    do_gather = dist is not None and not args.no_gather
    payload_every_step = c5 or args.gather_every_step
    need_pack = c5 or do_gather
    side = torch.cuda.Stream(dev) if (do_gather and not c5) else None
    bound = pk.pack_bound(B)
    packed = [torch.empty(bound, dtype=torch.uint8, device=dev) for _ in range(nbuf)] if need_pack else None
    totals = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(nbuf)] if need_pack else None
    recv_bufs = None
    if do_gather and rank == 0:
        recv_bufs = [None] + [torch.empty(pk.pack_bound(shard.shard_range(args.blocks, r, world)[1] if c5 else B), dtype=torch.uint8, device=dev)
                              for r in range(1, world)]
    gathered_bytes = [0]
    # c5: the containers of step i leave during step i + 1 (sizes by a device all-gather, read by the host one step later: no host
    # round trip inside a step; rspt_amd/shard.py LaggedGather); the last step's payload is flushed inside the timed region
    lag = shard.LaggedGather(dst=0, recv_bufs=recv_bufs, device=dev, timing=True) if (do_gather and c5) else None
    slot_free = [None, None]  # event: the gather that last used this slot's buffers has finished
    sizes_all = [torch.zeros((world, B), dtype=torch.int64, device=dev) for _ in range(2)] if (do_gather and not c5) else None

    decomp = args.op == "decompress"
    if decomp:  # the streams to decode: both batches compressed once, outside the timed region
        assert not c5 and dist is None, "--op decompress is a one-GPU c3 line"
        for s_ in range(2):
            pk.compress_batch(d_src[s_], d_dst[s_], d_sizes[s_], dst_stride)
        torch.cuda.synchronize()
        d_back = [torch.empty((B, pk.block_bytes), dtype=torch.uint8, device=dev) for _ in range(S)]  # (one output per batch in flight)
        d_used = [torch.empty(B, dtype=torch.int64, device=dev) for _ in range(S)]
        last_of_slot = [None] * S

    nstep = [0]  # steps so far: runs on through warm-up and timed loop, so that the buffer parity never jumps (the lagged gather's own
    #              slot parity is a step counter too)

    def one_step(_i):
        slot = nstep[0] & 1
        nstep[0] += 1
        if decomp:
            h = (nstep[0] - 1) % S  # --slots S: batch n is decoded on handle and stream n % S
            with torch.cuda.stream(slot_streams[h]):
                pks[h].decompress_batch(d_dst[slot], B, dst_stride, d_back[h], d_used[h])
            last_of_slot[h] = slot
            return
        if c5:
            # strong scaling: compress the shard, pack it, gather to rank 0 -- all inside the step.  --slots S: step n runs on handle
            # and stream n % S (no event between the slots: nothing is shared), on buffer set n % 2S
            n = nstep[0] - 1
            h, bs = n % S, n % nbuf
            with torch.cuda.stream(slot_streams[h]):
                pks[h].compress_batch(d_src[bs], d_dst[bs], d_sizes[bs], dst_stride)
                if lag is not None:
                    # the payload group posted a step ago read the container of two steps ago; this buffer set was last used 2S >= 2
                    # steps ago, and the gather stream runs in order: behind that event its container has left
                    lag.wait_slot_free(slot)
                pks[h].pack_batch(d_dst[bs], d_sizes[bs], packed[bs], totals[bs])
                if lag is not None:
                    lag.step(packed[bs], totals[bs])
            return
        if S > 1:  # one GPU, no exchange: batch n on handle and stream n % S, buffer set n % 2S
            n = nstep[0] - 1
            with torch.cuda.stream(slot_streams[n % S]):
                pks[n % S].compress_batch(d_src[n % nbuf], d_dst[n % nbuf], d_sizes[n % nbuf], dst_stride)
            return
        if slot_free[slot] is not None and not slot_free[slot].query():  # (two steps old: almost always done -- then no barrier packet)
            stream.wait_event(slot_free[slot])
        pk.compress_batch(d_src[slot], d_dst[slot], d_sizes[slot], dst_stride)
        if do_gather:
            if payload_every_step:
                pk.pack_batch(d_dst[slot], d_sizes[slot], packed[slot], totals[slot])
            ev = torch.cuda.Event()
            ev.record(stream)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                if payload_every_step:
                    got = shard.gather_containers(packed[slot], totals[slot], dst=0, recv_bufs=recv_bufs)
                    if got is not None:
                        gathered_bytes[0] = sum(n for _, n in got)
                else:
                    # the index only: every rank learns the size of every stream of the step (ncclAllGather of
                    # the per-block sizes); device tensors, no host round trip
                    dist.all_gather(list(sizes_all[slot].unbind(0)), d_sizes[slot])
                slot_free[slot] = torch.cuda.Event()
                slot_free[slot].record(side)

    def fence():
        if lag is not None:
            lag.flush()  # the last step's containers travel inside the timed region
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        one_step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(i)
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if lag is not None:
        gathered_bytes[0] = lag.gathered_bytes
    # --slots > 1: the same step with ONE step in flight, for the latency beside the throughput (no gather in this leg)
    single_ms = None
    if S > 1:
        torch.cuda.synchronize()
        with torch.cuda.stream(slot_streams[0]):
            for rep_ in range(2):
                if rep_ == 1:
                    torch.cuda.synchronize()
                    g0 = time.perf_counter()
                for i in range(args.steps):
                    if decomp:
                        pk.decompress_batch(d_dst[i & 1], B, dst_stride, d_back[0], d_used[0])
                        last_of_slot[0] = i & 1
                        continue
                    pk.compress_batch(d_src[i & 1], d_dst[i & 1], d_sizes[i & 1], dst_stride)
                    if need_pack:
                        pk.pack_batch(d_dst[i & 1], d_sizes[i & 1], packed[i & 1], totals[i & 1])
            torch.cuda.synchronize()
            single_ms = (time.perf_counter() - g0) / args.steps * 1e3
    # c3, N > 1: the streams of the last step travel to rank 0 once, outside the timed steps (sustained, rank 0's xGMI ingress
    # could take the output of only 2-3 GPUs at this rate); its time is reported beside the metric
    gather_ms = None
    if do_gather and not payload_every_step:
        last = (nstep[0] - 1) & 1
        torch.cuda.synchronize()
        dist.barrier()
        g0 = time.perf_counter()
        pk.pack_batch(d_dst[last], d_sizes[last], packed[last], totals[last])
        got = shard.gather_containers(packed[last], totals[last], dst=0, recv_bufs=recv_bufs)
        torch.cuda.synchronize()
        dist.barrier()
        gather_ms = (time.perf_counter() - g0) * 1e3
        if got is not None:
            gathered_bytes[0] = sum(n for _, n in got)

    # outside the timed region: both batches once more, their first and last streams against the oracle
    verified = None
    if decomp:  # the last decoded batch is the input again (lossless packers), and the decoder consumed every stream in full
        decode_ok = True
        for h in range(S):
            last = last_of_slot[h]
            if last is None:
                continue
            decode_ok = decode_ok and bool(torch.equal(d_used[h], d_sizes[last]))
            if args.packer in ("xdelta_hzr", "hzr"):
                decode_ok = decode_ok and bool(torch.equal(d_back[h].view(-1), d_src[last].view(-1)))
    for s in range(2):
        pk.compress_batch(d_src[s], d_dst[s], d_sizes[s], dst_stride)
    torch.cuda.synchronize()
    if rank == 0 and not args.no_verify:
        from oracle.oracle import Oracle

        orc = Oracle()
        big_dct = args.packer == "dct" and ns > 8192  # beyond the reference's own reach: the fp64 restatement is the checker
        po = None if big_dct else orc.packer(args.packer, args.bps, nch, ns, args.nb)
        verified = True
        for s in range(2):
            for b in sorted({0, B - 1}):
                x = d_le[s][b].cpu().numpy()
                want = orc.dct_big_compress(x, args.bps, nch, ns)[0] if big_dct else po.compress(x)
                n = int(d_sizes[s][b])
                got_b = d_dst[s][b, :n].cpu().numpy().tobytes() if 0 < n <= dst_stride else b""
                if args.packer == "dct" and ns > 8192:
                    verified = verified and abs(len(got_b) / max(1, len(want)) - 1) <= 0.01  # FFT path: CR gate, not bytes
                else:
                    verified = verified and got_b == want
        if decomp:
            verified = verified and decode_ok
        if not verified:
            sys.stderr.write("bench.py: a produced stream DIFFERS from the oracle\n")

    # per-kernel durations (HIP events on the launch stream), separate profiled pass over both batches
    pk.set_profiling(True)
    acc = {}
    reps = max(4, min(args.steps, 10))
    for i in range(reps):
        pk.compress_batch(d_src[i & 1], d_dst[i & 1], d_sizes[i & 1], dst_stride)
        for k, v in pk.stage_times().items():
            acc[k] = acc.get(k, 0.0) + v / reps
    pk.set_profiling(False)
    torch.cuda.synchronize()
    out_bytes = int((d_sizes[0].sum().item() + d_sizes[1].sum().item()) // 2)
    in_bytes = B * pk.block_bytes
    nb_now = pk.nb

    # second denominator: what a plain device copy reaches on this very GPU (bytes read + bytes written)
    copy_gbs = None
    try:
        a = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
        c = torch.empty_like(a)
        c.copy_(a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            c.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 2 * a.numel() * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del a, c
    except Exception:
        copy_gbs = None

    if rank == 0:
        alg_bytes = in_bytes + out_bytes  # bytes = input_bytes + output_bytes per launch
        if decomp:  # the decoder's kernels are not bracketed one by one: the whole step stands for "the kernel"
            acc = {"decompress_all": dt / args.steps * 1e3}
        dominant = max(acc, key=acc.get)
        plane_bytes = B * nb_now * nch * ns  # what the front end hands to the hzr stage
        # the dominant kernel's OWN algorithmic bytes (what it has to read and write once), over its own duration
        own = {"preprocess": in_bytes + plane_bytes, "hzr_encode": plane_bytes + out_bytes, "hzr_fused": plane_bytes + out_bytes,
               "hzr_hist": plane_bytes}.get(dominant, alg_bytes)
        own = min(own, alg_bytes)  # never more than SURVEY 8(d)'s whole-launch figure
        achieved = own / (acc[dominant] * 1e-3) / 1e9
        pipeline_gbs = alg_bytes / (dt / args.steps) / 1e9  # the WHOLE path: algorithmic bytes over the timed step (launch gaps included)
        kernel_sum_ms = sum(acc.values())
        traffic, traffic_note, traffic_pipeline = None, None, None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        default_shape = (args.packer, args.workload, B, nch, ns, args.nb, args.bps) == ("xdelta_hzr", "c3", 64, 64, 65536, 3, 4)
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if not default_shape:
                    traffic_note = "counters were collected on the default workload only"
                elif tj.get("_kernels_sha") != kernels_sha():
                    traffic_note = "stale: counters collected on other kernel sources (%s)" % tj.get("_kernels_sha")
                else:
                    traffic = tj.get(dominant)
                    traffic_pipeline = sum(int(v) for k_, v in tj.items() if not k_.startswith("_"))  # every kernel of the launch sequence
                    traffic_note = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes at kernels_sha %s" % tj.get("_kernels_sha")
            except Exception:
                traffic = None
        shape = "%dch x %d int%d" % (nch, ns, 8 * args.bps)
        samples_job = total_blocks * nch * ns
        res = {
            "metric": "MSamples/s %s (%s, %s)" % (args.op, args.packer, shape),
            "value": round(samples_job * args.steps / dt / 1e6, 1),
            "unit": "MSamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong" if c5 else "weak",
            "vs_baseline": None,
            "dtype": "int%d%s" % (8 * args.bps, " big-endian" if args.big_endian else ""),
            "data": "synthetic (parabolic sine + 4-bit counter-hash noise per SURVEY 8d; hashes differ from the survey's xorshift variant)",
            "verified": verified,
            "config": {
                "workload": ("%s nb=%d, %d blocks in total of %s (BASELINE configs[4]), contiguous shards, container pack + gather to rank 0 "
                             "inside the step, device-resident%s" % (args.packer, args.nb, total_blocks, shape,
                                                                "; %d steps in flight per rank, each on its own handle and stream" % S if S > 1 else "")) if c5 else
                            ("%s nb=%d, %d blocks/GPU/step of %s (BASELINE configs[2] shape, xdelta_hzr path), two alternating batches, "
                             "device-resident%s" % (args.packer, args.nb, B, shape,
                                                    "; %d batches in flight, each on its own handle and stream" % S if S > 1 else "")),
                "blocks_per_gpu": B,
                "steps_in_flight": S,
                "ms_per_step_one_in_flight": round(single_ms, 4) if single_ms is not None else None,
                "compression_ratio": round(in_bytes / out_bytes, 4),
                "gather": ("every step, inside the timed region%s" % (" (sizes by device all-gather, payload one step behind: no host sync in the step)" if lag is not None else "")
                           if payload_every_step else "sizes every step, payload once after the timed steps") if do_gather else False,
                "gathered_bytes": gathered_bytes[0],
                "gather_ms": round(gather_ms, 3) if gather_ms is not None else None,
                "gather_ms_per_step": (lambda v: round(v, 4) if v is not None else None)(lag.mean_payload_ms()) if lag is not None else None,
                "parallelism": "shard%d" % world,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": dominant,
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,  # HBM bytes of the DOMINANT kernel per launch (compare with kernel_algorithmic_bytes)
                "traffic_pipeline": traffic_pipeline,  # ... of the whole launch sequence (compare with algorithmic_bytes_per_launch)
                "traffic_over_algorithmic": round(traffic_pipeline / alg_bytes, 3) if traffic_pipeline else None,
                "traffic_note": traffic_note,
                "kernel_algorithmic_bytes": own,
                "algorithmic_bytes_per_launch": alg_bytes,
                "kernel_ms": {k: round(v, 4) for k, v in acc.items()},
                "kernel_sum_ms": round(kernel_sum_ms, 4),
                "pipeline_gbs": round(pipeline_gbs, 1),
                "pipeline_frac": round(pipeline_gbs / HBM_PEAK_GBS, 4),
                "device_copy_gbs": round(copy_gbs, 1) if copy_gbs else None,
                # which resource the STEP is bound by, from the counters rather than from the longest kernel: the share of the step
                # that moving the pipeline's measured HBM bytes at this GPU's copy rate accounts for (well under 1: not HBM)
                "hbm_time_share_of_step": round(traffic_pipeline / (copy_gbs * 1e9) / (dt / args.steps), 3) if (traffic_pipeline and copy_gbs) else None,
                "step_bound": (("hbm" if traffic_pipeline / (copy_gbs * 1e9) / (dt / args.steps) >= 0.7 else
                                "vector issue + barrier / LDS waits of the hzr stages (profiles/r03_notes.md 1b); HBM busy for the share above")
                               if (traffic_pipeline and copy_gbs) else None),
                "pipeline_frac_of_device_copy": round(pipeline_gbs / copy_gbs, 4) if copy_gbs else None,
            },
        }
        if cpu_res is not None:
            res["cpu_baseline"] = cpu_res
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    for q in pks:
        q.close()
    if rank == 0 and verified is False:
        sys.exit(3)


if __name__ == "__main__":
    main()
