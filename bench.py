#!/usr/bin/env python3
"""bench.py -- MSamples/s of xdelta_hzr compress on 64ch x 65536 x int32 blocks.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (rspt_hip_compress_batch_dev) over one
batch of `--blocks` synthetic blocks per GPU, inputs already resident in HBM.
For N > 1 the driver launches this file under torch.distributed.run, one rank
per GPU; blocks are independent, so every rank compresses its own shard (weak
scaling) and the compressed streams are gathered to rank 0 over RCCL.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra
objects: `roofline` (dominant kernel, HIP-event timed on the launch stream) and
`cpu_baseline` (the checker library timed on this node's host cores, N=1 only).
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=64, help="blocks per GPU per step (each 64ch x 65536 x int32 = 16 MiB)")
    ap.add_argument("--nch", type=int, default=64)
    ap.add_argument("--ns", type=int, default=65536)
    ap.add_argument("--nb", type=int, default=3)
    ap.add_argument("--bps", type=int, default=4, choices=[1, 2, 3, 4], help="bytes per sample of the input (the metric is quoted on 4)")
    ap.add_argument("--packer", default="xdelta_hzr", choices=["xdelta_hzr", "hzr", "hadamard", "dct"])
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the RCCL exchange altogether")
    ap.add_argument("--gather-every-step", action="store_true",
                    help="N>1: ship every step's streams to rank 0 (link-bound beyond 2-3 GPUs: see DESIGN.md); default: the sizes "
                         "index every step, the payload once after the timed steps")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--verify", action="store_true", help="check one block of the batch against the oracle before timing")
    return ap.parse_args()


def cpu_baseline(args, sample_native):
    """Time the checker on host cores: oracle/_ref (the compiled reference) when the
    prebuilt library travelled here, else our restatement.  Bounded sample."""
    from oracle import oracle as orc_mod

    kind = "reference" if orc_mod.have_ref() else "port"
    lib = orc_mod.Ref() if kind == "reference" else orc_mod.Oracle()
    nthreads = min(os.cpu_count() or 1, 16)
    samples_per_block = args.nch * args.ns
    counts = [0] * nthreads
    deadline = [0.0]

    def work(i):
        pk = lib.packer(args.packer, args.bps, args.nch, args.ns, args.nb)
        pk.compress(sample_native)  # first call pays page faults
        while time.perf_counter() < deadline[0]:
            pk.compress(sample_native)
            counts[i] += 1
        pk.close()

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
    # all cores, one packer instance per thread (ctypes releases the GIL)
    deadline[0] = time.perf_counter() + args.cpu_seconds * 2 / 3 + 2.0
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dtn = time.perf_counter() - t0
    return {
        "value": round(sum(counts) * samples_per_block / dtn / 1e6, 2),
        "unit": "MSamples/s",
        "cores": nthreads,
        "kind": kind,
        "value_1core": round(n1 * samples_per_block / dt1 / 1e6, 2),
        "sample": "%d + %d compress() calls of one %dch x %d x int32 synthetic block (%s), verify-decode included as in the reference"
        % (n1, sum(counts), args.nch, args.ns, args.packer),
    }


def main():
    args = parse_args()
    import torch

    from rspt_amd import api, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if "RANK" in os.environ:  # launched by torch.distributed.run (also with one rank: same code path)
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    B, nch, ns = args.blocks, args.nch, args.ns
    pk = api.SignalPacker(args.packer, args.bps, nch, ns, args.nb, device=local_rank)
    pk.reserve(B)
    # synthetic input, resident in HBM; every rank gets different blocks (SURVEY 8d generator)
    d_src = synth.synth_batch_native(B, nch, ns, first_block=rank * B, bps=args.bps, device=dev)
    dst_stride = (pk.max_compressed_size + 255) // 256 * 256
    d_dst = [torch.empty((B, dst_stride), dtype=torch.uint8, device=dev) for _ in range(2)]
    d_sizes = [torch.empty(B, dtype=torch.int64, device=dev) for _ in range(2)]
    stream = torch.cuda.current_stream(dev)

    if args.verify and rank == 0:
        from oracle.oracle import Oracle

        pk.compress_batch(d_src, d_dst[0], d_sizes[0], dst_stride)
        torch.cuda.synchronize()
        o = Oracle()
        po = o.packer(args.packer, 4, nch, ns, args.nb)
        want = po.compress(d_src[0].cpu().numpy())
        got = d_dst[0][0, : int(d_sizes[0][0])].cpu().numpy().tobytes()
        assert got == want, "GPU stream differs from the oracle"

    # gather plan (N>1): each rank packs its streams into a container on the device
    # (rspt_hip_pack_batch_dev) and the containers go to rank 0 over RCCL: sizes by
    # all_gather, payload by send/recv (rspt_amd/shard.py).  It runs on a side stream so
    # that step i's gather overlaps step i+1's compression.
    from rspt_amd import shard

    do_gather = dist is not None and not args.no_gather
    side = torch.cuda.Stream(dev) if do_gather else None
    bound = pk.pack_bound(B)
    packed = [torch.empty(bound, dtype=torch.uint8, device=dev) for _ in range(2)] if do_gather else None
    totals = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(2)] if do_gather else None
    recv_bufs = None
    if do_gather and rank == 0:
        recv_bufs = [None] + [torch.empty(bound, dtype=torch.uint8, device=dev) for _ in range(world - 1)]
    gathered_bytes = [0]
    slot_free = [None, None]  # event: the gather that last used this slot's buffers has finished

    sizes_all = [torch.zeros((world, B), dtype=torch.int64, device=dev) for _ in range(2)] if do_gather else None

    def one_step(i):
        slot = i & 1
        if slot_free[slot] is not None and not slot_free[slot].query():  # (two steps old: almost always done -- then no barrier packet)
            stream.wait_event(slot_free[slot])
        pk.compress_batch(d_src, d_dst[slot], d_sizes[slot], dst_stride)
        if do_gather:
            if args.gather_every_step:
                pk.pack_batch(d_dst[slot], d_sizes[slot], packed[slot], totals[slot])
            ev = torch.cuda.Event()
            ev.record(stream)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                if args.gather_every_step:
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

    # N > 1: the streams of the last step travel to rank 0 once, outside the timed steps (sustained, rank 0's xGMI ingress
    # could take the output of only 2-3 GPUs at this rate); its time is reported beside the metric
    gather_ms = None
    if do_gather and not args.gather_every_step:
        last = (args.steps - 1) & 1
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

    # per-kernel durations (HIP events on the launch stream), separate profiled pass
    pk.set_profiling(True)
    acc = {}
    reps = max(3, min(args.steps, 10))
    for i in range(reps):
        pk.compress_batch(d_src, d_dst[0], d_sizes[0], dst_stride)
        for k, v in pk.stage_times().items():
            acc[k] = acc.get(k, 0.0) + v / reps
    pk.set_profiling(False)
    torch.cuda.synchronize()
    out_bytes = int(d_sizes[0].sum().item())
    in_bytes = B * pk.block_bytes
    samples_per_step = B * nch * ns

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
        dominant = max(acc, key=acc.get)
        alg_bytes = in_bytes + out_bytes  # bytes = input_bytes + output_bytes per launch
        achieved = alg_bytes / (acc[dominant] * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        default_shape = (args.packer, B, nch, ns, args.nb, args.bps) == ("xdelta_hzr", 64, 64, 65536, 3, 4)  # what the counters were collected on
        if default_shape and os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dominant)
            except Exception:
                traffic = None
        res = {
            "metric": "MSamples/s compress (xdelta_hzr, 64ch x 65536 int32)",
            "value": round(world * samples_per_step * args.steps / dt / 1e6, 1),
            "unit": "MSamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int%d" % (8 * args.bps),
            "data": "synthetic",
            "config": {
                "workload": "%s nb=%d, %d blocks/GPU/step of %dch x %d x int%d (BASELINE configs[2] shape, xdelta_hzr path), device-resident"
                % (args.packer, args.nb, B, nch, ns, 8 * args.bps),
                "blocks_per_gpu": B,
                "compression_ratio": round(in_bytes / out_bytes, 4),
                "gather": ("every step" if args.gather_every_step else "sizes every step, payload once after the timed steps") if do_gather else False,
                "gathered_bytes": gathered_bytes[0],
                "gather_ms": round(gather_ms, 3) if gather_ms is not None else None,
                "parallelism": "shard%d" % world,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": dominant,
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes,
                "kernel_ms": {k: round(v, 4) for k, v in acc.items()},
                "pipeline_frac": round(alg_bytes / (sum(acc.values()) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "device_copy_gbs": round(copy_gbs, 1) if copy_gbs else None,
                "frac_of_device_copy": round(achieved / copy_gbs, 4) if copy_gbs else None,
            },
        }
        if world == 1 and not args.no_cpu:
            res["cpu_baseline"] = cpu_baseline(args, d_src[0].cpu().numpy())
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    pk.close()


if __name__ == "__main__":
    main()
