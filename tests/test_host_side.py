"""Host-side pieces around the path: big-endian ingest (lib_signalpacker/utils.cpp reverse_byte_order branches),
page-locked staging buffers for the host-pointer API, and the C++ factories' device placement."""
import os
import subprocess
import sys

import numpy as np
import pytest

import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _reverse_samples(native, bps):
    return np.ascontiguousarray(native.reshape(-1, bps)[:, ::-1]).reshape(-1)


@pytest.mark.parametrize("bps", [1, 2, 3, 4])
def test_reference_big_endian_ingest_is_a_byte_reversal(ref, bps):
    """convert_native_to_i32(reverse_byte_order=true) of the byte-reversed block == the little-endian conversion
    (utils.cpp:127-137,145-154,162-170,178-184): what the GPU's big-endian mode relies on -- pinned by the REAL reference"""
    nch, ns = 3, 257
    le = cases._rand_native(nch, ns, bps, 700 + bps, 1 << (8 * bps - 2))
    be = _reverse_samples(le, bps)
    a = ref.native_to_i32(le, ns, nch, bps, reverse_byte_order=False)
    b = ref.native_to_i32(be, ns, nch, bps, reverse_byte_order=True)
    assert (a == b).all()
    if bps > 1:  # (the reference's 1-byte reverse branch of convert_i32_to_native writes at offset +1: utils.cpp:116-118)
        assert ref.i32_to_native(a, bps, reverse_byte_order=True) == be.tobytes()


def test_cxx_shard_example_compiles(tmp_path):
    """tests/cxx/shard_devices.cpp: one packer per device on one thread each, through include/signal_packer.h alone"""
    from rspt_amd import build

    exe = tmp_path / "shard_devices"
    lib_dir = os.path.dirname(build.LIB)
    subprocess.check_call(["g++", "-std=c++11", "-pthread", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cxx", "shard_devices.cpp"),
                           "-o", str(exe), "-L" + lib_dir, "-lrspt_hip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    assert exe.exists()


@pytest.fixture(scope="module")
def api():
    from rspt_amd import api as a

    assert a.lib().rspt_hip_device_count() > 0, "no gfx950 device visible"
    return a


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ds3x20000_i24_xdelta", "esc_i16_big", "ecg12x8192_xdelta", "sine16384_i8_xdelta", "ds3x16384_i24_hadamard"])
def test_big_endian_feed(api, orc, golden, packer_cases, name):
    """a big-endian feed compresses to the stream of the byte-reversed block, and decompress hands big-endian samples back"""
    import torch

    c, g = packer_cases[name], golden["packers"][name]
    be = _reverse_samples(c["data"], c["bps"])
    pk = api.SignalPacker(c["kind"], c["bps"], c["nch"], c["ns"], c["nb"])
    pk.set_byte_order(big_endian=True)
    got = pk.compress(be)
    assert len(got) == g["size"] and orc.fnv1a(got) == g["fnv1a"]
    dec, used = pk.decompress(got)
    assert used == len(got)
    if g["lossless"]:
        assert dec == be.tobytes()
    # batched form, and back to little-endian on the same handle
    d = torch.from_numpy(np.stack([be, be])).cuda()
    d_dst, d_sizes = pk.compress_batch(d)
    torch.cuda.synchronize()
    assert d_dst[1, : int(d_sizes[1])].cpu().numpy().tobytes() == got or c["kind"] == "xdelta_hzr"  # (xdelta: nb may have escalated after call 1)
    pk.set_byte_order(big_endian=False)
    pk2 = api.SignalPacker(c["kind"], c["bps"], c["nch"], c["ns"], c["nb"])
    assert pk2.compress(c["data"]) == got
    pk.close()
    pk2.close()


@pytest.mark.gpu
def test_host_api_with_page_locked_buffers(api, orc):
    """rspt_hip_host_alloc: the host-pointer entry points on page-locked buffers (DMA at link rate) give the same bytes"""
    from rspt_amd import synth

    nch, ns = 12, 8192
    x = synth.synth_native(nch, ns, block_index=7).numpy().reshape(-1)
    pk = api.new_xdelta_hzr(4, nch, ns, 3)
    src, dst, back = api.HostBuffer(x.size), api.HostBuffer(2 * x.size), api.HostBuffer(x.size)
    src.a[:] = x
    n = pk.compress_into(src.a, dst.a)
    want = orc.packer("xdelta_hzr", 4, nch, ns, 3).compress(x)
    assert dst.a[:n].tobytes() == want
    used = pk.decompress_into(dst.a, back.a)
    assert used == n and back.a.tobytes() == x.tobytes()
    for b in (src, dst, back):
        b.close()
    pk.close()


@pytest.mark.gpu
@pytest.mark.parametrize("pinned", [False, True])
def test_compress_many_equals_a_loop_of_compress_calls(api, orc, pinned):
    """rspt_hip_compress_many: 150 blocks through the upload | compress | download pipeline (several chunks and a tail; the
    amplitudes step up in the middle, so nb escalates inside a chunk) == the oracle's packer fed the same blocks one by one"""
    nch, ns, bps, n = 8, 2048, 4, 150
    blocks = [cases._rand_native(nch, ns, bps, 4000 + i, 2000 if i < 70 else 1 << 27, walk=bool(i & 1)) for i in range(n)]
    po = orc.packer("xdelta_hzr", bps, nch, ns, 2)
    want = [po.compress(b) for b in blocks]
    pk = api.new_xdelta_hzr(bps, nch, ns, 2)
    stride = (pk.max_compressed_size + 63) // 64 * 64
    if pinned:
        hs, hd = api.HostBuffer(n * pk.block_bytes), api.HostBuffer(n * stride)
        src, out = hs.a, hd.a.reshape(n, stride)
    else:
        src, out = np.empty(n * pk.block_bytes, dtype=np.uint8), np.empty((n, stride), dtype=np.uint8)
    src[:] = np.concatenate(blocks)
    lens = pk.compress_many(src, out)
    for i in range(n):
        assert lens[i] == len(want[i]) and out[i, : lens[i]].tobytes() == want[i], i
    assert pk.nb == orc.packer_nb(po) > 2
    # and back: the same pipeline the other way round (every stream with the nb the sequence ended on)
    back = np.empty(n * pk.block_bytes, dtype=np.uint8)
    tail = slice(70, n)  # the blocks behind the escalation: written with the final nb, which is what the handle holds now
    used = pk.decompress_many(out[tail], back[: (n - 70) * pk.block_bytes])
    assert (used == lens[tail]).all()
    back[:] = 0
    used = pk.decompress_many(out[tail], back[: (n - 70) * pk.block_bytes], lengths=lens[tail])  # upload the streams only
    assert (used == lens[tail]).all()
    assert back[: (n - 70) * pk.block_bytes].tobytes() == np.concatenate(blocks[70:]).tobytes()
    damaged = out[tail].copy()
    damaged[3, 15] = 7  # stream 3: the first hzr block header [len-1:2][crc:4][mode:1] starts at 9 -> invalid encoding mode
    with pytest.raises(api.RsptHipError) as e:
        pk.decompress_many(damaged, back[: (n - 70) * pk.block_bytes])
    assert e.value.status == -6
    # a destination stride too short for some streams: those report the size they need, the others arrive
    short = int(np.median(lens))
    out2 = np.zeros((n, short), dtype=np.uint8)
    pk2 = api.new_xdelta_hzr(bps, nch, ns, 2)
    lens2 = pk2.compress_many(src, out2, raise_on_small=False)
    assert (lens2 == lens).all()
    for i in range(n):
        if lens[i] <= short:
            assert out2[i, : lens[i]].tobytes() == want[i], i
    with pytest.raises(api.RsptHipError) as e:
        pk2.compress_many(src, out2)
    assert e.value.status == -5
    pk.close()
    pk2.close()
    if pinned:
        hs.close()
        hd.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,bps,nch,ns", [("hzr", 2, 5, 3000), ("hadamard", 4, 3, 4096), ("dct", 4, 2, 1000)])
def test_many_entry_points_for_the_other_packers(api, orc, kind, bps, nch, ns):
    """compress_many / decompress_many are the per-call entry points in a pipeline, whatever the packer"""
    n = 9
    blocks = [cases._rand_native(nch, ns, bps, 5100 + i, 3000, walk=True) for i in range(n)]
    po = orc.packer(kind, bps, nch, ns)
    want = [po.compress(b) for b in blocks]
    pk = api.SignalPacker(kind, bps, nch, ns)
    stride = (pk.max_compressed_size + 63) // 64 * 64
    out = np.zeros((n, stride), dtype=np.uint8)
    lens = pk.compress_many(np.concatenate(blocks), out)
    for i in range(n):
        assert out[i, : lens[i]].tobytes() == want[i], (kind, i)
    back = np.empty(n * pk.block_bytes, dtype=np.uint8)
    used = pk.decompress_many(out, back, lengths=lens)
    assert (used == lens).all()
    for i in range(n):
        assert back[i * pk.block_bytes : (i + 1) * pk.block_bytes].tobytes() == bytes(po.decompress(want[i])[0]), (kind, i)
    pk.close()


@pytest.mark.gpu
def test_cxx_factories_follow_the_device_setting(api, tmp_path):
    """RSPT_HIP_DEVICE / rspt_cxx_set_device place the C++ factories' packers; the sharding example runs on every visible GPU"""
    from rspt_amd import build

    L = api.lib()
    ndev = L.rspt_hip_device_count()
    prev = L.rspt_cxx_set_device(ndev)  # one past the last device: no packer, an error on stderr -- never a fall-back to device 0
    assert prev == -1
    dead = api.CxxSignalPacker("xdelta_hzr", 4, 1, 64, 3)  # (like the reference, the factory itself does not throw)
    assert dead.compress(np.arange(64, dtype=np.int32).view(np.uint8)) == b""
    dead.close()
    L.rspt_cxx_set_device(ndev - 1)
    pk = api.CxxSignalPacker("xdelta_hzr", 4, 1, 64, 3)
    assert len(pk.compress(np.arange(64, dtype=np.int32).view(np.uint8))) > 0
    pk.close()
    L.rspt_cxx_set_device(-1)
    exe = tmp_path / "shard_devices"
    lib_dir = os.path.dirname(build.LIB)
    subprocess.check_call(["g++", "-std=c++11", "-pthread", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cxx", "shard_devices.cpp"),
                           "-o", str(exe), "-L" + lib_dir, "-lrspt_hip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], stdout=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode == 0, out.stdout
    assert "ok" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["c5", "c3", "c5-slots2"])
def test_bench_under_the_launcher_with_one_rank(api, workload):
    """The N > 1 branch of bench.py -- RANK set, init_process_group("nccl"), the side-stream exchange of sizes, the container
    pack + (lagged) gather of --workload c5 -- through RCCL with a world of one rank: the code the driver's 2/4/8-GPU runs take,
    on the one GPU a test box has.  Started as a child process (never an exec from this GPU-initialised process)."""
    import json
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", workload, "--steps", "3", "--warmup", "1", "--no-cpu"]
    slots = 2 if workload == "c5-slots2" else 1  # two steps in flight on two handles: the small-shard mode
    if slots > 1:
        workload = "c5"
        cmd[cmd.index("c5-slots2")] = "c5"
        cmd += ["--slots", "2", "--blocks", "128"]
    if workload == "c3":
        cmd += ["--blocks", "4"]
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln][-1]
    r = json.loads(line)
    assert r["n_gpus"] == 1 and r["verified"] is True and r["value"] > 0
    assert r["scaling"] == ("strong" if workload == "c5" else "weak")
    assert r["config"]["gather"]  # the exchange ran (a string describing it), not the --no-gather shortcut
    if workload == "c5":
        assert "no host sync" in r["config"]["gather"] and r["config"]["gathered_bytes"] > 0
        assert r["config"]["steps_in_flight"] == slots and (r["config"]["ms_per_step_one_in_flight"] is not None) == (slots > 1)


def _build_gather_example(tmp_path):
    from rspt_amd import build

    exe = tmp_path / "gather_devices"
    lib_dir = os.path.dirname(build.LIB)
    subprocess.check_call(["g++", "-std=c++11", "-pthread", "-w", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
                           os.path.join(ROOT, "tests", "cxx", "gather_devices.cpp"), "-o", str(exe), "-L" + lib_dir, "-lrspt_hip", "-L/opt/rocm/lib",
                           "-lrccl", "-lamdhip64", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cxx_gather_example_compiles(tmp_path):
    """tests/cxx/gather_devices.cpp: shards -> containers -> RCCL gather to rank 0 -> decode, through include/rspt_hip.h alone"""
    assert _build_gather_example(tmp_path).exists()


@pytest.mark.gpu
def test_cxx_gather_over_rccl(api, tmp_path):
    """the C ABI's gather (ncclAllGather of the sizes, grouped ncclSend / ncclRecv of the payload) with one rank per
    visible device, each rank's container decoded on rank 0"""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([str(_build_gather_example(tmp_path))], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ok" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("kind,bps,nch,ns,B", [("xdelta_hzr", 3, 64, 65536, 3), ("xdelta_hzr", 4, 64, 65536, 2), ("xdelta_hzr", 2, 64, 65536, 2),
                                                 ("xdelta_hzr", 3, 5, 20003, 3), ("hzr", 2, 7, 1001, 4), ("xdelta_hzr", 4, 3, 10, 5),
                                                 ("hadamard", 4, 64, 65536, 1), ("dct", 3, 3, 4096, 2)])
def test_big_endian_batches_equal_little_endian_ones(api, kind, bps, nch, ns, B):
    """Big-endian ingest is a flag of the front-end kernels (one v_perm_b32 per loaded sample: no byte-swap pass): full-size
    int24 / int32 / int16 batches (BASELINE shape; the int24 one ends with the sample whose dword is read one byte early), ragged
    and tiny shapes of the general kernel, and the transform packers' front ends produce, from the byte-reversed feed, exactly
    the streams of the little-endian feed -- which the other tests hold to the oracle."""
    import torch

    from rspt_amd import synth

    le = synth.synth_batch_native(B, nch, ns, first_block=300, bps=bps, ecg=True, device="cuda")
    be = le.view(B, -1, bps).flip(2).contiguous().view(B, -1)
    a, b = api.SignalPacker(kind, bps, nch, ns, 2), api.SignalPacker(kind, bps, nch, ns, 2)
    b.set_byte_order(big_endian=True)
    d1, s1 = a.compress_batch(le)
    d2, s2 = b.compress_batch(be)
    torch.cuda.synchronize()
    assert torch.equal(s1, s2) and int(s1.min()) > 0
    for i in range(B):
        n = int(s1[i])
        assert torch.equal(d1[i, :n], d2[i, :n]), i
    if kind in ("xdelta_hzr", "hzr"):  # and back: the big-endian handle hands big-endian samples back
        back, used = b.decompress_batch(d2, B, d2.shape[1])
        torch.cuda.synchronize()
        assert torch.equal(used, s2) and torch.equal(back.view(B, -1), be)
    a.close()
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("group,slots,pinned", [(1, 2, True), (4, 3, False), (3, 2, True)])
def test_feed_equals_a_loop_of_compress_calls(api, orc, group, slots, pinned):
    """rspt_hip_feed_*: blocks pushed one by one as they "arrive", finished streams polled whenever convenient -- never a blocking
    call until the final flush -- == the oracle's packer fed the same blocks one compress() call after the other (the amplitudes
    step up in the middle: nb escalates inside a group; one destination is too small and is reported, not written)"""
    nch, ns, bps, n = 8, 2048, 4, 23
    blocks = [cases._rand_native(nch, ns, bps, 5000 + i, 2000 if i < 9 else 1 << 27, walk=bool(i & 1)) for i in range(n)]
    po = orc.packer("xdelta_hzr", bps, nch, ns, 2)
    want = [po.compress(b) for b in blocks]
    pk = api.new_xdelta_hzr(bps, nch, ns, 2)
    cap = pk.max_compressed_size
    bufs = []
    if pinned:
        src = [api.HostBuffer(pk.block_bytes) for _ in range(n)]
        for s_, b in zip(src, blocks):
            s_.a[:] = b
        dst = [api.HostBuffer(cap) for _ in range(n)]
        bufs = src + dst
        src_a, dst_a = [s_.a for s_ in src], [d.a for d in dst]
    else:
        src_a, dst_a = [b.copy() for b in blocks], [np.zeros(cap, dtype=np.uint8) for _ in range(n)]
    small = 5
    dst_a[small] = dst_a[small][:16]  # too small for its stream
    got = {}

    def drain():
        while True:
            r = pk.feed_poll()
            if r is None:
                return
            got[r[0]] = r[1:]

    pk.feed_begin(group, slots)
    for i in range(n):
        while not pk.feed_push(src_a[i], dst_a[i]):  # ring full: take what is finished, then try again
            drain()
        if i % 5 == 4:
            drain()
        if i == 11:
            pk.feed_submit()  # a partly filled group goes out when nothing more is expected for a while
    pk.feed_flush()
    drain()
    assert sorted(got) == list(range(n))
    for i in range(n):
        ln, st = got[i]
        if i == small:
            assert st == -5 and ln == len(want[i])
        else:
            assert st == 0 and dst_a[i][:ln].tobytes() == want[i], i
    assert pk.feed_poll() is None
    pk.feed_end()
    # the handle is an ordinary packer again, its nb state where the feed left it
    assert pk.nb == orc.packer_nb(po) and pk.compress(blocks[0]) == po.compress(blocks[0])
    pk.close()
    for b in bufs:
        b.close()


@pytest.mark.gpu
def test_batch_entry_points_refuse_while_a_feed_is_open(api):
    """the feed owns the handle's plane workspace and copy streams: batch / many-block calls in between return RSPT_HIP_ERR_ARG
    (they used to run and race the feed's launches), and the handle is an ordinary packer again after rspt_hip_feed_end"""
    import torch

    from rspt_amd import synth

    nch, ns, B = 8, 2048, 3
    pk = api.new_xdelta_hzr(4, nch, ns, 2)
    d_src = synth.synth_batch_native(B, nch, ns, device="cuda")
    want_dst, want_sizes = pk.compress_batch(d_src)
    torch.cuda.synchronize()
    pk.feed_begin(2, 2)
    with pytest.raises(api.RsptHipError) as e:
        pk.compress_batch(d_src)
    assert e.value.status == -1
    with pytest.raises(api.RsptHipError):
        pk.decompress_batch(want_dst, B, want_dst.shape[1])
    host = d_src.cpu().numpy()
    out = np.zeros((B, pk.max_compressed_size), dtype=np.uint8)
    with pytest.raises(api.RsptHipError):
        pk.compress_many(host, out)
    pk.feed_end()
    got_dst, got_sizes = pk.compress_batch(d_src)
    torch.cuda.synchronize()
    assert torch.equal(got_sizes, want_sizes)
    for b in range(B):
        assert torch.equal(got_dst[b, : int(got_sizes[b])], want_dst[b, : int(want_sizes[b])])
    pk.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,bps,nch,ns", [("xdelta_hzr", 4, 64, 65536), ("xdelta_hzr", 3, 5, 20003), ("hzr", 2, 7, 1001),
                                             ("hadamard", 4, 12, 16384), ("dct", 4, 3, 4096), ("xdelta_hzr", 1, 3, 5000)])
def test_page_locked_buffers_are_used_in_place(api, orc, kind, bps, nch, ns):
    """With page-locked source / destination buffers rspt_hip_compress reads the samples across the link from the front end and
    the encoders write the stream into the caller's buffer; rspt_hip_decompress writes the samples there (no copy phases).  Same
    bytes as through the staged path -- for every packer, aligned and unaligned views (an unaligned source is staged), a
    destination that is too small (nothing written, the needed size reported), and escalation across calls."""
    from rspt_amd import synth

    x = synth.synth_native(nch, ns, block_index=21, bps=bps, ecg=True).numpy().reshape(-1)
    po = orc.packer(kind, bps, nch, ns, 2)
    pk = api.SignalPacker(kind, bps, nch, ns, 2)
    ref = api.SignalPacker(kind, bps, nch, ns, 2)  # the staged path: pageable buffers
    cap = pk.max_compressed_size
    src, dst, back = api.HostBuffer(x.size + 64), api.HostBuffer(cap + 64), api.HostBuffer(x.size + 64)
    for off in (0, 16, 5):  # (5: not 16-byte aligned -> staged; the destination may sit anywhere)
        s_view, d_view, b_view = src.a[off : off + x.size], dst.a[off : off + cap], back.a[off : off + x.size]
        s_view[:] = x
        d_view[:] = 0xEE
        n = pk.compress_into(s_view, d_view)
        want = ref.compress(x.copy())
        assert d_view[:n].tobytes() == want, (off, n, len(want))
        assert (d_view[n:] == 0xEE).all()  # nothing behind the stream was touched
        if kind in ("xdelta_hzr", "hzr"):
            assert want == po.compress(x)
        b_view[:] = 0
        used = pk.decompress_into(d_view, b_view)
        rb, ru = ref.decompress(want)
        assert used == n == ru and b_view.tobytes() == rb
    # too small (40 bytes: staged; 128 bytes: in place, the kernels are given exactly that much): reported, nothing written
    src.a[: x.size] = x
    for room in (40, 128):
        tiny = dst.a[:room]
        tiny[:] = 0x55
        with pytest.raises(api.RsptHipError) as e:
            pk.compress_into(src.a[: x.size], tiny)
        assert e.value.status == -5 and (tiny == 0x55).all(), room
    for b in (src, dst, back):
        b.close()
    pk.close()
    ref.close()


@pytest.mark.gpu
def test_batch_calls_can_be_captured_into_a_graph(api, orc):
    """The batch entry points only enqueue work (no host synchronisation, no allocation once rspt_hip_reserve has run), so a host
    may record a handle's steps into a HIP graph and replay them (small shards): two steps per graph, since a
    handle alternates between two plane-workspace sets.  The replayed steps give the bytes of the direct calls == the oracle's."""
    import torch

    from rspt_amd import synth

    nch, ns, B = 12, 8192, 24
    dev = torch.device("cuda", 0)
    pk = api.new_xdelta_hzr(4, nch, ns, 3)
    pk.reserve(B)
    stride = (pk.max_compressed_size + 255) // 256 * 256
    srcs = [synth.synth_batch_native(B, nch, ns, first_block=s * B, device=dev) for s in range(2)]
    dst = [torch.zeros((B, stride), dtype=torch.uint8, device=dev) for _ in range(2)]
    sz = [torch.zeros(B, dtype=torch.int64, device=dev) for _ in range(2)]
    cont = [torch.zeros(pk.pack_bound(B), dtype=torch.uint8, device=dev) for _ in range(2)]
    tot = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(2)]
    st = torch.cuda.ExternalStream(pk.stream_ptr, device=dev)
    want, want_cont = [], []
    with torch.cuda.stream(st):
        for s in range(2):
            pk.compress_batch(srcs[s], dst[s], sz[s], stride)
            pk.pack_batch(dst[s], sz[s], cont[s], tot[s])
    torch.cuda.synchronize()
    po = orc.packer("xdelta_hzr", 4, nch, ns, 3)
    for s in range(2):
        want.append([dst[s][b, : int(sz[s][b])].cpu().numpy().tobytes() for b in range(B)])
        want_cont.append(cont[s][: int(tot[s])].cpu().numpy().tobytes())
        assert want[s][0] == po.compress(srcs[s][0].cpu().numpy()) if s == 0 else True
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(dev)  # (the calls go to torch's current stream: the capturing one)
    with torch.cuda.graph(g, stream=side):
        for s in range(2):
            pk.compress_batch(srcs[s], dst[s], sz[s], stride)
            pk.pack_batch(dst[s], sz[s], cont[s], tot[s])
    for rep in range(3):
        for s in range(2):
            dst[s].zero_(), sz[s].zero_(), cont[s].zero_(), tot[s].zero_()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        for s in range(2):
            got = [dst[s][b, : int(sz[s][b])].cpu().numpy().tobytes() for b in range(B)]
            assert got == want[s], (rep, s)
            assert cont[s][: int(tot[s])].cpu().numpy().tobytes() == want_cont[s], (rep, s)
    # and the handle goes on with direct calls afterwards
    with torch.cuda.stream(st):
        pk.compress_batch(srcs[1], dst[1], sz[1], stride)
    torch.cuda.synchronize()
    assert [dst[1][b, : int(sz[1][b])].cpu().numpy().tobytes() for b in range(B)] == want[1]
    pk.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,nch,ns,B", [("xdelta_hzr", 12, 8192, 16), ("hadamard", 4, 65536, 3)])
def test_several_handles_in_flight_give_the_sequential_streams(api, kind, nch, ns, B):
    """Small batches in flight on several handles, each on its own stream and with no event between them (bench.py --slots,
    INTEGRATION.md): every step's streams are the ones a single handle produces for that batch, and they decode back in flight
    the same way."""
    import torch

    from rspt_amd import synth

    S, steps = 3, 9
    dev = torch.device("cuda", 0)
    pks = [api.SignalPacker(kind, 4, nch, ns, 3) for _ in range(S)]
    streams = [torch.cuda.ExternalStream(p.stream_ptr, device=dev) for p in pks]
    stride = (pks[0].max_compressed_size + 255) // 256 * 256
    srcs = [synth.synth_batch_native(B, nch, ns, first_block=n * B, device=dev) for n in range(steps)]
    dst = [torch.zeros((B, stride), dtype=torch.uint8, device=dev) for _ in range(steps)]
    sz = [torch.zeros(B, dtype=torch.int64, device=dev) for _ in range(steps)]
    back = [torch.zeros((B, pks[0].block_bytes), dtype=torch.uint8, device=dev) for _ in range(steps)]
    used = [torch.zeros(B, dtype=torch.int64, device=dev) for _ in range(steps)]
    for p in pks:
        p.reserve(B)
    for n in range(steps):
        with torch.cuda.stream(streams[n % S]):
            pks[n % S].compress_batch(srcs[n], dst[n], sz[n], stride)
    for n in range(steps):  # (the decode of step n follows its encode on the same stream)
        with torch.cuda.stream(streams[n % S]):
            pks[n % S].decompress_batch(dst[n], B, stride, back[n], used[n])
    torch.cuda.synchronize()
    one = api.SignalPacker(kind, 4, nch, ns, 3)
    for n in range(steps):
        d1, s1 = one.compress_batch(srcs[n], None, None, stride)
        torch.cuda.synchronize()
        assert torch.equal(s1, sz[n]), n
        for b in range(B):
            m = int(s1[b])
            assert torch.equal(d1[b, :m], dst[n][b, :m]), (n, b)
        assert torch.equal(used[n], sz[n]), n
        if kind == "xdelta_hzr":
            assert torch.equal(back[n], srcs[n]), n
        else:  # lossy: what the single handle decodes
            b1, _ = one.decompress_batch(d1, B, stride)
            torch.cuda.synchronize()
            assert torch.equal(back[n], b1), n
    for p in pks + [one]:
        p.close()
