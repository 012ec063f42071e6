"""Time-bounded randomised parity soak on the GPU: random packers, sample widths, geometries, amplitudes, byte orders and batch
sizes -- every produced stream against the CPU oracle (oracle/: the restatement pinned by the reference build), every decode against
the input (lossless packers) or the oracle's decode (lossy ones).  Not a test (it runs as long as it is told to); a mismatch prints
the seed of the case, which reproduces it.

    python tools/soak.py [seconds, default 300] [first seed, default 1] [host threads, default 1]

(This module lives in tests/ -- it is test infrastructure: it drives the CPU oracle -- and tools/soak.py is its launcher;
tests/test_gpu_soak.py runs a fixed range of seeds with the GPU suite.)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # (cases, streamtools, the fuzz generators)
import numpy as np
import torch

import cases
from streamtools import describe_mismatch, parse_stream
from test_gpu_fuzz import KINDS as BYTE_KINDS, _gen as gen_bytes
from oracle.oracle import Oracle
from rspt_amd import api

class Locked:
    """the CPU oracle behind one lock (threaded runs: the checker is not what is being tested for thread safety)"""

    def __init__(self, obj, lock):
        object.__setattr__(self, "_o", obj)
        object.__setattr__(self, "_k", lock)

    def __getattr__(self, name):
        v = getattr(self._o, name)
        if not callable(v):
            return v

        def call(*a, **kw):
            a = [x._o if isinstance(x, Locked) else x for x in a]
            with self._k:
                r = v(*a, **kw)
            return Locked(r, self._k) if type(r).__name__ == "Packer" else r

        return call


import threading

orc = Locked(Oracle(), threading.RLock())


def pick_shape(r, kind):
    if kind == "hadamard":
        ns = 1 << int(r.integers(4, 18))  # 16 .. 131072 (two-pass transform above 65536)
        nch = int(r.integers(1, max(2, min(40, (1 << 21) // ns))))
        return nch, ns
    if kind == "dct":
        ns = int(r.choice([16, 100, 257, 1000, 1024, 2048, 3000]))
        return int(r.integers(1, 9)), ns
    style = int(r.integers(0, 5))
    if r.integers(0, 40) == 0:  # now and then the BASELINE shapes at full size
        return (64, 65536) if r.integers(0, 2) else (12, 34199)
    if r.integers(0, int(os.environ.get("SOAK_WIDE_ONE_IN", "30"))) == 0:  # very wide / very long
        return (int(r.integers(300, 6000)), int(r.integers(1, 200))) if r.integers(0, 2) else (int(r.integers(1, 3)), int(r.integers(300000, 2200000)))
    if style == 0:  # many channels, short
        return int(r.integers(33, 200)), int(r.integers(1, 600))
    if style == 1:  # few channels, long, ragged
        return int(r.integers(1, 6)), int(r.integers(3000, 300000))
    if style == 2:  # the shapes of the BASELINE configs, cut down
        return int(r.choice([12, 64])), int(r.choice([8192, 4096, 16384, 65536 // 4]))
    if style == 3:  # tiny
        return int(r.integers(1, 5)), int(r.integers(1, 40))
    return int(r.integers(1, 70)), int(r.integers(16, 9000))


def iir_case(seed, r):
    """the pre-filter stage (rspt_test.cpp:116-136): both modes against the restatement, amplitudes up to full scale (the truncated
    double overflows int32 there: the reference's conversion, not the GPU's saturating one, decides what comes out)"""
    bps = int(r.choice([4, 4, 3, 2]))
    nch, ns = int(r.integers(1, 20)), int(r.choice([1, 5, 100, 2000, 2047, 2048, 2049, 5000, 40000]))
    n, d = (cases.IIR_BANDPASS, cases.IIR_BANDPASS, cases.IIR_LOWPASS, cases.IIR_HIGHPASS)[int(r.integers(0, 4))]
    init = int(r.choice([0, 3, 2000]))
    lim = 1 << (8 * bps - 1)
    amp = int(min(lim - 1, r.choice([100, 1 << 14, 1 << 21, 1 << 29, (1 << 31) - 1])))
    desc = "seed %d: iir int%d %dch x %d init %d amp %d" % (seed, 8 * bps, nch, ns, init, amp)
    data = cases._rand_native(nch, ns, bps, int(r.integers(1 << 30)), amp, walk=bool(r.integers(2)))
    pk = api.new_xdelta_hzr(bps, nch, ns, 3)
    bad = []
    for shared in (True, False):
        d_buf = torch.from_numpy(np.stack([data, data])).cuda()
        pk.iir_prefilter_batch(d_buf, n, d, init, per_channel=not shared)
        torch.cuda.synchronize()
        want = orc.iir_prefilter(data, bps, nch, ns, n, d, init, shared_state=shared)
        for b in range(2):
            got = d_buf[b].cpu().numpy().tobytes()
            if got != want:
                a_ = np.frombuffer(got, dtype=np.uint8).reshape(-1, bps)
                b_ = np.frombuffer(want, dtype=np.uint8).reshape(-1, bps)
                diff = np.nonzero((a_ != b_).any(axis=1))[0]
                bad.append("%s mode, block %d: %d of %d samples differ, first at %s" % ("shared" if shared else "per-channel", b, diff.size, a_.shape[0], diff[:4].tolist()))
    pk.close()
    return (2, bad), desc


def bytes_case(seed, r):
    """byte streams of every density through a 1-channel 8-bit hzr packer (plane 0 of its stream IS hzr_encode(data)): pieces of
    different kinds spliced together, zero stretches between them -- blocks with one populated segment, runs across block
    edges, Fill / PlainCopy / light / small blocks side by side; a batch of them, twice on the same handle"""
    pieces, total = [], int(r.integers(1, 400000))
    while sum(p.size for p in pieces) < total:
        k = str(r.choice(BYTE_KINDS + ["zeros", "zeros"]))
        ln = int(r.choice([1, 17, 300, 4096, 5000, 40000, 65536, 70000, 150000]))
        if ln < 1000 and k not in ("zeros", "dense", "noise", "const", "peaky"):
            k = "dense"  # (the other generators want room)
        pieces.append(np.zeros(ln, dtype=np.uint8) if k == "zeros" else gen_bytes(int(r.integers(1 << 30)), ln, k))
    base = np.concatenate(pieces)[:total]
    n = base.size
    desc = "seed %d: bytes n %d" % (seed, n)
    B = int(r.integers(1, 5))
    blocks = [base] + [np.roll(base, int(r.integers(1, n + 1))) for _ in range(B - 1)]
    pk = api.new_hzr(1, 1, n)
    po = orc.packer("hzr", 1, 1, n)
    want = [po.compress(b) for b in blocks]
    bad = []
    d_src = torch.from_numpy(np.stack(blocks)).cuda()
    for call in range(2):
        d_dst, d_sizes = pk.compress_batch(d_src)
        torch.cuda.synchronize()
        sizes = d_sizes.cpu().numpy()
        out = d_dst.cpu().numpy()
        d_out, d_used = pk.decompress_batch(d_dst, B, d_dst.shape[1])
        torch.cuda.synchronize()
        for i in range(B):
            got = out[i, : sizes[i]].tobytes()
            if got != want[i]:
                bad.append("call %d block %d of %d: stream differs: %s" % (call, i, B, describe_mismatch(got, want[i])[:600]))
            elif int(d_used[i]) != sizes[i] or d_out[i].cpu().numpy().tobytes() != blocks[i].tobytes():
                bad.append("call %d block %d of %d: decode differs (consumed %d of %d)" % (call, i, B, int(d_used[i]), sizes[i]))
    pk.close()
    po.close()
    return (2 * B, bad), desc


def feed_case(seed, r):
    """rspt_hip_feed_*: blocks pushed as they arrive, polled at random moments, a submit now and then == a loop of compress calls;
    then decompress_many of what came out (with the nb the sequence ended on: only streams written with it)"""
    kind = str(r.choice(["xdelta_hzr", "xdelta_hzr", "hzr"]))
    bps = int(r.choice([4, 3, 2, 1]))
    nch, ns = int(r.integers(1, 40)), int(r.integers(1, 6000))
    nb0 = int(r.integers(1, 5))
    n = int(r.integers(1, 40))
    group, slots = int(r.integers(1, 9)), int(r.integers(2, 5))
    desc = "seed %d: feed %s int%d %dch x %d nb %d, %d blocks, groups of %d, %d slots" % (seed, kind, 8 * bps, nch, ns, nb0, n, group, slots)
    lim = 1 << (8 * bps - 1)
    step = int(r.integers(0, n + 1))
    blocks = [cases._rand_native(nch, ns, bps, int(r.integers(1 << 30)), max(1, min(lim - 1, 2000 if i < step else 1 << 27)), walk=bool(i & 1)) for i in range(n)]
    po = orc.packer(kind, bps, nch, ns, nb0)
    want = [po.compress(b) for b in blocks]
    pk = api.SignalPacker(kind, bps, nch, ns, nb0)
    cap = pk.max_compressed_size
    dst = [np.zeros(cap, dtype=np.uint8) for _ in range(n)]
    got, bad = {}, []

    def drain():
        while True:
            q = pk.feed_poll()
            if q is None:
                return
            got[q[0]] = q[1:]

    pk.feed_begin(group, slots)
    for i in range(n):
        while not pk.feed_push(blocks[i], dst[i]):
            drain()
        if r.integers(0, 4) == 0:
            drain()
        if r.integers(0, 9) == 0:
            pk.feed_submit()
    pk.feed_flush()
    drain()
    pk.feed_end()
    if sorted(got) != list(range(n)):
        bad.append("polled %s of %d blocks" % (sorted(got), n))
    for i in range(n):
        if i in got and (got[i][1] != 0 or dst[i][: got[i][0]].tobytes() != want[i]):
            bad.append("block %d: status %d, length %d (want %d)%s" % (i, got[i][1], got[i][0], len(want[i]), "" if got[i][1] else ": stream differs"))
    if not bad and kind == "xdelta_hzr":
        nbf = pk.nb
        keep_i = [i for i in range(n) if len(parse_stream(want[i])["planes"]) == nbf]
        if keep_i:
            stride = (cap + 63) // 64 * 64
            st = np.zeros((len(keep_i), stride), dtype=np.uint8)
            for q, i in enumerate(keep_i):
                st[q, : len(want[i])] = np.frombuffer(want[i], dtype=np.uint8)
            back = np.zeros(len(keep_i) * pk.block_bytes, dtype=np.uint8)
            used = pk.decompress_many(st, back)
            for q, i in enumerate(keep_i):
                if used[q] != len(want[i]) or back[q * pk.block_bytes : (q + 1) * pk.block_bytes].tobytes() != blocks[i].tobytes():
                    bad.append("decompress_many: block %d differs (consumed %d of %d)" % (i, used[q], len(want[i])))
    pk.close()
    po.close()
    return (n, bad), desc


def foreign_case(seed, r):
    """Valid streams that neither encoder would write: some hzr blocks of the oracle's stream re-encoded as PlainCopy (raw bytes,
    mode 0) -- a decoder has to take any valid stream, whatever mode decisions its writer made (hzr_decode.c:335-567).  int8 samples
    through the hzr packer: plane 0 is the data, planes 1-3 its sign extension.  The oracle's decoder checks the crafted stream first."""
    import struct

    total = int(r.integers(1, 400000))
    pieces = []
    while sum(q.size for q in pieces) < total:
        k = str(r.choice(BYTE_KINDS + ["zeros"]))
        ln = int(r.choice([300, 4096, 5000, 40000, 65536, 70000, 150000]))
        if ln < 1000 and k not in ("zeros", "dense", "noise", "const", "peaky"):
            k = "dense"
        pieces.append(np.zeros(ln, dtype=np.uint8) if k == "zeros" else gen_bytes(int(r.integers(1 << 30)), ln, k))
    data = np.concatenate(pieces)[:total]
    n = data.size
    desc = "seed %d: foreign stream, bytes n %d" % (seed, n)
    po = orc.packer("hzr", 1, 1, n)
    s0 = po.compress(data)
    neg = (data.view(np.int8) < 0)
    raw = [data, np.where(neg, 0xFF, 0).astype(np.uint8)]  # plane 0, planes 1..3
    ps = parse_stream(s0)
    out = bytearray(s0[:1])
    changed = 0
    for k, pl in enumerate(ps["planes"]):
        body = bytearray(struct.pack("<I", pl["in_size"]))
        for j, (mode, plen, crc, off) in enumerate(pl["blocks"]):
            lo = j * 65536
            rb_ = raw[min(k, 1)][lo : lo + 65536].tobytes()
            if mode != 0 and r.integers(0, 3) == 0:
                body += struct.pack("<HIB", len(rb_) - 1, orc.crc32c(rb_), 0) + rb_
                changed += 1
            else:
                body += s0[off : off + 7 + plen]
        out += struct.pack("<I", len(body)) + body
    crafted = bytes(out)
    bad = []
    ref, used_ref, rc = po.decompress(crafted)
    if ref != data.tobytes() or used_ref != len(crafted):
        return None, desc + " (the crafted stream does not pass the oracle's decoder: rc %d, consumed %d of %d)" % (rc, used_ref, len(crafted))
    pk = api.new_hzr(1, 1, n)
    try:
        dec, used = pk.decompress(crafted)
        if used != len(crafted) or dec != data.tobytes():
            bad.append("%d of the blocks re-encoded as PlainCopy: decode differs (consumed %d of %d)" % (changed, used, len(crafted)))
    except api.RsptHipError as e:
        bad.append("%d of the blocks re-encoded as PlainCopy: %s" % (changed, e))
    # ... and in a batch beside the stream as written
    stride = (max(len(crafted), len(s0)) + 255) // 256 * 256
    st = np.zeros((2, stride), dtype=np.uint8)
    st[0, : len(crafted)] = np.frombuffer(crafted, dtype=np.uint8)
    st[1, : len(s0)] = np.frombuffer(s0, dtype=np.uint8)
    d_out, d_used = pk.decompress_batch(torch.from_numpy(st).cuda(), 2, stride)
    torch.cuda.synchronize()
    if d_used.cpu().tolist() != [len(crafted), len(s0)] or d_out[0].cpu().numpy().tobytes() != data.tobytes() or d_out[1].cpu().numpy().tobytes() != data.tobytes():
        bad.append("batch decode of the crafted stream beside the original differs (consumed %s)" % d_used.cpu().tolist())
    pk.close()
    po.close()
    return (2, bad), desc


def damaged_case(seed, r):
    """A valid stream with a few bits flipped, bytes overwritten or its tail cut off, with and without block verification: the
    call has to return -- an error, or bytes -- and the handle has to decode the pristine stream afterwards.  (What a damaged stream
    decodes to without verification is nobody's business; with it, a stream that decodes at all decodes to the original.)"""
    total = int(r.integers(200, 300000))
    pieces = []
    while sum(q.size for q in pieces) < total:
        k = str(r.choice(BYTE_KINDS + ["zeros"]))
        ln = int(r.choice([300, 4096, 5000, 40000, 65536, 70000, 150000]))
        if ln < 1000 and k not in ("zeros", "dense", "noise", "const", "peaky"):
            k = "dense"
        pieces.append(np.zeros(ln, dtype=np.uint8) if k == "zeros" else gen_bytes(int(r.integers(1 << 30)), ln, k))
    data = np.concatenate(pieces)[:total]
    n = data.size
    desc = "seed %d: damaged stream, bytes n %d" % (seed, n)
    po = orc.packer("hzr", 1, 1, n)
    s0 = po.compress(data)
    po.close()
    pk = api.new_hzr(1, 1, n)
    bad = []
    ntry = 0
    mode_bytes = [blk[3] + 6 for pl_ in parse_stream(s0)["planes"] for blk in pl_["blocks"]]
    for verify in (True, False):
        pk.set_verify(verify)
        for _ in range(6):
            d_ = bytearray(s0)
            how = int(r.integers(0, 4))
            if os.environ.get("SOAK_DAMAGE"):
                how = int(os.environ["SOAK_DAMAGE"])
            if how == 0:
                for _k in range(int(r.integers(1, 4))):
                    q = int(r.integers(0, 8 * len(d_)))
                    d_[q >> 3] ^= 1 << (q & 7)
            elif how == 1:
                q = int(r.integers(0, len(d_)))
                ln = int(r.integers(1, 9))
                d_[q : q + ln] = bytes(int(x) for x in r.integers(0, 256, len(d_[q : q + ln])))
            elif how == 2:
                d_ = d_[: int(r.integers(1, len(d_)))]
            else:  # a block header's length / mode fields
                ps = parse_stream(s0)
                pl = ps["planes"][int(r.integers(0, len(ps["planes"])))]
                off = pl["blocks"][int(r.integers(0, len(pl["blocks"])))][3]
                d_[off + int(r.choice([0, 1, 6]))] = int(r.integers(0, 256))
            ntry += 1
            try:
                dec, used = pk.decompress(bytes(d_), bounded=True)  # (a damaged length field or a cut: nothing behind the buffer may be read)
                # (the mode byte of a block header is not covered by the block's CRC: a Fill block turned "Huffman" passes the check in
                #  the reference too, and what the payload then decodes to is not an error anybody can see)
                touched_mode = len(d_) == len(s0) and any(d_[q_] != s0[q_] for q_ in mode_bytes)
                if verify and dec != data.tobytes() and bytes(d_) != s0 and not touched_mode:
                    bad.append("verification on: a damaged stream (%s) decoded to other bytes without an error" % ("flips", "bytes", "cut", "header")[how])
            except api.RsptHipError as e:
                if e.status != -6:
                    bad.append("status %d for a damaged stream" % e.status)
    pk.set_verify(True)
    dec, used = pk.decompress(s0)
    if used != len(s0) or dec != data.tobytes():
        bad.append("the handle no longer decodes the pristine stream")
    pk.close()
    return (ntry, bad), desc


THREADED = [False]


def fft_case(seed, r):
    """the dct packer's fp64 FFT kernels (every ns = 2^k above 8192; forced here at smaller sizes too, where the reference's own
    arithmetic is the checker): coefficients equal except truncation-boundary flips of one count, sizes within 1 %"""
    from test_gpu_dct_fft import _coeff_mismatch

    lg = int(r.integers(4, 16))
    ns, nch, bps = 1 << lg, int(r.integers(1, 7)), int(r.choice([4, 4, 3, 2]))
    if ns * nch > (1 << 17):
        nch = max(1, (1 << 17) // ns)
    amp = int(min((1 << (8 * bps - 1)) - 1, r.choice([60, 1 << 10, 1 << 14])))
    desc = "seed %d: dct (FFT kernels) int%d %dch x %d amp %d" % (seed, 8 * bps, nch, ns, amp)
    # (noise, not a walk: a walk of this length leaves the packer's range -- coefficients past their two planes -- and what the
    #  inverse makes of that is the out-of-range conversion of whichever fp64 transform runs, not a property worth a tolerance)
    data = cases._rand_native(nch, ns, bps, int(r.integers(1 << 30)), amp, walk=False)
    want = orc.packer("dct", bps, nch, ns).compress(data) if ns <= 8192 else orc.dct_big_compress(data, bps, nch, ns)[0]
    pk = api.SignalPacker(api.KIND_DCT | api.DCT_FORCE_FFT, bps, nch, ns, 2)
    got = pk.compress(data)
    bad = []
    frac, dmax = _coeff_mismatch(orc._o, got, want, bps, nch, ns)
    if dmax > 1 or frac > 4e-3 + 2.0 / (nch * ns):
        bad.append("coefficients: %.5f of them differ, by up to %d" % (frac, dmax))
    if abs(len(got) / len(want) - 1) > 0.01 + 40.0 / len(want):
        bad.append("stream %d bytes, the checker's %d" % (len(got), len(want)))
    dec, used = pk.decompress(got)
    if used != len(got):
        bad.append("decode consumed %d of %d" % (used, len(got)))
    else:
        ref = orc.packer("dct", bps, nch, ns).decompress(got)[0] if ns <= 8192 else orc.dct_big_decompress(got, bps, nch, ns)[0]
        a_ = orc.native_to_i32(np.frombuffer(dec, dtype=np.uint8), ns, nch, bps).astype(np.int64)
        b_ = orc.native_to_i32(np.frombuffer(ref, dtype=np.uint8), ns, nch, bps).astype(np.int64)
        if np.abs(a_ - b_).max() > 2:  # (float products summed in double against an fp64 FFT: a count or two at these magnitudes)
            bad.append("decoded samples differ from the checker's decode of the same stream by up to %d" % int(np.abs(a_ - b_).max()))
    pk.close()
    return (1, bad), desc


def one_case(seed, keep=None):
    r = np.random.default_rng(seed)
    kind = str(r.choice(["xdelta_hzr", "xdelta_hzr", "xdelta_hzr", "hzr", "hadamard", "dct", "iir"]))
    rb = np.random.default_rng(seed ^ 0x5EED0000)  # (a generator of its own: the seeds of the other kinds mean what they meant before this kind existed)
    if keep is None and rb.integers(0, 4) == 0:
        return bytes_case(seed, rb)
    if keep is None and rb.integers(0, 12) == 0:
        return feed_case(seed, rb)
    if keep is None and rb.integers(0, 12) == 0 and not THREADED[0]:  # (its checker reaches into the oracle past the lock)
        return fft_case(seed, rb)
    if keep is None and rb.integers(0, 10) == 0:
        return foreign_case(seed, rb)
    if keep is None and rb.integers(0, 12) == 0:
        return damaged_case(seed, rb)
    if kind == "iir":
        return iir_case(seed, r)
    bps = int(r.choice([4, 4, 3, 2, 1]))
    nch, ns = pick_shape(r, kind)
    nb0 = int(r.integers(1, 5)) if kind in ("xdelta_hzr", "hzr") else 3
    if kind == "hzr":
        nb0 = int(r.integers(1, bps + 1)) if bps < 4 else nb0
    be = bool(r.integers(0, 4) == 0) and bps > 1
    calls = int(r.integers(1, 4))
    desc = "seed %d: %s int%d %dch x %d nb %d%s" % (seed, kind, 8 * bps, nch, ns, nb0, " big-endian" if be else "")
    try:
        po = orc.packer(kind, bps, nch, ns, nb0)
    except ValueError:
        return None, desc + " (refused by the oracle)"
    try:
        pk = api.SignalPacker(kind, bps, nch, ns, nb0)
    except api.RsptHipError as e:
        po.close()
        return None, desc + " (refused by the library: %s)" % e
    if be:
        pk.set_byte_order(big_endian=True)
    bad = []
    nblocks = 0
    lim = 1 << (8 * bps - 1)
    for call in range(calls):
        B = int(r.integers(1, 5)) if nch * ns < (1 << 20) else 1
        if nch * ns <= 20000 and r.integers(0, 8) == 0:
            B = int(r.integers(5, 300))  # many small blocks in one launch: the work queues, k_layout, the container index
        amps = [int(min(lim - 1, r.choice([1, 3, 60, 1 << 7, 1 << 10, 1 << 14, 1 << 21, 1 << 29]))) for _ in range(B)]
        if B > 4:  # (mostly small amplitudes, a step somewhere: one escalation inside the batch, not one per block)
            step = int(r.integers(0, B + 1))
            amps = [min(a_, 60) if i < step else a_ for i, a_ in enumerate(amps)]
        blocks = [cases._rand_native(nch, ns, bps, int(r.integers(1 << 30)), max(1, a), walk=bool(r.integers(2))) for a in amps]
        if r.integers(0, 6) == 0:
            blocks[0] = np.zeros_like(blocks[0])  # an all-zero block now and then
        feed = [np.ascontiguousarray(b.reshape(-1, bps)[:, ::-1]).reshape(-1) if be else b for b in blocks]
        want, ref = [], []
        for b in blocks:  # (the oracle's object decodes with the nb it has reached, too: each stream right behind its compress call)
            want.append(po.compress(b))
            ref.append(po.decompress(want[-1])[0])
        if keep is not None:
            keep.append(dict(kind=kind, bps=bps, nch=nch, ns=ns, nb0=nb0, be=be, feed=feed, want=want))
        if r.integers(0, 2) == 0 or B == 1:  # the host-pointer entry point, block by block
            got, dec = [], []
            pinned = r.integers(0, 3) == 0  # page-locked buffers: read and written in place across the link, at odd offsets now and then
            if pinned:
                off = int(r.choice([0, 0, 16, 4, 1]))
                hsrc, hdst, hback = api.HostBuffer(pk.block_bytes + 64), api.HostBuffer(2 * pk.block_bytes + 8192), api.HostBuffer(pk.block_bytes + 64)
            for f in feed:  # (decoded at once: a stream carries no nb, the handle decodes with the nb it has reached -- like the reference's object)
                if pinned:
                    hsrc.a[off : off + f.size] = f
                    n_ = pk.compress_into(hsrc.a[off : off + f.size], hdst.a[off:])
                    got.append(hdst.a[off : off + n_].tobytes())
                    used_ = pk.decompress_into(hdst.a[off : off + n_], hback.a[off : off + f.size])
                    dec.append(hback.a[off : off + f.size].tobytes() if used_ == n_ else b"")
                else:
                    got.append(pk.compress(f))
                    dec.append(pk.decompress(got[-1])[0])
            if pinned:
                for hb_ in (hsrc, hdst, hback):
                    hb_.close()
        else:  # one device-resident batch
            d_src = torch.from_numpy(np.stack(feed)).cuda()
            d_dst, d_sizes = pk.compress_batch(d_src)
            torch.cuda.synchronize()
            sizes = d_sizes.cpu().numpy()
            out = d_dst.cpu().numpy()
            got = [out[i, : sizes[i]].tobytes() for i in range(B)]
            stride = d_dst.shape[1]
            d_out, d_used = pk.decompress_batch(d_dst, B, stride)
            torch.cuda.synchronize()
            dec = [d_out[i].cpu().numpy().tobytes() for i in range(B)]
            used = d_used.cpu().numpy()
            # the batch as one container and back: the index carries every stream's own nb, so ALL streams decode (on a fresh handle)
            d_packed, d_total = pk.pack_batch(d_dst, d_sizes)
            torch.cuda.synchronize()
            pk2 = api.SignalPacker(kind, bps, nch, ns, nb0)
            if be:
                pk2.set_byte_order(big_endian=True)
            try:
                p_out, p_used = pk2.decompress_packed(d_packed, nbytes=int(d_total))
                torch.cuda.synchronize()
                for i in range(B):
                    if int(p_used[i]) != sizes[i]:
                        bad.append("call %d block %d: container decode consumed %d of %d" % (call, i, int(p_used[i]), sizes[i]))
                    elif kind == "xdelta_hzr" and p_out[i].cpu().numpy().tobytes() != feed[i].tobytes():
                        bad.append("call %d block %d: container round trip differs" % (call, i))
            except api.RsptHipError as e:
                bad.append("call %d: container decode failed: %s" % (call, e))
            # ... and the same blocks through the host pipeline (upload | compress | download) on that fresh handle: a loop of compress calls
            if kind in ("xdelta_hzr", "hzr") and r.integers(0, 2) == 0:
                pk3 = api.SignalPacker(kind, bps, nch, ns, nb0)
                po3 = orc.packer(kind, bps, nch, ns, nb0)
                if be:
                    pk3.set_byte_order(big_endian=True)
                hs = np.concatenate(feed)
                ho = np.zeros((B, (pk3.max_compressed_size + 63) // 64 * 64), dtype=np.uint8)
                lens = pk3.compress_many(hs, ho)
                for i in range(B):
                    if ho[i, : lens[i]].tobytes() != po3.compress(blocks[i]):
                        bad.append("call %d block %d: compress_many stream differs" % (call, i))
                pk3.close()
                po3.close()
            pk2.close()
            # ... and into a destination too short for some of the streams: those report the size they need (bit 63), the others arrive,
            # nothing lands behind the destination
            if r.integers(0, 4) == 0:
                pk4, po4 = api.SignalPacker(kind, bps, nch, ns, nb0), orc.packer(kind, bps, nch, ns, nb0)
                if be:
                    pk4.set_byte_order(big_endian=True)
                want4 = [po4.compress(b_) for b_ in blocks]
                short = max(64, int(r.integers(16, max(len(w_) for w_ in want4) + 64)) // 16 * 16)
                flat = torch.full((B * short + 4096,), 0xA5, dtype=torch.uint8, device="cuda")
                sz4 = torch.zeros(B, dtype=torch.int64, device="cuda")
                pk4.compress_batch(d_src, flat[: B * short].view(B, short), sz4, short)
                torch.cuda.synchronize()
                h4, s4 = flat.cpu().numpy(), sz4.cpu().numpy()
                if (h4[B * short :] != 0xA5).any():
                    bad.append("call %d: short destination (stride %d): bytes written behind it" % (call, short))
                for i in range(B):
                    need = int(s4[i]) & ((1 << 63) - 1)
                    if len(want4[i]) <= short:
                        if int(s4[i]) != len(want4[i]) or h4[i * short : i * short + need].tobytes() != want4[i]:
                            bad.append("call %d block %d: short destination (stride %d): fitting stream differs (size word %d, want %d)" % (call, i, short, int(s4[i]), len(want4[i])))
                    elif int(s4[i]) >= 0 or need != len(want4[i]):
                        bad.append("call %d block %d: short destination (stride %d): size word %d for a stream of %d bytes" % (call, i, short, int(s4[i]), len(want4[i])))
                pk4.close()
                po4.close()
            for i in range(B):
                # the batch is decoded with the nb the handle ended on: streams written before an escalation inside this batch have
                # fewer planes and cannot be decoded by this handle any more (nor by the reference's object)
                if kind == "xdelta_hzr" and len(parse_stream(got[i])["planes"]) != pk.nb:
                    dec[i] = None
                elif used[i] != sizes[i]:
                    bad.append("call %d block %d: decode consumed %d of %d" % (call, i, used[i], sizes[i]))
        for i in range(B):
            nblocks += 1
            if got[i] != want[i]:
                hl = 3 * nch if kind in ("dct", "hadamard") else 0
                bad.append("call %d block %d of %d (amp %d, %s): stream differs (%d vs %d bytes): %s"
                           % (call, i, B, amps[i], "batch" if len(blocks) > 1 and dec[i] is not None and B > 1 else "-", len(got[i]), len(want[i]), describe_mismatch(got[i], want[i], hl)))
                continue
            if dec[i] is None:
                continue
            ref_dec = ref[i]
            ref_feed = np.ascontiguousarray(np.frombuffer(ref_dec, dtype=np.uint8).reshape(-1, bps)[:, ::-1]).reshape(-1).tobytes() if be else ref_dec
            if dec[i] != ref_feed:
                a_ = np.frombuffer(dec[i], dtype=np.uint8).reshape(-1, bps)
                b_ = np.frombuffer(ref_feed, dtype=np.uint8).reshape(-1, bps)
                diff = np.nonzero((a_ != b_).any(axis=1))[0]

                def val(rows):
                    le = rows[:, ::-1] if be else rows
                    v = np.zeros(rows.shape[0], dtype=np.int64)
                    for q in range(bps):
                        v |= le[:, q].astype(np.int64) << (8 * q)
                    return (v ^ (1 << (8 * bps - 1))) - (1 << (8 * bps - 1))

                src_ = val(feed[i].reshape(-1, bps)[diff[:4]])
                bad.append("call %d block %d (amp %d): decode differs from the oracle's in %d of %d samples; first at %s: ours %s, oracle %s, input %s"
                           % (call, i, amps[i], diff.size, a_.shape[0], diff[:4].tolist(), val(a_[diff[:4]]).tolist(), val(b_[diff[:4]]).tolist(), src_.tolist()))
            if kind == "xdelta_hzr" and dec[i] != feed[i].tobytes():
                bad.append("call %d block %d: lossless round trip differs" % (call, i))
        if kind == "xdelta_hzr" and pk.nb != orc.packer_nb(po):
            bad.append("call %d: nb %d, oracle %d" % (call, pk.nb, orc.packer_nb(po)))
    pk.close()
    po.close()
    return (nblocks, bad), desc


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    nthreads = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    if nthreads > 1:
        THREADED[0] = True
        # several host threads, each with its own handles and cases (the library's promise: one packer per thread at a time --
        # creation, compression and destruction of different packers may overlap freely)
        tot = [0, 0, 0]
        lock = threading.Lock()

        def worker(w):
            t0, seed = time.time(), seed0 + 1000000 * w
            while time.time() - t0 < budget:
                res, desc = one_case(seed)
                with lock:
                    if res is not None:
                        tot[0] += 1
                        tot[1] += res[0]
                        if res[1]:
                            tot[2] += 1
                            print("MISMATCH (thread %d) %s" % (w, desc))
                            for b in res[1][:6]:
                                print("    " + b)
                            sys.stdout.flush()
                seed += 1

        ths = [threading.Thread(target=worker, args=(w,)) for w in range(nthreads)]
        for t in ths:
            t.start()
        while any(t.is_alive() for t in ths):
            time.sleep(45)
            print("... %d cases (%d blocks), %d bad" % tuple(tot), flush=True)
        for t in ths:
            t.join()
        print("soak: %d threads from seed %d (+1000000 per thread), %d cases, %d blocks compared with the oracle, %d cases with a mismatch, %.0f s"
              % (nthreads, seed0, tot[0], tot[1], tot[2], budget))
        sys.exit(1 if tot[2] else 0)
    t0 = time.time()
    seed = seed0
    ncases = nblocks = nbad = nrefused = 0
    last = t0
    while time.time() - t0 < budget:
        res, desc = one_case(seed)
        if res is None:
            nrefused += 1
            if nrefused <= 8:
                print("refused: " + desc, flush=True)
        else:
            ncases += 1
            nblocks += res[0]
            if res[1]:
                nbad += 1
                print("MISMATCH " + desc)
                for b in res[1][:6]:
                    print("    " + b)
                sys.stdout.flush()
        if time.time() - last > 45:
            print("... %d cases (%d blocks), %d refused, %d bad, seed %d, %.0f s" % (ncases, nblocks, nrefused, nbad, seed, time.time() - t0), flush=True)
            last = time.time()
        seed += 1
    print("soak: seeds %d..%d, %d cases, %d blocks compared with the oracle, %d shapes refused, %d cases with a mismatch, %.0f s"
          % (seed0, seed - 1, ncases, nblocks, nrefused, nbad, time.time() - t0))
    sys.exit(1 if nbad else 0)


if __name__ == "__main__":
    main()
