"""Synthetic multi-channel sample blocks and the reference's shipped test data.

Integer-only generators, so every box produces identical bytes (hashes are
reproducible): a parabolic "sine" per channel plus 4-bit hash noise, after
SURVEY.md section 8(d).  One deliberate deviation from that sketch: the noise
is a counter-based hash of (block, sample, channel) instead of a sequential
xorshift32 stream, so that a block can be generated in parallel directly in HBM
(torch ops on the GPU) -- the signal statistics are the same.

The libm sine of the reference's own demos (README.md:55-58,
lib_rspt_test/rspt_test.cpp:180-223) is `sine_readme`.
"""
import lzma
import math
import os
import struct

import numpy as np
import torch

_GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _mix32(x):
    """lowbias32-style integer hash on int64 tensors holding uint32 values."""
    m = 0xFFFFFFFF
    x = x & m
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & m
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & m
    x = x ^ (x >> 16)
    return x


def synth_i32(nch, ns, block_index=0, ecg=False, device="cpu"):
    """[ns][nch] int32 tensor (interleaved, sample-major) -- SURVEY 8(d).

    per channel c: ph=(u32)(s*step_c), step_c=0xFFFFFFFF//(628+6c) (~sin(s/(100+c)));
    u=(i32)ph>>16; y=(u*(32768-|u|))>>15 (+-8192); x=((y*A_c)>>13)+noise-8,
    A_c=1000*(1+c%16)  =>  |x| <= 16008, which keeps a 65536-point WHT in int32.
    ecg=True adds a 40-sample triangular spike of height 8*A_c every 1000 samples.
    """
    dev = torch.device(device)
    s = torch.arange(ns, dtype=torch.int64, device=dev).view(ns, 1)
    c = torch.arange(nch, dtype=torch.int64, device=dev).view(1, nch)
    step = torch.div(torch.full_like(c, 0xFFFFFFFF), 628 + 6 * c, rounding_mode="floor")
    ph = (s * step) & 0xFFFFFFFF
    ph = torch.where(ph >= 0x80000000, ph - 0x100000000, ph)  # as int32
    u = ph >> 16
    y = (u * (32768 - u.abs())) >> 15
    amp = 1000 * (1 + c % 16)
    x = (y * amp) >> 13
    idx = (s * nch + c + (block_index * 0x9E3779B9 & 0xFFFFFFFF)) & 0xFFFFFFFF
    x = x + (_mix32(idx) & 15) - 8
    if ecg:
        t = s % 1000
        tri = torch.clamp(20 - (t - 20).abs(), min=0)  # 0..20..0 over 40 samples
        x = x + (tri * 8 * amp) // 20
    return x.to(torch.int32).contiguous()


def to_native(x_i32, bps=4):
    """[ns][nch] int32 -> interleaved little-endian bytes, bps bytes per sample
    (the layout convert_native_to_i32 reads, lib_signalpacker/utils.cpp:123-191)."""
    b = x_i32.contiguous().view(torch.uint8).view(*x_i32.shape, 4)
    return b[..., :bps].contiguous().view(-1)


def synth_native(nch, ns, block_index=0, bps=4, ecg=False, device="cpu"):
    """Interleaved LE bytes; for bps < 4 the low bps bytes of each int32 are kept."""
    return to_native(synth_i32(nch, ns, block_index, ecg, device), bps)


def synth_batch_native(nblocks, nch, ns, first_block=0, bps=4, ecg=False, device="cpu"):
    """nblocks independent blocks back to back: uint8 [nblocks, ns*nch*bps]."""
    out = torch.empty((nblocks, ns * nch * bps), dtype=torch.uint8, device=device)
    for b in range(nblocks):
        out[b] = synth_native(nch, ns, first_block + b, bps, ecg, device)
    return out


def sine_readme(ns, amplitude=1000.0, dtype=np.int32):
    """data_stream[i] = sin(i / 100.0) * amplitude  (README.md:57-58)."""
    return np.array([int(math.sin(i / 100.0) * amplitude) for i in range(ns)], dtype=dtype)


def _un7z(path):
    """Both shipped archives are single-folder LZMA2 with plain headers
    (SURVEY.md 8c): the packed stream starts at byte 32."""
    raw = open(path, "rb").read()
    off = struct.unpack("<Q", raw[12:20])[0]
    return lzma.decompress(
        raw[32 : 32 + off], format=lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_LZMA2, "dict_size": 2 << 20}]
    )


def ecg_12ch_i32():
    """lib_rspt_test/12_chan_32bit_34199_samples_r00000135fghd8.raw.7z:
    12 ch x 34199 samples x int32, interleaved (1,641,552 bytes)."""
    return _un7z(os.path.join(_GOLDEN, "ecg_12ch_i32_34199.raw.7z"))


def data_stream_3ch_i24():
    """lib_rspt_test/data_stream.7z: 3 ch x 20000 samples x 24 bit (180,000 bytes)."""
    return _un7z(os.path.join(_GOLDEN, "data_stream_3ch_i24_20000.7z"))
