"""Python host binding of the C ABI (include/rspt_hip.h) in librspt_hip.so.

Mirrors the reference's operator interface for this path -- the i_signal_packer
factories and compress()/decompress() of lib_rspt/signal_packer.h:29-73 -- so
that tests read like the reference's own harness (lib_rspt_test/rspt_test.cpp:58-112).
torch is used only for device memory and streams in the batched, device-resident
calls; nothing here computes.  There is NO CPU fallback: if the HIP library or a
gfx950 device is missing every constructor raises.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

KIND_HZR, KIND_XDELTA_HZR, KIND_DCT, KIND_HADAMARD = 0, 1, 2, 3
KINDS = {"hzr": 0, "xdelta_hzr": 1, "dct": 2, "hadamard": 3}
DCT_FORCE_FFT = 0x100  # RSPT_HIP_DCT_FORCE_FFT (test hook, include/rspt_hip.h)

# every symbol include/rspt_hip.h declares (tests check the library exports them all)
C_ABI_SYMBOLS = [
    "rspt_hip_status_string", "rspt_hip_last_hip_error", "rspt_hip_device_count", "rspt_hip_packer_create",
    "rspt_hip_packer_destroy", "rspt_hip_compress", "rspt_hip_decompress", "rspt_hip_decompress_bounded", "rspt_hip_max_compressed_size",
    "rspt_hip_block_bytes", "rspt_hip_current_nb", "rspt_hip_set_nb", "rspt_hip_set_verify", "rspt_hip_reserve", "rspt_hip_compress_batch_dev",
    "rspt_hip_decompress_batch_dev", "rspt_hip_decompress_packed_dev", "rspt_hip_pack_bound", "rspt_hip_pack_batch_dev", "rspt_hip_stream", "rspt_hip_synchronize", "rspt_hip_set_profiling", "rspt_hip_stage_count",
    "rspt_hip_stage_name", "rspt_hip_stage_times", "rspt_hip_debug_read", "rspt_hip_iir_prefilter_batch_dev", "rspt_hip_set_byte_order", "rspt_hip_host_alloc", "rspt_hip_host_free",
    "rspt_hip_compress_many", "rspt_hip_decompress_many", "rspt_hip_gather_sizes", "rspt_hip_gather_payload", "rspt_hip_gather_containers",
    "rspt_hip_gather_post_sizes", "rspt_hip_gather_post_payload", "rspt_hip_gather_wait",
    "rspt_hip_feed_begin", "rspt_hip_feed_push", "rspt_hip_feed_submit", "rspt_hip_feed_poll", "rspt_hip_feed_flush", "rspt_hip_feed_end",
]

_u8p = C.POINTER(C.c_uint8)
_szp = C.POINTER(C.c_size_t)
_lib = None


class RsptHipError(RuntimeError):
    def __init__(self, where, status, hip_error=0):
        self.status, self.hip_error = status, hip_error
        msg = lib().rspt_hip_status_string(status).decode() if _lib is not None else str(status)
        super().__init__("%s: %s (status %d, hipError %d)" % (where, msg, status, hip_error))


def lib():
    """Load librspt_hip.so.  A missing or stale library is rebuilt only where that is safe -- a single process with
    hipcc at hand; under a launcher (RANK set: the ranks of a torchrun job) a stale library is an error, never a
    concurrent compile.  Raises if the library cannot be had: there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    if os.environ.get("RSPT_HIP_LIB"):  # A/B timing of two builds in one session (tools/ab.sh); not a fallback
        path = os.environ["RSPT_HIP_LIB"]
    elif _build.stale():
        if "RANK" in os.environ:
            raise RuntimeError("rspt_amd: %s is missing or older than its sources; run `python __graft_entry__.py` (build()) "
                               "before launching ranks" % path)
        if os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
            path = _build.build()
        elif os.path.exists(path):  # built from other sources than the ones beside it, and no compiler to fix that: never run it silently
            raise RuntimeError("rspt_amd: %s does not match its sources (fingerprint %s) and there is no hipcc here to rebuild it"
                               % (path, _build.STAMP))
    if not os.path.exists(path):
        raise RuntimeError("rspt_amd: %s is missing and cannot be built here; there is no CPU fallback" % path)
    # One HIP runtime per process: torch ships its own libamdhip64 and the batch entry points take torch
    # tensors, so torch's copy has to be the one the loader binds first (a process that initialised
    # /opt/rocm's runtime before importing torch leaves torch without a visible device).
    import torch  # noqa: F401

    L = C.CDLL(path)
    L.rspt_hip_status_string.restype, L.rspt_hip_status_string.argtypes = C.c_char_p, [C.c_int]
    L.rspt_hip_last_hip_error.restype, L.rspt_hip_last_hip_error.argtypes = C.c_int, [C.c_void_p]
    L.rspt_hip_device_count.restype, L.rspt_hip_device_count.argtypes = C.c_int, []
    L.rspt_hip_packer_create.restype = C.c_int
    L.rspt_hip_packer_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]
    L.rspt_hip_packer_destroy.restype, L.rspt_hip_packer_destroy.argtypes = None, [C.c_void_p]
    L.rspt_hip_compress.restype, L.rspt_hip_compress.argtypes = C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, _szp]
    L.rspt_hip_decompress.restype, L.rspt_hip_decompress.argtypes = C.c_int, [C.c_void_p, C.c_void_p, _szp, C.c_void_p]
    L.rspt_hip_decompress_bounded.restype, L.rspt_hip_decompress_bounded.argtypes = C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, _szp, C.c_void_p]
    L.rspt_hip_compress_many.restype = C.c_int
    L.rspt_hip_compress_many.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, _szp]
    L.rspt_hip_decompress_many.restype = C.c_int
    L.rspt_hip_decompress_many.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, _szp, C.c_size_t, C.c_void_p, _szp]
    L.rspt_hip_max_compressed_size.restype, L.rspt_hip_max_compressed_size.argtypes = C.c_size_t, [C.c_void_p]
    L.rspt_hip_block_bytes.restype, L.rspt_hip_block_bytes.argtypes = C.c_size_t, [C.c_void_p]
    L.rspt_hip_current_nb.restype, L.rspt_hip_current_nb.argtypes = C.c_uint, [C.c_void_p]
    L.rspt_hip_set_nb.restype, L.rspt_hip_set_nb.argtypes = C.c_int, [C.c_void_p, C.c_uint]
    L.rspt_hip_set_verify.restype, L.rspt_hip_set_verify.argtypes = C.c_int, [C.c_void_p, C.c_int]
    L.rspt_hip_reserve.restype, L.rspt_hip_reserve.argtypes = C.c_int, [C.c_void_p, C.c_size_t]
    L.rspt_hip_compress_batch_dev.restype = C.c_int
    L.rspt_hip_compress_batch_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.rspt_hip_decompress_batch_dev.restype = C.c_int
    L.rspt_hip_decompress_batch_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    L.rspt_hip_decompress_packed_dev.restype = C.c_int
    L.rspt_hip_decompress_packed_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    L.rspt_hip_synchronize.restype, L.rspt_hip_synchronize.argtypes = C.c_int, [C.c_void_p]
    L.rspt_hip_stream.restype, L.rspt_hip_stream.argtypes = C.c_void_p, [C.c_void_p]
    L.rspt_hip_pack_bound.restype, L.rspt_hip_pack_bound.argtypes = C.c_size_t, [C.c_void_p, C.c_size_t]
    L.rspt_hip_pack_batch_dev.restype = C.c_int
    L.rspt_hip_pack_batch_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    L.rspt_hip_set_profiling.restype, L.rspt_hip_set_profiling.argtypes = C.c_int, [C.c_void_p, C.c_int]
    L.rspt_hip_stage_count.restype, L.rspt_hip_stage_count.argtypes = C.c_int, [C.c_void_p]
    L.rspt_hip_stage_name.restype, L.rspt_hip_stage_name.argtypes = C.c_char_p, [C.c_void_p, C.c_int]
    L.rspt_hip_stage_times.restype, L.rspt_hip_stage_times.argtypes = C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int]
    L.rspt_hip_debug_read.restype, L.rspt_hip_debug_read.argtypes = C.c_longlong, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    L.rspt_hip_set_byte_order.restype, L.rspt_hip_set_byte_order.argtypes = C.c_int, [C.c_void_p, C.c_int]
    L.rspt_hip_host_alloc.restype, L.rspt_hip_host_alloc.argtypes = C.c_void_p, [C.c_size_t]
    L.rspt_hip_host_free.restype, L.rspt_hip_host_free.argtypes = None, [C.c_void_p]
    L.rspt_hip_iir_prefilter_batch_dev.restype = C.c_int
    L.rspt_hip_iir_prefilter_batch_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_size_t, C.c_int,
                                                   C.c_int, C.c_void_p]
    L.rspt_hip_feed_begin.restype, L.rspt_hip_feed_begin.argtypes = C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t]
    L.rspt_hip_feed_push.restype, L.rspt_hip_feed_push.argtypes = C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.rspt_hip_feed_submit.restype, L.rspt_hip_feed_submit.argtypes = C.c_int, [C.c_void_p]
    L.rspt_hip_feed_poll.restype, L.rspt_hip_feed_poll.argtypes = C.c_int, [C.c_void_p, _szp, _szp, C.POINTER(C.c_int)]
    L.rspt_hip_feed_flush.restype, L.rspt_hip_feed_flush.argtypes = C.c_int, [C.c_void_p]
    L.rspt_hip_feed_end.restype, L.rspt_hip_feed_end.argtypes = C.c_int, [C.c_void_p]
    # the C++ factories behind the same library (include/signal_packer.h), via their C shim
    L.rspt_cxx_new.restype, L.rspt_cxx_new.argtypes = C.c_void_p, [C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t]
    L.rspt_cxx_delete.restype, L.rspt_cxx_delete.argtypes = None, [C.c_int, C.c_void_p]
    L.rspt_cxx_compress.restype, L.rspt_cxx_compress.argtypes = None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, _szp]
    L.rspt_cxx_decompress.restype, L.rspt_cxx_decompress.argtypes = C.c_int, [C.c_void_p, C.c_void_p, _szp, C.c_void_p]
    L.rspt_cxx_set_device.restype, L.rspt_cxx_set_device.argtypes = C.c_int, [C.c_int]
    _lib = L
    return L


def _as_u8(buf):
    a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf.view(np.uint8).reshape(-1)
    return np.ascontiguousarray(a)


class SignalPacker:
    """One i_signal_packer instance on one GPU."""

    def __init__(self, kind, bytes_per_channel, nr_of_channels, nr_of_samples_in_each_channel, nr_bytes_to_encode=3, device=0):
        self._L = lib()
        kind_flags = KINDS[kind] if isinstance(kind, str) else int(kind)
        self.kind = kind_flags & 0xFF
        self.bps, self.nch, self.ns = bytes_per_channel, nr_of_channels, nr_of_samples_in_each_channel
        h = C.c_void_p()
        rc = self._L.rspt_hip_packer_create(C.byref(h), kind_flags, self.bps, self.nch, self.ns, nr_bytes_to_encode, device)
        if rc != 0:
            raise RsptHipError("rspt_hip_packer_create", rc)
        self._h = h
        self.block_bytes = self._L.rspt_hip_block_bytes(h)
        self.max_compressed_size = self._L.rspt_hip_max_compressed_size(h)

    def _check(self, where, rc):
        if rc != 0:
            raise RsptHipError(where, rc, self._L.rspt_hip_last_hip_error(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._L.rspt_hip_packer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- i_signal_packer::compress / decompress (host buffers) ----------------
    def compress(self, src, dst_max_len=None):
        a = _as_u8(src)
        assert a.size == self.block_bytes, (a.size, self.block_bytes)
        cap = dst_max_len if dst_max_len is not None else 2 * self.block_bytes + 4096
        out = np.empty(cap, dtype=np.uint8)
        n = C.c_size_t(0)
        self._check("rspt_hip_compress", self._L.rspt_hip_compress(self._h, a.ctypes.data, out.ctypes.data, cap, C.byref(n)))
        return out[: n.value].tobytes()

    def compress_into(self, src, out):
        """host buffers as they are (numpy uint8 arrays, e.g. from host_alloc): -> stream length"""
        n = C.c_size_t(0)
        self._check("rspt_hip_compress", self._L.rspt_hip_compress(self._h, src.ctypes.data, out.ctypes.data, out.size, C.byref(n)))
        return n.value

    def compress_many(self, src, out, raise_on_small=True):
        """a sequence of blocks from host memory through the upload | compress | download pipeline (rspt_hip_compress_many):
        src = uint8 array of n * block_bytes, out = uint8 array [n, stride] -> array of the n stream lengths"""
        n = src.size // self.block_bytes
        assert src.size == n * self.block_bytes and out.ndim == 2 and out.shape[0] == n
        lens = (C.c_size_t * n)()
        rc = self._L.rspt_hip_compress_many(self._h, src.ctypes.data, n, out.ctypes.data, out.strides[0], lens)
        if rc != -5 or raise_on_small:  # RSPT_HIP_ERR_DST_TOO_SMALL: the lengths say which streams
            self._check("rspt_hip_compress_many", rc)
        return np.array(lens[:], dtype=np.int64)

    # -- a feed of blocks that arrive over time (rspt_hip_feed_*): push when a block is there, poll for finished streams ----
    def feed_begin(self, blocks_per_launch=1, slots=3):
        self._check("rspt_hip_feed_begin", self._L.rspt_hip_feed_begin(self._h, blocks_per_launch, slots))

    def feed_push(self, src, dst):
        """queue one block (uint8 arrays that stay alive until the block is polled); False: the ring is full, poll first"""
        rc = self._L.rspt_hip_feed_push(self._h, src.ctypes.data, dst.ctypes.data, dst.size)
        if rc == -8:  # RSPT_HIP_ERR_BUSY
            return False
        self._check("rspt_hip_feed_push", rc)
        return True

    def feed_submit(self):
        self._check("rspt_hip_feed_submit", self._L.rspt_hip_feed_submit(self._h))

    def feed_poll(self):
        """-> (seq, length, status) of one finished block in push order, or None when none is ready (never waits)"""
        seq, n, st = C.c_size_t(0), C.c_size_t(0), C.c_int(0)
        rc = self._L.rspt_hip_feed_poll(self._h, C.byref(seq), C.byref(n), C.byref(st))
        if rc == 0:
            return None
        if rc != 1:
            self._check("rspt_hip_feed_poll", rc)
        return seq.value, n.value, st.value

    def feed_flush(self):
        self._check("rspt_hip_feed_flush", self._L.rspt_hip_feed_flush(self._h))

    def feed_end(self):
        self._check("rspt_hip_feed_end", self._L.rspt_hip_feed_end(self._h))

    def decompress_many(self, streams, out, lengths=None):
        """streams = uint8 array [n, stride] (one stream per row), out = uint8 array of n * block_bytes -> bytes consumed per stream
        (rspt_hip_decompress_many: upload | decode | download pipeline); lengths (optional): what to upload of each stream"""
        n = streams.shape[0]
        assert streams.ndim == 2 and out.size == n * self.block_bytes
        used = (C.c_size_t * n)()
        lens = (C.c_size_t * n)(*[int(v) for v in lengths]) if lengths is not None else None
        self._check("rspt_hip_decompress_many",
                    self._L.rspt_hip_decompress_many(self._h, streams.ctypes.data, streams.strides[0], lens, n, out.ctypes.data, used))
        return np.array(used[:], dtype=np.int64)

    def decompress_into(self, stream, out):
        n = C.c_size_t(0)
        self._check("rspt_hip_decompress", self._L.rspt_hip_decompress(self._h, stream.ctypes.data, C.byref(n), out.ctypes.data))
        return n.value

    def decompress(self, stream, bounded=False):
        """bounded: rspt_hip_decompress_bounded with the length of `stream` (for streams that may be damaged)"""
        s = _as_u8(stream)
        out = np.empty(self.block_bytes, dtype=np.uint8)
        n = C.c_size_t(0)
        if bounded:
            self._check("rspt_hip_decompress_bounded", self._L.rspt_hip_decompress_bounded(self._h, s.ctypes.data, s.size, C.byref(n), out.ctypes.data))
        else:
            self._check("rspt_hip_decompress", self._L.rspt_hip_decompress(self._h, s.ctypes.data, C.byref(n), out.ctypes.data))
        return out.tobytes(), n.value

    @property
    def nb(self):
        return self._L.rspt_hip_current_nb(self._h)

    def set_nb(self, nb):
        self._check("rspt_hip_set_nb", self._L.rspt_hip_set_nb(self._h, nb))

    def set_byte_order(self, big_endian=True):
        """samples arrive (compress) and leave (decompress) with their bytes reversed (utils.cpp reverse_byte_order branches)"""
        self._check("rspt_hip_set_byte_order", self._L.rspt_hip_set_byte_order(self._h, int(bool(big_endian))))

    def set_verify(self, on=True):
        """check every block's CRC-32C on decompress (hzr_verify's job in the reference); off by default"""
        self._check("rspt_hip_set_verify", self._L.rspt_hip_set_verify(self._h, int(bool(on))))

    # -- device-resident batches (torch tensors carry the memory) --------------
    def reserve(self, nblocks):
        self._check("rspt_hip_reserve", self._L.rspt_hip_reserve(self._h, nblocks))

    def compress_batch(self, d_src, d_dst=None, d_sizes=None, dst_stride=None, stream=None):
        """d_src: uint8 cuda tensor [nblocks, block_bytes].  Returns (d_dst, d_sizes);
        asynchronous on `stream` (default: torch's current stream)."""
        import torch

        assert d_src.is_cuda and d_src.dtype == torch.uint8 and d_src.is_contiguous()
        nblocks = d_src.numel() // self.block_bytes
        assert nblocks * self.block_bytes == d_src.numel()
        if dst_stride is None:
            dst_stride = (self.max_compressed_size + 255) // 256 * 256 if d_dst is None else d_dst.numel() // nblocks
        if d_dst is None:
            d_dst = torch.empty((nblocks, dst_stride), dtype=torch.uint8, device=d_src.device)
        if d_sizes is None:
            d_sizes = torch.empty(nblocks, dtype=torch.int64, device=d_src.device)
        st = stream if stream is not None else torch.cuda.current_stream(d_src.device).cuda_stream
        rc = self._L.rspt_hip_compress_batch_dev(self._h, d_src.data_ptr(), nblocks, d_dst.data_ptr(), dst_stride, d_sizes.data_ptr(), st)
        self._check("rspt_hip_compress_batch_dev", rc)
        return d_dst, d_sizes

    def decompress_batch(self, d_streams, nblocks, src_stride, d_out=None, d_consumed=None, stream=None):
        import torch

        if d_out is None:
            d_out = torch.empty((nblocks, self.block_bytes), dtype=torch.uint8, device=d_streams.device)
        if d_consumed is None:
            d_consumed = torch.empty(nblocks, dtype=torch.int64, device=d_streams.device)
        st = stream if stream is not None else torch.cuda.current_stream(d_streams.device).cuda_stream
        rc = self._L.rspt_hip_decompress_batch_dev(self._h, d_streams.data_ptr(), src_stride, nblocks, d_out.data_ptr(), d_consumed.data_ptr(), st)
        self._check("rspt_hip_decompress_batch_dev", rc)
        return d_out, d_consumed

    def decompress_packed(self, d_packed, d_out=None, d_consumed=None, stream=None, nbytes=None):
        """Decompress every stream of a container (what pack_batch / the multi-GPU gather produce) on the device.
        Each stream is decoded with the nb of its own index entry.  Reads the 32-byte header to the host for the block
        count (a synchronisation).  `nbytes`: container length if shorter than the tensor."""
        import torch

        head = d_packed[:32].cpu().numpy().view(np.uint64)
        if int(head[0]) != 0x4B43415054505352:
            raise ValueError("not an RSPTPACK container")
        nblocks = int(head[1])
        plen = int(nbytes) if nbytes is not None else d_packed.numel()
        if nblocks == 0 or nblocks > 65535 or plen < 32 or nblocks > (plen - 32) // 16:  # (nothing is sized from an untrusted count)
            raise RsptHipError("rspt_hip_decompress_packed_dev", -6 if 0 < nblocks <= 65535 else -1)
        if d_out is None:
            d_out = torch.empty((nblocks, self.block_bytes), dtype=torch.uint8, device=d_packed.device)
        if d_consumed is None:
            d_consumed = torch.empty(nblocks, dtype=torch.int64, device=d_packed.device)
        st = stream if stream is not None else torch.cuda.current_stream(d_packed.device).cuda_stream
        rc = self._L.rspt_hip_decompress_packed_dev(self._h, d_packed.data_ptr(), plen, nblocks, d_out.data_ptr(), d_consumed.data_ptr(), st)
        self._check("rspt_hip_decompress_packed_dev", rc)
        return d_out, d_consumed

    def pack_bound(self, nblocks):
        return self._L.rspt_hip_pack_bound(self._h, nblocks)

    def pack_batch(self, d_dst, d_sizes, d_packed=None, d_total=None, stream=None):
        """streams of a batch -> one container (layout: include/rspt_hip.h); asynchronous."""
        import torch

        nblocks = d_sizes.numel()
        stride = d_dst.numel() // nblocks
        if d_packed is None:
            d_packed = torch.empty(self.pack_bound(nblocks), dtype=torch.uint8, device=d_dst.device)
        if d_total is None:
            d_total = torch.zeros(1, dtype=torch.int64, device=d_dst.device)
        st = stream if stream is not None else torch.cuda.current_stream(d_dst.device).cuda_stream
        rc = self._L.rspt_hip_pack_batch_dev(self._h, d_dst.data_ptr(), stride, d_sizes.data_ptr(), nblocks, d_packed.data_ptr(), d_total.data_ptr(), st)
        self._check("rspt_hip_pack_batch_dev", rc)
        return d_packed, d_total

    def iir_prefilter_batch(self, d_buf, n, d, init_nr_samples=2000, per_channel=False, stream=None):
        """The reference's pre-filter step (rspt_test.cpp:116-136) on device-resident blocks, in place; asynchronous."""
        import torch

        assert d_buf.is_cuda and d_buf.dtype == torch.uint8 and d_buf.is_contiguous()
        nblocks = d_buf.numel() // self.block_bytes
        assert nblocks * self.block_bytes == d_buf.numel()
        nn, dd = np.ascontiguousarray(n, dtype=np.float64), np.ascontiguousarray(d, dtype=np.float64)
        assert nn.size == dd.size
        st = stream if stream is not None else torch.cuda.current_stream(d_buf.device).cuda_stream
        rc = self._L.rspt_hip_iir_prefilter_batch_dev(self._h, d_buf.data_ptr(), nblocks, nn.ctypes.data_as(C.POINTER(C.c_double)),
                                                      dd.ctypes.data_as(C.POINTER(C.c_double)), nn.size, init_nr_samples, int(bool(per_channel)), st)
        self._check("rspt_hip_iir_prefilter_batch_dev", rc)
        return d_buf

    def synchronize(self):
        self._check("rspt_hip_synchronize", self._L.rspt_hip_synchronize(self._h))

    @property
    def stream_ptr(self):
        """the handle's own hipStream_t (rspt_hip_stream): what the host-pointer entry points run on; a caller with several
        handles in flight can launch each handle's batches on it (torch.cuda.ExternalStream(pk.stream_ptr)) instead of making
        further streams -- the runtime maps streams onto few hardware queues (GPU_MAX_HW_QUEUES, default 4)"""
        return int(self._L.rspt_hip_stream(self._h) or 0)

    def debug_read(self, which, nbytes):
        """test hook: workspace buffer `which` of the last batch call (see rspt_hip.h)"""
        out = np.zeros(nbytes, dtype=np.uint8)
        n = self._L.rspt_hip_debug_read(self._h, which, out.ctypes.data, nbytes)
        if n < 0:
            raise RsptHipError("rspt_hip_debug_read", int(n))
        return out[:n]

    # -- measurement -------------------------------------------------------------
    def set_profiling(self, on=True):
        self._L.rspt_hip_set_profiling(self._h, int(on))

    def stage_times(self):
        n = self._L.rspt_hip_stage_count(self._h)
        ms = (C.c_float * n)()
        self._check("rspt_hip_stage_times", self._L.rspt_hip_stage_times(self._h, ms, n))
        return {self._L.rspt_hip_stage_name(self._h, i).decode(): float(ms[i]) for i in range(n)}


class HostBuffer:
    """page-locked host memory (rspt_hip_host_alloc) as a numpy uint8 array: `.a`"""

    def __init__(self, nbytes):
        self._L = lib()
        self._p = self._L.rspt_hip_host_alloc(nbytes)
        if not self._p:
            raise MemoryError("rspt_hip_host_alloc(%d)" % nbytes)
        self.a = np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(self._p))

    def close(self):
        if getattr(self, "_p", None):
            self.a = None
            self._L.rspt_hip_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# factory names of lib_rspt/signal_packer.h:59-69
def new_xdelta_hzr(bytes_per_channel, nr_of_channels, nr_of_samples_in_each_channel, nr_bytes_to_encode, device=0):
    return SignalPacker(KIND_XDELTA_HZR, bytes_per_channel, nr_of_channels, nr_of_samples_in_each_channel, nr_bytes_to_encode, device)


def new_hzr(bytes_per_channel, nr_of_channels, nr_of_samples_in_each_channel, device=0):
    return SignalPacker(KIND_HZR, bytes_per_channel, nr_of_channels, nr_of_samples_in_each_channel, 4, device)


def new_dct(bytes_per_channel, nr_of_channels, nr_of_samples_in_each_channel, device=0):
    return SignalPacker(KIND_DCT, bytes_per_channel, nr_of_channels, nr_of_samples_in_each_channel, 2, device)


def new_hadamard(bytes_per_channel, nr_of_channels, nr_of_samples_in_each_channel, device=0):
    return SignalPacker(KIND_HADAMARD, bytes_per_channel, nr_of_channels, nr_of_samples_in_each_channel, 3, device)


class CxxSignalPacker:
    """Drives the C++ i_signal_packer factories of include/signal_packer.h themselves
    (through the tiny C shim in signal_packer_hip.cpp) -- the reference-facing surface."""

    def __init__(self, kind, bps, nch, ns, nb=3):
        self._L = lib()
        self.kind = KINDS[kind] if isinstance(kind, str) else int(kind)
        self.block_bytes = bps * nch * ns
        self._p = self._L.rspt_cxx_new(self.kind, bps, nch, ns, nb)
        if not self._p:
            raise RuntimeError("i_signal_packer factory failed")

    def close(self):
        if getattr(self, "_p", None):
            self._L.rspt_cxx_delete(self.kind, self._p)
            self._p = None

    def compress(self, src):
        a = _as_u8(src)
        cap = 2 * self.block_bytes + 4096
        out = np.empty(cap, dtype=np.uint8)
        n = C.c_size_t(0)
        self._L.rspt_cxx_compress(self._p, a.ctypes.data, out.ctypes.data, cap, C.byref(n))
        return out[: n.value].tobytes()

    def decompress(self, stream):
        s = _as_u8(stream)
        out = np.empty(self.block_bytes, dtype=np.uint8)
        n = C.c_size_t(0)
        rc = self._L.rspt_cxx_decompress(self._p, s.ctypes.data, C.byref(n), out.ctypes.data)
        return out.tobytes(), n.value, rc
