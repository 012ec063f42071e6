"""ctypes front-end to the checker libraries.  TEST INFRASTRUCTURE ONLY.

  * ``Oracle``  -> oracle/librspt_oracle.so   (our CPU restatement, rspt_oracle.c)
  * ``Ref``     -> oracle/_ref/librspt_ref.so (the real reference, compiled from
                   /root/reference by oracle/Makefile; may be absent)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module; the product (rspt_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "librspt_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "librspt_ref.so")

KIND_HZR, KIND_XDELTA_HZR, KIND_DCT, KIND_HADAMARD = 0, 1, 2, 3
KINDS = {"hzr": 0, "xdelta_hzr": 1, "dct": 2, "hadamard": 3}

_u8p = C.POINTER(C.c_uint8)
_szp = C.POINTER(C.c_size_t)


def build(force=False):
    """Compile the checker(s) (never the product)."""
    if force or not os.path.exists(ORACLE_SO) or (
        os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(_HERE, "rspt_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "librspt_oracle.so"], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/lib_rspt/signal_packer.h") and (
        force or not os.path.exists(REF_SO)
        or os.path.getmtime(REF_SO) < max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("ref_shim.cpp", "Makefile"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


def _ptr(a):
    return a.ctypes.data_as(_u8p)


def _as_u8(buf):
    a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf.view(np.uint8).reshape(-1)
    return np.ascontiguousarray(a)


class _Lib:
    prefix = ""

    def __init__(self, path):
        self.lib = C.CDLL(path)
        L, p = self.lib, self.prefix
        f = getattr(L, p + "hzr_max_compressed_size")
        f.restype, f.argtypes = C.c_size_t, [C.c_size_t]
        f = getattr(L, p + "hzr_encode")
        f.restype, f.argtypes = C.c_int, [_u8p, C.c_size_t, _u8p, C.c_size_t, _szp]
        f = getattr(L, p + "packer_new")
        f.restype, f.argtypes = C.c_void_p, [C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t]
        f = getattr(L, p + "packer_free")
        f.restype, f.argtypes = None, [C.c_void_p]
        f = getattr(L, p + "packer_compress")
        f.restype, f.argtypes = C.c_int, [C.c_void_p, _u8p, _u8p, C.c_size_t, _szp]
        f = getattr(L, p + "packer_decompress")
        f.restype, f.argtypes = C.c_int, [C.c_void_p, _u8p, _szp, _u8p]

    def hzr_max_compressed_size(self, n):
        return getattr(self.lib, self.prefix + "hzr_max_compressed_size")(n)

    def hzr_encode(self, data):
        a = _as_u8(data)
        cap = self.hzr_max_compressed_size(a.size)
        out = np.zeros(cap, dtype=np.uint8)
        n = C.c_size_t(0)
        src = a if a.size else np.zeros(1, dtype=np.uint8)
        ok = getattr(self.lib, self.prefix + "hzr_encode")(_ptr(src), a.size, _ptr(out), cap, C.byref(n))
        if not ok:
            raise RuntimeError("hzr_encode failed")
        return out[: n.value].tobytes()

    def packer(self, kind, bps, nch, ns, nb=3):
        return Packer(self, KINDS[kind] if isinstance(kind, str) else kind, bps, nch, ns, nb)


class Packer:
    """i_signal_packer-shaped handle (lib_rspt/signal_packer.h:29-73)."""

    def __init__(self, lib, kind, bps, nch, ns, nb):
        self._l, self.kind, self.bps, self.nch, self.ns = lib, kind, bps, nch, ns
        self._h = getattr(lib.lib, lib.prefix + "packer_new")(kind, bps, nch, ns, nb)
        if not self._h:
            raise ValueError("packer_new refused (kind=%d bps=%d nch=%d ns=%d nb=%d)" % (kind, bps, nch, ns, nb))
        self.in_bytes = bps * nch * ns

    def close(self):
        if self._h:
            getattr(self._l.lib, self._l.prefix + "packer_free")(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def compress(self, src, dst_max_len=None):
        a = _as_u8(src)
        assert a.size == self.in_bytes, (a.size, self.in_bytes)
        # the reference over-reads up to 3 bytes past the last sample for bps<4 (utils.cpp:155-189)
        padded = np.zeros(a.size + 8, dtype=np.uint8)
        padded[: a.size] = a
        cap = dst_max_len if dst_max_len is not None else 2 * self.in_bytes + 4096
        out = np.zeros(cap + 64, dtype=np.uint8)
        n = C.c_size_t(0)
        rc = getattr(self._l.lib, self._l.prefix + "packer_compress")(self._h, _ptr(padded), _ptr(out), cap, C.byref(n))
        if rc != 0:
            raise RuntimeError("compress failed rc=%d" % rc)
        return out[: n.value].tobytes()

    def decompress(self, stream):
        s = _as_u8(stream)
        padded = np.zeros(s.size + 16, dtype=np.uint8)
        padded[: s.size] = s
        out = np.zeros(self.in_bytes + 8, dtype=np.uint8)
        n = C.c_size_t(0)
        rc = getattr(self._l.lib, self._l.prefix + "packer_decompress")(self._h, _ptr(padded), C.byref(n), _ptr(out))
        return out[: self.in_bytes].tobytes(), n.value, rc


class Oracle(_Lib):
    prefix = "orc_"

    def __init__(self, path=None):
        build()
        super().__init__(path or ORACLE_SO)
        L = self.lib
        L.orc_crc32c.restype, L.orc_crc32c.argtypes = C.c_uint32, [_u8p, C.c_size_t]
        L.orc_fnv1a.restype, L.orc_fnv1a.argtypes = C.c_uint32, [_u8p, C.c_size_t]
        L.orc_hzr_decode.restype, L.orc_hzr_decode.argtypes = C.c_int, [_u8p, C.c_size_t, _u8p, C.c_size_t, _szp]
        L.orc_hzr_verify.restype, L.orc_hzr_verify.argtypes = C.c_int, [_u8p, C.c_size_t, _szp]
        L.orc_hzr_block_stats.restype = None
        L.orc_hzr_block_stats.argtypes = [_u8p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_int), _szp]
        L.orc_packer_nb.restype, L.orc_packer_nb.argtypes = C.c_uint, [C.c_void_p]
        L.orc_packer_set_fast_verify.restype, L.orc_packer_set_fast_verify.argtypes = None, [C.c_void_p, C.c_int]
        L.orc_packer_max_compressed_size.restype, L.orc_packer_max_compressed_size.argtypes = C.c_size_t, [C.c_void_p]
        L.orc_packer_last_enc.restype, L.orc_packer_last_enc.argtypes = C.POINTER(C.c_int32), [C.c_void_p]
        L.orc_prdn.restype, L.orc_prdn.argtypes = C.c_double, [_u8p, _u8p, C.c_size_t, C.c_size_t, C.c_size_t]
        L.orc_xdelta_needed_nb.restype = C.c_uint
        L.orc_xdelta_needed_nb.argtypes = [C.POINTER(C.c_int32), C.c_size_t, C.c_size_t, C.c_uint]
        L.orc_xdelta_forward.restype, L.orc_xdelta_forward.argtypes = None, [C.POINTER(C.c_int32), C.c_size_t]
        L.orc_native_to_i32.restype = None
        L.orc_native_to_i32.argtypes = [C.POINTER(C.c_int32), _u8p, C.c_size_t, C.c_size_t, C.c_size_t]
        L.orc_fwht.restype, L.orc_fwht.argtypes = None, [C.POINTER(C.c_int32), C.c_size_t]
        L.orc_average_32.restype, L.orc_average_32.argtypes = C.c_int32, [C.POINTER(C.c_int32), C.c_size_t]
        L.orc_iir_prefilter_native.restype = C.c_int
        L.orc_iir_prefilter_native.argtypes = [_u8p, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_size_t, C.c_int, C.c_int]

    def iir_prefilter(self, native, bps, nch, ns, n, d, init_nr_samples=2000, shared_state=True):
        """the reference's pre-filter step (rspt_test.cpp:116-136): interleaved native block -> filtered native block"""
        a = _as_u8(native).copy()
        nn, dd = np.ascontiguousarray(n, dtype=np.float64), np.ascontiguousarray(d, dtype=np.float64)
        rc = self.lib.orc_iir_prefilter_native(_ptr(a), bps, nch, ns, nn.ctypes.data_as(C.POINTER(C.c_double)), dd.ctypes.data_as(C.POINTER(C.c_double)),
                                               nn.size, init_nr_samples, int(bool(shared_state)))
        if rc != 0:
            raise ValueError("orc_iir_prefilter_native rc=%d" % rc)
        return a.tobytes()

    # ---- dct beyond the dense table (ns > 8192): fp64 restatement -----------------------------
    # Same definition as signal_packer_dct.cpp:76-100 with the cosines in fp64 instead of the float32
    # table (which cannot be built at this size); the sum is evaluated by scipy's fp64 DCT.
    # PARITY NOTE: no reference run exists for these sizes; at ns <= 8192 the same routine is checked
    # against the real reference under SURVEY 8(d)'s PRDN / CR gate (tests/test_oracle_golden.py).
    def dct_big_compress(self, native, bps, nch, ns):
        import scipy.fft

        L = self.lib
        L.orc_packer_compress_coeffs.restype = C.c_int
        L.orc_packer_compress_coeffs.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _u8p, C.c_size_t, _szp]
        x = self.native_to_i32(native, ns, nch, bps)
        means = np.array([self.average_32(x[c]) for c in range(nch)], dtype=np.int32)
        s = (x - means[:, None]).astype(np.int32).astype(np.float32).astype(np.float64)  # (float)src: dct.cpp:80
        csum = scipy.fft.dct(s, type=2, axis=1) * 0.5  # sum_x s[x] cos(pi (2x+1) i / 2n)
        ratio1 = np.sqrt(2.0 / ns)
        cs0 = np.float32(1 / np.sqrt(2))
        scale = np.full(ns, float(np.float32(1.0)) * ratio1 / 128.0)
        scale[0] = float(cs0) * ratio1 / 128.0
        coeffs = np.ascontiguousarray(np.trunc(csum * scale[None, :]).astype(np.int64).astype(np.int32))
        pk = self.packer("dct", bps, nch, ns)
        cap = self.packer_max_compressed_size(pk) + 64
        out = np.zeros(cap, dtype=np.uint8)
        n = C.c_size_t(0)
        rc = L.orc_packer_compress_coeffs(pk._h, coeffs.ctypes.data_as(C.POINTER(C.c_int32)), means.ctypes.data_as(C.POINTER(C.c_int32)),
                                          _ptr(out), cap, C.byref(n))
        pk.close()
        if rc:
            raise RuntimeError("compress_coeffs rc=%d" % rc)
        return out[: n.value].tobytes(), coeffs

    def dct_big_decompress(self, stream, bps, nch, ns):
        import scipy.fft

        L = self.lib
        L.orc_packer_decompress_coeffs.restype = C.c_int
        L.orc_packer_decompress_coeffs.argtypes = [C.c_void_p, _u8p, _szp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.orc_i32_to_native.restype = None
        L.orc_i32_to_native.argtypes = [_u8p, C.POINTER(C.c_int32), C.c_size_t, C.c_size_t, C.c_size_t]
        s = _as_u8(stream)
        padded = np.zeros(s.size + 16, dtype=np.uint8)
        padded[: s.size] = s
        coeffs = np.zeros((nch, ns), dtype=np.int32)
        means = np.zeros(nch, dtype=np.int32)
        used = C.c_size_t(0)
        pk = self.packer("dct", bps, nch, ns)
        rc = L.orc_packer_decompress_coeffs(pk._h, _ptr(padded), C.byref(used), coeffs.ctypes.data_as(C.POINTER(C.c_int32)),
                                            means.ctypes.data_as(C.POINTER(C.c_int32)))
        pk.close()
        if rc:
            raise RuntimeError("decompress_coeffs rc=%d" % rc)
        y = coeffs.astype(np.float32).astype(np.float64)
        y[:, 0] = (np.float32(1 / np.sqrt(2)) * coeffs[:, 0].astype(np.float32)).astype(np.float64)  # Cs[0]*dct[0] in float (dct.cpp:95)
        # sum_x Y[x] cos(pi (2i+1) x / 2n) = DCT-III with full weight on Y[0]: scipy's type 3 is Y0 + 2 sum_{x>0}
        y[:, 1:] *= 0.5
        rec = scipy.fft.dct(y, type=3, axis=1)
        rec = np.trunc(rec * (np.sqrt(2.0 / ns) * 128.0)).astype(np.int64).astype(np.int32)
        rec = np.ascontiguousarray((rec.astype(np.int64) + means[:, None]).astype(np.int32))  # wrap like int32 add
        out = np.zeros(bps * nch * ns + 8, dtype=np.uint8)
        L.orc_i32_to_native(_ptr(out), rec.ctypes.data_as(C.POINTER(C.c_int32)), ns, nch, bps)
        return out[: bps * nch * ns].tobytes(), used.value

    def crc32c(self, data):
        a = _as_u8(data)
        src = a if a.size else np.zeros(1, dtype=np.uint8)
        return self.lib.orc_crc32c(_ptr(src), a.size)

    def fnv1a(self, data):
        a = _as_u8(data)
        src = a if a.size else np.zeros(1, dtype=np.uint8)
        return self.lib.orc_fnv1a(_ptr(src), a.size)

    def hzr_decode(self, stream, out_size):
        s = _as_u8(stream)
        out = np.zeros(max(out_size, 1), dtype=np.uint8)
        used = C.c_size_t(0)
        ok = self.lib.orc_hzr_decode(_ptr(s), s.size, _ptr(out), out_size, C.byref(used))
        if not ok:
            raise RuntimeError("hzr_decode failed")
        return out[:out_size].tobytes(), used.value

    def hzr_verify(self, stream):
        s = _as_u8(stream)
        n = C.c_size_t(0)
        return bool(self.lib.orc_hzr_verify(_ptr(s), s.size, C.byref(n))), n.value

    def hzr_block_stats(self, block):
        a = _as_u8(block)
        hist = np.zeros(261, dtype=np.uint32)
        mode, plen = C.c_int(0), C.c_size_t(0)
        self.lib.orc_hzr_block_stats(_ptr(a), a.size, hist.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(mode), C.byref(plen))
        return hist, mode.value, plen.value

    def native_to_i32(self, native, ns, nch, bps):
        a = _as_u8(native)
        out = np.zeros(nch * ns, dtype=np.int32)
        self.lib.orc_native_to_i32(out.ctypes.data_as(C.POINTER(C.c_int32)), _ptr(a), ns, nch, bps)
        return out.reshape(nch, ns)

    def xdelta_forward(self, planar):
        v = np.ascontiguousarray(planar, dtype=np.int32).reshape(-1).copy()
        self.lib.orc_xdelta_forward(v.ctypes.data_as(C.POINTER(C.c_int32)), v.size)
        return v

    def xdelta_needed_nb(self, v, bps, nb_min=1):
        v = np.ascontiguousarray(v, dtype=np.int32).reshape(-1)
        return self.lib.orc_xdelta_needed_nb(v.ctypes.data_as(C.POINTER(C.c_int32)), v.size, bps, nb_min)

    def fwht(self, row):
        v = np.ascontiguousarray(row, dtype=np.int32).copy()
        self.lib.orc_fwht(v.ctypes.data_as(C.POINTER(C.c_int32)), v.size)
        return v

    def average_32(self, row):
        v = np.ascontiguousarray(row, dtype=np.int32)
        return self.lib.orc_average_32(v.ctypes.data_as(C.POINTER(C.c_int32)), v.size)

    def prdn(self, orig, dec, ns, nch, bps):
        a, b = _as_u8(orig), _as_u8(dec)
        return self.lib.orc_prdn(_ptr(a), _ptr(b), ns, nch, bps)

    def packer_nb(self, pk):
        return self.lib.orc_packer_nb(pk._h)

    def packer_set_fast_verify(self, pk, on=True):
        self.lib.orc_packer_set_fast_verify(pk._h, int(on))

    def packer_max_compressed_size(self, pk):
        return self.lib.orc_packer_max_compressed_size(pk._h)

    def packer_last_enc(self, pk):
        p = self.lib.orc_packer_last_enc(pk._h)
        return np.ctypeslib.as_array(p, shape=(pk.nch * pk.ns,)).copy()


class Ref(_Lib):
    prefix = "ref_"

    def __init__(self, path=None):
        build()
        path = path or REF_SO
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        super().__init__(path)
        L = self.lib
        L.ref_hzr_decode.restype, L.ref_hzr_decode.argtypes = C.c_int, [_u8p, C.c_size_t, _u8p, C.c_size_t]
        L.ref_hzr_verify.restype, L.ref_hzr_verify.argtypes = C.c_int, [_u8p, C.c_size_t, _szp]
        L.ref_iir_prefilter_native.restype = C.c_int
        L.ref_iir_prefilter_native.argtypes = [_u8p, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_size_t, C.c_int]
        L.ref_native_to_i32.restype = None
        L.ref_native_to_i32.argtypes = [C.POINTER(C.c_int32), _u8p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]
        L.ref_i32_to_native.restype = None
        L.ref_i32_to_native.argtypes = [_u8p, C.POINTER(C.c_int32), C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]

    def iir_prefilter(self, native, bps, nch, ns, n, d, init_nr_samples=2000):
        a = np.zeros(_as_u8(native).size + 8, dtype=np.uint8)  # (the reference over-reads up to 3 bytes for bps < 4)
        a[: _as_u8(native).size] = _as_u8(native)
        nn, dd = np.ascontiguousarray(n, dtype=np.float64), np.ascontiguousarray(d, dtype=np.float64)
        self.lib.ref_iir_prefilter_native(_ptr(a), bps, nch, ns, nn.ctypes.data_as(C.POINTER(C.c_double)), dd.ctypes.data_as(C.POINTER(C.c_double)),
                                          nn.size, init_nr_samples)
        return a[: bps * nch * ns].tobytes()

    def native_to_i32(self, native, ns, nch, bps, reverse_byte_order=False):
        a = np.zeros(_as_u8(native).size + 8, dtype=np.uint8)
        a[: _as_u8(native).size] = _as_u8(native)
        out = np.zeros((nch, ns), dtype=np.int32)
        self.lib.ref_native_to_i32(out.ctypes.data_as(C.POINTER(C.c_int32)), _ptr(a), ns, nch, bps, int(reverse_byte_order))
        return out

    def i32_to_native(self, planar, bps, reverse_byte_order=False):
        pl = np.ascontiguousarray(planar, dtype=np.int32)
        nch, ns = pl.shape
        out = np.zeros(bps * nch * ns + 8, dtype=np.uint8)
        self.lib.ref_i32_to_native(_ptr(out), pl.ctypes.data_as(C.POINTER(C.c_int32)), ns, nch, bps, int(reverse_byte_order))
        return out[: bps * nch * ns].tobytes()

    def hzr_decode(self, stream, out_size):
        s = _as_u8(stream)
        padded = np.zeros(s.size + 16, dtype=np.uint8)
        padded[: s.size] = s
        out = np.zeros(max(out_size, 1), dtype=np.uint8)
        ok = self.lib.ref_hzr_decode(_ptr(padded), s.size, _ptr(out), out_size)
        if not ok:
            raise RuntimeError("ref hzr_decode failed")
        return out[:out_size].tobytes()

    def hzr_verify(self, stream):
        s = _as_u8(stream)
        n = C.c_size_t(0)
        return bool(self.lib.ref_hzr_verify(_ptr(s), s.size, C.byref(n))), n.value


def have_ref():
    return os.path.exists(REF_SO)
