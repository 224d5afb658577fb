"""zzflate_amd -- MI355X-native DEFLATE encoder behind zzflate's entry points.

Host-side mirror of the reference interface (zzflate/zzflate.h:8-19) over the C ABI of
libzzflate_amd.so (include/zzflate_amd.h). Names follow the reference: ``Format``, ``Config``,
``ZzFlateEncode``, ``ZzFlateEncodeToCallback``, ``adler32x``, ``combine``, ``crc32``.

There is no CPU encode path: importing works without a GPU (so that the ABI can be inspected), but
every encode call raises ``ZzFlateError`` when no HIP device is usable.
"""
import ctypes
import enum
import os
from dataclasses import dataclass

from . import build as _build

__all__ = [
    "Format", "Config", "ZzFlateError", "ZzFlateEncode", "ZzFlateEncodeToCallback", "adler32x", "combine",
    "crc32", "crc32_combine", "bound", "Context", "lib", "DEFAULT_PACKET", "generate_host", "header", "trailer",
]

DEFAULT_PACKET = 32768
_ERR = (1 << 64) - 1


class Format(enum.IntEnum):  # zzflate.h:8
    Zlib = 0
    Gzip = 1
    Deflate = 2


@dataclass
class Config:  # zzflate.h:10-15
    format: Format = Format.Zlib
    level: int = 1
    threaded: bool = True


class _CConfig(ctypes.Structure):
    _fields_ = [("format", ctypes.c_int32), ("level", ctypes.c_uint8), ("threaded", ctypes.c_uint8)]


class ZzFlateError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"zzflate_amd error {code}: {msg}")
        self.code = code


def _load():
    # PyTorch wheels bundle their own HIP/HSA runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7). Two
    # HIP runtimes in one process do not work ("no ROCm-capable device"), so when torch is installed it is
    # imported first: the library's NEEDED libamdhip64.so.7 then binds to the runtime torch already loaded.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = _build.build()          # returns at once when the library is newer than every source under csrc/ and include/
    path = os.environ.get("ZZFLATE_AMD_LIB", path)      # diagnostics: an experimental build of the same sources
    L = ctypes.CDLL(path)
    u64, u32, i32, vp = ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p
    pu64 = ctypes.POINTER(ctypes.c_uint64)
    sig = {
        "zz_ctx_create": (i32, [i32, ctypes.POINTER(vp)]),
        "zz_ctx_destroy": (None, [vp]),
        "zz_ctx_workspace_bytes": (u64, [vp]),
        "zz_ctx_enable_timing": (None, [vp, i32]),
        "zz_ctx_set_warm_window": (i32, [vp, u32]),
        "zz_ctx_set_extended_levels": (i32, [vp, i32]),
        "zz_ctx_last_kernel_ms": (ctypes.c_double, [vp]),
        "zz_bound": (u64, [u64, i32, i32, u32]),
        "zz_encode": (i32, [vp, pu64, vp, u64, ctypes.POINTER(_CConfig)]),
        "zz_encode_callback": (i32, [vp, u64, ctypes.POINTER(_CConfig), vp, vp]),
        "zz_set_packet_size": (i32, [u32]),
        "zz_get_packet_size": (u32, []),
        "zz_encode_device": (i32, [vp, vp, u64, vp, u64, pu64, i32, i32, u32, vp]),
        "zz_encode_device_async": (i32, [vp, vp, u64, vp, u64, i32, i32, u32, vp]),
        "zz_encode_finish": (i32, [vp, pu64]),
        "zz_encode_stream_device": (i32, [vp, vp, u64, vp, u64, pu64, i32, i32, vp]),
        "zz_encode_ranges_device": (i32, [vp, vp, u64, vp, u64, pu64, i32, i32, u32, vp]),
        "zz_encode_stream_chunks_device": (i32, [vp, vp, u64, vp, u64, pu64, i32, i32, pu64, u32, ctypes.POINTER(u32), vp]),
        "zz_encode_shard_device": (i32, [vp, vp, u64, u64, i32, vp, u64, pu64, ctypes.POINTER(u32), i32, i32, u32, vp]),
        "zz_encode_shard_device_async": (i32, [vp, vp, u64, u64, i32, vp, u64, i32, i32, u32, vp]),
        "zz_encode_shard_finish": (i32, [vp, pu64, ctypes.POINTER(u32), i32]),
        "zz_encode_multi_device": (i32, [ctypes.POINTER(vp), i32, ctypes.POINTER(vp), pu64, pu64, vp, u64, pu64, i32, i32, u32]),
        "zz_verify_last_device": (i32, [vp, pu64, pu64, vp]),
        "zz_packet_extent_device": (i32, [vp, u64, pu64, pu64, vp]),
        "zz_header": (i32, [i32, vp]),
        "zz_trailer": (i32, [i32, u32, u64, vp]),
        "zz_adler32": (u32, [u32, vp, u64]),
        "zz_adler32_combine": (u32, [u32, u32, u64]),
        "zz_crc32": (u32, [vp, u64, u32]),
        "zz_crc32_combine": (u32, [u32, u32, u64]),
        "zz_generate_device": (i32, [vp, i32, u64, u64, vp, u64, vp]),
        "zz_generate_host": (i32, [i32, u64, u64, vp, u64]),
        "zz_debug_reset_devices": (None, []),
        "zz_debug_host_staging_bytes": (u64, []),
        "zz_debug_lds_atomic_order": (i32, [vp, u32, pu64, pu64]),
        "zz_debug_force_lds_order": (None, [i32]),
        "zz_debug_peer_state": (i32, [i32, i32]),
        "zz_debug_last_pulls": (None, [ctypes.POINTER(i32), ctypes.POINTER(i32)]),
        "zz_debug_lds_order_verdict": (i32, [i32]),
        "zz_last_error": (ctypes.c_char_p, []),
        "zz_version": (ctypes.c_char_p, []),
        "zz_build_flags": (ctypes.c_char_p, []),
        "zz_debug_force_lds_violation": (None, [i32]),
        "zz_debug_reset_lds_order": (None, [i32]),
        "zz_debug_l1_kernel": (i32, [vp]),
        "zz_debug_l2_kernel": (i32, [vp]),
    }
    # (diagnostic hooks an older experimental build named by ZZFLATE_AMD_LIB may lack: A/B runs of tools/abn.sh)
    optional = {"zz_build_flags", "zz_debug_force_lds_violation", "zz_debug_reset_lds_order", "zz_debug_l1_kernel", "zz_debug_l2_kernel"}
    for name, (res, args) in sig.items():
        if name in optional and not hasattr(L, name):
            continue
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    return L


lib = _load()
_CALLBACK = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint8), ctypes.c_uint64)


def _check(rc):
    if rc != 0:
        raise ZzFlateError(rc, lib.zz_last_error().decode())


def _cfg(config):
    return _CConfig(int(config.format), int(config.level), 1 if config.threaded else 0)


def bound(n, format=Format.Zlib, level=1, packet_size=DEFAULT_PACKET):
    return lib.zz_bound(n, int(format), int(level), packet_size)


def ZzFlateEncode(source, config, dest_capacity=None):
    """zzflate.h:17 -- returns the encoded bytes. ``dest_capacity`` plays the role of ``*destLen`` on
    entry; a destination that is too small raises (the C entry point sets ``*destLen = ~0``)."""
    import numpy as np
    src = source if isinstance(source, bytes) else bytes(source)
    cap = bound(len(src), config.format, config.level, lib.zz_get_packet_size()) if dest_capacity is None else dest_capacity
    dest = np.empty(max(cap, 1), dtype=np.uint8)           # not zero-filled: the library writes it
    n = ctypes.c_uint64(cap)
    c = _cfg(config)
    rc = lib.zz_encode(dest.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n), src, len(src), ctypes.byref(c))
    _check(rc)
    return dest[: n.value].tobytes()


def ZzFlateEncodeToCallback(source, config, callback):
    """zzflate.h:19 -- ``callback(chunk: bytes)`` is called for the header, each stream chunk and the trailer."""
    src = bytes(source)
    failure = []

    def tramp(_user, ptr, nbytes):
        # an exception must not vanish inside the ctypes trampoline: keep the first one, stop forwarding, re-raise below
        if not failure:
            try:
                callback(ctypes.string_at(ptr, nbytes))
            except BaseException as e:          # noqa: BLE001
                failure.append(e)
        return 0

    cb = _CALLBACK(tramp)
    c = _cfg(config)
    rc = lib.zz_encode_callback(src, len(src), ctypes.byref(c), ctypes.cast(cb, ctypes.c_void_p), None)
    if failure:
        raise failure[0]
    _check(rc)


def adler32x(start, data):  # adler.cpp:17-43
    b = bytes(data)
    return lib.zz_adler32(start, b, len(b))


def combine(first, second, len_second):  # adler.cpp:5-15
    return lib.zz_adler32_combine(first, second, len_second)


def crc32(data, start=0):  # crc.h:7
    b = bytes(data)
    return lib.zz_crc32(b, len(b), start)


def crc32_combine(crc1, crc2, len2):
    return lib.zz_crc32_combine(crc1, crc2, len2)


def header(format):
    buf = ctypes.create_string_buffer(10)
    n = lib.zz_header(int(format), buf)
    return buf.raw[:n]


def trailer(format, cks_total, n):
    buf = ctypes.create_string_buffer(8)
    k = lib.zz_trailer(int(format), cks_total, n, buf)
    return buf.raw[:k]


def generate_host(kind, seed, first_byte, n):
    buf = ctypes.create_string_buffer(max(n, 1))
    _check(lib.zz_generate_host(kind, seed, first_byte, buf, n))
    return buf.raw[:n]


GEN_TEXT, GEN_RANDOM, GEN_LOG, GEN_MIX = 0, 1, 2, 3


class Context:
    """Device context: workspace + timing. Buffers are torch tensors (uint8, on the context's device) or
    raw device pointers; PyTorch is only the allocator here."""

    def __init__(self, device=0):
        h = ctypes.c_void_p()
        _check(lib.zz_ctx_create(device, ctypes.byref(h)))
        self._h = h
        self.device = device

    def close(self):
        if self._h:
            lib.zz_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _ptr(t):
        return t if isinstance(t, int) else t.data_ptr()

    def _stream(self):
        try:
            import torch
            return torch.cuda.current_stream(self.device).cuda_stream     # this context's device, not torch's current one
        except Exception:
            return 0

    def set_warm_window(self, nbytes):
        """Levels >= 1: hash the last ``nbytes`` (0..32768) in front of every packet into its table before parsing it, so
        that matches may reach across packet boundaries (0 = cold packets = the reference's threaded stream)."""
        _check(lib.zz_ctx_set_warm_window(self._h, nbytes))

    def set_extended_levels(self, on=True):
        """Accept levels 4, 5, 6 (beyond the reference, which rejects them): hash chains of depth 2 / 4 / 8 over a window of
        8 / 32 / 32 KiB, one-step lazy matching, package-merge code lengths. Off by default, so that level > 3
        stays the reference's error."""
        _check(lib.zz_ctx_set_extended_levels(self._h, 1 if on else 0))

    def enable_timing(self, on=True):
        lib.zz_ctx_enable_timing(self._h, 1 if on else 0)

    def last_kernel_ms(self):
        return lib.zz_ctx_last_kernel_ms(self._h)

    def workspace_bytes(self):
        return lib.zz_ctx_workspace_bytes(self._h)

    def encode(self, src, n, dst, cap, format=Format.Zlib, level=1, packet_size=DEFAULT_PACKET, stream=None):
        """Whole stream on the device; returns the number of bytes written to ``dst``."""
        out = ctypes.c_uint64(0)
        st = self._stream() if stream is None else stream
        _check(lib.zz_encode_device(self._h, self._ptr(src), n, self._ptr(dst), cap, ctypes.byref(out), int(format),
                                    int(level), packet_size, st))
        return out.value

    def encode_async(self, src, n, dst, cap, format=Format.Zlib, level=1, packet_size=DEFAULT_PACKET, stream=None):
        """Enqueue ``encode`` on ``stream`` without waiting; ``finish()`` returns the byte count. One call per context at a
        time: use two contexts on two streams to keep two calls in flight."""
        st = self._stream() if stream is None else stream
        _check(lib.zz_encode_device_async(self._h, self._ptr(src), n, self._ptr(dst), cap, int(format), int(level), packet_size, st))

    def finish(self):
        out = ctypes.c_uint64(0)
        _check(lib.zz_encode_finish(self._h, ctypes.byref(out)))
        return out.value

    def encode_stream(self, src, n, dst, cap, format=Format.Zlib, level=1, stream=None):
        """The reference's sequential whole-buffer stream (threaded=false) into a caller-owned buffer of ``cap`` bytes
        (at level 1 the capacity decides the block lengths, encoder.cpp:331-337)."""
        out = ctypes.c_uint64(0)
        st = self._stream() if stream is None else stream
        _check(lib.zz_encode_stream_device(self._h, self._ptr(src), n, self._ptr(dst), cap, ctypes.byref(out), int(format),
                                           int(level), st))
        return out.value

    def encode_ranges(self, src, n, dst, cap, count, format=Format.Zlib, level=2, stream=None):
        """The reference's own threaded=true split for a machine with ``count`` hardware threads (zzflate.cpp:67-78,97-155):
        ``count`` ranges of ceil(n / count) bytes, one encoder (here: one wavefront) each. Levels 0, 2, 3."""
        out = ctypes.c_uint64(0)
        st = self._stream() if stream is None else stream
        _check(lib.zz_encode_ranges_device(self._h, self._ptr(src), n, self._ptr(dst), cap, ctypes.byref(out), int(format),
                                           int(level), int(count), st))
        return out.value

    def encode_stream_chunks(self, src, n, dst, cap, format=Format.Zlib, level=1, stream=None):
        """The sequential stream as ZzFlateEncodeToCallback produces it (library-owned 1,000,000-byte chunks decide the
        level-1 block lengths); returns (bytes written, [chunk sizes the reference's callback would see])."""
        out = ctypes.c_uint64(0)
        sizes = (ctypes.c_uint64 * 8192)()
        nch = ctypes.c_uint32(0)
        st = self._stream() if stream is None else stream
        _check(lib.zz_encode_stream_chunks_device(self._h, self._ptr(src), n, self._ptr(dst), cap, ctypes.byref(out), int(format),
                                                  int(level), sizes, 8192, ctypes.byref(nch), st))
        return out.value, list(sizes[: nch.value])

    def encode_shard(self, src, n, dst, cap, halo=0, is_last=True, checksum=Format.Zlib, level=1,
                     packet_size=DEFAULT_PACKET, stream=None):
        """One shard (contiguous packet range) of a stream; returns (bytes, checksum partial)."""
        out = ctypes.c_uint64(0)
        cks = ctypes.c_uint32(0)
        st = self._stream() if stream is None else stream
        _check(lib.zz_encode_shard_device(self._h, self._ptr(src), n, halo, 1 if is_last else 0, self._ptr(dst), cap,
                                          ctypes.byref(out), ctypes.byref(cks), int(checksum), int(level), packet_size, st))
        return out.value, cks.value

    def encode_shard_async(self, src, n, dst, cap, halo=0, is_last=True, checksum=Format.Zlib, level=1,
                           packet_size=DEFAULT_PACKET, stream=None):
        """Enqueue one shard and return; `finish_shard` waits for it. src, dst and the stream must stay alive."""
        st = self._stream() if stream is None else stream
        _check(lib.zz_encode_shard_device_async(self._h, self._ptr(src), n, halo, 1 if is_last else 0, self._ptr(dst), cap,
                                                int(checksum), int(level), packet_size, st))

    def finish_shard(self, checksum=Format.Zlib):
        out = ctypes.c_uint64(0)
        cks = ctypes.c_uint32(0)
        _check(lib.zz_encode_shard_finish(self._h, ctypes.byref(out), ctypes.byref(cks), int(checksum)))
        return out.value, cks.value

    def verify_last(self, stream=None):
        """Inflates every packet of the last encode / encode_shard call's output on the device and compares with its
        input (both tensors must still be alive): returns (bad packets, lowest bad packet or None)."""
        bad, first = ctypes.c_uint64(0), ctypes.c_uint64(0)
        st = self._stream() if stream is None else stream
        _check(lib.zz_verify_last_device(self._h, ctypes.byref(bad), ctypes.byref(first), st))
        return bad.value, (None if bad.value == 0 else first.value)

    def packet_extent(self, k, stream=None):
        """(offset behind the container header, bytes) of packet k in the stream the last encode / encode_shard call wrote."""
        off, nb = ctypes.c_uint64(0), ctypes.c_uint64(0)
        st = self._stream() if stream is None else stream
        _check(lib.zz_packet_extent_device(self._h, k, ctypes.byref(off), ctypes.byref(nb), st))
        return off.value, nb.value

    def generate(self, kind, seed, first_byte, buf, n, stream=None):
        st = self._stream() if stream is None else stream
        _check(lib.zz_generate_device(self._h, kind, seed, first_byte, self._ptr(buf), n, st))


def encode_multi(ctxs, srcs, ns, dst, cap, format=Format.Zlib, level=1, packet_size=DEFAULT_PACKET, halos=None):
    """zz_encode_multi_device: shard i = srcs[i][:ns[i]] on the device of ctxs[i]; the whole stream (header, shards in
    order, trailer) lands in `dst` on the device of ctxs[0]. One process, no torch.distributed. Returns the byte count."""
    k = len(ctxs)
    vp = ctypes.c_void_p
    cs = (vp * k)(*[c._h for c in ctxs])
    ps = (vp * k)(*[Context._ptr(t) for t in srcs])
    nn = (ctypes.c_uint64 * k)(*ns)
    hh = (ctypes.c_uint64 * k)(*(halos or [0] * k))
    out = ctypes.c_uint64(0)
    _check(lib.zz_encode_multi_device(cs, k, ps, nn, hh, Context._ptr(dst), cap, ctypes.byref(out), int(format), int(level), packet_size))
    return out.value
