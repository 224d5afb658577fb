import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
CORPUS = os.path.join(GOLDEN, "corpus")
CORPUS_FILES = sorted(os.listdir(CORPUS))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


u64, u32, ci, vp, cp = ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p


class Oracle:
    """ctypes face of oracle/libzzoracle.so (the C restatement). TEST INFRASTRUCTURE ONLY."""

    def __init__(self):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
        L = ctypes.CDLL(os.path.join(ROOT, "oracle", "libzzoracle.so"))
        L.zzo_encode.restype = u64; L.zzo_encode.argtypes = [vp, u64, cp, u64, ci, ci]
        L.zzo_encode_packets.restype = u64; L.zzo_encode_packets.argtypes = [vp, u64, cp, u64, ci, ci, u64]
        L.zzo_packet.restype = u64; L.zzo_packet.argtypes = [ci, cp, u64, u64, ci, vp, u64]
        L.zzo_encode_packets_warm.restype = u64; L.zzo_encode_packets_warm.argtypes = [vp, u64, cp, u64, ci, ci, u64, u64]
        L.zzo_packet_warm.restype = u64; L.zzo_packet_warm.argtypes = [ci, cp, u64, u64, ci, vp, u64, u64]
        L.zzo_encode_callback.restype = u64
        L.zzo_encode_callback.argtypes = [vp, u64, cp, u64, ci, ci, ctypes.POINTER(u64), ci, ctypes.POINTER(ci)]
        L.zzo_adler32.restype = u32; L.zzo_adler32.argtypes = [u32, cp, u64]
        L.zzo_adler_combine.restype = u32; L.zzo_adler_combine.argtypes = [u32, u32, u64]
        L.zzo_crc32.restype = u32; L.zzo_crc32.argtypes = [cp, u64, u32]
        L.zzo_crc32_combine.restype = u32; L.zzo_crc32_combine.argtypes = [u32, u32, u64]
        L.zzo_reverse.restype = u32; L.zzo_reverse.argtypes = [u32, ci]
        L.zzo_dist_bucket.restype = ci; L.zzo_dist_bucket.argtypes = [ci]
        L.zzo_bitstream.restype = u64
        L.zzo_encode_ranges.restype = u64; L.zzo_encode_ranges.argtypes = [vp, u64, cp, u64, ci, ci, u32]
        self.L = L

    def encode_ranges(self, d, fmt, lvl, count):
        """the reference's own threaded=true split for a machine with `count` hardware threads (zzflate.cpp:67-78,97-155)"""
        cap = 2 * len(d) + 4096
        b = ctypes.create_string_buffer(cap)
        n = self.L.zzo_encode_ranges(b, cap, d, len(d), fmt, lvl, count)
        assert n != 0xFFFFFFFFFFFFFFFF
        return b.raw[:n]

    def encode_ranges_raw(self, d, fmt, lvl, count):
        """the same, None where the restatement refuses the split (ranges that would start past the input's end, D10)"""
        cap = 2 * len(d) + 4096 + 16 * count
        b = ctypes.create_string_buffer(cap)
        n = self.L.zzo_encode_ranges(b, cap, d, len(d), fmt, lvl, count)
        return None if n == 0xFFFFFFFFFFFFFFFF else b.raw[:n]

    def encode(self, d, fmt, lvl, cap=None):
        """zzo_encode into a destination of `cap` bytes (default: roomy). At level 1 the capacity decides the block
        lengths (encoder.cpp:331-337); like the reference, the restatement silently leaves a truncated stream when the
        room runs out (D9) -- callers check with inflate before using such a result as the expectation."""
        cap = 2 * len(d) + 1024 if cap is None else cap
        b = ctypes.create_string_buffer(max(cap, 16) + 64)      # the trailer append is unchecked in the reference
        n = self.L.zzo_encode(b, cap, d, len(d), fmt, lvl)
        return None if n == u64(-1).value else b.raw[:n]

    def encode_packets(self, d, fmt, lvl, P=32768, warm=0):
        cap = 2 * len(d) + 1024 + 16 * (len(d) // P + 1)
        b = ctypes.create_string_buffer(cap)
        n = self.L.zzo_encode_packets_warm(b, cap, d, len(d), fmt, lvl, P, warm)
        return None if n == u64(-1).value else b.raw[:n]

    def packet(self, d, lvl, off, ln, final):
        cap = 2 * ln + 1024
        b = ctypes.create_string_buffer(cap)
        n = self.L.zzo_packet(lvl, d, off, ln, int(final), b, cap)
        return b.raw[:n]

    def encode_callback(self, d, fmt, lvl):
        cap = 2 * len(d) + 4096
        b = ctypes.create_string_buffer(cap)
        sizes = (u64 * 4096)()
        nc = ci(0)
        n = self.L.zzo_encode_callback(b, cap, d, len(d), fmt, lvl, sizes, 4096, ctypes.byref(nc))
        return b.raw[:n], list(sizes[: nc.value])


class Ref:
    """ctypes face of oracle/_ref/libzzref.so (the compiled, unmodified reference). TEST INFRASTRUCTURE ONLY."""

    def __init__(self, path):
        L = ctypes.CDLL(path)
        L.zzref_encode.restype = u64; L.zzref_encode.argtypes = [vp, u64, cp, u64, ci, ci, ci, u32]
        L.zzref_packet.restype = u64; L.zzref_packet.argtypes = [ci, cp, u64, u64, ci, vp, u64, u32]
        L.zzref_encode_callback.restype = u64
        L.zzref_encode_callback.argtypes = [vp, u64, cp, u64, ci, ci, ci, ctypes.POINTER(ci), u32]
        L.zzref_adler32x.restype = u32; L.zzref_adler32x.argtypes = [u32, cp, u64]
        L.zzref_combine.restype = u32; L.zzref_combine.argtypes = [u32, u32, u64]
        L.zzref_crc32.restype = u32; L.zzref_crc32.argtypes = [cp, u64, u32]
        L.zzref_reverse.restype = u32; L.zzref_reverse.argtypes = [u32, ci]
        L.zzref_bitstream.restype = u64
        self.L = L

    def encode(self, d, fmt, lvl, threaded=0, seed=1, cap=None):
        cap = 2 * len(d) + 1024 if cap is None else cap
        b = ctypes.create_string_buffer(max(cap, 16) + 64)      # the reference appends the trailer unchecked
        n = self.L.zzref_encode(b, cap, d, len(d), fmt, lvl, threaded, seed)
        return None if n == u64(-1).value else b.raw[:n]

    def packet(self, d, lvl, off, ln, final, seed=1):
        cap = 2 * ln + 1024
        b = ctypes.create_string_buffer(cap)
        n = self.L.zzref_packet(lvl, d, off, ln, int(final), b, cap, seed)
        return b.raw[:n]

    def encode_callback(self, d, fmt, lvl, seed=1):
        cap = 2 * len(d) + 4096
        b = ctypes.create_string_buffer(cap)
        nc = ci(0)
        n = self.L.zzref_encode_callback(b, cap, d, len(d), fmt, lvl, 0, ctypes.byref(nc), seed)
        return b.raw[:n], nc.value


@pytest.fixture(scope="session")
def oracle():
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    path = os.path.join(ROOT, "oracle", "_ref", "libzzref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref not built (reference checkout absent)")
    return Ref(path)


@pytest.fixture(scope="session")
def corpus():
    return {f: open(os.path.join(CORPUS, f), "rb").read() for f in CORPUS_FILES}


def synth(kind, n, seed):
    """small deterministic inputs that stress different parts of the encoder"""
    import random
    rng = random.Random(seed * 1000003 + sum(kind.encode()) * 7919 + n)
    if kind == "random":
        return bytes(rng.getrandbits(8) for _ in range(n))
    if kind == "zeros":
        return bytes(n)
    if kind == "words":
        words = [bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(1, 9))) for _ in range(50)]
        out = bytearray()
        while len(out) < n:
            out += rng.choice(words) + b" "
        return bytes(out[:n])
    if kind == "ab":
        return bytes(rng.choice(b"ab") for _ in range(n))
    if kind == "runs":
        out = bytearray()
        while len(out) < n:
            out += bytes([rng.getrandbits(8)]) * rng.randint(1, 600)
        return bytes(out[:n])
    if kind == "period":   # period <= 250: stays clear of reference defect D11
        base = bytes(rng.getrandbits(8) for _ in range(rng.randint(1, 250)))
        return (base * (n // len(base) + 1))[:n]
    if kind == "longperiod":   # period >= 259: triggers D11 in the reference (never fed to it)
        base = bytes(rng.getrandbits(8) for _ in range(rng.randint(259, 700)))
        return (base * (n // len(base) + 1))[:n]
    raise ValueError(kind)


SYNTH_KINDS = ["random", "zeros", "words", "ab", "runs", "period"]
EDGE_SIZES = [1, 2, 3, 4, 5, 7, 8, 9, 63, 64, 65, 255, 256, 257, 258, 259, 260, 261, 262, 263, 264, 265, 266, 300,
              511, 513, 1000, 4000, 16383, 16384, 16385, 16642, 16643, 20000, 32767, 32768, 32769, 33000, 40000,
              65535, 65536, 65537, 70000, 100000]
