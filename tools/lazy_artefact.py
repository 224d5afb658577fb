"""CPU measurement (no GPU): what the extended levels' `(q & 63) == 63` exception -- the last lane of a 64-position group never
defers to its successor, DESIGN.md 7 -- costs in compression ratio. Builds the oracle twice (as shipped, and with
-DZZO_LAZY_EVERYWHERE), encodes the corpus and synthetic inputs in packet mode at levels 4..6 and prints both sizes.
    python3 tools/lazy_artefact.py"""
import ctypes, os, subprocess, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import CORPUS, CORPUS_FILES, synth
os.makedirs("/tmp/lazy", exist_ok=True)
libs = {}
for name, flag in (("shipped", []), ("everywhere", ["-DZZO_LAZY_EVERYWHERE"])):
    so = f"/tmp/lazy/libzzo_{name}.so"
    subprocess.run(["cc", "-std=c11", "-O2", "-fPIC", "-shared", "-pthread", *flag, "-o", so, os.path.join(ROOT, "oracle", "zzoracle.c")], check=True)
    L = ctypes.CDLL(so)
    L.zzo_encode_packets_warm.restype = ctypes.c_uint64
    L.zzo_encode_packets_warm.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64]
    libs[name] = L
inputs = [(f, open(os.path.join(CORPUS, f), "rb").read()) for f in CORPUS_FILES]
inputs += [(f"{k}.{n}", synth(k, n, 9)) for k, n in (("words", 1 << 20), ("runs", 1 << 20), ("period", 1 << 18))]
tot = {}
for lvl in (4, 5, 6):
    for name, L in libs.items():
        s = 0
        for f, d in inputs:
            cap = 2 * len(d) + 4096
            b = ctypes.create_string_buffer(cap)
            w = L.zzo_encode_packets_warm(b, cap, d, len(d), 2, lvl, 32768, 0)
            assert zlib.decompressobj(-15).decompress(b.raw[:w]) == d
            s += w
        tot[(lvl, name)] = s
    n = sum(len(d) for _, d in inputs)
    a, b_ = tot[(lvl, "shipped")], tot[(lvl, "everywhere")]
    print(f"level {lvl}: {n} bytes in; as defined {a} ({a / n:.5f}); deferring at lane 63 too {b_} ({b_ / n:.5f}); difference {a - b_} bytes = {100.0 * (a - b_) / a:.4f} %")
