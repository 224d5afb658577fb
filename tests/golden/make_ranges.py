"""Generates tests/golden/ranges.json from the compiled, unmodified reference (oracle/_ref/libzzref.so): the reference's
OWN threaded=true streams (zzflate.cpp:67-78,97-155: hardware_concurrency() ranges of ceil(n / count) bytes) for the
Canterbury files and three synthetic inputs, zlib container, levels 0, 2, 3 (level 1 threaded is invalid in the reference,
SURVEY.md App. B D2). The range count is a property of the machine the reference runs on, so it is recorded with the
hashes; consumers (the oracle test on CPU, the -m gpu test of zz_encode_ranges_device) pass that count.

Run in the build container (where /root/reference exists and `make -C oracle` has built _ref):
    python tests/golden/make_ranges.py
"""
import ctypes
import hashlib
import json
import os
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from conftest import Ref, CORPUS, CORPUS_FILES, ROOT, synth  # noqa: E402

SYNTH = [("words", 6 << 20, 3), ("runs", 1 << 20, 5), ("random", 300000, 7)]     # (kind, bytes, seed): ranges above 500,000 bytes = several blocks per range


def main():
    ref = Ref(os.path.join(ROOT, "oracle", "_ref", "libzzref.so"))
    ref.L.zzref_hardware_concurrency.restype = ctypes.c_uint32
    count = ref.L.zzref_hardware_concurrency()
    G = {"count": count, "files": {}, "synth": {}}
    def streams(d):
        e = {}
        for lvl in (0, 2, 3):
            o = ref.encode(d, 0, lvl, threaded=1)
            assert zlib.decompress(o) == d, lvl
            e[str(lvl)] = [len(o), hashlib.sha256(o).hexdigest()]
        return e
    for f in CORPUS_FILES:
        d = open(os.path.join(CORPUS, f), "rb").read()
        G["files"][f] = streams(d)
    for kind, n, seed in SYNTH:
        d = synth(kind, n, seed)
        G["synth"][f"{kind}.{n}.{seed}"] = {"sha256": hashlib.sha256(d).hexdigest(), "streams": streams(d)}
    json.dump(G, open(os.path.join(HERE, "ranges.json"), "w"), indent=1, sort_keys=True)
    print("wrote ranges.json: count =", count)


if __name__ == "__main__":
    main()
