"""Writes tests/golden/extended.json: sizes and SHA-256 of the ORACLE's own extended-level (4, 5, 6) packet streams of the
corpus files. These are not reference outputs -- the reference has no such levels (zzflate.cpp:201,230-234) -- they pin the
definition in oracle/zzoracle.c ("Extended levels") against accidental change. Run from the repo root:
    python tests/golden/make_extended.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from conftest import Oracle, CORPUS_FILES, CORPUS  # noqa: E402

o = Oracle()
out = {"what": "oracle/zzoracle.c extended levels 4..6, zlib container, packet sizes 32768 and 4096: [bytes, sha256]", "files": {}}
for f in CORPUS_FILES:
    d = open(os.path.join(CORPUS, f), "rb").read()
    out["files"][f] = {}
    for lvl in (4, 5, 6):
        for P in (32768, 4096):
            s = o.encode_packets(d, 0, lvl, P)
            out["files"][f][f"{lvl}/{P}"] = [len(s), hashlib.sha256(s).hexdigest()]
json.dump(out, open(os.path.join(HERE, "extended.json"), "w"), indent=1, sort_keys=True)
print("wrote extended.json")
