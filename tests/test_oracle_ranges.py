"""The reference's own threaded=true split (zzflate.cpp:67-78,97-155) as the oracle restates it (zzo_encode_ranges), pinned by
the streams the unmodified reference produced in the build container (tests/golden/ranges.json, made by make_ranges.py from
oracle/_ref with hardware_concurrency() = its "count") and, where oracle/_ref is present, by the reference run live."""
import hashlib
import json
import os
import zlib

import pytest

from conftest import CORPUS, CORPUS_FILES, ROOT, synth

G = json.load(open(os.path.join(ROOT, "tests", "golden", "ranges.json")))


def cases():
    for f in CORPUS_FILES:
        yield f, lambda f=f: open(os.path.join(CORPUS, f), "rb").read(), G["files"][f]
    for key, e in G["synth"].items():
        kind, n, seed = key.split(".")
        yield key, lambda kind=kind, n=int(n), seed=int(seed): synth(kind, n, seed), e["streams"]


@pytest.mark.parametrize("name,load,want", list(cases()), ids=[c[0] for c in cases()])
def test_oracle_ranges_equal_the_reference_streams(oracle, name, load, want):
    d = load()
    for lvl in (0, 2, 3):
        o = oracle.encode_ranges(d, 0, lvl, G["count"])
        assert [len(o), hashlib.sha256(o).hexdigest()] == want[str(lvl)], (name, lvl)
        assert zlib.decompress(o) == d


def test_short_input_takes_the_single_encoder(oracle):
    # zzflate.cpp:84: fewer than 100 bytes per hardware thread -> no split
    d = synth("words", 799, 2)
    for lvl in (0, 2, 3):
        assert oracle.encode_ranges(d, 0, lvl, 8) == oracle.encode(d, 0, lvl)
    d = synth("words", 800, 2)
    assert oracle.encode_ranges(d, 0, 2, 8) != oracle.encode(d, 0, 2)


def test_reference_live_with_this_machines_count(oracle, ref):
    import ctypes
    ref.L.zzref_hardware_concurrency.restype = ctypes.c_uint32
    count = ref.L.zzref_hardware_concurrency()
    d = synth("words", 200000, 11)
    for lvl in (0, 2, 3):
        assert ref.encode(d, 0, lvl, threaded=1) == oracle.encode_ranges(d, 0, lvl, count), lvl


def test_counts_whose_ranges_would_start_past_the_input_are_refused(oracle):
    """divideInRanges (zzflate.cpp:67-78) puts boundary i at ceil(n / count) * i: with count near sqrt(n) or above, the trailing
    ranges start at or past the input's end (the reference reads out of bounds there). The restatement refuses
    exactly those -- (count - 1) * ceil(n / count) >= n -- and nothing else."""
    def splits(n, count):
        return (count - 1) * (-(-n // count)) < n
    assert not splits(20000, 192) and not splits(12801, 128)            # the advisor's examples
    for count in (128, 192, 256):
        bad = [n for n in range(100 * count, 100 * count + 400) if not splits(n, count)]
        good = [n for n in range(100 * count, 100 * count + 400) if splits(n, count)]
        assert bad and good
        for n in bad[:2] + bad[-1:]:
            d = synth("words", n, count)
            for lvl in (0, 2, 3):
                assert oracle.encode_ranges_raw(d, 0, lvl, count) is None, (count, n, lvl)
        for n in good[:1] + good[-1:]:
            d = synth("words", n, count)
            for lvl in (0, 2, 3):
                assert zlib.decompress(oracle.encode_ranges(d, 0, lvl, count)) == d, (count, n, lvl)
