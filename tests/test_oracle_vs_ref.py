"""Pins the oracle restatement (oracle/zzoracle.c) against the compiled, unmodified reference
(oracle/_ref/libzzref.so): identical bytes wherever the reference's output is a valid encoding of the
input, a valid encoding elsewhere. Skipped where the reference build is absent (the committed golden
vectors in test_oracle_golden.py cover that case)."""
import zlib

import pytest

from conftest import CORPUS_FILES, SYNTH_KINDS, synth

WBITS = {0: 15, 1: 31, 2: -15}


def valid(b, d, fmt=2, zdict=None):
    try:
        o = zlib.decompressobj(WBITS[fmt], zdict=zdict) if zdict else zlib.decompressobj(WBITS[fmt])
        return o.decompress(b) == d
    except zlib.error:
        return False


@pytest.mark.parametrize("fname", CORPUS_FILES)
def test_corpus_whole_and_packets(oracle, ref, corpus, fname):
    d = corpus[fname]
    for fmt in range(3):
        for lvl in range(4):
            assert oracle.encode(d, fmt, lvl) == ref.encode(d, fmt, lvl), (fname, fmt, lvl)
    P = 32768
    npk = (len(d) + P - 1) // P
    for lvl in range(4):
        for k in range(npk):
            off, ln = k * P, min(P, len(d) - k * P)
            a = oracle.packet(d, lvl, off, ln, k == npk - 1)
            assert any(a == ref.packet(d, lvl, off, ln, k == npk - 1, s) for s in (1, 2, 3)), (fname, lvl, k)


@pytest.mark.parametrize("kind", SYNTH_KINDS)
def test_synth(oracle, ref, kind):
    sizes = [1, 2, 3, 4, 5, 8, 9, 64, 258, 259, 263, 300, 1000, 4000, 16384, 16643, 32768, 32769, 40000, 70000]
    for n in sizes:
        d = synth(kind, n, 1)
        for lvl in range(4):
            a, b = oracle.encode(d, 2, lvl), ref.encode(d, 2, lvl)
            if a != b:   # only allowed where the reference's stream is invalid (App. B) and ours is valid
                assert valid(a, d) and not valid(b, d), (kind, n, lvl)
        for P in (32768, 1000):
            npk = (n + P - 1) // P
            for lvl in range(4):
                for k in range(npk):
                    off, ln = k * P, min(P, n - k * P)
                    zd = d[max(0, off - 32768):off] or None
                    a = oracle.packet(d, lvl, off, ln, k == npk - 1)
                    refs = [ref.packet(d, lvl, off, ln, k == npk - 1, s) for s in (1, 2, 3)]
                    if a not in refs:
                        assert valid(a, d[off:off + ln], 2, zd) and not any(valid(r, d[off:off + ln], 2, zd) for r in refs), (kind, n, P, lvl, k)


@pytest.mark.parametrize("fname", CORPUS_FILES)
def test_tight_destination_level1(oracle, ref, corpus, fname):
    """The level-1 block lengths follow from the destination's capacity (encoder.cpp:331-337); the reference's callers
    pass max(200, n) (zztest/Test.cpp:206). Identical bytes wherever the reference's multi-block stream is valid."""
    d = corpus[fname]
    n = len(d)
    same = 0
    for fmt in (0, 1, 2):
        for cap in (max(200, n), n * 95 // 100, n * 9 // 10, n * 85 // 100, n * 8 // 10, n * 3 // 4):
            a, b = oracle.encode(d, fmt, 1, cap=cap), ref.encode(d, fmt, 1, cap=cap)
            if a == b:
                same += 1
            else:
                assert valid(a, d, fmt) and not valid(b, d, fmt), (fname, fmt, cap)
    assert same >= 3      # the cut is crossed by a short match in a good third of the cases (D12)


def test_callback_api(oracle, ref, corpus):
    for fname in ("alice29.txt", "kennedy.xls", "sum"):
        d = corpus[fname]
        for lvl in range(4):
            (a, sizes), (b, nb) = oracle.encode_callback(d, 0, lvl), ref.encode_callback(d, 0, lvl)
            if a != b:
                assert valid(a, d, 0) and not valid(b, d, 0)
            else:
                assert len(sizes) == nb


def test_checksums(oracle, ref):
    import random
    rng = random.Random(5)
    for n in (0, 1, 3, 4, 5, 1000, 65521, 100000):
        d = bytes(rng.getrandbits(8) for _ in range(n))
        assert oracle.L.zzo_adler32(1, d, n) == ref.L.zzref_adler32x(1, d, n) == zlib.adler32(d)
        assert oracle.L.zzo_crc32(d, n, 0) == ref.L.zzref_crc32(d, n, 0) == zlib.crc32(d)
