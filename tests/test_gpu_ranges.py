"""The reference's own threaded=true split (zzflate.cpp:67-78 divideInRanges, :97-155) on the device: zz_encode_ranges_device
and, through ZZFLATE_RANGES, the drop-in host entry points. Expectations: the streams the unmodified reference wrote with
threaded=true in the build container (tests/golden/ranges.json; its hardware_concurrency() is the recorded count) and the
oracle's restatement for other counts, containers and sizes. Levels 0, 2, 3 (level 1 threaded is invalid in the reference, D2).
Needs a real MI355X: run with `-m gpu`."""
import hashlib
import json
import os
import zlib

import pytest
import torch

import zzflate_amd as zz
from conftest import CORPUS_FILES, ROOT, synth

pytestmark = pytest.mark.gpu
G = json.load(open(os.path.join(ROOT, "tests", "golden", "ranges.json")))
WBITS = {0: 15, 1: 31, 2: -15}
FORMATS = [zz.Format.Zlib, zz.Format.Gzip, zz.Format.Deflate]


def run(ctx, d, fmt, lvl, count):
    src = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
    cap = 2 * len(d) + 4096 + 16 * count
    dst = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    w = ctx.encode_ranges(src, len(d), dst, cap, count, FORMATS[fmt], lvl)
    return dst[:w].cpu().numpy().tobytes()


@pytest.mark.parametrize("fname", CORPUS_FILES)
def test_corpus_equals_the_reference_threaded_streams(oracle, corpus, fname):
    ctx = zz.Context(0)
    d = corpus[fname]
    for lvl in (0, 2, 3):
        got = run(ctx, d, 0, lvl, G["count"])
        assert [len(got), hashlib.sha256(got).hexdigest()] == G["files"][fname][str(lvl)], (fname, lvl)
        assert zlib.decompress(got) == d


@pytest.mark.parametrize("key", list(G["synth"].keys()))
def test_synthetic_equals_the_reference_threaded_streams(key):
    # (ranges above 500,000 bytes: several dynamic blocks per range, the table carried across them)
    kind, n, seed = key.split(".")
    d = synth(kind, int(n), int(seed))
    assert hashlib.sha256(d).hexdigest() == G["synth"][key]["sha256"]
    ctx = zz.Context(0)
    for lvl in (0, 2, 3):
        got = run(ctx, d, 0, lvl, G["count"])
        assert [len(got), hashlib.sha256(got).hexdigest()] == G["synth"][key]["streams"][str(lvl)], (key, lvl)


def test_other_counts_containers_and_edges_equal_the_oracle(oracle):
    ctx = zz.Context(0)
    for kind, n in (("words", 100000), ("runs", 70001), ("period", 40000), ("random", 9000), ("zeros", 300000)):
        d = synth(kind, n, 4)
        for count in (1, 2, 3, 16, 61):
            for fmt in (0, 1, 2):
                for lvl in (0, 2, 3):
                    want = oracle.encode_ranges(d, fmt, lvl, count)
                    assert run(ctx, d, fmt, lvl, count) == want, (kind, n, count, fmt, lvl)
    # fewer than 100 bytes per range: the single encoder (zzflate.cpp:84)
    d = synth("words", 799, 2)
    for lvl in (0, 2, 3):
        assert run(ctx, d, 0, lvl, 8) == oracle.encode(d, 0, lvl)


def test_long_ranges_many_blocks_each(oracle):
    # 24 MiB in 5 ranges of 4.8 MiB: ten dynamic blocks per range (500,000 bytes / 20,000 records each), the table carried across
    # them, ranges 1..4 with the bytes of the range in front as room for backward extension
    d = synth("words", 12 << 20, 21) + synth("runs", 4 << 20, 22) + synth("words", 8 << 20, 23)
    ctx = zz.Context(0)
    for lvl in (2, 3):
        assert run(ctx, d, 1, lvl, 5) == oracle.encode_ranges(d, 1, lvl, 5), lvl


def test_level1_is_refused_with_the_reason():
    ctx = zz.Context(0)
    d = synth("words", 50000, 1)
    with pytest.raises(zz.ZzFlateError) as e:
        run(ctx, d, 0, 1, 8)
    assert "invalid" in str(e.value)


def test_host_entry_point_with_ZZFLATE_RANGES(monkeypatch, oracle, corpus):
    d = corpus["alice29.txt"]
    monkeypatch.setenv("ZZFLATE_RANGES", str(G["count"]))
    for lvl in (0, 2, 3):
        got = zz.ZzFlateEncode(d, zz.Config(zz.Format.Zlib, lvl, True))
        assert [len(got), hashlib.sha256(got).hexdigest()] == G["files"]["alice29.txt"][str(lvl)]
    # level 1 keeps packet mode (the reference's threaded level-1 stream does not inflate)
    got = zz.ZzFlateEncode(d, zz.Config(zz.Format.Zlib, 1, True))
    assert got == oracle.encode_packets(d, 0, 1)
    monkeypatch.delenv("ZZFLATE_RANGES")
    assert zz.ZzFlateEncode(d, zz.Config(zz.Format.Zlib, 2, True)) == oracle.encode_packets(d, 0, 2)


def test_counts_whose_ranges_would_start_past_the_input_are_refused(oracle):
    """(count - 1) * ceil(n / count) >= n: divideInRanges (zzflate.cpp:67-78) would cut ranges that start past the input's end (the
    reference reads out of bounds there, D10). The device entry point refuses them with ZZ_E_ARG and says why, before anything is
    launched; every other count up to 256 equals the oracle; the host entry point falls through to packet mode."""
    ctx = zz.Context(0)
    for count in (128, 192, 256):
        ns = range(100 * count, 100 * count + 400)
        bad = [n for n in ns if (count - 1) * (-(-n // count)) >= n]
        good = [n for n in ns if (count - 1) * (-(-n // count)) < n]
        for n in bad[:2] + bad[-1:]:
            d = synth("words", n, count)
            for lvl in (0, 2, 3):
                with pytest.raises(zz.ZzFlateError) as e:
                    run(ctx, d, 0, lvl, count)
                assert e.value.code == -4 and "count too large" in str(e.value), (count, n, lvl)
        for n in good[:1] + good[-1:]:
            d = synth("words", n, count)
            for lvl in (0, 2, 3):
                assert run(ctx, d, 0, lvl, count) == oracle.encode_ranges(d, 0, lvl, count), (count, n, lvl)


def test_host_entry_point_serves_what_the_split_cannot_in_packet_mode(monkeypatch, oracle):
    # ZZFLATE_RANGES with a count the split cannot take (or above 4096): the call is served in packet mode, not failed
    d = synth("words", 20000, 5)
    for count in ("192", "5000"):
        monkeypatch.setenv("ZZFLATE_RANGES", count)
        for lvl in (0, 2):
            assert zz.ZzFlateEncode(d, zz.Config(zz.Format.Zlib, lvl, True)) == oracle.encode_packets(d, 0, lvl), (count, lvl)
