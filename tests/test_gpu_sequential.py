"""The reference's single-Encoder stream (threaded=false, zzflate.cpp:84-95) through the drop-in entry points, in
both forms the reference gives it -- the forms its own callers use (zztest/Test.cpp:202-282):
  * ZzFlateEncode into a caller-owned buffer: at level 1 the block lengths follow from the buffer's capacity
    (encoder.cpp:331-337), and the callers pass dest = max(200, input size);
  * ZzFlateEncodeToCallback: library-owned 1,000,000-byte chunks (outputbitstream.h:171-201) decide the level-1 block
    lengths, and the callback sees exactly those chunks.
Expectations: the oracle (pinned to the compiled reference in test_oracle_vs_ref.py) and the committed goldens.
Needs a real MI355X: run with `-m gpu`."""
import hashlib
import json
import os
import zlib

import pytest

import zzflate_amd as zz
from conftest import GOLDEN, CORPUS_FILES, SYNTH_KINDS, synth

pytestmark = pytest.mark.gpu
G = json.load(open(os.path.join(GOLDEN, "golden.json")))
WBITS = {0: 15, 1: 31, 2: -15}
FORMATS = [zz.Format.Zlib, zz.Format.Gzip, zz.Format.Deflate]


def h(b):
    return [len(b), hashlib.sha256(b).hexdigest()]


h_ = h


def inflates(b, d, fmt):
    try:
        return zlib.decompressobj(WBITS[fmt]).decompress(b) == d
    except zlib.error:
        return False


@pytest.mark.parametrize("fname", CORPUS_FILES)
def test_level1_tight_destination_matches_reference(oracle, corpus, fname):
    """ZzFlateEncode(level 1, threaded=false, dest = max(200, n)) -- zztest/Test.cpp:206-212 (zlib) and :254-258 (gzip) --
    and a few smaller destinations: the oracle's bytes, the reference's goldens where the reference is valid."""
    d = corpus[fname]
    n = len(d)
    for fmt in (0, 1, 2):
        for permille in (1000, 950, 900, 800, 700):
            cap = max(200, n * permille // 1000)
            want = oracle.encode(d, fmt, 1, cap=cap)
            assert inflates(want, d, fmt)
            got = zz.ZzFlateEncode(d, zz.Config(FORMATS[fmt], 1, False), dest_capacity=cap)
            assert got == want, (fname, fmt, cap, len(got), len(want))
            g = G["files"][fname]["tight"].get(str(fmt), {}).get(str(permille))
            if g:
                assert h(got) == g, (fname, fmt, permille)


def test_level1_destination_too_small_is_reported(oracle, corpus):
    """Where the reference runs out of room it silently leaves a truncated stream (D9); the drop-in says so."""
    d = synth("random", 50000, 7)
    for cap in (200, 25000, 50000, 56000):            # random bytes grow by 5.5 % at level 1
        want = oracle.encode(d, 0, 1, cap=cap)
        if inflates(want, d, 0):
            assert zz.ZzFlateEncode(d, zz.Config(zz.Format.Zlib, 1, False), dest_capacity=cap) == want
        else:
            with pytest.raises(zz.ZzFlateError) as e:
                zz.ZzFlateEncode(d, zz.Config(zz.Format.Zlib, 1, False), dest_capacity=cap)
            assert e.value.code == -2
    # a roomy destination: one block, SURVEY App. D bytes
    a = corpus["alice29.txt"]
    assert h(zz.ZzFlateEncode(a, zz.Config(zz.Format.Zlib, 1, False), dest_capacity=2 * len(a) + 1024)) == \
        G["files"]["alice29.txt"]["whole"]["0"]["1"]


@pytest.mark.parametrize("fname", CORPUS_FILES)
def test_callback_form_matches_reference(oracle, corpus, fname):
    """ZzFlateEncodeToCallback(threaded=false) at every level: same bytes AND the same callback sequence (header, the
    encoder's chunks, trailer) as the reference; golden = the reference's own stream hash and callback count."""
    d = corpus[fname]
    for lvl in range(4):
        want, sizes = oracle.encode_callback(d, 0, lvl)
        chunks = []
        zz.ZzFlateEncodeToCallback(d, zz.Config(zz.Format.Zlib, lvl, False), chunks.append)
        got = b"".join(chunks)
        assert got == want, (fname, lvl, len(got), len(want))
        assert [len(c) for c in chunks] == sizes, (fname, lvl)
        g = G["files"][fname]["callback"].get(str(lvl))
        if g:
            assert h(got) + [len(chunks)] == g, (fname, lvl)
        assert inflates(got, d, 0)


def test_callback_form_on_longer_inputs(oracle, corpus):
    """Several chunks at every level: ~4 MiB of corpus files back to back and synthetic inputs. At level 1 every chunk
    boundary changes the block cut, at levels 0, 2, 3 only the callback sequence."""
    big = (corpus["kennedy.xls"] + corpus["lcet10.txt"] + corpus["ptt5"] + corpus["plrabn12.txt"]) * 2
    cases = [big, synth("words", 2500000, 3), synth("runs", 3000000, 4), synth("random", 1200000, 5)]
    checked = 0
    for d in cases:
        for lvl in range(4):
            for fmt in (0, 1):
                want, sizes = oracle.encode_callback(d, fmt, lvl)
                # Where the reference gives up mid-stream (a block that needs > 2^18 bytes of a nearly full chunk) there is
                # nothing to be bit-exact with. None of these 32 cases is one -- asserted, so that the comparison below cannot
                # silently stop covering them.
                assert inflates(want, d, fmt), (len(d), lvl, fmt)
                checked += 1
                chunks = []
                zz.ZzFlateEncodeToCallback(d, zz.Config(FORMATS[fmt], lvl, False), chunks.append)
                assert b"".join(chunks) == want, (len(d), lvl, fmt)
                assert [len(c) for c in chunks] == sizes, (len(d), lvl, fmt)
                assert all(len(c) <= 1000000 for c in chunks)
    assert checked == 32


def test_stream_chunks_device_entry_point(oracle, corpus):
    import torch
    ctx = zz.Context(0)
    d = corpus["kennedy.xls"]
    n = len(d)
    src = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
    cap = 2 * n + 1024
    dst = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    for lvl in range(4):
        w, sizes = ctx.encode_stream_chunks(src, n, dst, cap, zz.Format.Gzip, lvl)
        want, osizes = oracle.encode_callback(d, 1, lvl)
        assert dst[:w].cpu().numpy().tobytes() == want
        assert [10] + sizes + [8] == osizes


@pytest.mark.parametrize("kind", SYNTH_KINDS)
def test_synthetic_tight_destinations(oracle, kind):
    for n in (300, 5000, 40000, 200000):
        d = synth(kind, n, 11)
        for cap in (max(200, n), max(200, n * 9 // 10), max(200, n // 2), 2 * n + 1024):
            want = oracle.encode(d, 2, 1, cap=cap)
            if inflates(want, d, 2):
                got = zz.ZzFlateEncode(d, zz.Config(zz.Format.Deflate, 1, False), dest_capacity=cap)
                assert got == want, (kind, n, cap)
            else:
                with pytest.raises(zz.ZzFlateError):
                    zz.ZzFlateEncode(d, zz.Config(zz.Format.Deflate, 1, False), dest_capacity=cap)


def test_stream_level2_batch_start_inside_the_last_records_of_a_block(oracle):
    """Regression (found by tools/fuzz_gpu.py, present since round 1): at levels 2,3 the sequential stream's token pass
    works one position at a time once fewer than 64 records remain before the 20,000-record cut (so that nothing behind
    the cut enters the hash table); when a new 16,384-byte batch began inside that stretch, the batch's second byte was
    entered into the table by the "what the last match covers" sweep instead of being probed (encoder.cpp:383: j starts
    at startPos + 1). The input below has such a batch start 326 bytes before the end of its second block."""
    import torch
    d = zlib.decompress(open(os.path.join(GOLDEN, "cases", "stream_l2_batch_start_near_record_cut.bin.z"), "rb").read())
    ctx = zz.Context(0)
    src = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
    cap = 2 * len(d) + 1024
    dst = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    for lvl in (2, 3):
        w = ctx.encode_stream(src, len(d), dst, cap, zz.Format.Deflate, lvl)
        assert dst[:w].cpu().numpy().tobytes() == oracle.encode(d, 2, lvl), lvl


def test_stream_level2_one_position_at_a_time_everywhere(oracle, corpus, tmp_path):
    """The same code path, everywhere: a build with -DZZ_ST_ALWAYS_CAREFUL takes the one-position-at-a-time form of the
    token pass for whole streams; it must still be the reference's stream."""
    import ctypes
    import subprocess
    import torch
    # built by __graft_entry__.build() (zzflate_amd/build.py build_careful); only compiled here if that build is missing or stale
    lib = zz._build.build_careful()
    L = ctypes.CDLL(lib)
    u64, vp, ci = ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int
    h = vp()
    assert L.zz_ctx_create(0, ctypes.byref(h)) == 0
    for fname in ("alice29.txt", "kennedy.xls", "ptt5", "lcet10.txt"):
        d = corpus[fname]
        src = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
        cap = 2 * len(d) + 1024
        dst = torch.zeros(cap, dtype=torch.uint8, device="cuda")
        out = u64(0)
        rc = L.zz_encode_stream_device(h, vp(src.data_ptr()), u64(len(d)), vp(dst.data_ptr()), u64(cap), ctypes.byref(out), ci(0), ci(2), vp(0))
        assert rc == 0
        assert h_(dst[:out.value].cpu().numpy().tobytes()) == G["files"][fname]["whole"]["0"]["2"], fname
    L.zz_ctx_destroy(h)
