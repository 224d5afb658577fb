"""Parity of the HIP path (through the C ABI) with the oracle, the committed golden vectors and inflate.
All tests here need a real MI355X: run with `-m gpu`."""
import hashlib
import json
import os
import zlib

import pytest

import zzflate_amd as zz
from conftest import GOLDEN, CORPUS_FILES, SYNTH_KINDS, EDGE_SIZES, synth, ROOT as ROOT_DIR

pytestmark = pytest.mark.gpu
G = json.load(open(os.path.join(GOLDEN, "golden.json")))
WBITS = {0: 15, 1: 31, 2: -15}
LEVELS = [0, 1, 2, 3]


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available()
    ctx = zz.Context(0)

    class Dev:
        def encode(self, data, fmt, lvl, P=32768):
            n = len(data)
            src = torch.frombuffer(bytearray(data) if n else bytearray(1), dtype=torch.uint8).cuda()
            cap = zz.bound(n, fmt, lvl, P)
            dst = torch.zeros(cap, dtype=torch.uint8, device="cuda")
            w = ctx.encode(src, n, dst, cap, fmt, lvl, P)
            return bytes(dst[:w].cpu().numpy().tobytes())

        def shard(self, data, off, n, halo, last, checksum, lvl, P=32768):
            buf = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
            cap = zz.bound(n, 2, lvl, P)
            dst = torch.zeros(cap, dtype=torch.uint8, device="cuda")
            w, cks = ctx.encode_shard(buf.data_ptr() + off, n, dst, cap, halo, last, checksum, lvl, P)
            assert ctx.verify_last() == (0, None)      # the device decoder follows distances into the halo
            return bytes(dst[:w].cpu().numpy().tobytes()), cks
        def stream(self, data, fmt, lvl):
            n = len(data)
            src = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
            cap = 2 * n + 1024
            dst = torch.zeros(cap, dtype=torch.uint8, device="cuda")
            w = ctx.encode_stream(src, n, dst, cap, fmt, lvl)
            return bytes(dst[:w].cpu().numpy().tobytes())
    d = Dev()
    d.ctx = ctx
    d.torch = torch
    return d


def h(b):
    return [len(b), hashlib.sha256(b).hexdigest()]


@pytest.mark.parametrize("fname", CORPUS_FILES)
@pytest.mark.parametrize("lvl", LEVELS)
def test_corpus_matches_oracle_and_golden(gpu, oracle, corpus, fname, lvl):
    d = corpus[fname]
    for fmt in range(3):
        got = gpu.encode(d, fmt, lvl)
        want = oracle.encode_packets(d, fmt, lvl)
        assert got == want, (fname, fmt, lvl, len(got), len(want))
        assert zlib.decompressobj(WBITS[fmt]).decompress(got) == d
    for P in (32768, 4096):
        g = G["files"][fname]["packets"][str(P)].get(str(lvl))
        if g:
            assert h(gpu.encode(d, 2, lvl, P)) == g, (fname, P, lvl)


@pytest.mark.parametrize("lvl", LEVELS)
def test_single_packet_equals_reference_whole_stream(gpu, corpus, lvl):
    """Inputs of at most one packet: packet mode == the reference's own ZzFlateEncode stream (goldens)."""
    for fname in ("grammar.lsp", "xargs.1", "fields.c", "cp.html"):
        for fmt in range(3):
            assert h(gpu.encode(corpus[fname], fmt, lvl)) == G["files"][fname]["whole"][str(fmt)][str(lvl)]
    for name, e in G["tiny"].items():
        d = bytes.fromhex(e["input_hex"])
        for key, want in e["whole"].items():
            fmt, l = map(int, key.split("."))
            if l == lvl:
                assert gpu.encode(d, fmt, lvl).hex() == want, (name, key)


@pytest.mark.parametrize("kind", SYNTH_KINDS + ["longperiod"])
@pytest.mark.parametrize("lvl", LEVELS)
def test_synthetic_edge_sizes(gpu, oracle, kind, lvl):
    for n in EDGE_SIZES:
        d = synth(kind, n, 1)
        got = gpu.encode(d, 0, lvl)
        assert got == oracle.encode_packets(d, 0, lvl), (kind, n, lvl)
        assert zlib.decompress(got) == d
    for P in (1000, 1024, 2048, 4096, 32767):
        d = synth(kind, 70000, 2)
        assert gpu.encode(d, 1, lvl, P) == oracle.encode_packets(d, 1, lvl, P), (kind, P, lvl)


@pytest.mark.parametrize("lvl", LEVELS)
def test_synth_goldens(gpu, lvl):
    for key, e in G["synth"].items():
        kind, n = key.split(".")
        want = e["packets"].get(str(lvl))
        if want:
            assert h(gpu.encode(synth(kind, int(n), 1), 2, lvl)) == want, key


@pytest.mark.parametrize("lvl", LEVELS)
def test_sequential_stream_matches_reference_goldens(gpu, oracle, corpus, lvl):
    """threaded=false on more than one packet: the reference's whole-buffer stream (ZzFlateEncode with an ample
    destination), on the device. Goldens are the reference's own outputs."""
    for fname in CORPUS_FILES:
        d = corpus[fname]
        for fmt in range(3):
            got = gpu.stream(d, fmt, lvl)
            assert h(got) == G["files"][fname]["whole"][str(fmt)][str(lvl)], (fname, fmt, lvl)
    for kind in SYNTH_KINDS:
        for n in (40000, 70000, 200000):
            d = synth(kind, n, 3)
            assert gpu.stream(d, 0, lvl) == oracle.encode(d, 0, lvl), (kind, n, lvl)
    # through the host entry point with threaded=false
    d = corpus["alice29.txt"]
    assert zz.ZzFlateEncode(d, zz.Config(zz.Format.Gzip, lvl, False)) == oracle.encode(d, 1, lvl)


def test_randomized_inputs_and_packet_sizes(gpu, oracle):
    """Seeded fuzz: structured random inputs, random lengths and packet sizes, every level and container."""
    import random
    rng = random.Random(20261003)
    kinds = SYNTH_KINDS + ["longperiod"]
    for it in range(120):
        kind = rng.choice(kinds)
        n = rng.choice([rng.randint(1, 300), rng.randint(300, 40000), rng.randint(40000, 200000)])
        d = synth(kind, n, 100 + it)
        if it % 3 == 0:   # splice two kinds: long repeats meeting noise
            e = synth(rng.choice(kinds), n, 500 + it)
            cut = rng.randint(0, n)
            d = d[:cut] + e[cut:]
        P = rng.choice([32768, 32768, 32768, 16384, 8192, 4096, 1000, 777, rng.randint(1, 32768)])
        lvl = rng.randint(0, 3)
        fmt = rng.randint(0, 2)
        got = gpu.encode(d, fmt, lvl, P)
        assert got == oracle.encode_packets(d, fmt, lvl, P), (it, kind, n, P, lvl, fmt)
        assert zlib.decompressobj(WBITS[fmt]).decompress(got) == d


def test_empty_input_is_a_valid_stream(gpu):
    """D8: the reference emits no block for empty input (invalid stream); we emit one empty final block."""
    for fmt in range(3):
        for lvl in LEVELS:
            o = gpu.encode(b"", fmt, lvl)
            assert zlib.decompressobj(WBITS[fmt]).decompress(o) == b""


def test_error_convention(gpu):
    torch = gpu.torch
    src = torch.zeros(100000, dtype=torch.uint8, device="cuda")
    dst = torch.zeros(200000, dtype=torch.uint8, device="cuda")
    with pytest.raises(zz.ZzFlateError) as e:
        gpu.ctx.encode(src, 100000, dst, 200000, 0, 4)          # level > 3: zzflate.cpp:230
    assert e.value.code == -1
    with pytest.raises(zz.ZzFlateError) as e:
        gpu.ctx.encode(src, 100000, dst, 1, 0, 1)               # no room for the header: zzflate.cpp:229
    assert e.value.code == -2
    src.random_(0, 256)
    with pytest.raises(zz.ZzFlateError) as e:
        gpu.ctx.encode(src, 100000, dst, 50000, 0, 1)           # too small for the stream (D9: detected)
    assert e.value.code == -2


def test_host_entry_points(gpu, oracle, corpus):
    """zz_encode / zz_encode_callback / the C++-named entry points on host buffers."""
    d = corpus["alice29.txt"]
    for lvl in LEVELS:
        cfg = zz.Config(zz.Format.Zlib, lvl, True)
        o = zz.ZzFlateEncode(d, cfg)
        assert o == oracle.encode_packets(d, 0, lvl)
        chunks = []
        zz.ZzFlateEncodeToCallback(d, cfg, chunks.append)
        assert b"".join(chunks) == o and len(chunks[0]) == 2 and len(chunks[-1]) == 4
    small = corpus["grammar.lsp"]
    assert zz.ZzFlateEncode(small, zz.Config(zz.Format.Gzip, 1, False)) == oracle.encode(small, 1, 1)
    assert zz.ZzFlateEncode(d, zz.Config(zz.Format.Zlib, 3, False)) == oracle.encode(d, 0, 3)   # sequential stream
    with pytest.raises(zz.ZzFlateError):
        zz.ZzFlateEncode(d, zz.Config(zz.Format.Zlib, 1, True), dest_capacity=100)


def test_host_entry_points_from_several_threads(gpu, oracle, corpus):
    """The reference is re-entrant for concurrent calls on distinct buffers: so are the drop-in entry
    points (they serialise on the default context)."""
    import threading
    names = ["alice29.txt", "lcet10.txt", "kennedy.xls", "ptt5", "plrabn12.txt", "asyoulik.txt"]
    want = {(nm, lvl): oracle.encode_packets(corpus[nm], 0, lvl) for nm in names for lvl in (1, 2)}
    bad = []

    def work(i):
        for rep in range(3):
            nm = names[(i + rep) % len(names)]
            lvl = 1 + (i + rep) % 2
            if zz.ZzFlateEncode(corpus[nm], zz.Config(zz.Format.Zlib, lvl, True)) != want[(nm, lvl)]:
                bad.append((i, nm, lvl))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(6)]
    for t in ts: t.start()
    for t in ts: t.join()
    assert not bad


@pytest.mark.parametrize("lvl", LEVELS)
def test_device_inflate_self_check(gpu, corpus, lvl):
    """zz_verify_last_device: every packet of a stream inflates, on the device, to its input; a
    damaged packet is found. Independent of the oracle: a plain RFC 1951 decoder, one lane per packet."""
    import torch
    for name, P in (("alice29.txt", 32768), ("kennedy.xls", 4096), ("ptt5", 32768), ("fields.c", 1000)):
        d = corpus[name]
        n = len(d)
        src = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
        cap = zz.bound(n, 0, lvl, P)
        dst = torch.zeros(cap, dtype=torch.uint8, device="cuda")
        w = gpu.ctx.encode(src, n, dst, cap, 0, lvl, P)
        assert gpu.ctx.verify_last() == (0, None), (name, lvl)
        # damage one byte in the middle of the stream: some packet must fail, and it must be near that byte
        pos = 2 + (w - 6) // 2
        dst[pos] ^= 0x5A
        bad, first = gpu.ctx.verify_last()
        assert bad >= 1 and first is not None, (name, lvl)
        dst[pos] ^= 0x5A
        assert gpu.ctx.verify_last() == (0, None)
    # the data generators at a size host zlib would need seconds for
    n = 64 << 20
    for kind in (zz.GEN_TEXT, zz.GEN_MIX, zz.GEN_RANDOM):
        src = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
        gpu.ctx.generate(kind, 0x5EED0002, 0, src, n)
        cap = zz.bound(n, 1, lvl, 32768)
        dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
        gpu.ctx.encode(src, n, dst, cap, 1, lvl)
        assert gpu.ctx.verify_last() == (0, None), (kind, lvl)


@pytest.mark.parametrize("kind,gib", [("text", 1), ("mix", 2), ("log", 1), ("random", 2)])
def test_full_size_round_trip_on_device(gpu, kind, gib):
    """BASELINE.json sizes (configs[1]: 1 GiB and beyond) at every level: every packet inflates on the device to its
    input, and sizes/checksum trailer are consistent with the per-level invariants. Host zlib could not do this in
    test time; sampled packets of the same generators are compared with the oracle in the 256 MiB test above."""
    import torch
    n = gib << 30
    K = {"text": zz.GEN_TEXT, "mix": zz.GEN_MIX, "log": zz.GEN_LOG, "random": zz.GEN_RANDOM}[kind]
    src = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    gpu.ctx.generate(K, 0x5EED0002, 0, src, n)
    dst = torch.empty(zz.bound(n, 1, 1, 32768), dtype=torch.uint8, device="cuda")
    sizes = {}
    for lvl in LEVELS:
        cap = zz.bound(n, 1, lvl, 32768)
        w = gpu.ctx.encode(src, n, dst, cap, 1, lvl)                    # gzip: CRC-32 + length trailer
        assert gpu.ctx.verify_last() == (0, None), (kind, lvl)
        tail = dst[w - 8:w].cpu().numpy().tobytes()
        assert int.from_bytes(tail[4:], "little") == n & 0xFFFFFFFF
        sizes[lvl] = w
    assert sizes[2] == sizes[3]                                         # one code path in the reference (encoder.cpp:506-527)
    assert sizes[0] == 10 + 8 + n + 10 * (n // 32768) - 5               # stored: 10 bytes per packet, 5 for the last
    if kind != "random":
        assert sizes[2] < sizes[1] < sizes[0]


@pytest.mark.parametrize("fmt", [0, 1, 2])
def test_host_slab_pipeline(gpu, oracle, corpus, fmt, monkeypatch):
    """Host buffers longer than one slab take the pipelined path (H2D -> encode -> D2H in slabs of
    whole packets): same bytes as one call, same 1,000,000-byte callback chunking as the reference."""
    monkeypatch.setenv("ZZFLATE_SLAB_MIB", "1")
    d = (corpus["lcet10.txt"] + corpus["ptt5"] + corpus["kennedy.xls"]) * 2 + b"tail"      # ~3.9 MiB: four slabs, ragged end
    F = [zz.Format.Zlib, zz.Format.Gzip, zz.Format.Deflate][fmt]
    for lvl in LEVELS:
        cfg = zz.Config(F, lvl, True)
        want = oracle.encode_packets(d, fmt, lvl)
        assert zz.ZzFlateEncode(d, cfg) == want, (fmt, lvl)
        chunks = []
        zz.ZzFlateEncodeToCallback(d, cfg, chunks.append)
        assert b"".join(chunks) == want
        hl, tl = {0: (2, 4), 1: (10, 8), 2: (0, 0)}[fmt]
        assert len(chunks[0]) == hl and len(chunks[-1]) == tl
        body = chunks[1:-1]
        assert all(len(c) == 1000000 for c in body[:-1]) and 0 < len(body[-1]) <= 1000000   # outputbitstream.h:183
    with pytest.raises(zz.ZzFlateError):
        zz.ZzFlateEncode(d, zz.Config(F, 1, True), dest_capacity=len(d) // 4)


def test_host_input_larger_than_the_staging_ring(gpu, oracle, corpus, monkeypatch):
    """the host entry points keep two slabs of input (plus 4 KiB in front of each, for level >= 2's
    backward match extension) and two of output on the device, whatever the input's size -- so inputs larger than HBM
    work. 24 MiB through 1 MiB slabs: same bytes as one call / the oracle, staging far smaller than the input."""
    monkeypatch.setenv("ZZFLATE_SLAB_MIB", "1")
    monkeypatch.setenv("ZZFLATE_DEVICES", "0")
    zz.lib.zz_debug_reset_devices()
    try:
        text = zz.generate_host(zz.GEN_TEXT, 0x5EED0002, 0, 6 << 20)
        d = text + corpus["kennedy.xls"] * 3 + zz.generate_host(zz.GEN_LOG, 0x5EED0005, 0, 5 << 20) + corpus["ptt5"] * 4 + \
            zz.generate_host(zz.GEN_MIX, 0x5EED0004, 0, 7 << 20) + b"ragged tail"
        assert len(d) > 22 << 20
        for lvl in LEVELS:
            want = oracle.encode_packets(d, 1, lvl)
            got = zz.ZzFlateEncode(d, zz.Config(zz.Format.Gzip, lvl, True))
            assert got == want, lvl
        assert zz.lib.zz_debug_host_staging_bytes() < 12 << 20        # two 1 MiB slabs each way (+ the simple path's buffers)
    finally:
        monkeypatch.undo()
        zz.lib.zz_debug_reset_devices()


@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
def test_multi_device_fan_out(gpu, oracle, corpus, monkeypatch, devices):
    """The device analogue of the reference's std::async fan-out (zzflate.cpp:97-155) behind ZzFlateEncode: slabs go
    round-robin to the devices of ZZFLATE_DEVICES (here the same GPU several times: several pipelines, several
    contexts, several host threads), the calling thread joins them in order. Same bytes as a single device."""
    monkeypatch.setenv("ZZFLATE_SLAB_MIB", "1")
    monkeypatch.setenv("ZZFLATE_DEVICES", devices)
    zz.lib.zz_debug_reset_devices()
    try:
        d = (corpus["lcet10.txt"] + corpus["ptt5"] + corpus["kennedy.xls"] + corpus["plrabn12.txt"]) * 4 + b"end"   # ~9.7 MiB: ten slabs
        for lvl in LEVELS:
            for fmt in (0, 1, 2):
                F = [zz.Format.Zlib, zz.Format.Gzip, zz.Format.Deflate][fmt]
                want = oracle.encode_packets(d, fmt, lvl)
                assert zz.ZzFlateEncode(d, zz.Config(F, lvl, True)) == want, (devices, lvl, fmt)
            chunks = []
            zz.ZzFlateEncodeToCallback(d, zz.Config(zz.Format.Zlib, lvl, True), chunks.append)
            assert b"".join(chunks) == oracle.encode_packets(d, 0, lvl)
        with pytest.raises(zz.ZzFlateError):
            zz.ZzFlateEncode(d, zz.Config(zz.Format.Zlib, 1, True), dest_capacity=len(d) // 4)
    finally:
        monkeypatch.undo()
        zz.lib.zz_debug_reset_devices()


def test_callback_may_call_back_into_the_library(gpu, oracle, corpus):
    """No lock is held while callbacks run: a callback that encodes something itself must not deadlock, and an
    exception raised by the callback reaches the caller."""
    d = corpus["alice29.txt"]
    inner = []

    def cb(chunk):
        if not inner:
            inner.append(zz.ZzFlateEncode(corpus["fields.c"], zz.Config(zz.Format.Zlib, 2, True)))
    zz.ZzFlateEncodeToCallback(d, zz.Config(zz.Format.Zlib, 1, True), cb)
    assert inner[0] == oracle.encode_packets(corpus["fields.c"], 0, 2)

    class Boom(Exception):
        pass

    def bad(chunk):
        raise Boom()
    with pytest.raises(Boom):
        zz.ZzFlateEncodeToCallback(d, zz.Config(zz.Format.Zlib, 1, True), bad)


@pytest.mark.parametrize("lvl", [0, 1, 2, 3])
def test_multi_device_entry_point_equals_the_single_call(gpu, oracle, corpus, lvl):
    """zz_encode_multi_device (the reference's fan-out + in-order join, zzflate.cpp:97-155, over devices of ONE process):
    three shards on three contexts -- all of device 0 here, which still runs the concurrent encodes, the peer copies into
    the final offsets and the host-side checksum fold -- must give the single-call stream and the oracle's."""
    torch = gpu.torch
    d = (corpus["lcet10.txt"] + corpus["kennedy.xls"][:300000] + corpus["alice29.txt"])[:720000]
    P = 4096 if lvl == 3 else 32768
    cuts = [0, 8 * 32768, 15 * 32768, len(d)]
    ctxs = [zz.Context(0) for _ in range(3)]
    halo = 65536
    srcs, keep, ns, halos = [], [], [], []
    for i in range(3):
        lo, hi = cuts[i], cuts[i + 1]
        h = min(halo, lo)
        t = torch.frombuffer(bytearray(d[lo - h:hi]), dtype=torch.uint8).cuda()
        keep.append(t)
        srcs.append(t[h:])
        ns.append(hi - lo)
        halos.append(h)
    for fmt in (0, 1, 2):
        cap = zz.bound(len(d), fmt, lvl, P)
        dst = torch.zeros(cap, dtype=torch.uint8, device="cuda")
        w = zz.encode_multi(ctxs, srcs, ns, dst, cap, fmt, lvl, P, halos=halos)
        got = dst[:w].cpu().numpy().tobytes()
        assert got == oracle.encode_packets(d, fmt, lvl, P), (lvl, fmt)
    # the pulls' way is asked for once per device pair and reported: same device = no peer needed; on a box with several
    # GPUs the second call must find the pair already enabled (idempotent) and count its pulls as peer to peer or staged
    import ctypes
    p2p, staged = ctypes.c_int(-1), ctypes.c_int(-1)
    zz.lib.zz_debug_last_pulls(ctypes.byref(p2p), ctypes.byref(staged))
    assert p2p.value + staged.value == 2 and zz.lib.zz_debug_peer_state(0, 0) == 1
    if torch.cuda.device_count() > 1:
        c1 = zz.Context(1)
        with torch.cuda.device(1):
            t1 = keep[1].to("cuda:1")
        for _ in range(2):
            w2 = zz.encode_multi([ctxs[0], c1], [srcs[0], t1[halos[1]:]], [ns[0], ns[1]], dst, cap, 0, lvl, P, halos=[0, halos[1]])
            assert dst[:w2].cpu().numpy().tobytes() == oracle.encode_packets(d[:cuts[2]], 0, lvl, P)
            assert zz.lib.zz_debug_peer_state(0, 1) in (1, -1)
    # too small a destination is reported, and the contexts stay usable
    dst = torch.zeros(1000, dtype=torch.uint8, device="cuda")
    with pytest.raises(zz.ZzFlateError):
        zz.encode_multi(ctxs, srcs, ns, dst, 1000, 0, lvl, P, halos=halos)
    cap = zz.bound(len(d), 0, lvl, P)
    dst = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    w = zz.encode_multi(ctxs[:1], [srcs[0]], [ns[0]], dst, cap, 0, lvl, P)
    assert dst[:w].cpu().numpy().tobytes() == oracle.encode_packets(d[:ns[0]], 0, lvl, P)
    with pytest.raises(zz.ZzFlateError):                     # a shard that is not a whole number of packets in front of the last
        zz.encode_multi(ctxs[:2], [srcs[0], srcs[1]], [ns[0] - 5, ns[1]], dst, cap, 0, lvl, P)


def test_nested_and_concurrent_callbacks_never_wait_for_the_pool(gpu, oracle, corpus):
    """The pool holds two contexts per device. A callback that encodes (depth 2: whose callback encodes again) and two
    threads whose callbacks both encode would wait for each other's contexts for ever if a nested call could block: it
    gets a temporary context instead (zz_api.hip pool_acquire)."""
    import threading
    d, small = corpus["alice29.txt"], corpus["fields.c"]
    want_small = oracle.encode_packets(small, 0, 2)
    want_xargs = oracle.encode_packets(corpus["xargs.1"], 0, 1)
    got = []

    def depth2(chunk):
        if len(got) < 1:
            def depth3(chunk2):
                if len(got) < 1:
                    got.append(zz.ZzFlateEncode(corpus["xargs.1"], zz.Config(zz.Format.Zlib, 1, True)))
            zz.ZzFlateEncodeToCallback(small, zz.Config(zz.Format.Zlib, 2, True), depth3)
    zz.ZzFlateEncodeToCallback(d, zz.Config(zz.Format.Zlib, 1, True), depth2)
    assert got == [want_xargs]

    # two threads, both inside a callback at the same time (a barrier makes sure of it), both encoding from there
    inside = threading.Barrier(2, timeout=60)
    results, errors = [None, None], []

    def worker(t):
        done = []

        def cb(chunk):
            if not done:
                done.append(1)
                inside.wait()                      # both outer calls hold a context of device 0 now: the pool is full
                results[t] = zz.ZzFlateEncode(small, zz.Config(zz.Format.Zlib, 2, True))
        try:
            zz.ZzFlateEncodeToCallback(d, zz.Config(zz.Format.Zlib, 1, True), cb)
        except Exception as e:                     # noqa: BLE001
            errors.append(e)
    ths = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    [t.start() for t in ths]
    [t.join(120) for t in ths]
    assert not any(t.is_alive() for t in ths), "nested calls from two threads blocked each other"
    assert not errors and results == [want_small, want_small]


@pytest.mark.parametrize("P", [1, 2, 3, 5, 8, 13, 15, 16, 17])
def test_tiny_packets_at_the_end_of_an_allocation(gpu, oracle, P):
    """Packets shorter than the 16-byte loads: bounds-checked loads are chosen by bytes, not by packet index, so
    nothing is read past the input even when it ends exactly at the end of its allocation."""
    torch = gpu.torch
    for n in (73, 256, 1000):
        d = synth("words", n, P)
        for lvl in (1, 2):
            # the source tensor has exactly n bytes: its last byte is the last byte of the allocation
            src = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
            cap = zz.bound(n, 2, lvl, P)
            dst = torch.zeros(cap, dtype=torch.uint8, device="cuda")
            w = gpu.ctx.encode(src, n, dst, cap, 2, lvl, P)
            assert dst[:w].cpu().numpy().tobytes() == oracle.encode_packets(d, 2, lvl, P), (P, n, lvl)


def test_two_calls_in_flight(gpu, oracle, corpus):
    """zz_encode_device_async / zz_encode_finish: two contexts on two streams, the second call enqueued before the first is
    finished; same bytes as the synchronous call, errors come out of finish, a context refuses a second call."""
    torch = gpu.torch
    d0, d1 = corpus["lcet10.txt"] * 8, corpus["kennedy.xls"] * 3
    ctxs = [zz.Context(0), zz.Context(0)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for lvl in LEVELS:
        bufs = []
        for i, d in enumerate((d0, d1)):
            src = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
            cap = zz.bound(len(d), 1, lvl)
            dst = torch.zeros(cap, dtype=torch.uint8, device="cuda")
            bufs.append((src, dst, cap, d))
        torch.cuda.synchronize()
        for i, (src, dst, cap, d) in enumerate(bufs):
            ctxs[i].encode_async(src, len(d), dst, cap, 1, lvl, stream=streams[i].cuda_stream)
        with pytest.raises(zz.ZzFlateError):
            ctxs[0].encode(bufs[0][0], len(d0), bufs[0][1], bufs[0][2], 1, lvl)      # one call per context at a time
        for i, (src, dst, cap, d) in enumerate(bufs):
            w = ctxs[i].finish()
            assert dst[:w].cpu().numpy().tobytes() == oracle.encode_packets(d, 1, lvl), (lvl, i)
            assert ctxs[i].verify_last() == (0, None)
    src, dst, cap, d = bufs[0]
    ctxs[0].encode_async(src, len(d), dst, 1000, 1, 1, stream=streams[0].cuda_stream)   # too small: reported by finish
    with pytest.raises(zz.ZzFlateError) as e:
        ctxs[0].finish()
    assert e.value.code == -2
    with pytest.raises(zz.ZzFlateError):
        ctxs[0].finish()                                                               # nothing enqueued


@pytest.mark.parametrize("warm", [258, 4096, 32768])
def test_warm_window_level1(gpu, oracle, corpus, warm):
    """SURVEY.md 8f.3 (beyond the reference): the last `warm` bytes in front of a packet are hashed into its table before
    the parse, so matches reach across packet boundaries. Bit-exact with the oracle's restatement of that rule, valid
    DEFLATE (host zlib and the device decoder), identical through shards and through the host entry points, and on text
    no larger than the cold-packet stream."""
    import torch
    ctx = zz.Context(0)
    ctx.set_warm_window(warm)

    def enc(d, fmt, P=32768):
        src = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
        cap = zz.bound(len(d), fmt, 1, P)
        dst = torch.zeros(cap, dtype=torch.uint8, device="cuda")
        w = ctx.encode(src, len(d), dst, cap, fmt, 1, P)
        assert ctx.verify_last() == (0, None)
        return dst[:w].cpu().numpy().tobytes()
    for fname in CORPUS_FILES:
        d = corpus[fname]
        got = enc(d, 0)
        assert got == oracle.encode_packets(d, 0, 1, warm=warm), (fname, warm)
        assert zlib.decompress(got) == d
    for fname in ("alice29.txt", "lcet10.txt", "plrabn12.txt", "asyoulik.txt"):
        assert len(enc(corpus[fname], 0)) < len(gpu.encode(corpus[fname], 0, 1)), fname
    for kind in SYNTH_KINDS + ["longperiod"]:
        for n, P in ((70000, 32768), (100000, 4096), (33000, 1000), (5000, 777)):
            d = synth(kind, n, 21)
            assert enc(d, 2, P) == oracle.encode_packets(d, 2, 1, P, warm=warm), (kind, n, P, warm)
    # level 0 has nothing to warm
    d = corpus["lcet10.txt"]
    src = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
    cap = zz.bound(len(d), 0, 0)
    dst = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    w = ctx.encode(src, len(d), dst, cap, 0, 0)
    assert dst[:w].cpu().numpy().tobytes() == oracle.encode_packets(d, 0, 0)
    # shards: the window is the shard's halo
    buf = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
    cut = 6 * 32768
    parts = []
    for off, n, last in ((0, cut, False), (cut, len(d) - cut, True)):
        cap = zz.bound(n, 2, 1)
        dst = torch.zeros(cap, dtype=torch.uint8, device="cuda")
        w, _ = ctx.encode_shard(buf.data_ptr() + off, n, dst, cap, halo=off, is_last=last, checksum=zz.Format.Deflate, level=1)
        assert ctx.verify_last() == (0, None)
        parts.append(dst[:w].cpu().numpy().tobytes())
    assert b"".join(parts) == oracle.encode_packets(d, 2, 1, warm=warm)
    # twice the same call, the same bytes
    assert enc(d, 1) == enc(d, 1)


def test_lds_serves_equal_addresses_in_lane_order(gpu):
    """What k_l6_matches' counting sort takes its places from (zz_level6.h): a returning LDS add hands the lanes of one
    wavefront instruction that hit one counter their counts in ascending lane order, and one wavefront's instructions theirs
    in issue order -- and the masked exchange (ds_mskor) the parsers insert with behaves the same. Not in the ISA manual, so
    checked on the device this runs on: 335 million values, keys with every kind of duplicate (one dword, one bank, one heavy key, all equal), 16 wavefronts per workgroup on every CU."""
    import ctypes
    ctx = zz.Context(0)
    bad, checked = ctypes.c_uint64(1), ctypes.c_uint64(0)
    assert zz.lib.zz_debug_lds_atomic_order(ctx._h, 128, ctypes.byref(bad), ctypes.byref(checked)) == 0
    assert checked.value == 512 * 1024 * 6 * 128 and bad.value == 0, (bad.value, checked.value)


def test_lds_order_guard_refuses_and_level1_falls_back(gpu, oracle, corpus):
    """Where the probe's verdict is "does not hold" (forced here through the debug hook), the modes that depend on the LDS's
    ordering are refused with ZZ_E_UNSUPPORTED and a message, level 1 runs its one-wavefront kernel -- the same bytes as the
    two-wavefront one and as the oracle --, and levels 0..3 with cold packets are untouched."""
    import torch
    ctx = zz.Context(0)
    assert zz.lib.zz_debug_lds_order_verdict(0) == 1          # the device this runs on passes the probe
    d = corpus["alice29.txt"] + corpus["kennedy.xls"][:300000]
    src = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
    cap = zz.bound(len(d), 0, 1, 32768)
    dst = torch.zeros(cap, dtype=torch.uint8, device="cuda")

    def enc(lvl):
        w = ctx.encode(src, len(d), dst, cap, 0, lvl, 32768)
        assert ctx.verify_last() == (0, None)
        return dst[:w].cpu().numpy().tobytes()
    two = enc(1)
    assert two == oracle.encode_packets(d, 0, 1, 32768)
    try:
        zz.lib.zz_debug_force_lds_order(0)
        for call in (lambda: ctx.set_warm_window(4096), lambda: ctx.set_extended_levels(True)):
            with pytest.raises(zz.ZzFlateError) as e:
                call()
            assert e.value.code == -5 and "lane order" in str(e.value)
        ctx.set_warm_window(0)                                 # switching off is always allowed
        ctx.set_extended_levels(False)
        assert enc(1) == two                                   # k_encode_l1 instead of k_encode_l1p: the same stream
        for lvl in (0, 2, 3):
            assert enc(lvl) == oracle.encode_packets(d, 0, lvl, 32768)
    finally:
        zz.lib.zz_debug_force_lds_order(-1)
    ctx.set_warm_window(4096)                                  # accepted again
    ctx.set_warm_window(0)


def test_lds_order_violation_seen_by_the_kernel_reruns_the_call(gpu, oracle, corpus):
    """k_encode_l1p checks the invariant it stands on as it runs (the slot must end up with the HIGHEST lane's entry: zz_level1p.h
    P2) and the warm window's pre-hash checks the same property on its last store per trip (level 1) or on a canary (levels 2, 3).
    A violation (forced here through the kernel's own report path: zz_packet_params::dbg_viol) must (a) take the device's verdict
    away, (b) run the level-1 CALL again on the one-wavefront kernel -- the stream handed out is the oracle's --, (c) fail a
    warm-window call with ZZ_E_UNSUPPORTED and a message instead of handing out a stream that may not be the defined one."""
    import torch
    ctx = zz.Context(0)
    d = corpus["lcet10.txt"] + corpus["ptt5"][:200000]
    src = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
    cap = zz.bound(len(d), 0, 1, 32768)
    dst = torch.zeros(cap, dtype=torch.uint8, device="cuda")

    def enc(lvl, fmt=0):
        w = ctx.encode(src, len(d), dst, cap, fmt, lvl, 32768)
        assert ctx.verify_last() == (0, None)
        return dst[:w].cpu().numpy().tobytes()
    try:
        assert zz.lib.zz_debug_lds_order_verdict(0) == 1 and zz.lib.zz_debug_l1_kernel(ctx._h) == 2
        want = oracle.encode_packets(d, 1, 1, 32768)
        assert enc(1, 1) == want                               # no violation: k_encode_l1p, verdict kept
        assert zz.lib.zz_debug_l1_kernel(ctx._h) == 2
        zz.lib.zz_debug_force_lds_violation(1)
        assert enc(1, 1) == want                               # violation reported by the kernel: the call ran again, same bytes
        assert zz.lib.zz_debug_lds_order_verdict(0) == 0 and zz.lib.zz_debug_l1_kernel(ctx._h) == 1
        assert enc(1, 1) == want                               # and every later call takes k_encode_l1 straight away
        with pytest.raises(zz.ZzFlateError) as e:              # the modes without a fallback are refused from here on
            ctx.set_warm_window(4096)
        assert e.value.code == -5
        # the asynchronous form: the rerun happens in zz_encode_finish
        zz.lib.zz_debug_reset_lds_order(0)
        assert zz.lib.zz_debug_lds_order_verdict(0) == 1
        zz.lib.zz_debug_force_lds_violation(1)
        ctx.encode_async(src, len(d), dst, cap, 1, 1, 32768)
        w = ctx.finish()
        assert dst[:w].cpu().numpy().tobytes() == want and zz.lib.zz_debug_lds_order_verdict(0) == 0
        # levels 2,3: k_encode_l2p enters a block's positions by one ordered LDS exchange per lane and checks what comes back; a
        # violation reruns the call on the two-wavefront kernel, which reads, writes, reads back and ranks same-hash lanes itself
        zz.lib.zz_debug_reset_lds_order(0)
        assert zz.lib.zz_debug_lds_order_verdict(0) == 1 and zz.lib.zz_debug_l2_kernel(ctx._h) == 2     # (asking for the verdict runs the probe again)
        for lvl in (2, 3):
            want2 = oracle.encode_packets(d, 0, lvl, 32768)
            zz.lib.zz_debug_reset_lds_order(0)
            assert zz.lib.zz_debug_lds_order_verdict(0) == 1
            assert enc(lvl) == want2
            zz.lib.zz_debug_force_lds_violation(1)
            assert enc(lvl) == want2                           # reported by the kernel, run again, same bytes
            assert zz.lib.zz_debug_lds_order_verdict(0) == 0 and zz.lib.zz_debug_l2_kernel(ctx._h) == 1
            assert enc(lvl) == want2                           # and from here on the two-wavefront kernel straight away
        # warm window, levels 1 and 2: no other form exists, the call fails loudly
        for lvl in (1, 2):
            zz.lib.zz_debug_reset_lds_order(0)
            ctx.set_warm_window(32768)
            assert enc(lvl) == oracle.encode_packets(d, 0, lvl, 32768, 32768)
            zz.lib.zz_debug_force_lds_violation(1)
            with pytest.raises(zz.ZzFlateError) as e:
                enc(lvl)
            assert e.value.code == -5 and "lane order" in str(e.value), lvl
            ctx.set_warm_window(0)
            assert zz.lib.zz_debug_lds_order_verdict(0) == 0
    finally:
        zz.lib.zz_debug_force_lds_violation(0)
        zz.lib.zz_debug_reset_lds_order(0)
        ctx.set_warm_window(0)
    assert zz.lib.zz_debug_lds_order_verdict(0) == 1 and zz.lib.zz_debug_l1_kernel(ctx._h) == 2


@pytest.mark.parametrize("args", [("1,2,3", "0"), ("1", "32768")], ids=["cold", "warm"])
def test_one_parser_kernels_stay_bit_exact(args):
    """k_encode_l1 / k_encode_l1w / k_encode_l2_t<0, false> are what ZZFLATE_L1_KERNEL=classic / ZZFLATE_L2_KERNEL=classic launch
    (A/B runs; k_encode_l1 is also the fallback behind the LDS-order check) now that the two-parser kernels are the default: the
    switches are read once per process, so a child process runs tests/gpu_quick.py (corpus files and synthetic edge sizes against
    the oracle) with both set."""
    import subprocess, sys
    env = dict(os.environ, ZZFLATE_L1_KERNEL="classic", ZZFLATE_L2_KERNEL="classic")
    r = subprocess.run([sys.executable, os.path.join(ROOT_DIR, "tests", "gpu_quick.py"), *args], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "bad 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_extended_levels_and_warm_window_at_level2(gpu, oracle, corpus):
    """levels 4, 5, 6 (off unless switched on) = hash chains of depth 2 / 4 / 8 over a window of 8 / 32 /
    32 KiB, one-step lazy matching, package-merge code lengths (zz_level6.h). Bit-exact with the oracle's definition
    (oracle/zzoracle.c "Extended levels"), valid DEFLATE (zlib and the device decoder), smaller than level 3 on the text
    files by more than 5 % at level 6; and a warm window is available at levels 2, 3 through zz_ctx_set_warm_window."""
    import torch
    ctx = zz.Context(0)

    def enc(d, fmt, lvl, P=32768):
        src = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
        cap = zz.bound(len(d), fmt, 2, P)
        dst = torch.zeros(cap, dtype=torch.uint8, device="cuda")
        w = ctx.encode(src, len(d), dst, cap, fmt, lvl, P)
        assert ctx.verify_last() == (0, None)
        return dst[:w].cpu().numpy().tobytes()
    d = corpus["alice29.txt"]
    with pytest.raises(zz.ZzFlateError) as e:       # the reference's error convention until switched on (zzflate.cpp:230)
        enc(d, 0, 6)
    assert e.value.code == -1
    ctx.set_extended_levels(True)
    with pytest.raises(zz.ZzFlateError):
        enc(d, 0, 7)
    for fname in CORPUS_FILES:
        d = corpus[fname]
        l3 = gpu.encode(d, 0, 3)
        for lvl in (4, 5, 6):
            got = enc(d, 0, lvl)
            assert got == oracle.encode_packets(d, 0, lvl), (fname, lvl)
            assert zlib.decompress(got) == d
            if fname.endswith(".txt"):
                assert len(got) < len(l3), (fname, lvl)
                if lvl == 6:
                    assert len(got) < 0.95 * len(l3), fname
        for P in (4096, 1000):
            assert enc(d[:200000], 1, 6, P) == oracle.encode_packets(d[:200000], 1, 6, P), (fname, P)
    for kind in SYNTH_KINDS + ["longperiod"]:
        for n, P in ((70000, 32768), (100000, 4096), (33000, 1000), (5000, 777)):
            d = synth(kind, n, 31)
            for lvl in (4, 5, 6):
                assert enc(d, 2, lvl, P) == oracle.encode_packets(d, 2, lvl, P), (kind, n, P, lvl)
    for n in EDGE_SIZES:
        d = synth("words", n, 5)
        assert enc(d, 2, 6) == oracle.encode_packets(d, 2, 6), n
        d = synth("runs", n, 6)
        assert enc(d, 2, 5, 4096) == oracle.encode_packets(d, 2, 5, 4096), n
    # shards: the window is the shard's halo
    d = corpus["lcet10.txt"] + corpus["ptt5"][:200000]
    buf = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
    cut = 6 * 32768
    parts = []
    for off, n, last in ((0, cut, False), (cut, len(d) - cut, True)):
        cap = zz.bound(n, 2, 2)
        dst = torch.zeros(cap, dtype=torch.uint8, device="cuda")
        w, _ = ctx.encode_shard(buf.data_ptr() + off, n, dst, cap, halo=off, is_last=last, checksum=zz.Format.Deflate, level=6)
        assert ctx.verify_last() == (0, None)
        parts.append(dst[:w].cpu().numpy().tobytes())
    assert b"".join(parts) == oracle.encode_packets(d, 2, 6)
    ctx.set_extended_levels(False)
    ctx.set_warm_window(8192)
    d = corpus["lcet10.txt"]
    for lvl in (2, 3):
        assert enc(d, 1, lvl) == oracle.encode_packets(d, 1, lvl, warm=8192)


def test_level2_match_that_overruns_the_batch_and_the_search_region(gpu, oracle):
    """Regression (found by tools/fuzz_gpu.py, case 7720 of seed 1234, present since round 1): a match that starts in the first
    16 KiB batch and ends behind n - 258 ends the token pass (encoder.cpp:225-234 leaves its loop with target <= 0); the kernel
    computed the next batch's length as an unsigned difference and went on probing -- a valid stream, one match too many. The
    input is period-375 data of 16,866 bytes; neighbouring sizes and periods ride along."""
    d = zlib.decompress(open(os.path.join(GOLDEN, "cases", "l2_match_overruns_batch_and_search_region.bin.z"), "rb").read())
    assert len(d) == 16866
    for lvl in (2, 3):
        for fmt in (0, 2):
            assert gpu.encode(d, fmt, lvl) == oracle.encode_packets(d, fmt, lvl), (lvl, fmt)
    for n in range(16384 + 258 - 40, 16384 + 258 + 300, 7):
        for per in (259, 300, 375, 511, 700):
            e = (d[:per] * (n // per + 1))[:n]
            assert gpu.encode(e, 2, 2) == oracle.encode_packets(e, 2, 2), (n, per)
    for P in (20000, 17000):
        assert gpu.encode(d * 3, 2, 3, P) == oracle.encode_packets(d * 3, 2, 3, P), P


def test_warm_window_candidates_at_the_very_start_of_the_stream(gpu, oracle):
    """Regression (found by tools/fuzz_gpu.py, case 3690 of seed 9191): with a warm window a candidate may sit in the first
    eight bytes of the STREAM while the packet that probes it has plenty of bytes in front of it, so the "fewer than 8 bytes
    in front of the candidate" path of the level-2 backward comparison (D4 clamp, encoder.cpp:92-102) must not be tied to
    the packet's own offset. 31,466 bytes from a short vocabulary in 1,024-byte packets."""
    import torch
    d = zlib.decompress(open(os.path.join(GOLDEN, "cases", "warm_candidate_in_first_bytes_of_stream.bin.z"), "rb").read())
    ctx = zz.Context(0)
    ctx.set_extended_levels(True)
    try:
        src = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
        for P in (1024, 777):
            for lvl, warm in ((5, 0), (6, 0), (4, 0), (2, 32768), (3, 1000)):
                ctx.set_warm_window(warm if lvl < 4 else 0)
                cap = zz.bound(len(d), 1, 3, P)
                dst = torch.zeros(cap, dtype=torch.uint8, device="cuda")
                w = ctx.encode(src, len(d), dst, cap, zz.Format.Gzip, lvl, P)
                got = dst[:w].cpu().numpy().tobytes()
                assert got == oracle.encode_packets(d, 1, lvl, P, warm=warm), (P, lvl)
                assert zlib.decompressobj(31).decompress(got) == d
    finally:
        ctx.set_warm_window(0)
        ctx.set_extended_levels(False)


def test_warm_window_through_the_host_entry_points(gpu, oracle, corpus):
    """ZZFLATE_WARM_WINDOW for ZzFlateEncode / ZzFlateEncodeToCallback; checked in a child process because the
    variable is read once. Slabs carry the window in their halo, so the slab pipeline gives the same bytes."""
    import subprocess, sys
    code = (
        "import os, sys; sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))\n"
        "import zzflate_amd as zz\nfrom conftest import Oracle, CORPUS\n"
        "d = b''.join(open(os.path.join(CORPUS, f), 'rb').read() for f in ('lcet10.txt', 'alice29.txt', 'plrabn12.txt')) * 3\n"
        "got = zz.ZzFlateEncode(d, zz.Config(zz.Format.Zlib, 1, True))\n"
        "assert got == Oracle().encode_packets(d, 0, 1, warm=32768), (len(got),)\nprint('ok')\n") % (ROOT_DIR, ROOT_DIR)
    env = dict(os.environ, ZZFLATE_WARM_WINDOW="32768", ZZFLATE_SLAB_MIB="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_extended_levels_in_batches(gpu, oracle, corpus):
    """The extended levels go through a call in batches of packets (k_l6_matches over a batch, then the encode kernel over the
    same packets; 1 GiB of input per batch by default: only the 8 GiB shard test gets there). With ZZFLATE_L6_BATCH_MIB=1 a few
    MiB make several batches -- with packets of 32768 and 4096 bytes, a last batch that is not full, one workgroup per packet
    and fewer -- and the stream must not change. In a child process: the variable is read once."""
    import subprocess, sys
    code = (
        "import os, sys, zlib; sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))\n"
        "import torch\nimport zzflate_amd as zz\nfrom conftest import Oracle, CORPUS\n"
        "d = b''.join(open(os.path.join(CORPUS, f), 'rb').read() for f in ('lcet10.txt', 'kennedy.xls', 'ptt5', 'plrabn12.txt')) * 2 + b'tail'\n"
        "ctx = zz.Context(0); ctx.set_extended_levels(True); o = Oracle()\n"
        "src = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()\n"
        "for lvl, P in ((6, 32768), (4, 4096), (5, 32768)):\n"
        "    cap = zz.bound(len(d), 0, 2, P); dst = torch.zeros(cap, dtype=torch.uint8, device='cuda')\n"
        "    w = ctx.encode(src, len(d), dst, cap, 0, lvl, P)\n"
        "    assert ctx.verify_last() == (0, None)\n"
        "    got = dst[:w].cpu().numpy().tobytes()\n"
        "    assert got == o.encode_packets(d, 0, lvl, P), (lvl, P, len(got))\n"
        "    assert zlib.decompress(got) == d\n"
        "print('ok', len(d))\n") % (ROOT_DIR, ROOT_DIR)
    env = dict(os.environ, ZZFLATE_L6_BATCH_MIB="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 13, 16, 19, 31, 32, 33, 64, 100, 117, 118, 119, 128, 200, 255, 256])
def test_code_length_runs_of_every_shape(gpu, oracle, k):
    """The run-length records of the code lengths (huffman.cpp:158-216) are written a lane per run from closed forms
    (zz_level2.h rle_lengths_w; tests/test_rle_closed_form.py checks the forms on the CPU): bytes uniform over an alphabet of k
    consecutive values give the literal code runs of k equal non-zero lengths (every remainder of (k - 1) / 6), a run of 256 - k
    zeros behind them (longer than 138 for small k, 3..10 and 11.. for large), and runs that cross the 64-symbol blocks the wave
    works in; with the alphabet at the top of the byte range the zero run comes first."""
    import random
    rnd = random.Random(1000 + k)
    for base in (0, 256 - k):
        d = bytes(base + rnd.randrange(k) for _ in range(40000))
        for lvl in (2, 3):
            assert gpu.encode(d, 0, lvl) == oracle.encode_packets(d, 0, lvl, 32768), (k, base, lvl)


@pytest.mark.parametrize("lvl", LEVELS)
def test_shards_concatenate_to_the_whole_stream(gpu, oracle, corpus, lvl):
    """Multi-GPU contract on one GPU: shards cut at packet boundaries + checksum combine == one call."""
    d = corpus["lcet10.txt"]
    P = 32768
    whole = {fmt: gpu.encode(d, fmt, lvl) for fmt in (0, 1)}
    cut = 5 * P
    for fmt in (0, 1):
        s0, c0 = gpu.shard(d, 0, cut, 0, False, fmt, lvl)
        s1, c1 = gpu.shard(d, cut, len(d) - cut, cut, True, fmt, lvl)
        if fmt == 0:
            total = zz.combine(zz.combine(1, c0, cut), c1, len(d) - cut)
        else:
            total = zz.crc32_combine(c0, c1, len(d) - cut)
        out = zz.header(fmt) + s0 + s1 + zz.trailer(fmt, total, len(d))
        assert out == whole[fmt], (fmt, lvl)


def test_generated_inputs_match_host(gpu):
    torch = gpu.torch
    for kind in (zz.GEN_TEXT, zz.GEN_RANDOM, zz.GEN_LOG, zz.GEN_MIX):
        n = 3 * 65536 + 1234
        buf = torch.zeros(n, dtype=torch.uint8, device="cuda")
        gpu.ctx.generate(kind, 0x5EED0002, 65536 * 7, buf, n)
        torch.cuda.synchronize()
        assert bytes(buf.cpu().numpy().tobytes()) == zz.generate_host(kind, 0x5EED0002, 65536 * 7, n)


@pytest.mark.parametrize("lvl", LEVELS)
@pytest.mark.parametrize("kind", [0, 1, 2])
def test_large_generated_roundtrip_and_sampled_parity(gpu, oracle, kind, lvl):
    """Full-size property checks: 256 MiB generated on the device, compressed, inflated on the host in
    pieces (round trip + trailer), and sampled packets compared bit-for-bit with the oracle."""
    torch = gpu.torch
    n = 256 << 20 if lvl == 1 else 64 << 20
    P = 32768
    src = torch.empty(n, dtype=torch.uint8, device="cuda")
    gpu.ctx.generate(kind, 0x5EED0000 + kind, 0, src, n)
    cap = zz.bound(n, 1, lvl, P)
    dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
    w = gpu.ctx.encode(src, n, dst, cap, 1, lvl, P)
    out = dst[:w].cpu().numpy().tobytes()
    inp = src.cpu().numpy().tobytes()
    o = zlib.decompressobj(31)
    pos = 0
    for i in range(0, len(out), 1 << 24):
        piece = o.decompress(out[i:i + (1 << 24)])
        assert piece == inp[pos:pos + len(piece)]
        pos += len(piece)
    assert pos == n and o.eof          # crc32 + isize verified by zlib
    # sampled packets vs the oracle: a prefix, a middle run, the tail
    for start_pk, cnt in ((0, 8), (n // P // 2, 8), (n // P - 8, 8)):
        off = start_pk * P
        last = off + cnt * P >= n
        got, _ = gpu.shard(inp[:off + cnt * P], off, cnt * P, off, last, 2, lvl)
        want = b"".join(oracle.packet(inp, lvl, off + k * P, P, last and k == cnt - 1) for k in range(cnt))
        assert got == want, (kind, lvl, start_pk)
