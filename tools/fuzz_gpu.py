"""Extended seeded fuzz against the oracle (manual; the pytest suite runs a 120-case version):
python tools/fuzz_gpu.py [cases] [seed]"""
import os, random, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import zzflate_amd as zz
from conftest import Oracle, synth, SYNTH_KINDS
o = Oracle()
ctx = zz.Context(0)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 777)
WB = {0: 15, 1: 31, 2: -15}
kinds = SYNTH_KINDS + ["longperiod"]
t0 = time.time(); bad = 0
for it in range(cases):
    kind = rng.choice(kinds)
    n = rng.choice([rng.randint(1, 300), rng.randint(300, 40000), rng.randint(40000, 400000)])
    d = synth(kind, n, 10000 + it)
    if it % 3 == 0:
        e = synth(rng.choice(kinds), n, 50000 + it); cut = rng.randint(0, n); d = d[:cut] + e[cut:]
    if it % 7 == 0:   # sprinkle short repeats of earlier content: backward extension and in-group candidates
        b = bytearray(d)
        for _ in range(rng.randint(1, 40)):
            if len(b) < 64: break
            a = rng.randrange(0, len(b) - 32); L = rng.randint(3, 300); c = rng.randrange(0, len(b))
            b[c:c + L] = b[a:a + L]
        d = bytes(b[:n]) if len(b) >= n else bytes(b)
    P = rng.choice([32768, 32768, 32768, 16384, 8192, 4096, 2048, 1024, 1000, 777, rng.randint(1, 32768)])
    lvl = rng.randint(0, 3); fmt = rng.randint(0, 2)
    src = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
    cap = zz.bound(len(d), fmt, lvl, P)
    dst = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    w = ctx.encode(src, len(d), dst, cap, fmt, lvl, P)
    got = dst[:w].cpu().numpy().tobytes()
    want = o.encode_packets(d, fmt, lvl, P)
    if got != want or zlib.decompressobj(WB[fmt]).decompress(got) != d:
        bad += 1
        print("MISMATCH", it, kind, len(d), P, lvl, fmt, flush=True)
    if it % 100 == 99: print(f"{it + 1} cases, {bad} bad, {time.time() - t0:.0f} s", flush=True)
print("bad", bad)
sys.exit(1 if bad else 0)
