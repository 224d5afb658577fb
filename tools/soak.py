"""Soak (manual): many GiB through levels 1..3 and the extended levels, every packet checked by the device decoder each time --
looks for timing-dependent faults in the two-wavefront hand-overs. python tools/soak.py [rounds] [gib]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import zzflate_amd as zz
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n = (int(sys.argv[2]) if len(sys.argv) > 2 else 4) << 30
ctx = zz.Context(0)
ctx.set_extended_levels(True)
src = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
dst = torch.empty(zz.bound(n, 1, 1, 32768), dtype=torch.uint8, device="cuda")
bad_total = 0; t0 = time.time(); done = 0; sizes = {}
for r in range(rounds):
    kind = [zz.GEN_TEXT, zz.GEN_MIX, zz.GEN_LOG, zz.GEN_RANDOM][r % 4]
    ctx.generate(kind, 0x5EED0000 + r // 4, 0, src, n)       # a new seed every four rounds
    for lvl, fmt in ((1, 0), (2, 1), (1, 1), (3, 0), (4 + r % 3, 2)):
        w = ctx.encode(src, n, dst, zz.bound(n, fmt, min(lvl, 3), 32768), fmt, lvl)
        bad, first = ctx.verify_last()
        bad_total += bad; done += n
        key = (kind, r // 4, lvl, fmt)
        if bad: print("BAD", key, bad, first, flush=True)
    print(f"round {r + 1}/{rounds}: {done / 2**30:.0f} GiB encoded and verified, {bad_total} bad packets, {time.time() - t0:.0f} s", flush=True)
print("bad", bad_total)
sys.exit(1 if bad_total else 0)
