"""Quick manual GPU check (not collected by pytest): a few encodes against the oracle."""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import zzflate_amd as zz
from conftest import Oracle, synth
o = Oracle()
ctx = zz.Context(0)
levels = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,1").split(",")]
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # warm window in bytes (levels >= 1)
if warm:
    ctx.set_warm_window(warm)
def enc(data, fmt, lvl, P=32768):
    n = len(data)
    src = torch.frombuffer(bytearray(data) if n else bytearray(1), dtype=torch.uint8).cuda()
    cap = zz.bound(n, fmt, lvl, P)
    dst = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    w = ctx.encode(src, n, dst, cap, fmt, lvl, P)
    return dst[:w].cpu().numpy().tobytes()
bad = 0
for f in ["grammar.lsp", "alice29.txt", "ptt5", "kennedy.xls"]:
    d = open(os.path.join(ROOT, "tests/golden/corpus", f), "rb").read()
    for lvl in levels:
        for fmt in (0, 1, 2):
            t = time.time(); got = enc(d, fmt, lvl); dt = time.time() - t
            want = o.encode_packets(d, fmt, lvl, 32768, warm if lvl else 0)
            ok = got == want
            if not ok:
                bad += 1
                i = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), None)
                print("MISMATCH", f, lvl, fmt, len(got), len(want), "first diff", i)
            else:
                print("ok", f, lvl, fmt, len(got), f"{dt*1e3:.1f} ms", flush=True)
for kind in ["random", "zeros", "words", "ab", "runs", "period"]:
    for n in (1, 5, 300, 70000):
        d = synth(kind, n, 1)
        for lvl in levels:
            got = enc(d, 0, lvl); want = o.encode_packets(d, 0, lvl, 32768, warm if lvl else 0)
            if got != want:
                bad += 1
                i = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), None)
                print("MISMATCH", kind, n, lvl, len(got), len(want), i)
print("bad", bad)
sys.exit(1 if bad else 0)
