"""Per-call latency on small device-resident inputs (manual)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import zzflate_amd as zz
ctx = zz.Context(0)
for n in (4096, 148481, 1 << 20, 16 << 20):
    src = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    ctx.generate(zz.GEN_TEXT, 1, 0, src, n)
    for lvl in (0, 1, 2):
        cap = zz.bound(n, 0, lvl, 32768)
        dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
        for _ in range(5): ctx.encode(src, n, dst, cap, 0, lvl)
        torch.cuda.synchronize(); t = time.perf_counter()
        R = 200
        for _ in range(R): ctx.encode(src, n, dst, cap, 0, lvl)
        dt = (time.perf_counter() - t) / R
        print(f"n {n:9d} level {lvl}: {dt * 1e6:8.1f} us per call = {n / dt / 1e9:7.3f} GB/s", flush=True)
