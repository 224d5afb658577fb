"""Throughput on degenerate inputs (manual): zeros, short periods, long periods -- no pathological slow paths."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import zzflate_amd as zz
ctx = zz.Context(0)
n = 256 << 20
rng = np.random.default_rng(1)
def period(k):
    base = rng.integers(0, 256, k, dtype=np.uint8)
    return np.tile(base, n // k + 1)[:n]
cases = {"zeros": np.zeros(n, np.uint8), "period 1 (0x41)": np.full(n, 0x41, np.uint8), "period 3": period(3), "period 7": period(7),
         "period 300": period(300), "period 5000": period(5000), "period 40000": period(40000),
         "two symbols random": rng.integers(0, 2, n, dtype=np.uint8) + 65,
         "four symbols random (DNA-like)": np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)],
         "sixteen symbols random": rng.integers(0, 16, n, dtype=np.uint8) + 65}
only = sys.argv[1:]          # optional substrings selecting cases
if only: cases = {k: v for k, v in cases.items() if any(o in k for o in only)}
for name, arr in cases.items():
    src = torch.from_numpy(arr).cuda()
    for lvl in (1, 2):
        cap = zz.bound(n, 0, lvl, 32768)
        dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
        w = ctx.encode(src, n, dst, cap, 0, lvl)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(3): w = ctx.encode(src, n, dst, cap, 0, lvl)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
        print(f"{name:22s} level {lvl}: {n / dt / 1e9:7.1f} GB/s  ratio {w / n:.4f}", flush=True)
