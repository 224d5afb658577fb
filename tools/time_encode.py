"""Time Context.encode alone (no output checks): python tools/time_encode.py [level] [mib] [gen]. For experiments."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import zzflate_amd as zz
lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = (int(sys.argv[2]) if len(sys.argv) > 2 else 1024) << 20
gen = {"text": zz.GEN_TEXT, "random": zz.GEN_RANDOM, "log": zz.GEN_LOG}[sys.argv[3] if len(sys.argv) > 3 else "text"]
ctx = zz.Context(0); ctx.enable_timing(True)
src = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
ctx.generate(gen, 0x5EED0002, 0, src, n)
cap = zz.bound(n, 0, lvl, 32768)
dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
for _ in range(2): w = ctx.encode(src, n, dst, cap, 0, lvl)
torch.cuda.synchronize(); t = time.perf_counter(); km = 0.0
R = 8
for _ in range(R): w = ctx.encode(src, n, dst, cap, 0, lvl); km += ctx.last_kernel_ms()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / R
print(f"level {lvl}: {n / dt / 1e9:.2f} GB/s whole call, kernel {km / R:.3f} ms, ratio {w / n:.4f}")
