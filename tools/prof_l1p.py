"""Diagnostic: per-phase cycles of k_encode_l1p's first parsing wavefront (even blocks) from a -DZZ_PROF build (s_memtime
stamps). Builds a separate library (never the shipped one), runs one encode of synthetic data, prints the shares.
    python3 tools/prof_l1p.py [MiB] [kind: 0 text, 1 random, 2 log, 3 mix, 100 + f: family f of the mix]"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
lib = os.environ.get("ZZ_PROF_LIB") or os.path.join(ROOT, "abl", "prof.so")            # built in the build container (tools/mkab.sh prof -DZZ_PROF) ...
if not os.path.exists(lib):                            # ... or here
  lib = os.path.join(ROOT, "gpurun_out", "libzz_prof.so")
  os.makedirs(os.path.dirname(lib), exist_ok=True)
  subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DZZ_PROF", "-o", lib,
                os.path.join(ROOT, "zzflate_amd/csrc/zz_api.hip"), os.path.join(ROOT, "zzflate_amd/csrc/zz_cxx_shim.cpp")], check=True)
L = ctypes.CDLL(lib)
u64, vp, ci, u32 = ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32
h = vp()
assert L.zz_ctx_create(0, ctypes.byref(h)) == 0
L.zz_bound.restype = u64
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = mib << 20
src = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
if kind >= 100:     # one family of the twelve-family mix (kind - 100), its 16 MiB segment tiled
    seg = 16 << 20
    mix = torch.empty(12 * seg + 64, dtype=torch.uint8, device="cuda")
    L.zz_generate_device(h, ci(3), u64(0x5EED0004), u64(0), vp(mix.data_ptr()), u64(12 * seg), vp(0))
    src[:n].copy_(mix[(kind - 100) * seg:(kind - 99) * seg].repeat(n // seg))
else:
    L.zz_generate_device(h, ci(kind), u64(0x5EED0002), u64(0), vp(src.data_ptr()), u64(n), vp(0))
cap = L.zz_bound(u64(n), ci(0), ci(1), u32(32768))
dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
out = u64(0)
for it in range(2):
    rc = L.zz_encode_device(h, vp(src.data_ptr()), u64(n), vp(dst.data_ptr()), u64(cap), ctypes.byref(out), ci(0), ci(1), u32(32768), vp(0))
    assert rc == 0
    sets = (ctypes.c_ulonglong * 64)()
    L.zz_debug_read_prof_sets(h, sets)
    prof = list(sets[0:16])
names = ["probe + compare (P1 P2)", "wait: walk of the block in front", "R: LDS reads back, told picked", "R: comparison", "R: events, hop pointers", "walk (W)", "publish", "wait: own walk's barrier", "repair + tokens (P4)"]
idx = [0, 1, 7, 8, 2, 3, 4, 5, 6]
b = max(1, prof[10])
tot = sum(prof[i] for i in idx)
print(f"input {mib} MiB kind {kind}: ratio {out.value / n:.4f}; blocks of wavefront 0: {prof[10]}; cycles per block {tot / b:.0f} (two barriers: one block period = half)")
print(f"per block: event lanes {prof[12] / b:.2f}; out-of-line path {prof[11] / b:.3f} times (positions carried in {prof[15] / b:.2f}); told spins {prof[13] / b:.3f}; chain walks {prof[14] / b:.4f}")
for i, nm in zip(idx, names):
    print(f"  {nm:36s} {100.0 * prof[i] / tot:6.2f} %   {prof[i] / b:8.0f} cyc/block")

# the second parsing wavefront (odd blocks) and the emitter, from their own counter sets
p1 = list(sets[16:32]); b1 = max(1, p1[10]); tot1 = sum(p1[i] for i in idx)
print(f"wavefront 1 (odd blocks): blocks {p1[10]}; cycles per block {tot1 / b1:.0f}")
for i, nm in zip(idx, names):
    print(f"  {nm:36s} {100.0 * p1[i] / max(1, tot1):6.2f} %   {p1[i] / b1:8.0f} cyc/block")
e = list(sets[32:48]); be = max(1, e[10])
print(f"emitter: blocks {e[10]}; per block: asleep in the barrier {e[0] / be:.0f} cyc, emitting {e[1] / be:.0f} cyc")
if e[11]:      # per packet (stamps around the emitter's packet: what happens outside the block loop)
    pk = e[11]
    print(f"emitter, per packet ({pk} packets): launch to the table-clear barrier {e[2] / pk:.0f} | Adler-32 {e[3] / pk:.0f} | header + wait for the parsers' first probe (B_0) {e[4] / pk:.0f} | "
          f"B_1 {e[5] / pk:.0f} | the block loop {(e[0] + e[1]) / pk:.0f} | end of block, alignment, flush {e[6] / pk:.0f} cycles")
