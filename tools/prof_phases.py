"""Diagnostic: per-phase cycle shares of k_encode_l1 from a -DZZ_PROF build (s_memtime stamps).
Builds a separate library (never the shipped one), runs one encode of synthetic text, prints shares."""
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
L.zz_ctx_set_extended_levels(h, 1)
L.zz_bound.restype = u64
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 0
level = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n = mib << 20
src = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
if kind == 9:   # real English text: the Canterbury text files tiled (cold table per packet, so tiling is harmless)
    base = b"".join(open(os.path.join(ROOT, "tests/golden/corpus", f), "rb").read() for f in ("alice29.txt", "asyoulik.txt", "lcet10.txt", "plrabn12.txt"))
    host = (base * (n // len(base) + 1))[:n]
    src[:n].copy_(torch.frombuffer(bytearray(host), dtype=torch.uint8))
elif kind in (12, 14):   # random bytes over 2 / 4 symbols (the DNA-like family of the mix is the latter)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    src[:n].copy_(torch.randint(0, kind - 10, (n,), generator=g, device="cuda", dtype=torch.uint8) + 65)
else:
    L.zz_generate_device(h, ci(kind), u64(0x5EED0002), u64(0), vp(src.data_ptr()), u64(n), vp(0))
cap = L.zz_bound(u64(n), ci(0), ci(level), u32(32768))
dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
out = u64(0)
for it in range(2):
    rc = L.zz_encode_device(h, vp(src.data_ptr()), u64(n), vp(dst.data_ptr()), u64(cap), ctypes.byref(out), ci(0), ci(level), u32(32768), vp(0))
    assert rc == 0
    sets = (ctypes.c_ulonglong * 64)()
    if hasattr(L, "zz_debug_read_prof_sets"):
        L.zz_debug_read_prof_sets(h, sets)
        prof = list(sets[0:16])
    else:
        prof = (ctypes.c_ulonglong * 16)()
        L.zz_debug_read_prof(h, prof)
if level in (2, 3) and sets[16 + 10]:
    # the two-parser token pass (zz_level2p.h): one counter set per parsing wavefront, one for the helper
    for w in (0, 1):
        p = list(sets[16 * (1 + w):16 * (2 + w)]); b = max(1, p[10])
        print(f"parser {w}: blocks {p[10]}; per block: matches {p[11] / b:.2f} (out-of-line {p[12] / b:.3f}); cycles: loop top {p[5] / b:.0f} | enter + compare (P) {p[0] / b:.0f} | "
              f"wait: walk in front {p[1] / b:.0f} | walk {p[2] / b:.0f} | symbols + hand-over {p[3] / b:.0f} | wait: own barrier {p[4] / b:.0f} | sum {sum(p[0:6]) / b:.0f}")
    e = list(sets[48:64]); be = max(1, e[10])
    print(f"helper: blocks {e[10]}; per block: asleep in the barrier {e[0] / be:.0f} cyc, busy {e[1] / be:.0f} cyc")
if level >= 2:
    names = ["init+adler", "token pass", "counter unpack", "huffman" if not (prof[6] and not prof[9]) else "huffman (what the parts below leave)", "codes", "emit"]
    tot = sum(prof[:6]) + (sum(prof[6:9]) if prof[6] and not prof[9] else 0)
    print(f"level {level} input {mib} MiB kind {kind}: ratio {out.value / n:.4f}; packets {prof[10]}, tokens/packet {prof[11] / max(1, prof[10]):.0f}, cycles/packet {tot / max(1, prof[10]):.0f} = {tot / max(1, prof[10]) / 32768:.1f} cyc/byte")
    for i, nm in enumerate(names):
        print(f"  {nm:18s} {100.0 * prof[i] / tot:6.2f} %   {prof[i] / max(1, prof[10]):10.0f} cyc/packet")
    pk = max(1, prof[10])
    if prof[6] and not prof[9]:     # the code construction in parts (stamps 6, 7, 8 inside it; the rest is under "huffman")
        print("  code construction (cycles/packet): literal / length code's lengths %.0f | distance code's (or the wait for them) %.0f | run lengths %.0f | code-length code + bit count %.0f" % (
            prof[6] / pk, prof[7] / pk, prof[8] / pk, prof[3] / pk))
    if level >= 4:
        print("  k_l6_matches (cycles/packet, wavefront 0): zero %.0f | histogram %.0f | scan %.0f | rounds before the packet's own blocks %.0f | rounds with chain copies %.0f | window and packet into LDS %.0f | compares %.0f | between packets %.0f" % (
            prof[6] / pk, prof[7] / pk, prof[8] / pk, prof[9] / pk, prof[12] / pk, prof[15] / pk, prof[14] / pk, prof[13] / pk))
        sys.exit(0)
    if not (prof[6] and not prof[9]): print("  token pass detail (cycles/packet): loop top %.0f | insert + dup sets %.0f | compare loads + wait %.0f | walk %.0f | publish %.0f | finish block %.0f" % (
        prof[6] / pk, prof[7] / pk, prof[8] / pk, prof[9] / pk, prof[12] / pk, prof[13] / pk))
    sys.exit(0)
names = ["loop top", "hash+probe issue", "emit prev group", "readback+dup loop", "wait cand load", "info VALU", "walk", "repair+pack", "wait wnext"]
tot = sum(prof[:9])
g = max(1, prof[10])
print(f"input {mib} MiB kind {kind}: ratio {out.value / n:.4f}; groups {prof[10]}")
print(f"per group: event lanes {prof[12]/g:.2f} (with an in-group candidate {prof[15]/g:.2f}); walk left for the C++ path {prof[11]/g:.3f} times (hard {prof[13]/g:.3f}, 16-or-more {prof[14]/g:.3f})")
print(f"bytes/group {n / g:.1f}, cycles/group {tot / g:.0f}")
for i, nm in enumerate(names):
    print(f"  {nm:16s} {100.0 * prof[i] / tot:6.2f} %   {prof[i] / max(1, prof[10]):8.0f} cyc/group")
