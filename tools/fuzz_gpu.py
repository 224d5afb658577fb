"""Extended seeded fuzz against the oracle (manual; the pytest suite runs a 120-case version): packet mode (cold, warm
window, extended levels), the sequential stream into roomy and tight destinations, the callback form's chunks.
python tools/fuzz_gpu.py [cases] [seed] [--seconds S]     stops by itself after S seconds (exit 0 unless a case was bad), so that a
                                                          run under a time box ends with its own summary line, not with the box's kill
ZZ_FUZZ_DUMP=<case> python tools/fuzz_gpu.py [cases] [seed]   (no GPU needed) writes that case's input to gpurun_out/fuzz_case_<case>.in"""
import os, random, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import zzflate_amd as zz
from conftest import Oracle, synth, SYNTH_KINDS
DUMP = int(os.environ.get("ZZ_FUZZ_DUMP", "-1"))
o = Oracle()
ctx = zz.Context(0) if DUMP < 0 else None
BUDGET = None
if "--seconds" in sys.argv:
    i = sys.argv.index("--seconds"); BUDGET = float(sys.argv[i + 1]); del sys.argv[i:i + 2]
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 777)
WB = {0: 15, 1: 31, 2: -15}
kinds = SYNTH_KINDS + ["longperiod"]
t0 = time.time(); bad = 0
done = 0
for it in range(cases):
    if BUDGET is not None and time.time() - t0 > BUDGET: break
    done = it + 1
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
    mode = it % 4
    if DUMP >= 0:
        # consume the same random numbers as a real run, without a GPU
        if it == DUMP:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            open(os.path.join(ROOT, "gpurun_out", f"fuzz_case_{it}.in"), "wb").write(d)
            print("case", it, kind, len(d), "P", P, "lvl", lvl, "fmt", fmt, "mode", mode)
            sys.exit(0)
        if mode == 3 and len(d) > 0:
            if rng.random() < 0.5:
                rng.randint(40, 99); rng.choice([0, 1, 2])
        else:
            if mode >= 1: rng.choice([0, 0, 258, 1000, 4096, 32768])
            if mode == 2 and rng.random() < 0.4: rng.choice([4, 5, 6])
        continue
    src = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
    if os.environ.get("ZZ_FUZZ_RANGES") == "1" and len(d) > 0:
        # every case through the reference's own threaded split (zz_encode_ranges_device) with a count of its own (a separate
        # run: ZZ_FUZZ_RANGES=1 python tools/fuzz_gpu.py ...), levels 0, 2, 3
        count = 1 + (it * 13 + len(d)) % 40
        rl = (0, 2, 3)[it % 3]
        want = o.encode_ranges(d, fmt, rl, count)
        cap = 2 * len(d) + 4096 + 16 * count
        dst = torch.zeros(cap, dtype=torch.uint8, device="cuda")
        w = ctx.encode_ranges(src, len(d), dst, cap, count, fmt, rl)
        got = dst[:w].cpu().numpy().tobytes()
        if got != want or zlib.decompressobj(WB[fmt]).decompress(got) != d:
            bad += 1
            print("MISMATCH ranges", it, kind, len(d), count, rl, fmt, flush=True)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            open(os.path.join(ROOT, "gpurun_out", f"fuzz_fail_{it}.in"), "wb").write(d)
        if (it + 1) % 100 == 0: print(f"{it + 1} cases, {bad} bad, {time.time() - t0:.0f} s", flush=True)
        continue
    if mode == 3 and len(d) > 0:
        # the reference's single-Encoder stream (threaded = false) into a tight or roomy destination, or in chunks
        if rng.random() < 0.5:
            cap = rng.choice([2 * len(d) + 1024, max(200, len(d)), max(200, len(d) * rng.randint(40, 99) // 100)])
            want = o.encode(d, fmt, lvl, cap=cap)
            ok_ref = False
            try:
                ok_ref = zlib.decompressobj(WB[fmt]).decompress(want) == d
            except zlib.error:
                pass
            dst = torch.zeros(cap + 64, dtype=torch.uint8, device="cuda")
            try:
                w = ctx.encode_stream(src, len(d), dst, cap, fmt, lvl)
                got = dst[:w].cpu().numpy().tobytes()
            except zz.ZzFlateError:
                got = None
            if (ok_ref and got != want) or (not ok_ref and got is not None and lvl == 1):
                bad += 1
                print("MISMATCH stream", it, kind, len(d), cap, lvl, fmt, got is None, flush=True)
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                open(os.path.join(ROOT, "gpurun_out", f"fuzz_fail_{it}.in"), "wb").write(d)
                open(os.path.join(ROOT, "gpurun_out", f"fuzz_fail_{it}.got"), "wb").write(got or b"")
        else:
            want, sizes = o.encode_callback(d, fmt, lvl)
            cap = 2 * len(d) + 4096
            dst = torch.zeros(cap, dtype=torch.uint8, device="cuda")
            w, chunks = ctx.encode_stream_chunks(src, len(d), dst, cap, fmt, lvl)
            got = dst[:w].cpu().numpy().tobytes()
            valid = True
            try:
                valid = zlib.decompressobj(WB[fmt]).decompress(want) == d
            except zlib.error:
                valid = False
            if valid and (got != want or chunks != sizes[1:-1]):
                bad += 1
                print("MISMATCH chunks", it, kind, len(d), lvl, fmt, flush=True)
    else:
        # packet mode; every other case with a warm window or an extended level
        warm = rng.choice([0, 0, 258, 1000, 4096, 32768]) if mode >= 1 else 0
        elvl = lvl
        if mode == 2 and rng.random() < 0.4:
            elvl = rng.choice([4, 5, 6]); warm = 0
        if os.environ.get("ZZ_FUZZ_EXT") == "1":          # every packet-mode case at an extended level (a separate, own random stream)
            elvl = 4 + (it * 7 + len(d)) % 3; warm = 0
        ctx.set_warm_window(warm)
        ctx.set_extended_levels(elvl > 3)
        cap = zz.bound(len(d), fmt, min(elvl, 3), P)
        dst = torch.zeros(cap, dtype=torch.uint8, device="cuda")
        w = ctx.encode(src, len(d), dst, cap, fmt, elvl, P)
        got = dst[:w].cpu().numpy().tobytes()
        want = o.encode_packets(d, fmt, elvl if elvl > 3 else lvl, P, warm=0 if elvl > 3 else warm)
        vb, _ = ctx.verify_last() if len(d) else (0, None)
        if got != want or zlib.decompressobj(WB[fmt]).decompress(got) != d or vb:
            bad += 1
            print("MISMATCH", it, kind, len(d), P, elvl, fmt, warm, vb, flush=True)
    if it % 100 == 99: print(f"{it + 1} cases, {bad} bad, {time.time() - t0:.0f} s", flush=True)
print(f"{done} cases in {time.time() - t0:.0f} s" + (f" (time budget {BUDGET:.0f} s)" if BUDGET is not None else "") + f", bad {bad}")
sys.exit(1 if bad else 0)
