#!/usr/bin/env python3
"""Where the encode kernels touch their scratch (spill) memory: compiles zz_api.hip to gfx950 assembly and lists, per
kernel, every scratch_load / scratch_store with the loop depth of the basic block it sits in (the compiler's own
"; in Loop: Header=... Depth=N" / "=>This Inner Loop Header: Depth=N" annotations), plus the -Rpass-analysis=
kernel-resource-usage remarks. CPU-only (hipcc cross-compiles).

    python3 tools/scratch_report.py > profiles/r03_resource_usage.txt
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "zzflate_amd", "csrc", "zz_api.hip")
KERNELS = ["k_encode_l1E", "k_encode_l1pE", "k_encode_l1pwE", "k_encode_l1wE", "k_encode_l2_tILj0ELb0ELb0E", "k_encode_l2_tILj0ELb0ELb1E", "k_encode_l2_tILj32768ELb0ELb0E", "k_encode_l2_tILj32768ELb1ELb0E", "k_l6_matchesILi2E", "k_l6_matchesILi4E", "k_l6_matchesILi8E",
           "k_stream_l1E", "k_stream_l2E", "k_encode_l0E"]


def main():
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "zz.s")
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
                            "-Rpass-analysis=kernel-resource-usage", "-o", asm, SRC], capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr)
        print("# hipcc --offload-arch=gfx950 -O3 -std=c++17 -Rpass-analysis=kernel-resource-usage  (ROCm 7.2)")
        print("# source: zzflate_amd/csrc/zz_api.hip and the headers it includes\n")
        cur = None
        for line in r.stderr.splitlines():
            m = re.search(r"remark: (.*) \[-Rpass-analysis", line)
            if not m:
                continue
            t = m.group(1)
            if t.startswith("Function Name:"):
                cur = t
                print()
            print(t)
        text = open(asm).read()
        print("\n\n# ---- scratch accesses by loop depth (depth 0 = straight-line kernel code outside every loop) ----")
        for k in KERNELS:
            m = re.search(r"^(_ZN2zz\d+%s\w*):" % re.escape(k), text, flags=re.M)
            if not m:
                continue
            name = m.group(1)
            body = text[m.end():]
            body = body[:body.index(".Lfunc_end")]
            depth, label = 0, "(entry)"
            per = {}
            nloads = nstores = 0
            for ln in body.splitlines():
                lm = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", ln)
                if lm:
                    label = lm.group(1)
                    c = lm.group(2) or ""
                    dm = re.search(r"Depth=(\d+)", c)
                    depth = int(dm.group(1)) if dm else 0
                    continue
                # (continuation comment lines of a block header carry the innermost loop's depth)
                cm = re.match(r"^\s*;\s+(Parent Loop|=>This|Child Loop).*Depth=(\d+)", ln)
                if cm and cm.group(1) == "=>This":
                    depth = int(cm.group(2))
                if "scratch_load" in ln or "scratch_store" in ln:
                    kind = "store" if "scratch_store" in ln else "load"
                    per.setdefault((depth, kind), []).append(label)
                    if kind == "store":
                        nstores += 1
                    else:
                        nloads += 1
            print(f"\n{name}: {nstores} scratch_store, {nloads} scratch_load instructions")
            for (d, kind), labels in sorted(per.items()):
                uniq = sorted(set(labels), key=lambda s: [int(x) for x in re.findall(r"\d+", s)])
                print(f"  depth {d}: {len(labels):3d} scratch_{kind:5s} in blocks {', '.join(uniq[:12])}{' ...' if len(uniq) > 12 else ''}")


if __name__ == "__main__":
    main()
