#!/usr/bin/env python3
"""Summarise the rocprofv3 passes of tools/collect_profiles.sh.

    python3 tools/summarize_pmc.py gpurun_out/prof/<tag> <tag> [--dest profiles]

Writes <dest>/<tag>_pmc.json (per-kernel mean of every counter, per dispatch), <dest>/<tag>_kernel_stats_l{0,1}.csv
(the --stats kernel table) and <dest>/traffic.json: HBM bytes per launch of the dominant kernels, corrected as
MI355X_MICROARCH.md prescribes for gfx950 -- FETCH_SIZE counts 128-byte requests as 64, so
traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (both counters are in KiB).
"""
import argparse, csv, glob, hashlib, json, os, re, shutil, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash():
    """SHA-256 over the kernel sources: bench.py reports profiles/traffic.json only while it still describes them"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "zzflate_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def short(name):
    """kernel name without return type and parameter list; template arguments stay (k_encode_l2_t<0u, false> is level 2,
    k_encode_l2_t<32768u, true> the encode kernel of the extended levels)"""
    name = re.sub(r"^void ", "", name.strip('"'))
    depth = 0
    for i, ch in enumerate(name):
        if ch == "<": depth += 1
        elif ch == ">": depth -= 1
        elif ch == "(" and depth == 0: return name[:i]
    return name


def counters(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = collections.defaultdict(float)
        for row in csv.DictReader(open(f)):
            key = (row.get("Dispatch_Id"), short(row["Kernel_Name"]), row["Counter_Name"])
            per_dispatch[key] += float(row["Counter_Value"])     # one row per XCD/instance: sum them
        for (_, k, c), v in per_dispatch.items():
            acc[k][c].append(v)
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src"); ap.add_argument("tag"); ap.add_argument("--dest", default="profiles")
    ap.add_argument("--git-sha", default="unknown", help="commit the profiled tree was built from")
    a = ap.parse_args()
    os.makedirs(a.dest, exist_ok=True)
    out = {}
    for sub in sorted(os.listdir(a.src)):
        p = os.path.join(a.src, sub)
        if os.path.isdir(p) and not sub.startswith("stats"):
            out[sub] = counters(p)
    json.dump(out, open(os.path.join(a.dest, f"{a.tag}_pmc.json"), "w"), indent=1, sort_keys=True)
    for lv in ("l0", "l1"):
        for f in glob.glob(os.path.join(a.src, f"stats_{lv}", "**", "*kernel_stats.csv"), recursive=True):
            shutil.copy(f, os.path.join(a.dest, f"{a.tag}_kernel_stats_{lv}.csv"))
    traffic = {}
    def t(fetch, write, kern):
        f = out.get(fetch, {}).get(kern, {}).get("FETCH_SIZE"); w = out.get(write, {}).get(kern, {}).get("WRITE_SIZE")
        return int((2 * f + w) * 1024) if f is not None and w is not None else None
    traffic["level1_text_1024MiB"] = t("fetch_l1", "write_l1", "zz::k_encode_l1p")
    traffic["level2_text_1024MiB"] = t("fetch_l1", "write_l1", "zz::k_encode_l2_t<0u, false, true>")
    traffic["level0_random_1024MiB"] = t("fetch_l0", "write_l0", "zz::k_encode_l0")
    # wave-level instructions per launch (SQ_INSTS_*): what the issue-rate bound in bench.py's roofline is computed from
    def insts(run, kern):
        c = out.get(run, {}).get(kern, {})
        keys = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_VMEM")
        return int(sum(c[k] for k in keys)) if all(k in c for k in keys) else None
    traffic["instructions_level1_text_1024MiB"] = insts("sq_l1", "zz::k_encode_l1p")
    traffic["instructions_level2_text_1024MiB"] = insts("sq_l1", "zz::k_encode_l2_t<0u, false, true>")
    # the CU's one scalar unit: SALU + branch instructions per launch (bench.py roofline.scalar)
    def scalar(run, kern):
        c = out.get(run, {}).get(kern, {})
        return int(c["SQ_INSTS_SALU"] + c["SQ_INSTS_BRANCH"]) if "SQ_INSTS_SALU" in c and "SQ_INSTS_BRANCH" in c else None
    traffic["scalar_instructions_level1_text_1024MiB"] = scalar("sq_l1", "zz::k_encode_l1p")
    traffic["scalar_instructions_level2_text_1024MiB"] = scalar("sq_l1", "zz::k_encode_l2_t<0u, false, true>")
    traffic["git_sha"] = a.git_sha
    traffic["source_sha256"] = source_hash()
    json.dump(traffic, open(os.path.join(a.dest, "traffic.json"), "w"), indent=1)
    print(json.dumps(traffic))


if __name__ == "__main__":
    main()
