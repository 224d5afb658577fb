#!/bin/bash
# Level 1: throughput and L2<->fabric fetch traffic against resident workgroups per CU (dynamic LDS pads the workgroup).
# The question behind it: would staging a packet's window in LDS pay? A staged window costs LDS, i.e.
# resident packets: 8 KiB of window = six workgroups per CU instead of nine, the whole 32 KiB = three. The padded runs are the
# UPPER bound of such a kernel (they pay the occupancy and do none of the staging work), and their FETCH_SIZE shows how much of
# the candidate traffic is L2 capacity (288 resident packets x 32 KiB per XCD against 4 MiB).
#   bash tools/traffic_vs_occupancy.sh > gpurun_out/traffic_vs_occupancy.txt     (on a GPU box, from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for pad in 0 2500 9000 22000 36000; do
  wg=$((163840 / (17932 + pad)))
  out=$(ZZFLATE_L1_PAD_LDS=$pad timeout -k 10 200 python3 $R/bench.py --steps 5 --no-cpu --no-extra --no-sequential 2>&1 | tail -1)
  val=$(echo "$out" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])")
  rm -rf /tmp/tvo_$pad
  ZZFLATE_L1_PAD_LDS=$pad timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/tvo_$pad -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-extra --no-sequential > /tmp/tvo_$pad.log 2>&1
  f=$(python3 - <<PY
import csv, glob, collections
per = collections.defaultdict(float)
for fn in glob.glob('/tmp/tvo_$pad/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        if 'k_encode_l1p' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE':
            per[(fn, r['Dispatch_Id'])] += float(r['Counter_Value'])     # one row per XCD: summed (tools/summarize_pmc.py does the same)
n = len(per)
# FETCH_SIZE is in KiB and counts a 128-byte request as 64 on gfx950 (MI355X_MICROARCH.md): x 2
print(round(2 * sum(per.values()) / max(n, 1) * 1024 / 1e9, 2) if n else 'n/a', n)
PY
)
  echo "pad $pad B  workgroups/CU $wg  whole-call GB/s, kernel ms: $val  fetched GB per GiB launch (2 x FETCH_SIZE), dispatches: $f"
done
