#!/bin/bash
# One-off: the unit-busy counters of k_encode_l1p and the instruction counts, two/three --pmc passes.
#   bash tools/busy_pass.sh r05
set -o pipefail
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof/${TAG}_busy
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu --no-sequential --no-extra --min-seconds 0 --steps 2 --warmup 1"
run() { name=$1; shift; echo "[busy] $name"; timeout -k 10 200 rocprofv3 "$@" > "$OUT/$name.log" 2>&1 || echo "[busy] $name failed: $(tail -2 $OUT/$name.log)"; }
run busy1_l1  --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES SQ_WAVES \
              --output-format csv -d "$OUT/busy1_l1" -- $B
run busy2_l1  --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE \
              --output-format csv -d "$OUT/busy2_l1" -- $B
run sq_l1     --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY \
              --output-format csv -d "$OUT/sq_l1" -- $B
cd "$R" && python3 - "$OUT" <<'PY'
import sys, json, os
sys.path.insert(0, "tools")
import summarize_pmc as s
out = {sub: s.counters(os.path.join(sys.argv[1], sub)) for sub in sorted(os.listdir(sys.argv[1])) if os.path.isdir(os.path.join(sys.argv[1], sub))}
keep = {sub: {k: v for k, v in ks.items() if "encode" in k} for sub, ks in out.items()}
json.dump(keep, open(os.path.join(sys.argv[1], "busy.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(keep, indent=1, sort_keys=True))
PY
