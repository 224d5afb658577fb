#!/bin/bash
# Collect the per-round profile set on a GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r02 $(git rev-parse --short HEAD)
# Kernel-trace/stats and every --pmc group run as separate rocprofv3 passes (never combined); the program is
# `python3 bench.py ...` directly after `--`. Raw output lands in gpurun_out/prof/<tag>/; tools/summarize_pmc.py
# turns it into profiles/<tag>_pmc.json, profiles/<tag>_kernel_stats.csv and profiles/traffic.json.
set -eo pipefail
TAG=${1:-r03}
SHA=${2:-unknown}          # commit of the tree being profiled (the GPU box has no .git: pass $(git rev-parse --short HEAD))
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu --no-sequential --min-seconds 0"
run() { name=$1; shift; echo "[collect] $name"; timeout -k 10 280 rocprofv3 "$@" > "$OUT/$name.log" 2>&1; }
run stats_l1  --kernel-trace --stats --output-format csv -d "$OUT/stats_l1" -- $B --steps 5 --warmup 1
run fetch_l1  --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch_l1" -- $B --steps 2 --warmup 1
run write_l1  --pmc WRITE_SIZE --output-format csv -d "$OUT/write_l1" -- $B --steps 2 --warmup 1
run sq_l1     --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY \
              --output-format csv -d "$OUT/sq_l1" -- $B --steps 2 --warmup 1
# which unit is busy: the scalar unit's, the VALU's and the LDS's active cycles beside the wave cycles (SQ counters
# count quad-cycles: MI355X_MICROARCH.md); two passes of <= 8 SQ counters; a name this ROCm does not know fails that pass only
run busy1_l1  --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES SQ_WAVES \
              --output-format csv -d "$OUT/busy1_l1" -- $B --steps 2 --warmup 1 --no-extra || echo "[collect] busy1_l1 failed (counter names?)"
run busy2_l1  --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE \
              --output-format csv -d "$OUT/busy2_l1" -- $B --steps 2 --warmup 1 --no-extra || echo "[collect] busy2_l1 failed (counter names?)"
run stats_l0  --kernel-trace --stats --output-format csv -d "$OUT/stats_l0" -- $B --steps 5 --warmup 1 --level 0 --gen random --no-extra
run fetch_l0  --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch_l0" -- $B --steps 2 --warmup 1 --level 0 --gen random --no-extra
run write_l0  --pmc WRITE_SIZE --output-format csv -d "$OUT/write_l0" -- $B --steps 2 --warmup 1 --level 0 --gen random --no-extra
cd "$R" && python3 tools/summarize_pmc.py "$OUT" "$TAG" --dest "$R/gpurun_out/prof/${TAG}_summary" --git-sha "$SHA"
echo "[collect] done"
