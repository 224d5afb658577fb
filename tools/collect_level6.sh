#!/bin/bash
# Counters of the extended level 6 on the mixed corpus (run through gpurun from the repo root): separate rocprofv3 passes,
# the program directly after `--`; tools/summarize_pmc.py reduces them to gpurun_out/prof/<tag>_summary/<tag>_pmc.json
set -eo pipefail
TAG=${1:-r03_level6}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu --no-extra --level 6 --gen mix"
run() { name=$1; shift; echo "[collect] $name"; timeout -k 10 280 rocprofv3 "$@" > "$OUT/$name.log" 2>&1; }
run stats_l1  --kernel-trace --stats --output-format csv -d "$OUT/stats_l1" -- $B --steps 3 --warmup 1
run fetch_l6  --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch_l6" -- $B --steps 2 --warmup 1
run write_l6  --pmc WRITE_SIZE --output-format csv -d "$OUT/write_l6" -- $B --steps 2 --warmup 1
run sq_l6     --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY \
              --output-format csv -d "$OUT/sq_l6" -- $B --steps 2 --warmup 1
cd "$R" && python3 tools/summarize_pmc.py "$OUT" "$TAG" --dest "$R/gpurun_out/prof/${TAG}_summary" --git-sha "${2:-unknown}"
echo "[collect] done"
