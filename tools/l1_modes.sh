#!/bin/bash
# diagnostic: level-1 throughput of the two kernels alone and together (ZZFLATE_L1_MODE), and with LDS padding that
# changes how many workgroups of either kind a CU holds. Usage: tools/l1_modes.sh [extra bench args]
run() {
  out=$(env "$@" timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu --no-extra $EXTRA 2>&1 | tail -1)
  echo "$out" | TAG="$*" python -c "import sys,json,os; d=json.loads(sys.stdin.read()); print(os.environ['TAG'], '->', d['value'], 'GB/s  kernel_ms', d['roofline']['kernel_ms'], 'ratio', d['ratio'], 'bad', d['check']['device_inflate']['bad'])" || echo "$* -> FAILED: $out"
}
EXTRA="$*"
run ZZFLATE_L1_MODE=lds
run ZZFLATE_L1_MODE=global
run ZZFLATE_L1_MODE=both
run ZZFLATE_L1_MODE=both ZZFLATE_L1G_PAD_LDS=3000
run ZZFLATE_L1_MODE=both ZZFLATE_L1G_PAD_LDS=8000
run ZZFLATE_L1_MODE=both ZZFLATE_L1_PAD_LDS=2500
run ZZFLATE_L1_MODE=global ZZFLATE_L1G_PAD_LDS=8000
run ZZFLATE_L1_MODE=global ZZFLATE_L1G_PAD_LDS=14000
