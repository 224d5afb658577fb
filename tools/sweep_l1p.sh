#!/bin/bash
# Build variants of the library with sets of -D flags and bench them on one box (run on a GPU box):
#   tools/sweep_l1p.sh "-DZZ_L1P_LEN_LATE=0 -DZZ_L1P_TOK_EARLY=0" "-DZZ_L1P_LEN_LATE=1 -DZZ_L1P_TOK_EARLY=0" ... [-- gens]
R=${GRAFT_REPO_ROOT:-$(pwd)}
VARS=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do VARS+=("$1"); shift; done
shift || true
GENS=${@:-text mix}
mkdir -p /tmp/sweep
i=0
for v in "${VARS[@]}"; do
  lib=/tmp/sweep/lib_$i.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $v -o $lib $R/zzflate_amd/csrc/zz_api.hip $R/zzflate_amd/csrc/zz_cxx_shim.cpp 2>/dev/null || { echo "build failed: $v"; exit 1; }
  i=$((i+1))
done
for r in 1 2; do
  i=0
  for v in "${VARS[@]}"; do
    for g in $GENS; do
      echo -n "[$v] $g: "
      ZZFLATE_AMD_LIB=/tmp/sweep/lib_$i.so timeout -k 5 200 python3 $R/bench.py --steps 5 --no-cpu --no-extra --gen $g 2>&1 | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "GB/s kernel_ms", d["roofline"]["kernel_ms"], "bad", d["check"]["device_inflate"]["bad"])'
    done
    i=$((i+1))
  done
done
