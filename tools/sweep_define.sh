#!/bin/bash
# Build variants of the library with one macro set to each given value and bench them (run on a GPU box):
#   tools/sweep_define.sh ZZ_L1_PREFETCH 128 384 1024 -- --steps 10 --no-cpu --no-extra
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
M=$1; shift
VALS=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do VALS+=("$1"); shift; done
shift || true
mkdir -p $R/gpurun_out/sweep
for v in "${VALS[@]}"; do
  lib=$R/gpurun_out/sweep/libzz_${M}_${v}.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -D${M}=${v} -o $lib $R/zzflate_amd/csrc/zz_api.hip $R/zzflate_amd/csrc/zz_cxx_shim.cpp
  echo "## ${M}=${v}"
  ZZFLATE_AMD_LIB=$lib timeout -k 5 200 python $R/bench.py "$@" 2>&1 | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["kernel_ms"], (d.get("level2") or {}).get("value"))'
done
