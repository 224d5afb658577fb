#!/bin/bash
# Timing probe (run on a GPU box): what a two-stage pipeline over ONE packet could reach at most. The build with
# -DZZ_L1_PIPE_PROBE runs two parsing wavefronts per packet on alternate blocks of 64 positions (block g + 1 probed and
# compared while block g is walked, two barriers per block) and leaves out everything a correct version would add:
# the cross-block same-hash resolution, the exchange, the carried match end. Its output is NOT a valid stream.
#   bash tools/pipe_probe.sh [gen ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=/tmp/pipe_probe
mkdir -p $D
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DZZ_L1_PIPE_PROBE -o $D/lib.so $R/zzflate_amd/csrc/zz_api.hip $R/zzflate_amd/csrc/zz_cxx_shim.cpp || exit 1
line() { python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], "GB/s  kernel_ms", d["roofline"]["kernel_ms"])'; }
for gen in ${@:-text mix logs}; do
  for r in 1 2; do
    echo -n "$gen shipped: "; timeout -k 10 200 python3 $R/bench.py --steps 5 --no-cpu --no-extra --gen $gen 2>&1 | line
    echo -n "$gen probe  : "; ZZFLATE_AMD_LIB=$D/lib.so timeout -k 10 200 python3 $R/bench.py --steps 5 --no-cpu --no-extra --gen $gen 2>&1 | line
  done
done
