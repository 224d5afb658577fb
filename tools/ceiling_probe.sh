#!/bin/bash
# Ceiling probe (run on a GPU box): level-1 throughput against the number of parsing wavefronts a CU holds.
#   * the shipped kernel with its workgroups padded (ZZFLATE_L1_PAD_LDS): 9, 8, 7, 6, 5 workgroups per CU;
#   * the same kernel with a (wrong) 12- and 10-bit hash, i.e. 8 / 2 KiB tables: sixteen workgroups per CU, the
#     wave-slot limit. The output is no longer the reference's; only the kernel time means anything.
# What a design with more than nine parsing wavefronts per CU could reach at most, at the shipped instruction mix.
R=${GRAFT_REPO_ROOT:-$(pwd)}
GEN=${1:-text}
line() { python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], "GB/s  kernel_ms", d["roofline"]["kernel_ms"])'; }
for pad in 0 2400 5000 9000 14500; do
  echo -n "13 bits, pad $pad ($((163840 / (17920 + pad))) workgroups per CU): "
  ZZFLATE_L1_PAD_LDS=$pad timeout -k 10 200 python3 $R/bench.py --steps 5 --no-cpu --no-extra --gen $GEN 2>&1 | line
done
for bits in 12 10; do
  D=/tmp/probe_$bits
  rm -rf $D; mkdir -p $D/zzflate_amd; cp -r $R/zzflate_amd/csrc $D/zzflate_amd/; cp -r $R/include $D/
  sed -i "s/#define ZZ_HASH_BITS 13/#define ZZ_HASH_BITS $bits/" $D/zzflate_amd/csrc/zz_common.h
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o $D/lib.so $D/zzflate_amd/csrc/zz_api.hip $D/zzflate_amd/csrc/zz_cxx_shim.cpp || exit 1
  echo -n "$bits bits (16 workgroups per CU): "
  ZZFLATE_AMD_LIB=$D/lib.so ZZ_BENCH_NO_CHECK=1 timeout -k 10 200 python3 $R/bench.py --steps 5 --no-cpu --no-extra --gen $GEN 2>&1 | line
done
