#!/bin/bash
# Which unit of a CU is busy (run on a GPU box): SQ busy/active counters of the level-1 kernel, for the shipped build
# (nine workgroups per CU) and for the ceiling-probe build with a 10-bit hash (sixteen per CU, output not the reference's).
#   bash tools/units_probe.sh [gen]        -> gpurun_out/units/*.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
GEN=${1:-text}
OUT=$R/gpurun_out/units
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_avail.txt 2>&1 || true
D=/tmp/probe_10
if [ ! -f $D/lib.so ]; then
  rm -rf $D; mkdir -p $D/zzflate_amd; cp -r $R/zzflate_amd/csrc $D/zzflate_amd/; cp -r $R/include $D/
  sed -i "s/#define ZZ_HASH_BITS 13/#define ZZ_HASH_BITS 10/" $D/zzflate_amd/csrc/zz_common.h
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o $D/lib.so $D/zzflate_amd/csrc/zz_api.hip $D/zzflate_amd/csrc/zz_cxx_shim.cpp || exit 1
fi
B="python3 $R/bench.py --no-cpu --no-extra --steps 2 --warmup 1 --gen $GEN"
pass() {  # name lib counters...
  name=$1; lib=$2; shift 2
  echo "[units] $name"
  ZZFLATE_AMD_LIB=$lib timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- $B > $OUT/$name.log 2>&1 || echo "[units] $name failed"
}
for v in 13:$R/zzflate_amd/libzzflate_amd.so 10:$D/lib.so; do
  bits=${v%%:*}; lib=${v#*:}
  pass a_$bits $lib SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM
  pass b_$bits $lib SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU
  pass c_$bits $lib SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_IFETCH_LEVEL
done
cd $R && python3 tools/summarize_pmc.py $OUT units --dest $OUT/summary > /dev/null 2>&1
python3 - <<EOF
import json
d = json.load(open("$OUT/summary/units_pmc.json"))
for p in sorted(d):
    for k, cs in d[p].items():
        if "k_encode_l1" in k:
            print(p, k, json.dumps({c: round(v) for c, v in cs.items()}))
EOF
