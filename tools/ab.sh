#!/bin/bash
# A/B two builds of the library on the same box, alternating: tools/ab.sh ab_libs/old.so ab_libs/new.so [bench args]
# prints: whole-call GB/s, encode kernel ms, level-2 GB/s (when the line carries one)
R=${GRAFT_REPO_ROOT:-$(pwd)}
A=$1; B=$2; shift 2
for r in 1 2 3; do
  for L in $A $B; do
    echo -n "$L: "
    ZZFLATE_AMD_LIB=$R/$L timeout -k 5 200 python $R/bench.py --steps 10 --no-cpu "$@" 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'], (d.get('level2') or {}).get('value'))"
  done
done
