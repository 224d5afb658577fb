#!/bin/bash
# tools/abn.sh for runs whose figure is not level 1's: prints value, kernel ms, ratio, bad packets for any bench arguments
#   tools/abn2.sh "lib1 lib2 ..." --level 2 [--gen mix]
R=${GRAFT_REPO_ROOT:-$(pwd)}
LIBS=$1; shift
for r in 1 2; do
  for L in $LIBS; do
    echo -n "$L: "
    ZZFLATE_AMD_LIB=$R/$L timeout -k 5 120 python $R/bench.py --steps 5 --no-cpu --no-extra "$@" 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'], d['ratio'], d['check']['device_inflate']['bad'])"
  done
done
