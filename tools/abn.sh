#!/bin/bash
# like ab.sh for any number of builds: tools/abn.sh "lib1 lib2 ..." [bench args]; three alternating rounds
R=${GRAFT_REPO_ROOT:-$(pwd)}
LIBS=$1; shift
for r in 1 2 3; do
  for L in $LIBS; do
    echo -n "$L: "
    ZZFLATE_AMD_LIB=$R/$L timeout -k 5 200 python $R/bench.py --steps 10 --no-cpu "$@" 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'], d['ratio'], d['check']['device_inflate']['bad'], (d.get('level2') or {}).get('value'))"
  done
done
