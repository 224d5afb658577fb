#!/bin/bash
# diagnostic: level-1 throughput vs resident workgroups per CU (extra dynamic LDS pads the workgroup)
for pad in 0 2500 5000 9000 14500 22000 36000; do
  out=$(ZZFLATE_L1_PAD_LDS=$pad timeout -k 10 200 python bench.py --steps 3 --no-cpu --no-extra 2>&1 | tail -1)
  echo "$out" | PAD=$pad python -c "import sys,json,os; d=json.loads(sys.stdin.read()); p=int(os.environ['PAD']); print('pad', p, 'WGs/CU', 163840//(17932+p), d['value'], 'GB/s kernel_ms', d['roofline']['kernel_ms'])"
done
