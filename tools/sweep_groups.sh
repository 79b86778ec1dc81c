#!/bin/bash
# env-group / hardware-queue sweep of the bench
for q in 4 8; do for g in 3 4 6; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python bench.py --steps 150 --warmup 30 --no-cpu-baseline --groups $g 2>&1 | grep "^{" > /tmp/b.json
  python -c "import json; d=json.load(open('/tmp/b.json')); print('queues', $q, 'groups', d['config']['groups'], round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['achieved']), round(d['roofline']['avg_launch_ms'],3))"
done; done
