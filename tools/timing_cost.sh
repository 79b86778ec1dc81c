#!/bin/bash
for rep in 1 2; do for flag in "" "--no-kernel-timing"; do
  timeout -k 10 300 python bench.py --steps 200 --warmup 30 --no-cpu-baseline $flag 2>&1 | grep "^{" > /tmp/b.json
  python -c "import json; d=json.load(open('/tmp/b.json')); print('$flag', round(d['value']), round(d['ms_per_step'],3))"
done; done
