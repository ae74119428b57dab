#!/bin/bash
# masked forwards/s of the default workload at the small per-GPU batches of SURVEY C2 (B = 1, 4, 16) and the bench default
for b in 1 4 16 48; do timeout 250 python bench.py --steps 20 --warmup 5 --batch $b --no-cpu-baseline --attr-batch 0 --train-batch 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$b', d['value'], 'fwd/s', d['ms_per_step'], 'ms/step')"; done
