for tb in 8 16 32 64; do timeout 250 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --attr-batch 110 --train-batch $tb 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train-batch $tb', d['secondary']['train_explainer_step']['value'])"; done
