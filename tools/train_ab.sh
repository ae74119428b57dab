cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5k
for cfg in "r4paths:AG_WS_ROUTE=0" "auto:"; do
  tag=${cfg%%:*}; ev=${cfg#*:}
  for part in auto 0; do
    echo "== $tag partition=$part" >> gpurun_out/r5k/train.txt
    if [ -n "$ev" ]; then export $ev; else unset AG_WS_ROUTE; fi
    AG_TRAIN_PARTITION=$part STEPS=36 python tools/train_step_bench.py vit_base duo_bert_base froyo_vit_base 2>/dev/null >> gpurun_out/r5k/train.txt
  done
done
unset AG_WS_ROUTE
AG_BENCH_SKIP=sweep,fp32,vendor,configs,ledger python bench.py --attr-batch 0 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['secondary']['train_explainer_step']; print('skip_all', t['value'], t['one_stream_value'], t['graph_replay_value'])" >> gpurun_out/r5k/train.txt
AG_BENCH_SKIP=sweep,fp32,vendor,ledger python bench.py --attr-batch 0 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['secondary']['train_explainer_step']; print('with_configs', t['value'], t['one_stream_value'], t['graph_replay_value'])" >> gpurun_out/r5k/train.txt
cat gpurun_out/r5k/train.txt
