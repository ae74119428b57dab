#!/bin/bash
# A/B of environment-selected kernel variants on one box: bench line + FETCH_SIZE / WRITE_SIZE / L2 hit per GEMM class.
# usage (on the GPU box): tools/pmc_variants.sh "<VAR=val ...>" "<VAR=val ...>" ...   (empty string = defaults)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-secondary"
i=0
for v in "$@"; do
  i=$((i+1)); d=$R/gpurun_out/pmcv_$i; rm -rf $d; mkdir -p $d
  echo "=== variant $i: [$v]"
  for r in 1 2; do (export $v; python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']; print(d['value'], d['ms_per_step'], {n:x['avg_us'] for n,x in k.items() if 'gemm<bias' in n and 'f32' not in n})"); done
  (export $v; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d/fetch -- python3 $R/bench.py $ARGS > /dev/null 2>&1)
  (export $v; rocprofv3 --pmc WRITE_SIZE --output-format csv -d $d/write -- python3 $R/bench.py $ARGS > /dev/null 2>&1)
  (export $v; rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $d/tcc -- python3 $R/bench.py $ARGS > /dev/null 2>&1)
  python3 $R/tools/summarize_profile.py $d $d.json | grep "gemm_ring_kernel<[012], [12]"
  find $d -name "*.csv" -delete
done
