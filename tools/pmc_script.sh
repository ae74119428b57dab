#!/bin/bash
# rocprofv3 PMC of a python dev script: tools/pmc_script.sh <script.py> <kernel-name-substring> "<counters pass 1>" "<counters pass 2>" ...
R=$GRAFT_REPO_ROOT; S=$1; K=$2; shift 2
cd /tmp && export TMPDIR=/tmp
i=0
for c in "$@"; do
  i=$((i+1)); d=$R/gpurun_out/pmcs_$i; rm -rf $d
  rocprofv3 --pmc $c --output-format csv -d $d -- python3 $R/$S > /dev/null 2>&1
  python3 - "$d" "$K" <<'PY'
import csv, glob, sys, collections
d, k = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if k in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({c: round(sum(v) / len(v), 1) for c, v in acc.items()}, "launches", {c: len(v) for c, v in acc.items()})
PY
  find $d -name "*.csv" -delete
done
