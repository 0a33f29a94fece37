#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_map; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/dev/prof_map.py 300 > $O/run.log 2>&1 || exit 1
f=$(find $O -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'P'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:10]:
    print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,2), r['MinNs'])
P
