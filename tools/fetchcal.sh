#!/bin/bash
# usage: tools/fetchcal.sh <tag>  -- FETCH_SIZE / read-request counters of tools/ubench/fetchcal (known byte counts)
TAG=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/fetchcal_${TAG}/p$i -- $R/tools/ubench/fetchcal > $R/gpurun_out/fetchcal_${TAG}_p$i.log 2>&1 || tail -3 $R/gpurun_out/fetchcal_${TAG}_p$i.log
done
python3 - <<PY | tee $R/gpurun_out/${TAG}_fetchcal.txt
import csv, glob, os
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("$R/gpurun_out/fetchcal_${TAG}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
known = 1 << 30
for k in sorted(acc):
    if "k_" not in k: continue
    kb = (known // 8256) * 8256 if "gather" in k else known
    line = f"{k:60s} known {kb/1e6:9.1f} MB"
    for c in sorted(acc[k]):
        v = sum(acc[k][c]) / len(acc[k][c])
        line += f"  {c} {v:.1f}"
        if c == "FETCH_SIZE": line += f" (= {v*1024/1e6:.1f} MB, known/FETCH = {kb/(v*1024):.3f})"
    print(line)
PY
