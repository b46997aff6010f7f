#!/bin/bash
# GPU box: per-kernel breakdown of one rank's step of a multi-GPU plan simulated on ONE GPU (tools/plan_sim.py <plan>),
# rocprofv3 kernel trace + stats -> gpurun_out/<tag>_<plan>_kernel_stats.csv.   usage: tools/plan_prof.sh <tag> <plan>...
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for plan in "$@"; do
  name=${plan/:/_}
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_${name}_trace -- python3 tools/plan_sim.py $plan > gpurun_out/${tag}_${name}.log 2>&1
  st=$(ls gpurun_out/${tag}_${name}_trace/*/*kernel_stats.csv | head -1)
  cp $st gpurun_out/${tag}_${name}_kernel_stats.csv
  rm -rf gpurun_out/${tag}_${name}_trace
  grep "ms/step" gpurun_out/${tag}_${name}.log
  python3 - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/${tag}_${name}_kernel_stats.csv")))
for r in rows[:16]:
    print(f"{r['Name'][:88]:88s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:9.1f} us  {float(r['TotalDurationNs'])/1e6:8.2f} ms {float(r['Percentage']):5.1f}%")
PY
done
