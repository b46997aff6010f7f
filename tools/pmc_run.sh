#!/bin/bash
# usage: tools/pmc_run.sh <outdir> <counters...> -- <python args>   (GPU box; counters in their own pass, no traces mixed in)
out=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc "${ctrs[@]}" --output-format csv -d gpurun_out/$out -- python3 "$@" > gpurun_out/$out.log 2>&1
python3 - <<PY
import csv, glob, collections
rows = []
for f in glob.glob("gpurun_out/$out/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for r in rows:
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "gemm" in k or "attn" in k or "Cijk" in k or "conv" in k:
        print(k)
        for c, v in sorted(d.items()):
            print(f"    {c:32s} {v / cnt[(k, c)]:16.1f}  (avg over {cnt[(k, c)]} dispatches)")
PY
