#!/bin/bash
# GPU box: kernel trace of tools/attn_fp8_bench.py (pre-pass vs main kernel vs the bf16 kernels)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fp8prof -- python3 tools/attn_fp8_bench.py > gpurun_out/fp8prof.log 2>&1
st=$(ls gpurun_out/fp8prof/*/*kernel_stats.csv | head -1)
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$st")))[:8]:
    print(f"{r['Name'][:90]:90s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:10.1f} us")
PY
cp $st gpurun_out/r03_attn_fp8_kernel_stats.csv; rm -rf gpurun_out/fp8prof
grep "TFLOP" gpurun_out/fp8prof.log
