#!/bin/bash
# GPU box: kernel trace of the Wan2.2-5B bench step with MXFP8 linears + fp8 attention operands (bench.py --mxfp8 --fp8-attention)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/wfp8 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vae --no-secondary --mxfp8 --fp8-attention > gpurun_out/wfp8.log 2>&1
st=$(ls gpurun_out/wfp8/*/*kernel_stats.csv | head -1)
cp $st gpurun_out/r03_wan_fp8_kernel_stats.csv; rm -rf gpurun_out/wfp8
python3 - <<PY
import csv
rows = [r for r in csv.DictReader(open("gpurun_out/r03_wan_fp8_kernel_stats.csv")) if "at::" not in r["Name"] and "mfma_peak" not in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print(f"{r['Name'][:96]:96s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:9.1f} us {float(r['TotalDurationNs'])/1e6/3:8.2f} ms/step")
print("library kernels per step:", tot / 1e6 / 3, "ms")
PY
grep -o '"ms_per_step": [0-9.]*' gpurun_out/wfp8.log | head -1
