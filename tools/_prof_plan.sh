cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/plan8 -- python3 tools/plan_sim.py interleave:8 > gpurun_out/plan8.log 2>&1 < /dev/null
grep "ms/step" gpurun_out/plan8.log
f=$(find gpurun_out/plan8 -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then head -16 "$f" | cut -c1-150; cp "$f" gpurun_out/plan8_kernel_stats.csv; fi
