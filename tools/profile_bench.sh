#!/bin/bash
# GPU box: rocprofv3 kernel-trace stats + HBM traffic counters (separate passes) of the bench command -> gpurun_out/<tag>_*
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vae --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace -- python3 $CMD > gpurun_out/${tag}_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -- python3 $CMD > gpurun_out/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -- python3 $CMD > gpurun_out/${tag}_write.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/${tag}_sq -- python3 $CMD > gpurun_out/${tag}_sq.log 2>&1
python3 - <<PY
import csv, glob, collections, json
tag = "$tag"
out = {}
st = glob.glob(f"gpurun_out/{tag}_trace/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(st[0])))
out["kernel_stats"] = [{"name": r["Name"][:110], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                        "total_ms": float(r["TotalDurationNs"]) / 1e6, "pct": float(r["Percentage"])} for r in rows[:14]]
pm = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("fetch", "write", "sq"):
    for f in glob.glob(f"gpurun_out/{tag}_{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "(anonymous namespace)::" in k and "at::native" not in k:   # every kernel of libframeino_hip.so
                pm[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out["pmc_avg_per_dispatch"] = {k: {c: sum(v) / len(v) for c, v in d.items()} | {"dispatches": len(next(iter(d.values())))}
                               for k, d in pm.items()}
json.dump(out, open(f"gpurun_out/{tag}_summary.json", "w"), indent=1)
for l in open(f"gpurun_out/{tag}_trace.log"):
    if l.startswith("{"):
        open(f"gpurun_out/{tag}_bench_line.json", "w").write(l)
print(json.dumps(out["pmc_avg_per_dispatch"], indent=1)[:3000])
PY

# ---- Wan VAE decode + encode (once-per-clip stages): kernel trace, then the SQ counters in their own pass ----
VCMD="tools/vae_bench.py --encode"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_vae_trace -- python3 $VCMD > gpurun_out/${tag}_vae_trace.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/${tag}_vae_sq -- python3 $VCMD > gpurun_out/${tag}_vae_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_vae_fetch -- python3 $VCMD > gpurun_out/${tag}_vae_fetch.log 2>&1
python3 - <<PY
import csv, glob, collections, json
tag = "$tag"
out = {"decode_encode_log": [l.strip() for l in open(f"gpurun_out/{tag}_vae_trace.log") if l.startswith(("decode", "encode"))]}
st = glob.glob(f"gpurun_out/{tag}_vae_trace/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(st[0])))
out["kernel_stats"] = [{"name": r["Name"][:110], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                        "total_ms": float(r["TotalDurationNs"]) / 1e6, "pct": float(r["Percentage"])} for r in rows[:10]]
pm = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for d in ("vae_sq", "vae_fetch"):
    for f in glob.glob(f"gpurun_out/{tag}_{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "(anonymous namespace)::" in k and "at::native" not in k:
                pm[k[:90]][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[(k[:90], r["Counter_Name"])] += 1
out["pmc_sum_over_dispatches"] = {k: dict(v) | {"dispatches": max(cnt[(k, c)] for c in v)} for k, v in pm.items()}
for k, v in out["pmc_sum_over_dispatches"].items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and v.get("GRBM_GUI_ACTIVE"):
        v["mfma_busy_frac"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8 * 1024)
json.dump(out, open(f"gpurun_out/{tag}_vae_summary.json", "w"), indent=1)
cp = st[0]
import shutil; shutil.copy(cp, f"gpurun_out/{tag}_vae_kernel_stats.csv")
print(json.dumps({k: v.get("mfma_busy_frac") for k, v in out["pmc_sum_over_dispatches"].items()}, indent=1)[:1500])
PY
st=$(ls gpurun_out/${tag}_trace/*/*kernel_stats.csv | head -1); cp $st gpurun_out/${tag}_kernel_stats.csv
# raw traces are large (gpurun merges at most 64 MiB back): keep the summaries only
rm -rf gpurun_out/${tag}_trace gpurun_out/${tag}_fetch gpurun_out/${tag}_write gpurun_out/${tag}_sq gpurun_out/${tag}_vae_trace gpurun_out/${tag}_vae_sq gpurun_out/${tag}_vae_fetch
ls -la gpurun_out/
