#!/bin/bash
# Everything the round's profile record of one step-kernel version consists of, in one GPU call:
#   bash tools/profile_version.sh r02 v30        (on the GPU box; writes gpurun_out/profiles_r02_v30/, copy into profiles/)
#     rNN_bench_vNN.json                  bench.py, unprofiled (the numbers the docs quote)
#     rNN_kernel_stats_vNN.csv            rocprofv3 --kernel-trace --stats of the same command
#     rNN_bench_vNN_under_rocprof.json    ... and what bench.py itself printed under the profiler
#     rNN_kernel_trace_vNN_step65536.json per-dispatch durations of sf_step_kernel at this workload's grid
#     rNN_pmc_traffic_vNN.json            FETCH_SIZE / WRITE_SIZE passes -> HBM bytes per launch (tools/pmc_report.py)
#     rNN_pmc_sq_vNN.txt                  SQ instruction mix, wait / active cycles, TA_BUSY (tools/pmc_step.sh)
#     step_kernel_latest.json             what bench.py replays as roofline.traffic / kernel_ms_rocprof
set -e
RND=$1; VER=$2
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles_${RND}_${VER}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="--steps 2000 --warmup 200 --no-cpu-baseline --numpy-api 0 --no-configs --steady-seconds 0.5"
python3 $R/bench.py --steps 12000 --warmup 200 > $OUT/${RND}_bench_${VER}.json 2> $OUT/bench.err
echo "bench done"
# the traced run launches its timed blocks as HIP graphs (512 launches per replay): launched one by one under the profiler the
# kernels are no longer back to back (its per-dispatch work is on the host) and every one starts on an idle chip
BT="--steps 512 --warmup 100 --repeats 12 --launch graph --timed-only"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py $BT > $OUT/${RND}_bench_${VER}_under_rocprof.json 2>> $OUT/bench.err
echo "kernel trace done"
SF_PMC_CALIB=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pf -- python3 $R/bench.py $B --rollout-k 0 --image-envs 0 > /dev/null 2>> $OUT/bench.err
SF_PMC_CALIB=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pw -- python3 $R/bench.py $B --rollout-k 0 --image-envs 0 > /dev/null 2>> $OUT/bench.err
echo "pmc traffic done"
cd $R
python3 tools/pmc_report.py $OUT/pf $OUT/pw --envs 65536 --out $OUT/${RND}_pmc_traffic_${VER}.json > /dev/null
python3 - "$OUT" "$RND" "$VER" <<'PY'
import csv, glob, json, os, sys
out, rnd, ver = sys.argv[1:4]
st = glob.glob(os.path.join(out, "kt", "**", "*kernel_stats.csv"), recursive=True)
if st:
    open(os.path.join(out, "%s_kernel_stats_%s.csv" % (rnd, ver)), "w").write(open(st[0]).read())
tr = glob.glob(os.path.join(out, "kt", "**", "*kernel_trace.csv"), recursive=True)[0]
per = {}
for r in csv.DictReader(open(tr)):
    if "sf_step_kernel" in r["Kernel_Name"]:
        k = "%s grid %s" % (r["Kernel_Name"].split("(")[0], r.get("Grid_Size") or r["Grid_Size_X"])
        per.setdefault(k, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rep = {}
for k, v in per.items():
    v.sort()
    rep[k] = {"calls": len(v), "mean_ns": sum(v) / len(v), "median_ns": v[len(v) // 2], "min_ns": v[0], "max_ns": v[-1]}
rep["note"] = ("per-dispatch durations (End - Start timestamps) from the same rocprofv3 --kernel-trace --stats run as the "
               "kernel_stats csv, whose sf_step_kernel row mixes this workload's launches with those of bench.py's other legs; "
               "command: rocprofv3 --kernel-trace --stats -- python bench.py --steps 512 --warmup 100 --repeats 8 --launch graph --no-cpu-baseline")
json.dump(rep, open(os.path.join(out, "%s_kernel_trace_%s_step65536.json" % (rnd, ver)), "w"), indent=1)
# (a split launch -- sf_step_kernel<..., 1256>, 131 072 threads for 65 536 envs -- where the library steps that way)
main = [k for k in rep if k.endswith("grid 131072") and ", false, 1, false, 1256>" in k] or \
       [k for k in rep if k.endswith("grid 65536") and ", false, 1, false, 256>" in k]
t = json.load(open(os.path.join(out, "%s_pmc_traffic_%s.json" % (rnd, ver))))
sys.path.insert(0, os.getcwd())
from spacefortress_amd import _lib
latest = {"version": "%s_%s" % (rnd, ver), "sf_build_id": _lib.lib().sf_build_id().decode(),
          "workload": t["workload"], "kernel": "sf_step_kernel",
          "traffic_bytes_per_launch": t["traffic_bytes_per_launch"], "read_bytes_per_launch": t["read_bytes_per_launch"],
          "write_bytes_per_launch": t["write_bytes_per_launch"], "pmc_file": "%s_pmc_traffic_%s.json" % (rnd, ver),
          "kernel_ms_rocprof": rep[main[0]]["mean_ns"] * 1e-6 if main else None,
          "kernel_ms_rocprof_median": rep[main[0]]["median_ns"] * 1e-6 if main else None,
          "trace_file": "%s_kernel_trace_%s_step65536.json" % (rnd, ver),
          "note": "what bench.py replays as roofline.traffic / roofline.kernel_ms_rocprof (marked stale when the loaded library's "
                  "sf_build_id is not this one); rewritten by tools/profile_version.sh with every profiled kernel version"}
json.dump(latest, open(os.path.join(out, "step_kernel_latest.json"), "w"), indent=1)
print(json.dumps(latest))
PY
bash tools/pmc_step.sh gpurun_out/profiles_${RND}_${VER}/sq > /dev/null
cp $OUT/sq/pmc_step.txt $OUT/${RND}_pmc_sq_${VER}.txt
rm -rf $OUT/kt $OUT/pf $OUT/pw $OUT/sq
ls $OUT
