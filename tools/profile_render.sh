#!/bin/bash
# The round's profile record of the render kernel, in one GPU call:  bash tools/profile_render.sh r02 v16   (on the GPU box;
# writes gpurun_out/profiles_render_r02_v16/, copy into profiles/)
#   rNN_render_probe_vNN.txt          tools/image_probe.py 16384 300 image, unprofiled (step+render and render alone, HIP events)
#   rNN_render_kernel_stats_vNN.csv   rocprofv3 --kernel-trace --stats of the same command
#   rNN_pmc_render_vNN.txt            SQ instruction mix, wait / active cycles (tools/pmc_render.sh)
#   rNN_pmc_render_traffic_vNN.json   FETCH_SIZE / WRITE_SIZE passes -> HBM-side bytes per launch of the frame kernel
#   image_kernel_latest.json          what bench.py replays as image_obs.kernel_ms_rocprof / roofline.traffic (with the library's sf_build_id)
set -e
RND=$1; VER=$2
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles_render_${RND}_${VER}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/image_probe.py 16384 300 image > $OUT/${RND}_render_probe_${VER}.txt 2>/dev/null
echo "probe done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/tools/image_probe.py 16384 300 image > /dev/null 2>&1
ST=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
cp $ST $OUT/${RND}_render_kernel_stats_${VER}.csv
TR=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
cd $R
python3 - "$TR" "$OUT" "$RND" "$VER" <<'PY'
import csv, json, os, sys
tr, out, rnd, ver = sys.argv[1:5]
sys.path.insert(0, os.getcwd())
from spacefortress_amd import _lib
d = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(tr))
           if "sf_render_kernel<true>" in r["Kernel_Name"])
rep = {"version": "%s_%s" % (rnd, ver), "sf_build_id": _lib.lib().sf_build_id().decode(), "envs": 16384,
       "kernel": "sf_render_kernel<true>", "calls": len(d), "kernel_ms_rocprof": sum(d) / len(d) * 1e-6,
       "kernel_ms_rocprof_median": d[len(d) // 2] * 1e-6, "stats_file": "%s_render_kernel_stats_%s.csv" % (rnd, ver),
       "note": "rocprofv3 --kernel-trace of tools/image_probe.py 16384 300 image (youturn, random actions, 400 steps in): per-dispatch "
               "End - Start of sf_render_kernel<true>; what bench.py replays as image_obs.kernel_ms_rocprof"}
json.dump(rep, open(os.path.join(out, "image_kernel_latest.json"), "w"), indent=1)
print(json.dumps(rep))
PY
rm -rf $OUT/kt
echo "kernel trace done"
# HBM-side traffic of the frame kernel: FETCH_SIZE and WRITE_SIZE in passes of their own (they do not fit one pass), with the
# guide's gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3: FETCH_SIZE tallies a 128-byte request as 64 bytes: doubled;
# WRITE_SIZE is exact for wide stores); counter unit KiB; per launch of sf_render_kernel<true>
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pf -- python3 $R/tools/image_probe.py 16384 100 image > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pw -- python3 $R/tools/image_probe.py 16384 100 image > /dev/null 2>&1
cd $R
python3 - "$OUT" "$RND" "$VER" <<'PY'
import csv, glob, json, os, sys
out, rnd, ver = sys.argv[1:4]
def mean(d):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "sf_render_kernel<true>" in r["Kernel_Name"]]
    return sum(v) / len(v), len(v)
fe, nf = mean(os.path.join(out, "pf"))
wr, nw = mean(os.path.join(out, "pw"))
rep = json.load(open(os.path.join(out, "image_kernel_latest.json")))
rep.update({"FETCH_SIZE_KiB_mean": fe, "WRITE_SIZE_KiB_mean": wr, "pmc_launches": [nf, nw],
            "read_bytes_per_launch": 2.0 * fe * 1024, "write_bytes_per_launch": wr * 1024,
            "traffic_bytes_per_launch": 2.0 * fe * 1024 + wr * 1024,
            "traffic_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per pass, over tools/image_probe.py 16384 100 image; "
                            "FETCH_SIZE doubled (gfx950: a 128-byte request is tallied as 64 bytes), WRITE_SIZE as read; KiB -> bytes; "
                            "per launch of sf_render_kernel<true> (16 384 frames)"})
json.dump(rep, open(os.path.join(out, "image_kernel_latest.json"), "w"), indent=1)
json.dump({k: rep[k] for k in ("version", "sf_build_id", "envs", "kernel", "FETCH_SIZE_KiB_mean", "WRITE_SIZE_KiB_mean", "pmc_launches",
                               "read_bytes_per_launch", "write_bytes_per_launch", "traffic_bytes_per_launch", "traffic_note")},
          open(os.path.join(out, "%s_pmc_render_traffic_%s.json" % (rnd, ver)), "w"), indent=1)
print("traffic per launch: %.1f MB read + %.1f MB written" % (rep["read_bytes_per_launch"] / 1e6, rep["write_bytes_per_launch"] / 1e6))
PY
rm -rf $OUT/pf $OUT/pw
echo "pmc traffic done"
bash tools/pmc_render.sh spacefortress_amd/libsfmi.so gpurun_out/profiles_render_${RND}_${VER}/${RND}_pmc_render_${VER}.txt > /dev/null 2>&1
echo "pmc done"
cat $OUT/${RND}_render_probe_${VER}.txt
head -8 $OUT/${RND}_render_kernel_stats_${VER}.csv
