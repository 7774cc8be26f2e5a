#!/bin/bash
# per-kernel durations of the image path (step kernel, explosion pre-pass, frame kernel):  bash tools/kt_image.sh [n_envs]   (GPU box)
set -e
R=$GRAFT_REPO_ROOT
N=${1:-16384}
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/kt_image_tmp
rm -rf $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/tools/image_probe.py $N 100 image > /dev/null 2>&1
cd $R
python3 - <<PY
import csv, glob
f = glob.glob("$D/**/*kernel_stats.csv", recursive=True)[0]
print("n_envs = $N")
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("sf_render_kernel", "sf_explosion_kernel", "sf_step_kernel")):
        print("  %-60s calls %6s  mean %8.1f us  min %8.1f  max %8.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
rm -rf $D
