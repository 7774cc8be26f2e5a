#!/bin/bash
# Per-kernel durations of the image step for one library:  bash tools/kt_render.sh LIB LABEL [probe args]   (GPU box)
# rocprofv3 --kernel-trace --stats of tools/image_probe.py (16 384 envs, 300 steps: sf_step + render per step, then the render
# launch alone); writes gpurun_out/kt_LABEL.csv (the kernel stats) and prints the rows of sf_step / sf_render.
LIB=$1; LABEL=$2; shift 2
ARGS=${@:-16384 300 image}
R=$GRAFT_REPO_ROOT
export SFMI_LIB_PATH=$R/$LIB
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$LABEL
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$LABEL -- python3 $R/tools/image_probe.py $ARGS > /tmp/kt_$LABEL.log 2>&1
ST=$(find /tmp/kt_$LABEL -name "*kernel_stats.csv" | head -1)
cp $ST $R/gpurun_out/kt_$LABEL.csv
echo "== $LABEL ($LIB)"; tail -2 /tmp/kt_$LABEL.log
grep -E "sf_step|sf_render|sf_drawrec" $R/gpurun_out/kt_$LABEL.csv | cut -c1-200
