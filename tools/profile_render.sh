#!/bin/bash
# The round's profile record of the render kernel, in one GPU call:  bash tools/profile_render.sh r02 v16   (on the GPU box;
# writes gpurun_out/profiles_render_r02_v16/, copy into profiles/)
#   rNN_render_probe_vNN.txt          tools/image_probe.py 16384 300 image, unprofiled (step+render and render alone, HIP events)
#   rNN_render_kernel_stats_vNN.csv   rocprofv3 --kernel-trace --stats of the same command
#   rNN_pmc_render_vNN.txt            SQ instruction mix, wait / active cycles (tools/pmc_render.sh)
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
rm -rf $OUT/kt
echo "kernel trace done"
cd $R
bash tools/pmc_render.sh spacefortress_amd/libsfmi.so gpurun_out/profiles_render_${RND}_${VER}/${RND}_pmc_render_${VER}.txt > /dev/null 2>&1
echo "pmc done"
cat $OUT/${RND}_render_probe_${VER}.txt
head -8 $OUT/${RND}_render_kernel_stats_${VER}.csv
