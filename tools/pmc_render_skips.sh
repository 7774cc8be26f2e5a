#!/bin/bash
# Instructions per frame of the render kernel with parts of draw_strokes compiled out (SF_RENDER_SKIP: 16 = no resample pass,
# 32 = no coverage integrals, 64 = no compositing, 112 = none of the three): rocprofv3 --pmc over build/abl/libsfmi_skipK.so
# (tools/variant.py skipK -DSF_RENDER_SKIP=K) and libsfmi_cur.so.   bash tools/pmc_render_skips.sh OUT.txt   (GPU box)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/pmc_skips_tmp
rm -rf $D; : > $R/$1
for v in cur skip16 skip32 skip64 skip112; do
  export SFMI_LIB_PATH=$R/build/abl/libsfmi_$v.so
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $D/$v -- python3 $R/tools/image_probe.py 16384 40 image > /dev/null 2>&1
  ( echo "== $v"; cd $R; python3 tools/pmc_sum.py $D/$v "sf_render_kernel" ) >> $R/$1
done
rm -rf $D
