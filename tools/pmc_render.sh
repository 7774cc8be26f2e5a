#!/bin/bash
# SQ instruction mix of sf_render_kernel for a libsfmi variant:  bash tools/pmc_render.sh LIB OUTTXT   (on the GPU box)
set -e
LIB=$1; OUT=$2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export SFMI_LIB_PATH=$R/$LIB
D=$R/gpurun_out/pmc_render_tmp
rm -rf $D
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $D/p1 -- python3 $R/tools/image_probe.py 16384 40 image > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $D/p2 -- python3 $R/tools/image_probe.py 16384 40 image > /dev/null 2>&1
cd $R
( echo "== $LIB"; python3 tools/pmc_sum.py $D/p1 "sf_render_kernel"; python3 tools/pmc_sum.py $D/p2 "sf_render_kernel" ) > $OUT 2>&1
rm -rf $D
cat $OUT
