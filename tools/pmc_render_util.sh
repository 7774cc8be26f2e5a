#!/bin/bash
# lane utilisation of sf_render_kernel's vector instructions:  bash tools/pmc_render_util.sh OUTTXT   (on the GPU box)
set -e
OUT=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/pmc_render_tmp
rm -rf $D
rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1 || true
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD --kernel-trace --output-format csv -d $D/p1 -- python3 $R/tools/image_probe.py 16384 40 image > $R/gpurun_out/pmc_util_run.log 2>&1
cd $R
( python3 tools/pmc_sum.py $D/p1 "sf_render_kernel" ) > $OUT 2>&1
rm -rf $D
cat $OUT
