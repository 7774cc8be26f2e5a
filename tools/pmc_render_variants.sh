#!/bin/bash
# instruction counts of sf_render_kernel per ablation variant (build/abl/libsfmi_render_*.so):
#   bash tools/pmc_render_variants.sh OUTTXT variant...     (on the GPU box)
set -e
OUT=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
: > $R/$OUT
for v in "$@"; do
  export SFMI_LIB_PATH=$R/build/abl/libsfmi_render_$v.so
  D=$R/gpurun_out/pmc_render_tmp
  rm -rf $D
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $D/p1 -- python3 $R/tools/image_probe.py 16384 40 image > /dev/null 2>&1
  ( echo "== $v"; cd $R; python3 tools/pmc_sum.py $D/p1 "sf_render_kernel" ) >> $R/$OUT 2>&1
  rm -rf $D
done
cat $R/$OUT
