#!/bin/bash
# Instructions per frame of sf_render_kernel, of every kind, by phase:  bash tools/pmc_render_stops.sh OUT.txt   (GPU box)
# rocprofv3 --pmc SQ_INSTS_* over the early-return builds build/abl/libsfmi_stop1..5.so (tools/variant.py stopK
# -DSF_RENDER_STOP=K: the kernel returns behind phase K) and build/abl/libsfmi_cur.so (the full kernel); per-phase counts are
# the differences; stop41..44 = behind the set-up / the records / the rounds / everything but the resample pass of draw_strokes,
# stop43nc = stop43 without coverage and compositing (-DSF_RENDER_SKIP=96: the cheap rounds alone).  OUT.txt is truncated first.
set -e
OUT=${1:?usage: pmc_render_stops.sh OUT.txt}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/pmc_stops_tmp
rm -rf $D
: > $R/$OUT
for v in ${STOPS:-stop1 stop2 stop3 stop41 stop42 stop43 stop43nc stop44 stop4 stop5 cur}; do
  export SFMI_LIB_PATH=$R/build/abl/libsfmi_$v.so
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $D/$v -- python3 $R/tools/image_probe.py 16384 40 image > /dev/null 2>&1
  ( echo "== $v"; cd $R; python3 tools/pmc_sum.py $D/$v "sf_render_kernel" ) >> $R/$OUT
done
rm -rf $D
cat $R/$OUT
