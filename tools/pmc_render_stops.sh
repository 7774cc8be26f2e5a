set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/pmc_stops_tmp
rm -rf $D
for v in stop1 stop2 stop3 stop4 stop5 msh; do
  export SFMI_LIB_PATH=$R/build/abl/libsfmi_$v.so
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $D/$v -- python3 $R/tools/image_probe.py 16384 40 image > /dev/null 2>&1
  ( echo "== $v"; cd $R; python3 tools/pmc_sum.py $D/$v "sf_render_kernel" ) >> $R/gpurun_out/pmc_render_phase_stops_v27.txt
done
rm -rf $D
