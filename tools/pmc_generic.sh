R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pg
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pg -- python3 $R/tools/geometry_probe.py 16384 20 image-raw > /dev/null 2>&1
cd $R; python3 tools/pmc_sum.py /tmp/pg sf_render_generic
rm -rf /tmp/pg2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg2 -- python3 $R/tools/geometry_probe.py 16384 20 image-raw > /dev/null 2>&1
grep -h "generic" $(find /tmp/pg2 -name "*kernel_stats.csv") | cut -c1-160
