#!/bin/bash
# Where the render kernel's waves wait: instruction fetch, LDS, vector memory, launch.  bash tools/pmc_render_deep.sh LIB OUT  (GPU box)
set -e
LIB=$1; OUT=$2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export SFMI_LIB_PATH=$R/$LIB
D=$R/gpurun_out/pmc_render_deep_tmp
rm -rf $D
run() { rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $D/p$N -- python3 $R/tools/image_probe.py 16384 40 image > /dev/null 2>$D.err$N || { echo "pass $N failed: $*"; tail -3 $D.err$N; }; N=$((N+1)); }
N=1
run SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU
run SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES
run SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS
run SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
run TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum
run SPI_RA_REQ_NO_ALLOC_CSN SPI_RA_LDS_CU_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_CSN_BUSY SPI_CSN_WAVE GRBM_GUI_ACTIVE
cd $R
( echo "== $LIB"; for p in $D/p*; do python3 tools/pmc_sum.py $p "sf_render_kernel"; done ) > $OUT 2>&1
rm -rf $D $D.err*
cat $OUT
