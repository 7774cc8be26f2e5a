#!/bin/bash
# SQ counters of sf_step_kernel (two passes, 8 counters each): per-wave instruction mix and where the cycles go.
#   bash tools/pmc_step.sh OUTDIR        (on the GPU box; writes OUTDIR/pmc_step.txt)
set -e
OUT=${1:-gpurun_out/pmc_step}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="--steps 400 --warmup 50 --no-cpu-baseline --rollout-k 0 --image-envs 0 --kernel-timing-launches 1 --repeats 1"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $R/$OUT/p1 -- python3 $R/bench.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $R/$OUT/p2 -- python3 $R/bench.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$OUT/p3 -- python3 $R/bench.py $ARGS > /dev/null 2>&1 || true
cd $R
( python3 tools/pmc_sum.py $OUT/p1 "sf_step_kernel"; python3 tools/pmc_sum.py $OUT/p2 "sf_step_kernel"; python3 tools/pmc_sum.py $OUT/p3 "sf_step_kernel" ) > $OUT/pmc_step.txt 2>&1
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
cat $OUT/pmc_step.txt
