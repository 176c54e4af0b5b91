#!/bin/bash
# SQ issue/occupancy counters of the two render kernels (two --pmc passes, 8 SQ slots each); summarised by
# tools/summarize_sq.py into profiles/<tag>_sq.json.   gpurun -- 'bash tools/profile_sq.sh r01d'
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/sq1_$TAG -- python3 bench.py --no-cpu-baseline --no-seam --no-make-image --no-extras --no-configs --steps 10 --warmup 2 --repeats 1 > gpurun_out/sq1_$TAG.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d gpurun_out/sq2_$TAG -- python3 bench.py --no-cpu-baseline --no-seam --no-make-image --no-extras --no-configs --steps 10 --warmup 2 --repeats 1 > gpurun_out/sq2_$TAG.log 2>&1
rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_SALU SQ_INST_LEVEL_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/sq3_$TAG -- python3 bench.py --no-cpu-baseline --no-seam --no-make-image --no-extras --no-configs --steps 10 --warmup 2 --repeats 1 > gpurun_out/sq3_$TAG.log 2>&1
rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_TC_DATA_READ_REQ SQC_DCACHE_BUSY_CYCLES SQC_TC_STALL SQC_DCACHE_INPUT_VALID_READYB --output-format csv -d gpurun_out/sq4_$TAG -- python3 bench.py --no-cpu-baseline --no-seam --no-make-image --no-extras --no-configs --steps 10 --warmup 2 --repeats 1 > gpurun_out/sq4_$TAG.log 2>&1
ls gpurun_out/sq1_$TAG/*/ gpurun_out/sq2_$TAG/*/ | head
