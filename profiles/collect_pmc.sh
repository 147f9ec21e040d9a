#!/bin/bash
# Counter evidence for the GEMM analysis in DESIGN.md (section 6): isolated launches of the step's contraction shapes
# (profiles/gemm_shapes_probe.py) under rocprofv3, ONE counter group per pass (8 SQ slots / 4 TCC slots per pass),
# --pmc never combined with a trace domain.  Run through gpurun from the repo root:
#   gpurun --timeout 900 -- 'bash profiles/collect_pmc.sh r02'
# then `python3 profiles/summarize_pmc.py r02` here writes profiles/r02_gemm_pmc.csv.
set -e
TAG=${1:-r02}
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
python3 profiles/gemm_shapes_probe.py > "$OUT/timing.txt" 2>&1
cat "$OUT/timing.txt"
P="python3 profiles/gemm_shapes_probe.py --reps 2 --no-time"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $P > "$OUT/trace.log" 2>&1
echo "trace pass done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES \
    --output-format csv -d "$OUT/sq" -- $P > "$OUT/sq.log" 2>&1
echo "sq pass done"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES --output-format csv -d "$OUT/lds" -- $P > "$OUT/lds.log" 2>&1
echo "lds pass done"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum --output-format csv -d "$OUT/tc" -- $P > "$OUT/tc.log" 2>&1
echo "tc pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $P > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $P > "$OUT/write.log" 2>&1
echo "hbm passes done"
ls "$OUT"/*/* | head -30
