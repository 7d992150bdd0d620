#!/bin/bash
# PMC passes for the match kernel (separate runs per counter group; kernel-trace only).
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
cd /tmp
timeout 150 rocprofv3 -L 2>/dev/null | grep -E "^\s*(Name|name)?.*\b(SQ_WAVES|SQ_INSTS_VALU|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_WAIT_ANY|SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_ANY|SQ_ACTIVE_INST_VALU|SQ_INSTS_VMEM|SQ_INSTS_LDS|SQ_INSTS_SALU|FETCH_SIZE|WRITE_SIZE|TCC_HIT_sum|TCC_MISS_sum|TCP_TCC_READ_REQ_sum|SQ_INST_CYCLES_VMEM|GRBM_GUI_ACTIVE|SQ_LDS_BANK_CONFLICT|SQ_INSTS_VMEM_RD|TCC_EA0_RDREQ_sum|TCC_REQ_sum)\b" | head -40 > $OUT/counters_available.txt
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --lanes ${LANES:-4}"
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq -o sq -- $CMD > $OUT/sq.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/fetch -o fetch -- $CMD > $OUT/fetch.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -o tcc -- $CMD > $OUT/tcc.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq2 -o sq2 -- $CMD > $OUT/sq2.log 2>&1
find $OUT -name "*.csv" | head; du -sh $OUT
# keep only the counter csv + small logs (kernel-trace CSVs can be large)
find $OUT -name "*kernel_trace.csv" -size +8M -delete
