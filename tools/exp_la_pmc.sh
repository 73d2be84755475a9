# Counter passes over the 8K lookahead window (one run per pass)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out
B=$GRAFT_REPO_ROOT/bench.py
cd /tmp && export TMPDIR=/tmp
ARGS="--workload la8k --steps 1 --warmup 0 --cpu-pairs 0 --ref-pairs 0 --no-other-paths --no-clock-probe"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/la_sq -- python3 $B $ARGS > $OUT/la_sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM --output-format csv -d $OUT/la_sq2 -- python3 $B $ARGS > $OUT/la_sq2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/la_ta -- python3 $B $ARGS > $OUT/la_ta.log 2>&1
