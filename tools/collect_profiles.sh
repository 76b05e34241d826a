#!/bin/bash
# Collects the rocprofv3 summaries kept under profiles/ (run on the GPU box from the repository root):
# kernel trace of the default bench, HBM and instruction-mix counters of the fused loss kernel (one
# counter group per run), kernel trace and matrix-core counters of the K=256 image configuration.
set -e
R=$(pwd)
O=$R/gpurun_out/prof
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary"
K="$R/tools/kbench.py --kind gaussian --bits 8 --B 128 --L 1024"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o bench -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench.log 2>&1
echo "bench trace done"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O -o fetch -- python3 $B > $O/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O -o write -- python3 $B > $O/write.log 2>&1
echo "hbm counters done"
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $O -o sq -- python3 $B > $O/sq.log 2>&1
echo "sq counters done"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o dense -- python3 $K --reps 100 > $O/dense.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_F32 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $O -o mfma -- python3 $K --reps 3 > $O/mfma.log 2>&1
echo "dense done"
cd $R
python3 tools/pmc_summary.py $O/pmc_hbm.json $O/fetch_counter_collection.csv $O/write_counter_collection.csv > $O/pmc_hbm.txt
python3 tools/pmc_summary.py $O/pmc_sq.json $O/sq_counter_collection.csv > $O/pmc_sq.txt
python3 tools/pmc_summary.py $O/pmc_mfma.json $O/mfma_counter_collection.csv > $O/pmc_mfma.txt
rm -f $O/*_counter_collection.csv $O/*_kernel_trace.csv $O/*agent_info.csv
ls -la $O
