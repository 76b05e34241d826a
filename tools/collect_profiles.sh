#!/bin/bash
# Collects the rocprofv3 summaries kept under profiles/ (run on the GPU box from the repository root): kernel
# trace of the bench, HBM / instruction-mix / LDS counters of every kernel it launches (one counter group per
# run, each run a process of its own), kernel trace and matrix-core counters of the K=256 image configuration.
# Every profiled run launches eagerly (--no-graph): the kernels and their durations are those of the graph
# replays, and the tracer then sees plain single launches.  The first failing step ends the script.  Progress
# goes to stdout; results land in gpurun_out/prof (copy what is to be kept into profiles/).
set -e
R=$(pwd)
O=$R/gpurun_out/prof
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
export D3PM_BENCH_PROGRESS=1
B="$R/bench.py --no-graph --steps 20 --warmup 5 --no-cpu-baseline"
K="$R/tools/kbench.py --kind gaussian --bits 8 --B 128 --L 1024"
echo "== kernel trace of the bench"
timeout -k 10 170 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o bench -- python3 -u $R/bench.py --no-graph --steps 100 --warmup 20 --no-cpu-baseline > $O/bench.json
echo "== FETCH_SIZE"
timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O -o fetch -- python3 -u $B > $O/fetch.json
echo "== WRITE_SIZE"
timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O -o write -- python3 -u $B > $O/write.json
echo "== SQ instruction mix"
timeout -s KILL 170 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $O -o sq -- python3 -u $B > $O/sq.json
echo "== SQ LDS / memory instructions"
timeout -s KILL 170 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O -o lds -- python3 -u $B > $O/lds.json
echo "== dense K=256: kernel trace"
timeout -k 10 170 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o dense -- python3 -u $K --reps 100 > $O/dense.log
echo "== dense K=256: matrix-core counters"
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_F32 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $O -o mfma -- python3 -u $K --reps 3 > $O/mfma.log
echo "== summaries"
cd $R
python3 tools/pmc_summary.py $O/pmc_hbm.json $O/fetch_counter_collection.csv $O/write_counter_collection.csv > $O/pmc_hbm.txt
python3 tools/pmc_summary.py $O/pmc_sq.json $O/sq_counter_collection.csv > $O/pmc_sq.txt
python3 tools/pmc_summary.py $O/pmc_lds.json $O/lds_counter_collection.csv > $O/pmc_lds.txt
python3 tools/pmc_summary.py $O/pmc_mfma.json $O/mfma_counter_collection.csv > $O/pmc_mfma.txt
rm -f $O/*_counter_collection.csv $O/*_kernel_trace.csv $O/*agent_info.csv
ls -la $O
