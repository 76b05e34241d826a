#!/bin/bash
# Collects the rocprofv3 summaries kept under profiles/ (run on the GPU box from the repository root).
#   tools/collect_profiles.sh kernels   kernel trace + HBM / instruction-mix / LDS counters of tools/kprofile.py
#                                       (every hot kernel at the bench's shapes, eager launches)
#   tools/collect_profiles.sh bench     kernel trace of bench.py --no-graph --no-secondary (the headline step)
#   tools/collect_profiles.sh dense     kernel trace + matrix-core counters of the K=256 image configuration
#   tools/collect_profiles.sh roofline  kernel trace of bench.py --roofline-only: the dominant kernel over ten rotating
#                                       buffer sets and nothing else of it (the launch sequence roofline.frac is quoted on)
# One counter group per run, each run a process of its own with its own time limit; the first failing step ends
# the script.  Every profiled program launches eagerly: the kernels and their durations are those of the HIP-graph
# replays, and the tracer sees plain single launches.  Results land in gpurun_out/prof_<what> (copy what is to be
# kept into profiles/).
set -e
WHAT=${1:-kernels}
R=$(pwd)
O=$R/gpurun_out/prof_$WHAT
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
export D3PM_BENCH_PROGRESS=1
sum() { (cd $R && python3 tools/pmc_summary.py "$@"); }
case $WHAT in
kernels)
  P="$R/tools/kprofile.py --reps 50"
  echo "== kernel trace"
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o trace -- python3 -u $R/tools/kprofile.py --reps 200 > $O/trace.log
  echo "== FETCH_SIZE"
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O -o fetch -- python3 -u $P > $O/fetch.log
  echo "== WRITE_SIZE"
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O -o write -- python3 -u $P > $O/write.log
  echo "== SQ instruction mix"
  timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $O -o sq -- python3 -u $P > $O/sq.log
  echo "== SQ LDS / memory instructions"
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O -o lds -- python3 -u $P > $O/lds.log
  echo "== summaries"
  sum $O/pmc_hbm.json $O/fetch_counter_collection.csv $O/write_counter_collection.csv > $O/pmc_hbm.txt
  sum $O/pmc_sq.json $O/sq_counter_collection.csv > $O/pmc_sq.txt
  sum $O/pmc_lds.json $O/lds_counter_collection.csv > $O/pmc_lds.txt
  ;;
bench)
  echo "== kernel trace of the bench step (eager)"
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o bench -- python3 -u $R/bench.py --no-graph --no-secondary --no-cpu-baseline --steps 100 --warmup 20 > $O/bench.json
  ;;
roofline)
  echo "== kernel trace of the rotating-buffer launch sequence"
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o roofline -- python3 -u $R/bench.py --roofline-only > $O/roofline.json
  cat $O/roofline.json
  ;;
dense)
  K="$R/tools/kbench.py --kind gaussian --bits 8 --B 128 --L 1024"
  echo "== dense K=256: kernel trace"
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o dense -- python3 -u $K --reps 100 > $O/dense.log
  for bits in 7 6; do
    echo "== dense K=$((1 << bits)): kernel trace"
    timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o dense_k$((1 << bits)) -- python3 -u $R/tools/kbench.py --kind gaussian --bits $bits --B 128 --L 1024 --reps 100 > $O/dense_k$((1 << bits)).log
  done
  echo "== dense K=256: matrix-core counters"
  timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_F32 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $O -o mfma -- python3 -u $K --reps 3 > $O/mfma.log
  sum $O/pmc_mfma.json $O/mfma_counter_collection.csv > $O/pmc_mfma.txt
  ;;
*) echo "unknown: $WHAT"; exit 2;;
esac
rm -f $O/*_counter_collection.csv $O/*_kernel_trace.csv $O/*agent_info.csv
ls -la $O
