#!/bin/bash
# Round-4 profiles on a GPU box (gpurun): kernel trace + stats of the default bench command, then PMC passes (each on its own, with
# --kernel-trace only) on tools/run_chunks.py: HBM bytes (FETCH_SIZE, WRITE_SIZE) and SQ counters.  Output under gpurun_out/r04/.
set -o pipefail
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r04
mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3 -o c3 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err || exit 1
echo "stats done"
for prec in f16x2; do
  rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch_$prec -o f -- python3 tools/run_chunks.py $prec 1005 2 > /dev/null 2>&1 || exit 2
  rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write_$prec -o w -- python3 tools/run_chunks.py $prec 1005 2 > /dev/null 2>&1 || exit 3
  echo "traffic $prec done"
done
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d $OUT/pmc_sq1 -o s -- python3 tools/run_chunks.py f16x2 1005 2 > /dev/null 2>&1 || exit 4
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD -d $OUT/pmc_sq2 -o s -- python3 tools/run_chunks.py f16x2 1005 2 > /dev/null 2>&1 || exit 5
echo "sq done"
python3 tools/traffic_summary.py $OUT/pmc_fetch_f16x2 $OUT/pmc_write_f16x2 1005 $OUT/traffic_f16x2.json > $OUT/traffic_f16x2.txt
python3 tools/pmc_summary.py $OUT/pmc_sq1 > $OUT/pmc_sq1.txt
python3 tools/pmc_summary.py $OUT/pmc_sq2 > $OUT/pmc_sq2.txt
find $OUT/stats_c3 -name "*kernel_stats.csv" -exec cp {} $OUT/c3_kernel_stats.csv \;
# raw traces are large: keep the summaries only
rm -rf $OUT/stats_c3 $OUT/pmc_fetch_* $OUT/pmc_write_* $OUT/pmc_sq1 $OUT/pmc_sq2
ls -la $OUT
