#!/bin/bash
# quick PMC look at the f16x2 pass (1005 windows, second repetition counted): tools/pmc_quick.sh <outdir> [kernel-name filter]
# two counter sets, each its own rocprofv3 run with --kernel-trace only (no --stats, no other trace domain beside the counters)
set -o pipefail
export TMPDIR=/tmp
R=$PWD; OUT=$R/$1; F=${2:-.}
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d $OUT/sq1 -o s -- python3 tools/run_chunks.py f16x2 1005 2 > /dev/null 2>&1 || exit 4
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR -d $OUT/sq2 -o s -- python3 tools/run_chunks.py f16x2 1005 2 > /dev/null 2>&1 || exit 5
rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $OUT/sq3 -o s -- python3 tools/run_chunks.py f16x2 1005 2 > /dev/null 2>&1 || exit 6
python3 tools/pmc_summary.py $OUT/sq1 | grep -E "$F" > $OUT/pmc_sq1.txt
python3 tools/pmc_summary.py $OUT/sq2 | grep -E "$F" > $OUT/pmc_sq2.txt
python3 tools/pmc_summary.py $OUT/sq3 | grep -E "$F" > $OUT/pmc_sq3.txt
rm -rf $OUT/sq1 $OUT/sq2 $OUT/sq3
cat $OUT/pmc_sq1.txt $OUT/pmc_sq2.txt $OUT/pmc_sq3.txt
