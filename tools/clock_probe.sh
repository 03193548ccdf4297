#!/bin/bash
# effective shader clock of every kernel of an f16x2 pass: GRBM_GUI_ACTIVE / 8 XCDs / duration (MI355X guide, DVFS give-back), one rocprofv3 --pmc run
# usage: tools/clock_probe.sh <outfile> [ENV=VAL ...]   (dev library: SOFTSPOKEN_LIB is set here)
set -o pipefail
export TMPDIR=/tmp
R=$PWD; OUT=$R/$1; shift
D=$(mktemp -d /tmp/clk.XXXX)
export SOFTSPOKEN_LIB=$R/softspoken_amd/libsoftspoken_hip_dev.so
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE -d $D -o s -- python3 tools/run_chunks.py f16x2 1005 3 > /dev/null 2>&1 || exit 4
python3 - $D > $OUT <<'PY'
import csv, glob, os, sys, collections
d = sys.argv[1]
gui = collections.defaultdict(list); dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": gui[r["Kernel_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    n = len(dur[k]); h = n - n // 3 if n >= 3 else 0          # skip the first repetition
    us = sum(dur[k][n // 3:]) / max(1, n - n // 3); g = sum(gui[k][n // 3:]) / max(1, n - n // 3) if k in gui else 0
    if us > 20:
        tot += us * (n - n // 3) / 2
        print("%-100s n=%2d %8.1f us  %5.2f GHz" % (k.replace("void ss::", "")[:100], (n - n // 3) // 2, us, g / 8 / us / 1e3))
print("sum of kernel time per pass: %.1f us" % tot)
PY
rm -rf $D
cat $OUT
