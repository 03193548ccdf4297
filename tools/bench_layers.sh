#!/bin/bash
# usage: tools/bench_layers.sh <tag>   -- bench.py on the GPU box, prints value + per-layer table
tag=$1
timeout -k 10 300 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || { tail -c 800 gpurun_out/bench_$tag.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/bench_$tag.json"))
print("$tag", d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"].get("frac"))
for x in d["layers"]: print("  ", x["layer"].replace("conv3x3_","").replace("_kernel",""), x["us"], x["tflops"])
PY
