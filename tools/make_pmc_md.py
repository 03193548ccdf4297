"""Assemble profiles/rNN_pmc.md from the summaries tools/profile_rNN.sh leaves under gpurun_out/rNN/.
usage: python tools/make_pmc_md.py r03"""
import os, sys
r = sys.argv[1]
src = os.path.join("gpurun_out", r)
rd = lambda n: open(os.path.join(src, n)).read().rstrip()
out = f"""# Round {int(r[1:])} -- rocprofv3 PMC summaries (MI355X)

Target of every pass: `python3 tools/run_chunks.py f16x2 1005 2` (1005 windows = one 10-minute recording per pass of the network, second
repetition counted).  Each counter set is its own `rocprofv3 --kernel-trace --output-format csv --pmc ...` run (tools/profile_{r}.sh: no
`--stats`, no other trace domain beside the counters); raw CSVs are scratch.  Kernel stats of the default bench command under
`rocprofv3 --kernel-trace --stats`: profiles/{r}_bench_c3_f16x2_kernel_stats.csv (its line: profiles/{r}_bench_c3_f16x2_under_rocprof.json).

## HBM traffic per launch, f16x2 (FETCH_SIZE x 1024 x 2, WRITE_SIZE x 1024: MI355X_MICROARCH.md 'HBM'; JSON: profiles/{r}_traffic_f16x2.json)

```
{rd('traffic_f16x2.txt')}
```

An instantiation that serves several layers shows their average (n = launches per pass).  `<1, 4, false, true, ..., 4>` = the A launches of
conv3_1 / conv4_1 / conv_bottleneck / encoder_out over the shared two-slot bank ring; `conv3x3_upsr_kernel` = conv6.A / conv7.A / conv8.A (upsampled
input half at low resolution, three-slot ring of half-chunk entries filled by LDS-DMA); `conv3x3_ups_kernel` = conv9_1.A (the same over resident
banks); `<1, 4, true, false, false, false, ..., 4>` = conv2_1.A (no r tensor) as four 4-wave tiles per workgroup over resident banks;
`<1, 8, true, false, false, false, 4, false, true, ...>` = conv9_1.B (flatten, projection in B); `<1, 8, false, false, false, true, 1, ...>` =
conv2_1.B with the projection; `<1, 8, true, false, false, true, 0, true, ...>` = conv1_1.B (first conv in the loader).

## SQ counters, pass 1 (wave-cycle shares, matrix-pipe busy at the 2.4 GHz price)

```
{rd('pmc_sq1.txt')}
```

## SQ counters, pass 2 (LDS, instruction counts)

```
{rd('pmc_sq2.txt')}
```
"""
open(os.path.join("profiles", f"{r}_pmc.md"), "w").write(out)
print("wrote", os.path.join("profiles", f"{r}_pmc.md"))
