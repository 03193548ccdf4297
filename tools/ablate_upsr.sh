#!/bin/bash
# Timing-only ablations of conv4_ups.hip's ring form (dev build, SOFTSPOKEN_DBG bits 16..20; results are wrong on purpose):
# per-layer microseconds of conv6.A / conv7.A / conv8.A with one ingredient of a beat removed.   usage: tools/ablate_upsr.sh [n_files]
nf=${1:-20}
for spec in base=0 no_mfma=65536 no_lds_reads=131072 no_patch_loads=262144 no_ring_dma=524288 no_stores=1048576 no_mfma_no_reads=196608 only_sync=2031616; do
  label=${spec%%=*}; bits=${spec#*=}
  env SOFTSPOKEN_LIB=$PWD/softspoken_amd/libsoftspoken_hip_dev.so SOFTSPOKEN_DBG=$bits timeout -k 10 200 python - $nf <<'PY' 2>&1 | awk -v L=$label '{printf "%-18s %s\n", L, $0}'
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from softspoken_amd import synth, native, checkpoint
nf = int(sys.argv[1])
blob = checkpoint.pack_state_dict(synth.make_state_dict(0))
x = synth.to_pcm16(synth.synth_audio(3000, 600.0, 16000, 1))
c = native.Context(blob, 0, precision="f16x2", profile=True)
for rep in range(2):
    c.reset()
    ids = [c.add_pcm(x, native.PCM_S16, 16000, 1, len(x)) for _ in range(nf)]
    try: c.run()
    except Exception as e: pass
    if rep == 0: c.reset_stats()
out = []
for s in c.kernel_stats():
    if s["launches"] and "upsr" in s["name"]:
        out.append("%s %.0f" % (s["name"].split("/")[-1], 1e3 * s["total_ms"] / s["launches"]))
print("  ".join(out))
PY
done
