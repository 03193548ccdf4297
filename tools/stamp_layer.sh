#!/bin/bash
# stage-segment stamps of one layer's launch (dev build): tools/stamp_layer.sh <layer> [SOFTSPOKEN_DBG bits]
layer=$1; bits=${2:-0}
env SOFTSPOKEN_LIB=$PWD/softspoken_amd/libsoftspoken_hip_dev.so SOFTSPOKEN_STAMP_LAYER=$layer SOFTSPOKEN_DBG=$bits timeout -k 10 200 python - <<'PY' 2>&1 | grep stamps | tail -1
import os, sys
sys.path.insert(0, os.getcwd())
from softspoken_amd import synth, native, checkpoint
blob = checkpoint.pack_state_dict(synth.make_state_dict(0))
x = synth.to_pcm16(synth.synth_audio(3000, 600.0, 16000, 1))
c = native.Context(blob, 0, precision="f16x2", profile=True)
for rep in range(2):
    c.reset(); c.add_pcm(x, native.PCM_S16, 16000, 1, len(x))
    try: c.run()
    except Exception: pass
PY
