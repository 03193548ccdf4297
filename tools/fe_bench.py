"""Time the front-end kernel alone (features of N windows), development aid."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softspoken_amd import synth, native, checkpoint
nwin = int(sys.argv[1]) if len(sys.argv) > 1 else 256
blob = checkpoint.pack_state_dict(synth.make_state_dict(0))
ctx = native.Context(blob, 0, bf16=True, profile=True, chunk=nwin)
x = synth.synth_audio(5, 3.0 + 0.6 * nwin, 22050, 1, with_silence=False)[0].astype(np.float32)
fid = ctx.add_f32_22k(x)
starts = np.arange(nwin, dtype=np.int64) * 13230
for _ in range(2): ctx.features(fid, starts)
ctx.reset_stats()
for _ in range(5): ctx.features(fid, starts)
for s in ctx.kernel_stats():
    if s["name"] == "frontend":
        us = 1e3 * s["total_ms"] / s["launches"]
        print("dbg=%s frontend %.1f us per %d windows  -> %.0f GB/s" % (os.environ.get("SOFTSPOKEN_FEDBG", "0"), us, nwin, s["bytes"] / s["total_ms"] / 1e6))
ctx.close()
