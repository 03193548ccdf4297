"""Run a few chunks of the conv stack on random audio (profiling target for rocprofv3 --pmc).
usage: python tools/run_chunks.py [f16x2|bf16|fp32] [windows] [reps]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softspoken_amd import synth, native, checkpoint
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
nwin = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
blob = checkpoint.pack_state_dict(synth.make_state_dict(0))
ctx = native.Context(blob, 0, precision=prec, chunk=nwin)
x = synth.synth_audio(5, 3.0 + 0.6 * nwin, 22050, 1, with_silence=False)[0].astype(np.float32)
fid = ctx.add_f32_22k(x)
starts = np.arange(nwin, dtype=np.int64) * 13230
for _ in range(reps):
    ctx.infer_windows(fid, starts)
ctx.close()
print("done")
