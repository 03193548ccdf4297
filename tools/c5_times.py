"""Kernel times of the C5 leg (48 kHz stereo -> decode + mixdown + resample + front-end), HIP events: python tools/c5_times.py [files]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softspoken_amd import synth, native, checkpoint
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 8
blob = checkpoint.pack_state_dict(synth.make_state_dict(0))
x2 = synth.to_pcm16(synth.synth_audio(5000, 120.0, 48000, 2, with_silence=False))
x5 = np.concatenate([x2] * 5)
fr = np.array([x5.shape[0]] * nf, dtype=np.int64)
c = native.Context(blob, 0, precision="bf16", profile=True)
p = np.concatenate([x5] * nf)
d = c.device_alloc(p.nbytes); c.device_upload(d, p)
st = native.plan_windows(600.0)
for rep in range(3):
    c.reset()
    first = c.add_pcm_batch_device(d, native.PCM_S16, 48000, 2, fr)
    for k in range(nf):
        c.features(first + k, st, discard=True)
    c.sync()
    if rep == 0: c.reset_stats()
for s in c.kernel_stats():
    if s["launches"]: print("%-40s %8.3f ms per pass" % (s["name"], s["total_ms"] / 2))
