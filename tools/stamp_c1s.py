"""Row-segment stamps of conv1s.hip (dev build): python tools/stamp_c1s.py   (env: SOFTSPOKEN_C1S_TOKEN, SOFTSPOKEN_C1S_ROWS, ...)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SOFTSPOKEN_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "softspoken_amd", "libsoftspoken_hip_dev.so"))
os.environ["SOFTSPOKEN_STAMP_LAYER"] = "conv1_1.B"
from softspoken_amd import synth, native, checkpoint
blob = checkpoint.pack_state_dict(synth.make_state_dict(0))
x = synth.to_pcm16(synth.synth_audio(3000, 600.0, 16000, 1))
c = native.Context(blob, 0, precision="f16x2", profile=True)
for rep in range(2):
    c.reset(); c.add_pcm(x, native.PCM_S16, 16000, 1, len(x))
    c.run()
