import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from softspoken_amd import synth, native, checkpoint
from oracle import oracle_np as O
g = np.load("tests/golden/c1_logits.npz")
pcm = synth.to_pcm16(synth.synth_audio(1001, 60.0, 16000, 1))
sig, _, _ = O.load_audio_from_bytes(synth.wav_bytes(pcm, 16000))
ctx = native.Context(checkpoint.pack_state_dict(synth.make_state_dict(0)), 0, precision="f16x2")
fid = ctx.add_f32_22k(sig)
assert ctx.run()
d = np.abs(ctx.window_logits(fid) - g["logits"])
print("MAXDIFF", float(d.max()), "REGIONS", len(ctx.regions(fid)))
