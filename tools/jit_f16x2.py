import sys, hashlib, numpy as np
sys.path.insert(0, '/root/repo')
from softspoken_amd import synth, native, checkpoint
blob = checkpoint.pack_state_dict(synth.make_state_dict(0))
x = synth.to_pcm16(synth.synth_audio(78, 600.0, 16000, 1))
sig = synth.synth_audio(7, 40.0, 22050, 1).astype(np.float32).ravel()
starts = (np.arange(33) * 13230).astype(np.int64)
def h(a): return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
out = []
c = native.Context(blob, 0, precision="f16x2")
fid = c.add_pcm(x[:16000 * 75], native.PCM_S16, 16000, 1, 16000 * 75); assert c.run(); out.append(h(c.window_logits(fid))); c.close()
c = native.Context(blob, 0, precision="f16x2", chunk=7)
fid = c.add_f32_22k(sig); _, m = c.infer_windows(fid, starts); out.append(h(m)); c.close()
print("HASHES", " ".join(out))
