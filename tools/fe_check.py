"""Front-end kernel: parity with the reference-made feature fixture and the torch oracle, and its rate (development aid).
SOFTSPOKEN_LIB=<dev library> SOFTSPOKEN_FEDBG=256 runs the first kernel for A/B."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softspoken_amd import synth, native, checkpoint
from oracle import oracle_np as O
sd_np = synth.make_state_dict(0); sd = synth.to_torch_state_dict(sd_np)
blob = checkpoint.pack_state_dict(sd_np)
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "c1_features.npz"))
pcm = synth.to_pcm16(synth.synth_audio(1001, 60.0, 16000, 1))
sig, _, _ = O.load_audio_from_bytes(synth.wav_bytes(pcm, 16000))
starts = O.plan_windows(60.0); padded = O.pad_3s(sig)
c = native.Context(blob, 0, precision="bf16", profile=True)
fid = c.add_f32_22k(sig)
f = c.features(fid, starts[g["window_index"]])
print("golden windows: max |d| %.3e" % np.abs(f - g["feats"]).max())
allf = c.features(fid, starts)
x = torch.stack([torch.from_numpy(padded[s:s + 66150]) for s in starts])
ref = O.mel_features(x, sd["mel_spectrogram.spectrogram.window"], sd["mel_spectrogram.mel_scale.fb"]).numpy()
d = np.abs(allf - ref)
print("all 105 windows vs oracle: max %.3e, > 1e-5: %d, mean %.3e; window 0 zero: %s; finite %s" % (d.max(), (d > 1e-5).sum(), d.mean(), not allf[0].any(), np.isfinite(allf).all()))
# rate: a 10-min file's 1005 windows, 8 files
x10 = synth.to_pcm16(synth.synth_audio(3000, 600.0, 16000, 1))
c.reset(); ids = [c.add_pcm(x10, native.PCM_S16, 16000, 1, len(x10)) for _ in range(4)]
st = native.plan_windows(600.0)
for rep in range(3):
    if rep == 1: c.reset_stats()
    for i in ids: c.features(i, st, discard=True)
c.sync()
for s in c.kernel_stats():
    if s["name"] == "frontend" and s["launches"]:
        us = 1e3 * s["total_ms"] / s["launches"]
        print("frontend: %.1f us per 1005 windows -> %.2f M windows/s, %.0f GB/s algorithmic = %.1f %% of 8 TB/s" % (us, 1005 / us, 1005 * 395672 / us / 1e3, 1005 * 395672 / us / 1e3 / 80))
