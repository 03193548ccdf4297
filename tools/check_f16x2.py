"""f16x2 mode against the reference-made goldens and the fp32 mode (development aid)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softspoken_amd import synth, native, checkpoint
from oracle import oracle_np as O
blob = checkpoint.pack_state_dict(synth.make_state_dict(0))
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "c1_logits.npz"))
gy = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "c1_layers.npz"))
pcm = synth.to_pcm16(synth.synth_audio(1001, 60.0, 16000, 1))
sig, _, _ = O.load_audio_from_bytes(synth.wav_bytes(pcm, 16000))
starts = O.plan_windows(60.0)
res = {}
for prec in ("fp32", "f16x2"):
    c = native.Context(blob, 0, precision=prec)
    fid = c.add_f32_22k(sig)
    _, m = c.infer_windows(fid, starts)
    res[prec] = m
    d = np.abs(m - g["logits"])
    print(prec, "max |logit - golden| %.3e  mean %.3e  finite %s" % (d.max(), d.mean(), np.isfinite(m).all()), flush=True)
    spec, m2 = c.infer_windows(fid, starts[gy["window_index"]], want_spec=True)
    print(prec, "spec row64 max diff %.3e, mask %.3e" % (np.abs(spec[:, :, 64, :] - gy["spec_row64"]).max(), np.abs(m2 - gy["mask"]).max()), flush=True)
    assert c.run()
    print(prec, "regions equal golden:", c.regions(fid) == [tuple(r) for r in g["regions"].tolist()], flush=True)
    c.close()
print("f16x2 vs fp32 max diff %.3e" % np.abs(res["f16x2"] - res["fp32"]).max())
# throughput on a 10-min file
x = synth.to_pcm16(synth.synth_audio(3000, 600.0, 16000, 1))
for prec in ("fp32", "f16x2", "bf16"):
    c = native.Context(blob, 0, precision=prec)
    files = [x] * 10
    for rep in range(2):
        c.reset()
        ids = [c.add_pcm(f, native.PCM_S16, 16000, 1, len(f)) for f in files]
        c.sync(); t0 = time.perf_counter()
        assert c.run()
        dt = time.perf_counter() - t0
    print(f"{prec}: 10 x 600 s: run {dt:.3f} s -> {6000 / dt:.0f} audio-s/s ({10050 / dt:.0f} windows/s), device {c.last_run_device_ms():.0f} ms", flush=True)
    c.close()
