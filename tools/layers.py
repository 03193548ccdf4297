"""Per-layer times of one precision mode on 10 x 10 min (development aid): python tools/layers.py f16x2|bf16|fp32 [n_files]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softspoken_amd import synth, native, checkpoint
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x2"
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 10
blob = checkpoint.pack_state_dict(synth.make_state_dict(0))
x = synth.to_pcm16(synth.synth_audio(3000, 600.0, 16000, 1))
c = native.Context(blob, 0, precision=prec, profile=True)
for rep in range(3):
    c.reset()
    ids = [c.add_pcm(x, native.PCM_S16, 16000, 1, len(x)) for _ in range(nf)]
    c.sync(); t0 = time.perf_counter()
    try:
        assert c.run()
    except native.SoftspokenError as e:      # (timing-only ablation builds compute garbage: SS_ERR_RANGE after the kernels have run)
        if rep == 0: print('run():', e)
    dt = time.perf_counter() - t0
    if rep == 0: c.reset_stats()
tot = 0.0
for s in c.kernel_stats():
    if s["launches"]:
        us = 1e3 * s["total_ms"] / s["launches"]
        tot += s["total_ms"] / 2
        print("%-95s n=%3d %9.1f us %7.1f TF %7.1f GB/s" % (s["name"][-95:], s["launches"] // 2, us, s["flops"] / max(s["total_ms"], 1e-9) / 1e9, s["bytes"] / max(s["total_ms"], 1e-9) / 1e6))
print(f"{prec}: {nf} x 600 s: {nf * 600 / dt:.0f} audio-s/s ({nf * 1005 / dt:.0f} windows/s); kernel time {tot:.1f} ms per run")
