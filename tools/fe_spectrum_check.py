"""Dev build only: the front-end kernel's power spectrum |X[k]|^2 (k < 768) against numpy, 128 bins per run (SOFTSPOKEN_FEDBG = 512 + 1024 sel)."""
import os, sys, subprocess, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    sel = int(sys.argv[1])
    from softspoken_amd import synth, native, checkpoint
    blob = checkpoint.pack_state_dict(synth.make_state_dict(0))
    rng = np.random.default_rng(3)
    sig = (rng.standard_normal(66150 + 13230) * 0.1).astype(np.float32)
    c = native.Context(blob, 0, precision="bf16")
    fid = c.add_f32_22k(sig, padded=True)
    f = c.features(fid, np.array([0, 13230]))            # [2][128 bins of the selection][256 frames]
    win = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(512) / 512)).astype(np.float32)
    worst = 0.0; bad = []; byr = [0.0] * 4
    for w, s0 in enumerate((0, 13230)):
        x = sig[s0:s0 + 66150].astype(np.float64)
        xp = np.concatenate([x[256:0:-1], x])               # reflect left edge: frame t covers xp[256 t .. 256 t + 511]
        for t in range(256):
            fr = xp[256 * t: 256 * t + 512] * win
            ref = np.abs(np.fft.rfft(np.concatenate([fr, np.zeros(1536)]))) ** 2
            got = f[w, :, t]
            want = ref[128 * sel: 128 * sel + 128]
            err = np.abs(got - want) / (np.abs(want).max() + 1e-30)
            if err.max() > 1e-4 and len(bad) < 6: bad.append((w, t, int(err.argmax()) + 128 * sel, float(err.max())))
            worst = max(worst, float(err.max()))
            for r in range(4):
                byr[r] = max(byr[r], float(err[r::4].max()))
            if t == 3 and w == 0:
                print("frame 3 got/want ratio, bins %d..: " % (128 * sel), np.round(got[:16] / (want[:16] + 1e-30), 4).tolist())
    print(json.dumps(dict(sel=sel, worst=worst, by_r=byr, bad=bad[:3])))
else:
    from softspoken_amd import build
    for sel in range(6):
        e = dict(os.environ, SOFTSPOKEN_LIB=build.DEV_LIB, SOFTSPOKEN_FEDBG=str(512 + 1024 * sel))
        r = subprocess.run([sys.executable, __file__, str(sel)], env=e, capture_output=True, text=True)
        print([l for l in r.stdout.splitlines() if l.startswith("{")] or r.stderr[-500:])
