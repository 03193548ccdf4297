"""Quick stage-by-stage check of the HIP path against the oracle on a GPU box (development aid;
the real tests live in tests/).  Usage: python tools/gpu_check.py [fp32|bf16|all]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softspoken_amd import synth, native, checkpoint      # noqa: E402
from oracle import oracle_np as O                         # noqa: E402

torch.set_grad_enabled(False)
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "all"
    sd_np = synth.make_state_dict(0)
    sd = synth.to_torch_state_dict(sd_np)
    blob = checkpoint.pack_state_dict(sd_np)
    pcm = synth.to_pcm16(synth.synth_audio(1001, 60.0, 16000, 1))
    wav = synth.wav_bytes(pcm, 16000)
    sig22, _, info = O.load_audio_from_bytes(wav)
    padded = O.pad_3s(sig22)
    starts = O.plan_windows(60.0)
    g = np.load(os.path.join(GOLD, "c1_logits.npz"))
    gf = np.load(os.path.join(GOLD, "c1_features.npz"))

    ctx = native.Context(blob, 0, bf16=False, profile=True)
    # ---- A2: decode + resample on device vs oracle
    fid, winfo = ctx.add_wav_bytes(wav)
    dev_sig = ctx.read_signal(fid)
    print("resample: n", len(dev_sig), len(sig22), "maxdiff", np.abs(dev_sig - sig22).max())
    # ---- A3: features vs golden (reference-side restated mel) on the oracle signal
    ctx.reset()
    fid = ctx.add_f32_22k(sig22)
    pick = gf["window_index"]
    feats = ctx.features(fid, starts[pick])
    d = np.abs(feats - gf["feats"])
    print("features: maxdiff", d.max(), "mean", d.mean(), "golden max", gf["feats"].max())
    allf = ctx.features(fid, starts)
    x = torch.stack([torch.from_numpy(padded[s:s + 66150]) for s in starts])
    of = O.mel_features(x, sd["mel_spectrogram.spectrogram.window"], sd["mel_spectrogram.mel_scale.fb"]).numpy()
    d = np.abs(allf - of)
    print("features(all 105 windows) vs oracle: maxdiff", d.max(), "n>1e-4:", int((d > 1e-4).sum()), "n>1e-5:", int((d > 1e-5).sum()))
    if mode in ("fp32", "all"):
        spec, mask = ctx.infer_windows(fid, starts, want_spec=False)
        d = np.abs(mask - g["logits"])
        print("fp32 logits vs golden: maxdiff", d.max(), "mean", d.mean())
        spec, mask2 = ctx.infer_windows(fid, starts[pick], want_spec=True)
        gl = np.load(os.path.join(GOLD, "c1_layers.npz"))
        print("fp32 mask(2 windows) vs golden", np.abs(mask2 - gl["mask"]).max(), "spec row64 maxdiff", np.abs(spec[:, :, 64, :] - gl["spec_row64"]).max())
        t0 = time.time()
        ok = ctx.run()
        t1 = time.time()
        a, idx = ctx.avg(fid)
        print("run ok", ok, "wall %.3fs" % (t1 - t0), "device ms", ctx.last_run_device_ms(), "avg maxdiff", np.abs(a - g["avg"]).max(), len(a), len(g["avg"]))
        reg = ctx.regions(fid)
        print("regions", reg)
        print("golden ", g["regions"].tolist())
        csv = O.CSV_HEADER + "\n" + native.format_csv_rows("/data/site a", "c1_seed1001.wav", reg, 1)
        print("csv identical:", csv == str(g["csv"]))
        for s in ctx.kernel_stats():
            if s["launches"]:
                print("  %-20s n=%4d  %9.3f ms  %8.2f TFLOP/s  %8.1f GB/s" % (s["name"], s["launches"], s["total_ms"],
                      s["flops"] / max(s["total_ms"], 1e-9) / 1e9, s["bytes"] / max(s["total_ms"], 1e-9) / 1e6))
    ctx.close()
    if mode in ("bf16", "all"):
        ctx = native.Context(blob, 0, bf16=True, profile=True)
        fid = ctx.add_f32_22k(sig22)
        spec, mask = ctx.infer_windows(fid, starts)
        d = np.abs(mask - g["logits"])
        print("bf16 logits vs golden: maxdiff", d.max(), "mean", d.mean(), "logit std", g["logits"].std())
        ctx.reset_stats()
        ok = ctx.run()
        ok = ctx.run()
        print("bf16 run device ms", ctx.last_run_device_ms(), "regions", ctx.regions(fid))
        for s in ctx.kernel_stats():
            if s["launches"]:
                print("  %-20s n=%4d  %9.3f ms  %8.2f TFLOP/s  %8.1f GB/s" % (s["name"], s["launches"], s["total_ms"],
                      s["flops"] / max(s["total_ms"], 1e-9) / 1e9, s["bytes"] / max(s["total_ms"], 1e-9) / 1e6))
        ctx.close()


if __name__ == "__main__":
    main()
