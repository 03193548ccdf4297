"""Generate tests/golden/*.npz by running the REFERENCE's own classes on CPU.  Runs only in the
build container (needs /root/reference); the outputs are committed, the reference never travels.

What is the reference's own code here, and what is a stand-in:
  * SpecUNet_2D (root/code/backend/pytorch_neural_nets.py:79-197) -- imported and run as is: every
    conv/BN/pool/upsample/concat/head op of the goldens is executed by the reference's class.
  * NNDetector.average_overlapping_detections / find_speech_regions
    (root/code/frontend/NNDetector.py:153-190, 103-143) -- imported and run as is (via __new__,
    skipping __init__, which needs a project manager).
  * torchaudio is NOT in this image.  `torchaudio.transforms.MelSpectrogram` is replaced by an
    in-memory module that restates its documented algorithm on torch.stft; the
    mel front-end goldens are therefore "parity unpinned" at the torchaudio boundary.
  * sounddevice / librosa / soundfile are absent; empty modules satisfy the imports of
    voice_activity.py -- none of their functions is called.
  * DataFrame.to_csv text comes from pandas with the reference's column dtypes
    (silencer_ui.py:779-788) -- silencer_ui itself needs PySide6 and is not imported.

Usage:  python tests/golden/make_golden.py [--c3] [--hostile-only]
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from softspoken_amd import synth            # noqa: E402
from oracle import oracle_np as O           # noqa: E402


def install_standins():
    import torch.nn as nn

    class _Spectrogram(nn.Module):
        def __init__(self, n_fft, win_length, hop_length):
            super().__init__()
            self.n_fft, self.win_length, self.hop_length = n_fft, win_length, hop_length
            self.register_buffer("window", torch.hann_window(win_length))

        def forward(self, x):
            s = torch.stft(x, n_fft=self.n_fft, hop_length=self.hop_length, win_length=self.win_length,
                           window=self.window, center=True, pad_mode="reflect", normalized=False,
                           onesided=True, return_complex=True)
            return s.abs().pow(2.0)

    class _MelScale(nn.Module):
        def __init__(self):
            super().__init__()
            self.register_buffer("fb", torch.from_numpy(synth.mel_filterbank()))

        def forward(self, spec):
            return torch.matmul(spec.transpose(-1, -2), self.fb).transpose(-1, -2)

    class MelSpectrogram(nn.Module):
        def __init__(self, sample_rate=16000, n_fft=400, win_length=None, hop_length=None, f_min=0.0,
                     f_max=None, n_mels=128, **kw):
            super().__init__()
            assert (sample_rate, n_fft, win_length, hop_length, n_mels, f_max) == (22050, 2048, 512, 256, 128, 8000)
            self.spectrogram = _Spectrogram(n_fft, win_length, hop_length)
            self.mel_scale = _MelScale()

        def forward(self, x):
            return self.mel_scale(self.spectrogram(x))

    ta = types.ModuleType("torchaudio")
    tat = types.ModuleType("torchaudio.transforms")
    tat.MelSpectrogram = MelSpectrogram
    ta.transforms = tat
    sys.modules["torchaudio"] = ta
    sys.modules["torchaudio.transforms"] = tat
    for name in ("sounddevice", "librosa", "librosa.display", "soundfile"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["librosa"].display = sys.modules["librosa.display"]


def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)      # (the fixtures depend on it at the 1e-6 level: oneDNN's reduction order)
    install_standins()
    sys.path.insert(0, REF)
    from root.code.backend.pytorch_neural_nets import SpecUNet_2D
    from root.code.frontend.NNDetector import NNDetector
    from root.code.backend import settings

    if "--hostile-only" in sys.argv:                 # (leaves the other fixtures as they are)
        torch.set_grad_enabled(False)
        det = NNDetector.__new__(NNDetector)
        return make_hostile(SpecUNet_2D, det, settings)
    sd_np = synth.make_state_dict(0)
    sd = synth.to_torch_state_dict(sd_np)
    model = SpecUNet_2D()
    ref_keys = list(model.state_dict().keys())
    assert sorted(ref_keys) == sorted(sd.keys()), "synthetic checkpoint key layout != reference state_dict"
    for k, v in model.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), (k, v.shape, sd[k].shape)
    model.load_state_dict(sd, strict=True)
    model.eval()
    torch.set_grad_enabled(False)

    # ---- C1: one 60 s 16 kHz mono file --------------------------------------
    pcm = synth.to_pcm16(synth.synth_audio(1001, 60.0, 16000, 1))
    wav = synth.wav_bytes(pcm, 16000)
    sig22, _, info = O.load_audio_from_bytes(wav)             # build's own A2 (parity boundary = this signal)
    duration = info["frames"] / info["sr"]
    padded = O.pad_3s(sig22)
    starts = O.plan_windows(duration)
    W = len(starts)

    # features of two windows (a loud one and the one holding the 1e-3 burst / silence edge)
    sig_t = torch.from_numpy(padded)
    feats_all = []
    logits = []
    specs_first = None
    for s0 in range(0, W, settings.prediction_batch_size):
        idx = starts[s0:s0 + settings.prediction_batch_size]
        sl = torch.stack([sig_t[int(i):int(i) + 66150] for i in idx])
        spec, mask = model(sl)                                # the reference forward
        logits.append(mask.numpy())
        feats_all.append(model.sqrt_log10_nonzero(model.mel_spectrogram(sl))[:, :, :256].numpy())
        if specs_first is None:
            specs_first = spec[:2].numpy().copy()
    logits = np.vstack(logits)                                # (W,1,256)
    feats_all = np.concatenate(feats_all)                     # (W,128,256)

    # layer-wise taps for two windows through the reference module (forward hooks)
    taps = {}
    hooks = []
    for name in ["conv1_1", "conv2_1", "conv3_1", "conv4_1", "conv_bottleneck", "encoder_out", "conv6",
                 "conv7", "conv8", "conv9_1", "relu_flatten"]:
        mod = getattr(model, name)
        hooks.append(mod.register_forward_hook(lambda m, i, o, name=name: taps.__setitem__(name, o.detach().clone())))
    pick = [5, 60]
    sl = torch.stack([sig_t[int(starts[i]):int(starts[i]) + 66150] for i in pick])
    spec2, mask2 = model(sl)
    for h in hooks:
        h.remove()
    tap_stats = {k: np.array([[float(v[b].double().mean()), float(v[b].double().abs().max()),
                               float(v[b].double().pow(2).mean().sqrt())] for b in range(2)]) for k, v in taps.items()}

    # ---- reference post-processing --------------------------------------------------------------
    det = NNDetector.__new__(NNDetector)
    fkey = "/data/site a/c1_seed1001.wav"
    secs = len(padded) / settings.vad_resample
    avg = det.average_overlapping_detections({fkey: logits}, secs)
    regions = det.find_speech_regions({fkey: avg}, break_duration=0.5)
    avg_vals = np.array([a for a, _ in avg[fkey]], dtype=np.float64)
    avg_strs = np.array([t for _, t in avg[fkey]])
    reg_strs = np.array(regions[fkey], dtype=str).reshape(-1, 2)
    reg_f = np.array([(float(s) - 3, float(e) - 3) for s, e in regions[fkey]], dtype=np.float64).reshape(-1, 2)
    empty_avg = det.average_overlapping_detections({fkey: np.array([])}, secs)
    assert empty_avg[fkey] == []

    # CSV text through pandas with the reference's frame schema (silencer_ui.py:779-788; worker.py:107-125)
    import pandas as pd
    column_types = {'ID': 'int64', 'file_path': str, 'file_name': str, 'start_time': str, 'end_time': str,
                    'erase': int, 'user_comment': str, 'review_datetime': 'datetime64[ns]'}
    df = pd.DataFrame(columns=column_types.keys()).astype(column_types)
    nid = 1
    for (s, e) in reg_f:
        df.loc[len(df)] = {'ID': nid, 'file_path': os.path.dirname(fkey), 'file_name': os.path.basename(fkey),
                           'start_time': float(s), 'end_time': float(e), 'erase': 0, 'user_comment': '',
                           'review_datetime': ''}
        nid += 1
    csv = df.to_csv(index=False)
    empty_csv = pd.DataFrame(columns=column_types.keys()).astype(column_types).to_csv(index=False)

    frac = float((avg_vals > settings.threshold).mean())
    print(f"W={W} bins={len(avg_vals)} regions={len(reg_f)} frac>thr={frac:.3f} "
          f"logit range [{logits.min():.3f},{logits.max():.3f}]")

    np.savez_compressed(os.path.join(HERE, "c1_logits.npz"), starts=starts, logits=logits.astype(np.float32),
                        avg=avg_vals, avg_time_str=avg_strs, regions_str=reg_strs, regions=reg_f,
                        csv=np.array(csv), empty_csv=np.array(empty_csv), file_key=np.array(fkey),
                        duration=np.array(duration), n_padded=np.array(len(padded)))
    np.savez_compressed(os.path.join(HERE, "c1_features.npz"), window_index=np.array(pick),
                        feats=feats_all[pick].astype(np.float32),
                        feat_stats=np.stack([feats_all.mean(axis=(1, 2)), feats_all.max(axis=(1, 2))], 1).astype(np.float64))
    np.savez_compressed(os.path.join(HERE, "c1_layers.npz"), window_index=np.array(pick),
                        mask=mask2.numpy(), spec_stats=np.array([[float(spec2[b].double().mean()),
                                                                  float(spec2[b].double().abs().max())] for b in range(2)]),
                        spec_row64=spec2[:, :, 64, :].numpy(),
                        flatten=taps["relu_flatten"].numpy(),
                        conv4_1=taps["conv4_1"].numpy(), encoder_out=taps["encoder_out"].numpy(),
                        **{"stats_" + k: v for k, v in tap_stats.items()})
    # the 22.05 kHz signal itself is regenerated in tests from the seed (same oracle resampler);
    # store a checksum so drift in the generator is caught.
    np.savez_compressed(os.path.join(HERE, "c1_signal_check.npz"),
                        sum=np.array(float(np.sum(sig22.astype(np.float64)))),
                        sumsq=np.array(float(np.sum(sig22.astype(np.float64) ** 2))),
                        n=np.array(len(sig22)), head=sig22[:64], mid=sig22[500000:500064])
    # sparse fb triplets + window: an independent statement of the front-end tables
    fb = sd_np["mel_spectrogram.mel_scale.fb"]
    r, c = np.nonzero(fb)
    np.savez_compressed(os.path.join(HERE, "mel_tables.npz"), rows=r.astype(np.int32), cols=c.astype(np.int32),
                        vals=fb[r, c], window=sd_np["mel_spectrogram.spectrogram.window"])
    print("goldens written to", HERE)
    if "--c3" in sys.argv or not os.path.exists(os.path.join(HERE, "c3_recording.npz")):
        make_c3(model, det, settings)
    if not os.path.exists(os.path.join(HERE, "c1_hostile.npz")):
        make_hostile(SpecUNet_2D, det, settings)


def make_hostile(SpecUNet_2D, det, settings):
    """The C1 file through the reference's SpecUNet_2D with the "hostile-scale" synthetic checkpoint (synth.HOSTILE_GAINS: BatchNorm
    gains put the outputs of conv3_1 ... conv7 near 1e-3 or 1e3, the convolutions behind them undo it; the same function up to fp32
    rounding, so its logits are its own): per-window logits, the reference's averaging / regions, per-block output magnitudes."""
    sd = synth.to_torch_state_dict(synth.make_state_dict(0, hostile=True))
    model = SpecUNet_2D()
    model.load_state_dict(sd, strict=True)
    model.eval()
    pcm = synth.to_pcm16(synth.synth_audio(1001, 60.0, 16000, 1))
    sig22, _, info = O.load_audio_from_bytes(synth.wav_bytes(pcm, 16000))
    padded = O.pad_3s(sig22)
    starts = O.plan_windows(info["frames"] / info["sr"])
    sig_t = torch.from_numpy(padded)
    taps, hooks = {}, []
    for name in ["conv3_1", "conv4_1", "conv_bottleneck", "encoder_out", "conv6", "conv7", "conv8"]:
        hooks.append(getattr(model, name).register_forward_hook(
            lambda m, i, o, name=name: taps.__setitem__(name, max(taps.get(name, 0.0), float(o.abs().max())))))
    logits = []
    for s0 in range(0, len(starts), settings.prediction_batch_size):
        idx = starts[s0:s0 + settings.prediction_batch_size]
        sl = torch.stack([sig_t[int(i):int(i) + 66150] for i in idx])
        _, mask = model(sl)
        logits.append(mask.numpy())
    for h in hooks:
        h.remove()
    logits = np.vstack(logits)
    fkey = "/data/site a/c1_seed1001.wav"
    secs = len(padded) / settings.vad_resample
    avg = det.average_overlapping_detections({fkey: logits}, secs)
    regions = det.find_speech_regions({fkey: avg}, break_duration=0.5)
    avg_vals = np.array([a for a, _ in avg[fkey]], dtype=np.float64)
    reg_f = np.array([(float(s) - 3, float(e) - 3) for s, e in regions[fkey]], dtype=np.float64).reshape(-1, 2)
    print("hostile checkpoint: |block output| max", {k: f"{v:.3g}" for k, v in taps.items()},
          f"regions={len(reg_f)} logit range [{logits.min():.3f},{logits.max():.3f}]")
    np.savez_compressed(os.path.join(HERE, "c1_hostile.npz"), logits=logits.astype(np.float32), avg=avg_vals, regions=reg_f,
                        absmax_names=np.array(list(taps.keys())), absmax=np.array(list(taps.values())))


def make_c3(model, det, settings):
    """One full BASELINE config-3 recording (10 min, 16 kHz mono, seed 3000: the recording bench.py's headline repeats) through the
    reference's own SpecUNet_2D.forward in batches of settings.prediction_batch_size, NNDetector.average_overlapping_detections and
    find_speech_regions (pytorch_neural_nets.py:142-197, NNDetector.py:153-190,103-143), rows as worker.py:100-125 makes them, CSV
    text from pandas: all 1005 windows' logits, every averaged bin, the regions and the table."""
    import pandas as pd
    pcm = synth.to_pcm16(synth.synth_audio(3000, 600.0, 16000, 1))
    wav = synth.wav_bytes(pcm, 16000)
    sig22, _, info = O.load_audio_from_bytes(wav)
    duration = info["frames"] / info["sr"]
    padded = O.pad_3s(sig22)
    starts = O.plan_windows(duration)
    sig_t = torch.from_numpy(padded)
    logits = []
    for s0 in range(0, len(starts), settings.prediction_batch_size):
        idx = starts[s0:s0 + settings.prediction_batch_size]
        sl = torch.stack([sig_t[int(i):int(i) + 66150] for i in idx])
        _, mask = model(sl)
        logits.append(mask.numpy())
    logits = np.vstack(logits)
    fkey = "/data/site b/c3_seed3000.wav"
    secs = len(padded) / settings.vad_resample
    avg = det.average_overlapping_detections({fkey: logits}, secs)
    regions = det.find_speech_regions({fkey: avg}, break_duration=0.5)
    avg_vals = np.array([a for a, _ in avg[fkey]], dtype=np.float64)
    reg_f = np.array([(float(s) - 3, float(e) - 3) for s, e in regions[fkey]], dtype=np.float64).reshape(-1, 2)
    column_types = {'ID': 'int64', 'file_path': str, 'file_name': str, 'start_time': str, 'end_time': str,
                    'erase': int, 'user_comment': str, 'review_datetime': 'datetime64[ns]'}
    df = pd.DataFrame(columns=column_types.keys()).astype(column_types)
    for k, (s, e) in enumerate(reg_f):
        df.loc[len(df)] = {'ID': k + 1, 'file_path': os.path.dirname(fkey), 'file_name': os.path.basename(fkey),
                           'start_time': float(s), 'end_time': float(e), 'erase': 0, 'user_comment': '', 'review_datetime': ''}
    print(f"C3 recording: W={len(starts)} bins={len(avg_vals)} regions={len(reg_f)} logit range [{logits.min():.3f},{logits.max():.3f}]")
    np.savez_compressed(os.path.join(HERE, "c3_recording.npz"), logits=logits.astype(np.float32), avg=avg_vals, regions=reg_f,
                        csv=np.array(df.to_csv(index=False)), file_key=np.array(fkey), duration=np.array(duration),
                        n_padded=np.array(len(padded)), sig_sum=np.array(float(np.sum(sig22.astype(np.float64)))))


if __name__ == "__main__":
    main()
