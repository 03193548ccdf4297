"""NNDetector on MI355X -- the reference's detector class (root/code/frontend/NNDetector.py:11-190), same
constructor, attributes, methods and return types, backed by libsoftspoken_hip.so.

What stays identical for callers (silencer_ui.py:225-234, worker.py:65-97):
  plan_detection_job() -> {file: int64 start indexes}                        (reference :55-82)
  process_batch(audio, idxs) -> (speech_pred, mask_pred) numpy               (:84-101)
  average_overlapping_detections(dets, secs) -> {file: [(avg, "t.tttt")]}    (:153-190)
  find_speech_regions({file: {file: [...]}}, break) -> {file: [(str, str)]}  (:103-143)
  load_checkpoint(model, path) -> epoch + 1                                   (:42-53)
What changes underneath: the padded signal is uploaded to HBM once per file instead of once per
batch (the reference re-creates the device tensor in every process_batch call, :90), and
`detect_files` runs a whole job inside the library (windows of all files batched together,
averaging on the GPU).  A missing checkpoint is an error here unless allow_untrained=True; the
reference prints and carries on with random weights (:51-53).
"""
from __future__ import annotations

import logging
import math
import os

import numpy as np
import torch

from root.code.backend import settings
from root.code.backend.pytorch_neural_nets import SpecUNet_2D
from root.code.backend.voice_activity import add_file_to_context, get_audio_data
from softspoken_amd import native as _native

try:                                   # 64-bit content hash at memory speed when the wheel is there, a 64-bit BLAKE2 from the stdlib otherwise
    from xxhash import xxh3_64_intdigest as _hash_bytes
except Exception:                      # pragma: no cover
    import hashlib

    def _hash_bytes(buf):              # (a 32-bit checksum could serve a stale file on a collision)
        return int.from_bytes(hashlib.blake2b(buf, digest_size=8).digest(), "little")


def _content_hash(a: np.ndarray) -> int:
    return _hash_bytes(memoryview(a).cast("B"))


_BINS_PER_SECOND = 256 / 3           # NNDetector.py:185
_TIME_RESOLUTION = 3 / 256           # NNDetector.py:172


class NNDetector():
    def __init__(self, project_manager, allow_untrained: bool = False, checkpoint_path: str | None = None):
        # the reference picks "cuda" when torch sees a GPU (:22); ROCm builds of torch report the MI355X as cuda
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        logging.info(f"Device: {self.device} (compute: libsoftspoken_hip on gfx950)")
        torch.set_grad_enabled(False)

        self.project_manager = project_manager
        self.model = SpecUNet_2D(compute_spec_output=True)
        path = checkpoint_path or os.path.join(settings.model_dir, settings.model_name)
        if self.load_checkpoint(self.model, path) < 0 and not allow_untrained:
            raise FileNotFoundError(
                f"model checkpoint not found: {path} (the reference would silently run untrained weights here; "
                "pass allow_untrained=True to do that)")
        self.model.eval()

        self.files_to_process = self.project_manager.get_unprocessed_list()
        self.detections_project = {f: [] for f in self.files_to_process}
        self._resident = None     # (key, file_id) of the signal currently in HBM for process_batch
        self._stage = None        # ingest: (context, [device staging buffer, capacity] x 2) for file_prefetch
        self._stage_turn = 0

    # -- checkpoint ---------------------------------------------------------------------------------------
    def load_checkpoint(self, model, file_path='checkpoint.pth'):
        if not os.path.exists(file_path):
            print("No checkpoint found. Starting training from scratch.")
            return -1
        checkpoint = torch.load(file_path, map_location="cpu", weights_only=True)
        model.load_state_dict(checkpoint['model_state_dict'])
        return checkpoint['epoch'] + 1

    # -- planning -------------------------------------------------------------------------------------------
    def plan_detection_job(self):
        plan = self.detections_project
        for file in plan.keys():
            logging.info(f"Analyzing file: {file}")
            (audio_len_seconds, _) = get_audio_data(file)
            rate = settings.vad_resample
            padded_len = round(audio_len_seconds * rate) + (3 * 2 * rate)
            per_window = rate * 3
            per_step = math.floor(rate * settings.step_size)
            count = int(np.ceil((padded_len - per_window) / per_step))
            plan[file] = np.arange(count) * per_step
        return plan

    # -- inference ------------------------------------------------------------------------------------------
    def _resident_file(self, audio_data, ctx=None):
        """Upload `audio_data` (already 3 s-padded, float32) unless the signal in HBM still is this one: same length and the same
        content by a checksum over EVERY sample (an address or a few probe samples would serve a stale file for a buffer that
        was reused or edited in place), in a context whose arena nobody has reset since (SpecUNet_2D.forward and detect_files
        do).  The checksum of a 10-minute file costs a few milliseconds per call; the reference re-uploads the file instead (:90).
        ctx: the context to serve from (default: the detector's own; the fp32 side context after an SS_ERR_RANGE on this signal)."""
        a = np.ascontiguousarray(audio_data, dtype=np.float32)
        key = (a.size, _content_hash(a))
        if ctx is None:
            ctx = self.model.hip_context()
        r = self._resident
        if r is None or r[0] != key or r[2] is not ctx or r[3] != ctx.reset_generation():
            ctx.reset()
            fid = ctx.add_f32_22k(a, padded=True)
            self._resident = (key, fid, ctx, ctx.reset_generation())
        return ctx, self._resident[1], key

    def process_batch(self, audio_data, batch_indexes):
        holder = {}

        def run(ctx):                  # (after an SS_ERR_RANGE the context is another one, in fp32: the signal is uploaded to it)
            ctx, fid, holder["key"] = self._resident_file(audio_data, ctx)
            return ctx.infer_windows(fid, np.asarray(batch_indexes, dtype=np.int64), want_spec=self.model.compute_spec_output)
        m = self.model
        # a signal the f16x2 mode has refused once stays on the fp32 side context: its later batches do not pay the refusal again
        if self._resident is not None and self._resident[2] is m._fp32_tmp and m._fp32_tmp is not None and m._fp32_tmp.alive:
            a = np.ascontiguousarray(audio_data, dtype=np.float32)
            if self._resident[0] == (a.size, _content_hash(a)):
                return run(m._fp32_tmp)
        try:
            return run(m.hip_context())
        except _native.NativeError as e:
            if e.code != _native.SS_ERR_RANGE or m.effective_precision() != "f16x2":
                raise
            if m.range_refused(holder.get("key"), e):
                return run(m.hip_context())
            return run(m.fp32_context())

    # -- post-processing (host side of the path) ---------------------------------------------------------
    def average_overlapping_detections(self, detections, audio_length_seconds, padding=0, min_count=1):
        averaged = {}
        for file, per_window in detections.items():
            n_bins = int(round(audio_length_seconds * 256 / 3))
            total = np.zeros(n_bins + 2 * padding)
            hits = np.zeros(n_bins + 2 * padding)
            for i, logits in enumerate(per_window):
                at = padding + int(round(i * settings.step_size / _TIME_RESOLUTION))
                total[at:at + 256] += np.asarray(logits).reshape(-1)
                hits[at:at + 256] += 1
            keep = np.nonzero(hits >= min_count)[0]
            means = total[keep] / hits[keep]
            averaged[file] = [(m, f"{i / _BINS_PER_SECOND:.4f}") for m, i in zip(means, keep)]
        return averaged

    def find_speech_regions(self, averaged_detections, break_duration=0.5):
        found = {}
        for file, nested in averaged_detections.items():
            series = nested[file]
            runs, first, last = [], None, None
            for value, stamp in series:
                if value > settings.threshold:
                    if first is None:
                        first = stamp
                    last = stamp
                elif first is not None:
                    runs.append((first, last))
                    first = None
            if first is not None:
                runs.append((first, last))
            merged = []
            for run in runs:
                if merged and float(run[0]) - float(merged[-1][1]) <= break_duration:
                    merged[-1] = (merged[-1][0], run[1])
                else:
                    merged.append(run)
            found[file] = merged
        return found

    def extract_filename(self, file_path):
        return os.path.basename(file_path).rsplit('.', 1)[0]

    # -- one file at a time, the device a file ahead of the host (not in the reference: its loop is synchronous, worker.py:49-139) ----
    # file_prefetch(k + 1) while file k computes: header walk + asynchronous upload of the samples into one of two staging buffers
    # in HBM (the library's copy stream); file_begin(k + 1) as soon as file k's run has ended: decode + resample + every pass of
    # the network + averaging enqueued, nothing waited for; the caller then files the rows of file k (pandas, CSV) while the device
    # works, and reads k + 1's progress (file_poll) and regions (file_end) when its turn comes.  ProcessWorker.run drives it.
    def _staging(self, ctx, need):
        if self._stage is None:
            self._stage = {}
        for k in [k for k, v in self._stage.items() if v[0] is not ctx and not v[0].alive]:     # (staging of contexts that were closed)
            del self._stage[k]
        st = self._stage.setdefault(id(ctx), [ctx, [[0, 0], [0, 0]], 0])
        slot = st[1][st[2] & 1]
        st[2] += 1
        if slot[1] < need:
            if slot[0]:
                ctx.device_free(slot[0])
            slot[1] = int(need * 1.25) + 4096
            slot[0] = ctx.device_alloc(slot[1])
        return slot

    def file_prefetch(self, file, which=0, ctx=None):
        """-> handle for file_begin.  Allowed while another file's run is in flight.  which: the context (0 / 1) the file will run on."""
        from root.code.backend.voice_activity import _map_file
        ctx = ctx or self.model.hip_context(which)
        buf = _map_file(file)
        info = _native.wav_parse(buf)                                   # raises on a file that is not a WAV: the caller reports and skips it
        slot = self._staging(ctx, info.frames * info.channels * (info.bits // 8) + 64)
        infos = ctx.upload_wav_batch_async([buf], slot[0], slot[1])
        return (ctx, slot[0], infos[0], buf)

    class _FileToken:
        """A file in flight: its context, file id, and whether the reference's progress values have all been reported."""
        __slots__ = ("ctx", "fid", "file", "brk", "buf", "which", "reported")

        def __init__(self, ctx, fid, file, brk, buf, which):
            self.ctx, self.fid, self.file, self.brk, self.buf, self.which, self.reported = ctx, fid, file, brk, buf, which, False

    def file_begin(self, file, handle=None, break_duration=0.5, which=0, ctx=None):
        """Enqueue everything for `file` -> token for file_poll / file_end.  No other file may be in flight on the same context
        (`which`); one file may be in flight on each.  ctx: run on this context instead (the fp32 side context of a re-run)."""
        want = ctx or self.model.hip_context(which)
        if handle is None or handle[0] is not want:   # (a fall-back to fp32 in between: the staging belonged to the old context)
            handle = self.file_prefetch(file, which, ctx=want)
        ctx, dev, info, _buf = handle
        ctx.reset()
        if which == 0:
            self._resident = None
        fid = ctx.add_pcm_device(dev, info.format, info.sample_rate, info.channels, info.frames)
        ctx.run_begin(settings.threshold, break_duration, track=True)
        return self._FileToken(ctx, fid, file, break_duration, _buf, which)     # (_buf: the mapped file stays alive while its samples may still be in flight)

    def file_poll(self, token, progress=None, block=True):
        """Report the reference's progress values (worker.py:82-84: done = 32, 64, ..., total windows) that have completed."""
        if token.ctx.alive:
            token.ctx.run_poll(progress, block)
            if block:
                token.reported = True

    def file_end(self, token, progress=None):
        """-> [(start_s, end_s)] of the file (worker.py:100's "-3 s" applied).  When the f16x2 mode reports a value it cannot
        represent (SS_ERR_RANGE) the file is run again in fp32: this file alone on the model's fp32 side context, or -- from the
        second such file on -- with the detector switched to fp32 (SpecUNet_2D.range_refused).  progress: a file that is run again
        reports the reference's progress values through it unless file_poll has already reported them all."""
        ctx, fid, file, brk, which = token.ctx, token.fid, token.file, token.brk, token.which

        def again(on=None):                   # on the (new) context of the file's own turn: the other one may have the next file in flight
            t2 = self.file_begin(file, None, brk, which, ctx=on)
            t2.ctx.run_poll(None if token.reported else progress, True)
            token.reported = True
            t2.ctx.run_end()
            return [(float(s), float(e)) for s, e in t2.ctx.regions(t2.fid)]

        if not ctx.alive:                     # a switch to fp32 while this file was in flight on the other context closed it
            return again()
        try:
            ctx.run_end()
        except _native.NativeError as e:
            if e.code != _native.SS_ERR_RANGE or ctx.precision != "f16x2":
                raise
            if self.model.range_refused(file, e):
                return again()
            return again(self.model.fp32_context())
        return [(float(s), float(e)) for s, e in ctx.regions(fid)]

    def file_abort(self, token):
        """Wait for a file in flight and drop its results (stop requested, or its poll / end raised)."""
        try:
            if token.ctx.alive:
                token.ctx.run_end()
        except Exception:
            pass

    # -- whole-job fast path (not in the reference) -----------------------------------------------------
    def detect_files(self, files, progress=None, stop_flag=None, break_duration=0.5):
        """Run the library's job loop over `files` -> {file: [(start_s, end_s)]} with the worker's "-3 s"
        already applied (worker.py:100), or None if stopped.  Windows of all files share batches."""
        def run(ctx):
            ctx.reset()
            self._resident = None
            ids = []
            for f in files:
                fid, _ = add_file_to_context(ctx, f)
                ids.append(fid)
            if not ctx.run(settings.threshold, break_duration, progress, stop_flag):
                return None
            if not ids:
                return {}
            counts, reg = ctx.regions_batch(ids[0], len(ids))          # ids are consecutive after the reset above
            out, at = {}, 0
            for f, n in zip(files, counts.tolist()):
                out[f] = [(float(s), float(e)) for s, e in reg[at:at + n]]
                at += n
            return out
        # f16x2 reports values it cannot represent (SS_ERR_RANGE) instead of returning scores: the job is then run again in fp32,
        # which, like the reference's fp32 (pytorch_neural_nets.py:142-197), cannot fail on magnitude
        return self.model.with_range_fallback(run, key=("job",) + tuple(files))
