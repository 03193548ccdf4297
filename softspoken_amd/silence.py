"""Headless silencer: what the reference's SilenceWorker.run does
(root/code/frontend/silencer_ui.py:918-1015) without Qt -- every recording that has review rows with
erase == 1 is rewritten as `<name>_silenced.wav` with those intervals zeroed.

The samples go through `ss_silence_pcm` (decode, zero, 16-bit encode in one device pass); this module
only groups the rows, maps the file and writes header + samples.
"""
from __future__ import annotations

import os

from root.code.backend import voice_activity
from . import native


class SilenceJob:
    """Signals of the reference's worker (silencer_ui.py:908-916) as plain callbacks; `stop()` as there."""

    def __init__(self, review_df, output_dir, ctx=None, file_started=None, file_complete=None,
                 overall_progress=None, finished=None):
        self.review_df, self.output_dir = review_df, output_dir
        self.ctx = ctx
        self.file_started, self.file_complete = file_started, file_complete
        self.overall_progress, self.finished = overall_progress, finished
        self.stop_requested = False
        self.errors: dict[str, str] = {}
        self.outputs: list[str] = []

    def stop(self):
        self.stop_requested = True

    def _emit(self, cb, *a):
        if cb is not None:
            cb(*a)

    def run(self):
        erase = self.review_df[self.review_df['erase'] == 1]
        if erase.empty:
            self._emit(self.finished)
            return self.outputs
        ctx = self.ctx or voice_activity.audio_context()
        groups = erase.groupby(['file_path', 'file_name'])
        total, done = len(groups), 0
        for (fpath, fname), rows in groups:
            if self.stop_requested:
                break
            src = os.path.join(fpath, fname)
            self._emit(self.file_started, src)
            out_path = os.path.join(self.output_dir, f"{os.path.splitext(fname)[0]}_silenced.wav")
            try:
                buf = voice_activity._map_file(src)
                info = native.wav_parse(buf)
                pcm = buf[info.data_offset: info.data_offset + info.data_bytes]
                regions = [(float(s), float(e)) for s, e in zip(rows['start_time'], rows['end_time'])]
                with voice_activity._audio_lock:
                    out = ctx.silence_pcm(pcm, info.format, info.sample_rate, info.channels, info.frames, regions)
                with open(out_path, "wb") as fh:
                    fh.write(native.wav_header_pcm16(info.sample_rate, info.channels, info.frames))
                    fh.write(out.tobytes())
                self.outputs.append(out_path)
                self._emit(self.file_complete, out_path)
            except Exception as exc:            # the reference logs and moves on to the next file (:963-969, :996-997)
                self.errors[src] = str(exc)
                print(f"Error silencing {src}: {exc}")
            done += 1
            self._emit(self.overall_progress, int(done / total * 100))
        self._emit(self.finished)
        return self.outputs


def silence_files(review_df, output_dir, ctx=None):
    """-> list of written paths."""
    return SilenceJob(review_df, output_dir, ctx=ctx).run()
