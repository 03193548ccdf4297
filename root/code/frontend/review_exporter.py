"""Label exporters for reviewed detections -- same names, arguments, file layout and file bytes as the
reference's root/code/frontend/review_exporter.py (Transform :31-50, ReviewExportManager :53-126,
AudacityTxtTransform :129-215, KaleidoscopeCsvTransform :218-338, RavenTxtTransform :341-481), so
`save_review` (review_detections.py:141-166) can register and call them unchanged.

The one thing that differs is where a recording's length comes from: the reference asks soundfile
(`sf.info`, :26-28); here the header is read by the library's own WAV parser (`ss_wav_parse`), the
same one the detector uses, so no audio library is needed for an export.
"""
from __future__ import annotations

import os
from abc import ABC, abstractmethod
from pathlib import Path
from typing import Dict, Union

import pandas as pd


def _wav_duration(path: Union[str, Path]) -> float:
    """Seconds of audio in a WAV file, from its header (reference :26-28)."""
    from root.code.backend import voice_activity
    return voice_activity.get_audio_data(str(path))[0]


class Transform(ABC):
    """One application-specific export.  `__call__` gets a copy of the review table and returns a
    DataFrame (saved as CSV by the manager), str/bytes (written verbatim) or None (wrote its own files)."""

    name: str = "unnamed"
    extension: str = ".csv"

    @abstractmethod
    def __call__(self, df: pd.DataFrame, **kwargs):
        raise NotImplementedError


class ReviewExportManager:
    """Registry of transforms over one review table (reference :53-126)."""

    def __init__(self, df: pd.DataFrame):
        self.df = df
        self._registry: Dict[str, Transform] = {}

    def register_transform(self, transform: Transform) -> None:
        if transform.name in self._registry:
            raise KeyError(f"Transform '{transform.name}' already registered")
        self._registry[transform.name] = transform

    def transform(self, cls):
        """Class decorator form of register_transform."""
        self.register_transform(cls())
        return cls

    def export(self, name, dst, make_dirs: bool = True, **kwargs) -> Path:
        if name not in self._registry:
            raise KeyError(f"No transform named '{name}' registered")
        tr = self._registry[name]
        dst = Path(dst)
        if dst.is_dir():
            dst = dst / f"review{tr.extension}"
        if make_dirs:
            dst.parent.mkdir(parents=True, exist_ok=True)
        out = tr(self.df.copy(), **kwargs)
        if out is None:
            return dst
        if isinstance(out, pd.DataFrame):
            out.to_csv(dst, index=False)
        elif isinstance(out, str):
            with dst.open("w") as fh:
                fh.write(out)
        elif isinstance(out, bytes):
            with dst.open("wb") as fh:
                fh.write(out)
        else:
            raise TypeError(f"Unsupported return type from transform ({type(out).__name__}).")
        return dst

    def export_all(self, dst_dir, **kwargs):
        return {name: self.export(name, dst_dir, **kwargs) for name in self._registry}


def _need(df, cols, who):
    missing = set(cols) - set(df.columns)
    if missing:
        raise ValueError(f"{who}: DataFrame missing column(s): {missing}")


def _numeric_times(df):
    df = df.copy()
    for col in ("start_time", "end_time"):
        df[col] = pd.to_numeric(df[col], errors="coerce")
    return df


def _optional(df, col):
    """Column `col`, or blanks on a fresh 0..n-1 index when the table has no such column."""
    return df[col] if col in df.columns else pd.Series([""] * len(df))


class AudacityTxtTransform(Transform):
    """`<base_dir>/Audacity Outputs/<project_name>/<wav stem>.txt`: one label track per recording,
    tab-separated `start  end  comment`, no header (reference :129-215)."""

    name = "audacity"
    extension = ".txt"

    def __call__(self, df, *, base_dir, project_name, comment: str = "Human", precision: int = 6, **kwargs):
        folder = Path(base_dir) / "Audacity Outputs" / project_name
        folder.mkdir(parents=True, exist_ok=True)
        _need(df, ("file_name", "start_time", "end_time"), "AudacityTxtTransform")
        df = _numeric_times(df)
        df.sort_values(["file_name", "start_time"], inplace=True)
        for wav, rows in df.groupby("file_name", sort=False):
            text = "".join(f"{s:.{precision}f}\t{e:.{precision}f}\t{comment}\n"
                           for s, e in zip(rows["start_time"], rows["end_time"]))
            (folder / f"{Path(wav).stem}.txt").write_text(text)
        return None


class KaleidoscopeCsvTransform(Transform):
    """`<base_dir>/Kaleidoscope Outputs/<project_name>/<project_name>.csv` with the columns Kaleidoscope
    needs (INDIR, FOLDER, IN FILE*, OFFSET, DURATION, TOP1MATCH*, MANUAL ID) plus end_time, erase and
    review_datetime for traceability (reference :218-338)."""

    name = "kaleidoscope"
    extension = ".csv"

    def __call__(self, df, *, base_dir, project_name, precision: int = 6, human_label: str = "Human", **kwargs):
        folder = Path(base_dir) / "Kaleidoscope Outputs" / project_name
        folder.mkdir(parents=True, exist_ok=True)
        _need(df, ("file_path", "file_name", "start_time", "end_time"), "KaleidoscopeCsvTransform")
        df = _numeric_times(df)
        paths = [str(p) for p in df["file_path"]]
        indir = os.path.commonpath(paths)
        if not indir.endswith(os.sep):
            indir += os.sep
        rel = [os.path.relpath(p, indir) for p in paths]
        rel = ["" if r == "." else r for r in rel]
        if indir.endswith("\\"):
            indir = indir[:-1]
        table = pd.DataFrame({
            "INDIR": indir,
            "FOLDER": rel,
            "IN FILE*": df["file_name"],
            "OFFSET": df["start_time"].round(precision),
            "DURATION": (df["end_time"] - df["start_time"]).round(precision),
            "TOP1MATCH*": human_label,
            "MANUAL ID": _optional(df, "user_comment"),
            "end_time": df["end_time"].round(precision),
            "erase": _optional(df, "erase"),
            "review_datetime": _optional(df, "review_datetime"),
        })
        table.to_csv(folder / f"{project_name}.csv", index=False)
        return None


class RavenTxtTransform(Transform):
    """`<base_dir>/Raven Outputs/<project_name>/<project_name>_listfile.txt` (recordings in order of
    first appearance) and `<project_name>.txt` (tab-delimited selection table whose times run on
    through the list: each recording is offset by the summed length of those before it)
    (reference :341-481)."""

    name = "raven"
    extension = ".txt"

    def __call__(self, df, *, base_dir, project_name, precision: int = 6, annotation_label: str = "Human",
                 low_freq: int = 0, high_freq: int = 8000, **kwargs):
        folder = Path(base_dir) / "Raven Outputs" / project_name
        folder.mkdir(parents=True, exist_ok=True)
        _need(df, ("file_path", "file_name", "start_time", "end_time"), "RavenTxtTransform")
        df = df.copy()
        if len(df):
            df["abs_path"] = [str(Path(d) / f) for d, f in zip(df["file_path"], df["file_name"])]
        else:
            df["abs_path"] = pd.Series([], dtype=object)
        recordings = pd.unique(df["abs_path"])
        (folder / f"{project_name}_listfile.txt").write_text("\n".join(recordings) + "\n")

        offset, total = {}, 0.0
        for wav in recordings:
            try:
                seconds = _wav_duration(wav)
            except Exception:
                # unreadable recording: its last detection end stands in for its length
                seconds = df.loc[df["abs_path"] == wav, "end_time"].max(skipna=True).item()
            offset[wav] = total
            total += seconds

        def shifted(col):
            vals = [offset[p] + float(t) for p, t in zip(df["abs_path"], df[col])]
            return pd.Series(vals, index=df.index, dtype="float64").round(precision)

        table = pd.DataFrame({
            "Selection": range(1, len(df) + 1),
            "View": "Spectrogram 1",
            "Channel": 1,
            "Begin Time (s)": shifted("start_time"),
            "End Time (s)": shifted("end_time"),
            "Low Freq (Hz)": low_freq,
            "High Freq (Hz)": high_freq,
            "Annotation": annotation_label,
            "Begin Path": df["abs_path"],
            "erase": _optional(df, "erase"),
            "user_comment": _optional(df, "user_comment"),
            "review_datetime": _optional(df, "review_datetime"),
        })
        if "confidence" in df.columns:
            table["confidence"] = df["confidence"]
        table.to_csv(folder / f"{project_name}.txt", sep="\t", index=False, lineterminator="\n")
        return None
