"""Headless review table: what the reference's ReviewDetectionsScreen does to the detector's CSV
between "Run Voice Detector" and the exporters, without Qt.

    seed        review_detections.py:220-237   existing review CSV, else detections CSV + length filter
    filter      :764-770                        keep rows with end - start > settings.minimum_detection_len
    ID column   :62-86                          ID first; blank IDs continue after the largest one
    populate    :968-999                        sort by (file_name, start_time), times rounded to 3 dp,
                                                cells become text ("Yes"/"" for erase, "" for NaN)
    label       :683-709                        erase flag + review_datetime stamp on one row
    save        :92-166                         text cells -> DataFrame -> CSV, then the three exporters

The screen keeps its rows in a QTableWidget of strings and rebuilds the DataFrame from those strings on
every save; the same text round trip is kept here (`self.cells`), because it decides what lands in the
CSV (for example 0.30000000000000004 -> "0.3", NaN -> "", erase 1 -> "Yes" -> 1).
"""
from __future__ import annotations

import datetime
import os
from pathlib import Path

import pandas as pd

from root.code.backend import settings
from root.code.frontend import review_exporter

REVIEW_COLUMNS = ["ID", "file_path", "file_name", "start_time", "end_time", "erase", "user_comment",
                  "review_datetime"]


def filter_by_minimum_detection_len(df: pd.DataFrame, minimum=None) -> pd.DataFrame:
    """Rows strictly longer than `minimum` seconds (review_detections.py:770; settings.py:26)."""
    if minimum is None:
        minimum = settings.minimum_detection_len
    return df[(df['end_time'] - df['start_time']) > minimum]


def ensure_id_column_first(df: pd.DataFrame) -> pd.DataFrame:
    """review_detections.py:62-71."""
    df = df.copy()
    if "ID" not in df.columns:
        df.insert(0, "ID", range(1, len(df) + 1))
        return df
    return df[["ID"] + [c for c in df.columns if c != "ID"]]


def assign_missing_ids(df: pd.DataFrame) -> pd.DataFrame:
    """Blank / non-numeric IDs get max+1, max+2, ... in row order (review_detections.py:73-86)."""
    df = df.copy()
    ids = pd.to_numeric(df["ID"], errors="coerce")
    top = ids.max(skipna=True)
    nxt = int(top) + 1 if pd.notna(top) else 1
    for row in ids.index[ids.isna()]:
        ids.at[row] = nxt
        nxt += 1
    df["ID"] = ids.astype(int)
    return df


class ReviewTable:
    """`project_manager` needs `current_project` with 'detections_file', 'review_file', 'name', and
    `projects_folder` (the attributes the screen reads, review_detections.py:220-221,138,150,156)."""

    def __init__(self, project_manager):
        self.project_manager = project_manager
        proj = project_manager.current_project
        detections, review = proj.get('detections_file'), proj.get('review_file')
        if review is not None and os.path.exists(review):
            self.csv_data = pd.read_csv(review)
        elif detections is not None and os.path.exists(detections):
            self.csv_data = filter_by_minimum_detection_len(pd.read_csv(detections))
        else:
            self.csv_data = pd.DataFrame(columns=REVIEW_COLUMNS)
        self.csv_data = ensure_id_column_first(self.csv_data)
        self.headers: list[str] = []
        self.cells: list[list[str]] = []
        self.populate_table()

    # ---- DataFrame -> text cells (populate_table, review_detections.py:968-999) ----------------
    def populate_table(self):
        self.csv_data.sort_values(by=['file_name', 'start_time'], ignore_index=True, inplace=True)
        self.csv_data[['start_time', 'end_time']] = self.csv_data[['start_time', 'end_time']].round(3)
        self.headers = list(self.csv_data.columns)
        self.cells = []
        for _, row in self.csv_data.iterrows():
            line = []
            for col in self.headers:
                v = row[col]
                if col == "erase":
                    line.append("Yes" if v == 1 else "")
                elif pd.isna(v):
                    line.append("")
                else:
                    line.append(str(v))
            self.cells.append(line)

    # ---- labelling (apply_label_to_current_detection, :683-709) --------------------------------
    def apply_label(self, index: int, erase_flag: int, now: datetime.datetime | None = None):
        stamp = (now or datetime.datetime.now()).strftime("%Y-%m-%d %H:%M:%S")
        self.csv_data.at[index, "erase"] = erase_flag
        self.csv_data.at[index, "review_datetime"] = stamp
        self.cells[index][self.headers.index("erase")] = "Yes" if erase_flag == 1 else ""
        self.cells[index][self.headers.index("review_datetime")] = stamp

    def set_comment(self, index: int, text: str):
        self.cells[index][self.headers.index("user_comment")] = text

    def add_row(self, file_path, file_name, start_time, end_time, at=None):
        """A manually drawn detection: blank ID, times as %.3f text (review_detections.py:606-626)."""
        line = {"file_path": file_path, "file_name": file_name,
                "start_time": f"{start_time:.3f}", "end_time": f"{end_time:.3f}"}
        self.cells.insert(len(self.cells) if at is None else at, [line.get(h, "") for h in self.headers])

    # ---- text cells -> DataFrame -> files (save_review, :92-166) -------------------------------
    def save_review(self, persist: bool = True) -> pd.DataFrame:
        df = pd.DataFrame(self.cells, columns=self.headers)
        df = assign_missing_ids(ensure_id_column_first(df))
        for col in ("start_time", "end_time"):
            if col in df.columns:
                df[col] = pd.to_numeric(df[col], errors="coerce")
        if "erase" in df.columns:
            df["erase"] = df["erase"].apply(lambda x: 1 if x.strip().lower() == "yes" else 0)
        self.csv_data = df
        if persist:
            proj = self.project_manager.current_project
            df.to_csv(proj['review_file'], index=False)
            exporter = review_exporter.ReviewExportManager(df)
            for tr in (review_exporter.AudacityTxtTransform(), review_exporter.KaleidoscopeCsvTransform(),
                       review_exporter.RavenTxtTransform()):
                exporter.register_transform(tr)
                exporter.export(tr.name, dst=".", base_dir=Path(self.project_manager.projects_folder),
                                project_name=proj["name"])
        return df
