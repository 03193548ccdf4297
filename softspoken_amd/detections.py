"""Headless stand-in for the reference's DetectionProject (root/code/frontend/silencer_ui.py:775-817):
the detections DataFrame schema and its CSV round trip, without Qt.  The GUI's own class keeps
working with the worker; this one is for command-line jobs, the bench and the tests.
"""
from __future__ import annotations

import os

import numpy as np
import pandas as pd

COLUMN_TYPES = {
    'ID': 'int64', 'file_path': str, 'file_name': str, 'start_time': str, 'end_time': str,
    'erase': int, 'user_comment': str, 'review_datetime': 'datetime64[ns]',
}
CSV_HEADER = ",".join(COLUMN_TYPES)


class ProjectSettings:
    """Minimal object with the one attribute DetectionProject reads (silencer_ui.py:792)."""

    def __init__(self, detections_file):
        self.current_project = {'detections_file': detections_file}


class DetectionProject:
    def __init__(self, project_settings):
        self.settings = project_settings
        self.columns = COLUMN_TYPES.keys()
        path = self.settings.current_project['detections_file']
        if os.path.exists(path):
            self.df = pd.read_csv(path)
            if 'ID' not in self.df.columns:
                self.df.insert(0, 'ID', range(1, len(self.df) + 1))
            else:
                ids = pd.to_numeric(self.df['ID'], errors='coerce')
                missing = ids.isna()
                if missing.any():
                    top = ids.dropna().max()
                    nxt = (int(top) if not np.isnan(top) else 0) + 1
                    for row in self.df.index[missing]:
                        ids.at[row] = nxt
                        nxt += 1
                self.df['ID'] = ids.astype('int64')
            if 'review_datetime' in self.df.columns:
                self.df['review_datetime'] = pd.to_datetime(self.df['review_datetime'], errors='coerce')
            self.df = self.df.reindex(columns=self.columns).astype(COLUMN_TYPES)
        else:
            self.df = pd.DataFrame(columns=self.columns).astype(COLUMN_TYPES)

    def save_detections(self):
        self.df.to_csv(self.settings.current_project['detections_file'], index=False)
