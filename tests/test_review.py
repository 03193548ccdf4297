"""SURVEY.md 8(f) N1 + N2: the headless review table and the label exporters against
tests/golden/review_case.json, which the reference's own ReviewDetectionsScreen methods and
review_exporter wrote (tests/golden/make_review_golden.py).  Text files, so the bar is byte equality."""
import datetime
import json
import os
import types
import wave

import numpy as np
import pandas as pd
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CASE = json.load(open(os.path.join(HERE, "golden", "review_case.json")))


def _write_wav(path, seconds, rate):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with wave.open(path, "wb") as fh:
        fh.setnchannels(1)
        fh.setsampwidth(2)
        fh.setframerate(rate)
        fh.writeframes(np.zeros(int(round(seconds * rate)), dtype="<i2").tobytes())


def _collect(root):
    out = {}
    for base, _, files in os.walk(os.path.join(root, "projects")):
        for f in files:
            rel = os.path.relpath(os.path.join(base, f), root)
            if rel.endswith("_detections.csv"):
                continue
            with open(os.path.join(base, f), newline="") as fh:
                out[rel] = fh.read().replace(root, "@ROOT@")
    return out


@pytest.fixture()
def project(tmp_path, monkeypatch):
    root = str(tmp_path)
    for d, f, seconds, rate in CASE["recordings"]:
        if seconds is not None:
            _write_wav(os.path.join(root, "audio", d, f), seconds, rate)
    proj = os.path.join(root, "projects")
    os.makedirs(proj)
    det = os.path.join(proj, f"{CASE['project']}_detections.csv")
    with open(det, "w", newline="") as fh:
        fh.write(CASE["stages"]["detections_csv"].replace("@ROOT@", root))
    monkeypatch.chdir(root)
    pm = types.SimpleNamespace(
        current_project={"name": CASE["project"], "detections_file": det,
                         "review_file": os.path.join(proj, f"{CASE['project']}_review.csv")},
        projects_folder=proj)
    return root, pm


def _same(got, want):
    assert sorted(got) == sorted(want)
    for name in want:
        assert got[name] == want[name], name


def test_review_workflow_matches_reference(project):
    from softspoken_amd.review import ReviewTable
    from root.code.backend import settings
    assert settings.minimum_detection_len == CASE["minimum_detection_len"]
    root, pm = project
    now = datetime.datetime.fromisoformat(CASE["fixed_now"])

    t = ReviewTable(pm)                       # first opening: detections CSV, short rows dropped
    t.save_review(persist=True)
    _same(_collect(root), CASE["stages"]["first_save"])

    t.apply_label(1, 1, now=now)
    t.save_review(persist=True)
    t.apply_label(3, 0, now=now)
    t.set_comment(0, 'two people, "quoted", far')
    t.save_review(persist=True)
    _same(_collect(root), CASE["stages"]["after_labels"])

    t2 = ReviewTable(pm)                      # second opening reads the review CSV back
    d, f, _, _ = CASE["recordings"][0]
    t2.add_row(os.path.join(root, "audio", d), f, 2.0, 2.25, at=2)
    df = t2.save_review(persist=True)
    _same(_collect(root), CASE["stages"]["reopened"])
    assert df["ID"].tolist() == [7, 4, 8, 3, 6, 1, 5] and df["erase"].tolist() == [0, 1, 0, 0, 0, 0, 0]


def test_filter_is_strict_and_ids_continue():
    from softspoken_amd import review
    df = pd.DataFrame({"ID": [5, None, "x", 2], "start_time": [0.0, 1.0, 2.0, 3.0],
                       "end_time": [0.1, 1.1000001, 2.05, 4.0]})
    kept = review.filter_by_minimum_detection_len(df, 0.1)
    assert kept.index.tolist() == [1, 3]                      # 0.1 itself is not longer than 0.1
    ids = review.assign_missing_ids(df)["ID"].tolist()
    assert ids == [5, 6, 7, 2]
    assert review.ensure_id_column_first(df[["start_time", "ID"]]).columns.tolist() == ["ID", "start_time"]
    fresh = review.ensure_id_column_first(df[["start_time"]])
    assert fresh["ID"].tolist() == [1, 2, 3, 4]


def test_export_manager_contract(tmp_path):
    from root.code.frontend import review_exporter as rx
    df = pd.DataFrame({"file_path": ["/a"], "file_name": ["x.wav"], "start_time": [0.5], "end_time": [1.0]})
    m = rx.ReviewExportManager(df)

    class Table(rx.Transform):
        name, extension = "table", ".csv"

        def __call__(self, df, **kw):
            return df[["start_time"]]

    class Text(rx.Transform):
        name, extension = "text", ".txt"

        def __call__(self, df, **kw):
            return "hello\n"

    class Bad(rx.Transform):
        name = "bad"

        def __call__(self, df, **kw):
            return 3

    m.register_transform(Table())
    m.transform(Text)
    m.register_transform(Bad())
    with pytest.raises(KeyError):
        m.register_transform(Table())
    with pytest.raises(KeyError):
        m.export("nope", tmp_path)
    assert m.export("table", tmp_path).read_text() == "start_time\n0.5\n"
    assert m.export("text", tmp_path / "deep" / "out.txt").read_text() == "hello\n"
    with pytest.raises(TypeError):
        m.export("bad", tmp_path)
    with pytest.raises(ValueError, match="missing column"):
        rx.AudacityTxtTransform()(df.drop(columns=["file_name"]), base_dir=tmp_path, project_name="p")
    with pytest.raises(ValueError, match="missing column"):
        rx.RavenTxtTransform()(df.drop(columns=["file_path"]), base_dir=tmp_path, project_name="p")
