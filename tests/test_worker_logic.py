"""ProcessWorker.run's file pipeline (root/code/backend/worker.py) against a scripted detector, no GPU: the reference's signal order per
file (worker.py:49-139 of the reference), files alternating between two device contexts, the second context given up when its workspace
does not fit (SS_ERR_NOMEM), a file whose poll raises aborted on its context, stop between files."""
import pandas as pd
import pytest

from root.code.backend import settings
from root.code.backend.worker import ProcessWorker


class _Err(Exception):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


class _Model:
    def __init__(self):
        self.dropped = 0

    def drop_second_context(self):
        self.dropped += 1


class _Det:
    """Records every call; contexts refuse a second file in flight, as the library does (SS_ERR_STATE)."""

    def __init__(self, nomem_on_ctx1=False, poll_fails=(), begin_fails=()):
        self.model = _Model()
        self.calls, self.busy = [], {}
        self.nomem_on_ctx1, self.poll_fails, self.begin_fails = nomem_on_ctx1, set(poll_fails), set(begin_fails)

    def file_prefetch(self, file, which=0):
        self.calls.append(("prefetch", file, which))
        return ("handle", file, which)

    def file_begin(self, file, handle=None, break_duration=0.5, which=0):
        self.calls.append(("begin", file, which))
        if file in self.begin_fails:
            raise ValueError("not a RIFF/WAVE file")
        if which == 1 and self.nomem_on_ctx1:
            raise _Err(6, "activation workspace (44000 MiB): out of memory")
        assert self.busy.get(which) is None, f"context {which} already has {self.busy[which]} in flight"
        self.busy[which] = file
        return {"file": file, "which": which}

    def file_poll(self, token, progress=None, block=True):
        self.calls.append(("poll", token["file"]))
        if token["file"] in self.poll_fails:
            raise RuntimeError("device lost")
        for done in (32, 64, 70):
            progress(done, 70)

    def file_end(self, token, progress=None):
        self.calls.append(("end", token["file"]))
        assert self.busy[token["which"]] == token["file"]
        self.busy[token["which"]] = None
        k = int(token["file"][1:])
        return [(1.0 * k, 1.0 * k + 0.5), (10.0 + k, 11.0 + k)]

    def file_abort(self, token):
        self.calls.append(("abort", token["file"]))
        self.busy[token["which"]] = None


class _Project:
    def __init__(self):
        self.df = pd.DataFrame(columns=['ID', 'file_path', 'file_name', 'start_time', 'end_time', 'erase', 'user_comment', 'review_datetime'])
        self.saves = 0

    def save_detections(self):
        self.saves += 1


def _job(det, files, monkeypatch, n_ctx=2, stop_after=None):
    monkeypatch.setattr(settings, "hip_file_contexts", n_ctx)
    proj = _Project()
    w = ProcessWorker(det, proj, {f: None for f in files})
    ev = []
    w.signals.fileStarted.connect(lambda f: ev.append(("start", f)))
    w.signals.fileProgressChanged.connect(lambda p: ev.append(("prog", round(p, 3))))
    w.signals.fileDone.connect(lambda f: (ev.append(("done", f)), w.stop() if f == stop_after else None))
    w.signals.overallProgressChanged.connect(lambda p: ev.append(("overall", round(p, 3))))
    w.signals.message.connect(lambda m: ev.append(("msg", m)))
    w.signals.finished.connect(lambda: ev.append(("finished",)))
    w.run()
    return ev, proj


FILES = ["f0", "f1", "f2", "f3", "f4"]


def test_two_contexts_keep_the_reference_order_and_the_one_context_rows(monkeypatch):
    ev2, p2 = _job(_Det(), FILES, monkeypatch, 2)
    ev1, p1 = _job(_Det(), FILES, monkeypatch, 1)
    assert ev2 == ev1 and p2.df.equals(p1.df) and p2.saves == 5
    per_file = [e for e in ev2 if e[0] in ("start", "done")]
    assert per_file == [(k, f) for f in FILES for k in ("start", "done")]          # fileStarted -> ... -> fileDone, file by file
    assert list(p2.df["ID"]) == list(range(1, 11)) and ev2[-1] == ("finished",)
    assert [e[1] for e in ev2 if e[0] == "overall"] == [20.0, 40.0, 60.0, 80.0, 100.0]
    det = _Det()
    _job(det, FILES, monkeypatch, 2)
    begins = [(c[1], c[2]) for c in det.calls if c[0] == "begin"]
    assert begins == [("f0", 0), ("f1", 1), ("f2", 0), ("f3", 1), ("f4", 0)]      # alternating contexts
    # the device is a file ahead: f2 begins (on f0's context) before f0's rows are filed, i.e. right after f0's end
    order = [c[:2] for c in det.calls if c[0] in ("begin", "end")]
    assert order.index(("begin", "f2")) == order.index(("end", "f0")) + 1


def test_second_context_without_memory_degrades_to_one_context(monkeypatch, caplog):
    det = _Det(nomem_on_ctx1=True)
    ev, proj = _job(det, FILES, monkeypatch, 2)
    ref, pref = _job(_Det(), FILES, monkeypatch, 1)
    assert ev == ref and proj.df.equals(pref.df)                                   # no file lost, no message
    assert det.model.dropped == 1
    begins = [(c[1], c[2]) for c in det.calls if c[0] == "begin"]
    assert begins == [("f0", 0), ("f1", 1), ("f1", 0), ("f2", 0), ("f3", 0), ("f4", 0)]
    assert any("continuing with one context" in r.getMessage() for r in caplog.records)


def test_failed_files_are_reported_skipped_and_their_context_stays_usable(monkeypatch):
    det = _Det(poll_fails={"f1"}, begin_fails={"f3"})
    ev, proj = _job(det, FILES, monkeypatch, 2)
    msgs = [e[1] for e in ev if e[0] == "msg"]
    assert len(msgs) == 2 and "f1: device lost" in msgs[0] and "f3: not a RIFF/WAVE file" in msgs[1]
    assert [e[1] for e in ev if e[0] == "done"] == ["f0", "f2", "f4"]
    assert ("abort", "f1") in det.calls                                            # run_end on its context: f3 could begin there
    assert [e[1] for e in ev if e[0] == "overall"][-1] == 100.0 and len(proj.df) == 6


def test_stop_between_files_discards_what_is_in_flight(monkeypatch):
    det = _Det()
    ev, proj = _job(det, FILES, monkeypatch, 2, stop_after="f1")
    assert [e[1] for e in ev if e[0] == "done"] == ["f0", "f1"] and ev[-1] == ("finished",)
    assert len(proj.df) == 4
    in_flight = {c[1] for c in det.calls if c[0] == "begin"} - {c[1] for c in det.calls if c[0] == "end"}
    assert in_flight == {c[1] for c in det.calls if c[0] == "abort"}               # every file begun and not ended is aborted


def test_range_policy_first_input_on_the_side_context_second_switches_for_good(monkeypatch, caplog):
    """SpecUNet_2D.with_range_fallback / range_refused (pytorch_neural_nets.py of the drop-in): the first input the f16x2 mode refuses
    (SS_ERR_RANGE) is run again on the fp32 side context and the detector stays in f16x2; a second input switches it for good."""
    import logging
    from root.code.backend.pytorch_neural_nets import SpecUNet_2D
    from softspoken_amd import native

    class Ctx:
        def __init__(self, precision):
            self.precision, self.alive = precision, True

    m = SpecUNet_2D(precision="f16x2")
    main, side = Ctx("f16x2"), Ctx("fp32")
    monkeypatch.setattr(m, "hip_context", lambda which=0: Ctx("fp32") if m.effective_precision() == "fp32" else main)
    monkeypatch.setattr(m, "fp32_context", lambda: side)
    seen = []

    def fn(ctx, bad=True):
        seen.append(ctx.precision)
        if bad and ctx.precision == "f16x2":
            raise native.NativeError(native.SS_ERR_RANGE, "an activation left the f16 range")
        return ctx.precision
    with caplog.at_level(logging.WARNING):
        assert m.with_range_fallback(fn, key="file a") == "fp32" and seen == ["f16x2", "fp32"]
        assert m.effective_precision() == "f16x2"                                   # one input says nothing about the checkpoint
        assert m.with_range_fallback(lambda c: fn(c, bad=False), key="file b") == "f16x2"
        assert m.with_range_fallback(fn, key="file a") == "fp32" and m.effective_precision() == "f16x2"   # the same input again: still one
        assert m.with_range_fallback(fn, key="file c") == "fp32" and m.effective_precision() == "fp32"    # a second input: for good
    assert sum("cannot represent an input" in r.getMessage() for r in caplog.records) == 1
    assert sum("cannot represent this checkpoint" in r.getMessage() for r in caplog.records) == 1
    with pytest.raises(native.NativeError):                                         # other statuses are not swallowed
        m2 = SpecUNet_2D(precision="f16x2")
        monkeypatch.setattr(m2, "hip_context", lambda which=0: main)
        m2.with_range_fallback(lambda c: (_ for _ in ()).throw(native.NativeError(native.SS_ERR_NOMEM, "oom")))
