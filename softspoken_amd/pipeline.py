"""Jobs in flight (ss_run_begin / ss_run_end, include/softspoken.h).

A job has a device half (decode, resample, windows through the network, averaging: enqueued by `Context.run_begin`) and
a host half (wait, regions, whatever the caller does with them).  Run back to back the device idles during every host
half.  Two ways around that:

  * ONE context: the results of an ended run stay readable while the next job is added and started, so the order is
    end(k) -> submit(k + 1) -> results(k).  The device idles only between the end of job k and the first launch of job
    k + 1; kernels of different jobs never overlap (per-kernel timings stay the kernels' own).
  * TWO contexts on the same device alternate: job k + 1 is submitted while job k still runs, so the tail of one job
    overlaps the head of the next as well (a few per cent more, and kernel traces of such a run show shared time).

The reference's worker (root/code/backend/worker.py:49) is a serial per-file loop and stays that way behind its own
signals; this is for callers that hand over many files at once (bench.py, batch tools).
"""
from __future__ import annotations


def run_jobs(contexts, jobs, submit, end, results):
    """submit(context, job) -> token enqueues a job's device half (ending in context.run_begin());
    end(context, job, token) waits for it (context.run_end()); results(context, job, token) -> value reads the ended run.
    Values are yielded in job order; at most one job per context is in flight."""
    pending = None
    for k, job in enumerate(jobs):
        c = contexts[k % len(contexts)]
        ended = None
        if pending is not None and pending[0] is c:     # the context is busy with the previous job: end it, read it after the submit
            end(*pending)
            ended, pending = pending, None
        token = submit(c, job)
        if ended is not None:
            yield results(*ended)
        if pending is not None:                         # another context's job: it ran while this one was being submitted
            end(*pending)
            yield results(*pending)
        pending = (c, job, token)
    if pending is not None:
        end(*pending)
        yield results(*pending)
