"""Jobs in flight on alternating library contexts (ss_run_begin / ss_run_end, include/softspoken.h).

A job has a device half (decode, resample, windows through the network, averaging: enqueued by `Context.run_begin`) and
a host half (wait, regions, whatever the caller does with them).  With one context the two alternate and the device
idles during every host half; with two contexts on the same device job k's host half runs while job k+1's kernels do.
The reference's worker (root/code/backend/worker.py:49) is a serial per-file loop and stays that way behind its own
signals; this is for callers that hand over many files at once (bench.py, batch tools).
"""
from __future__ import annotations


def run_jobs(contexts, jobs, submit, collect):
    """submit(context, job) -> token enqueues a job's device half (ending in context.run_begin());
    collect(context, job, token) -> result ends it (starting with context.run_end()).
    Results are yielded in job order; at most one job per context is in flight."""
    pending = None
    for k, job in enumerate(jobs):
        c = contexts[k % len(contexts)]
        if pending is not None and pending[0] is c:     # a single context: finish its job before it takes the next
            yield collect(*pending)
            pending = None
        token = submit(c, job)
        if pending is not None:
            yield collect(*pending)
        pending = (c, job, token)
    if pending is not None:
        yield collect(*pending)
