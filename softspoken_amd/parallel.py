"""Multi-GPU jobs: one process per GPU, files sharded across ranks, ONE exchange at job end.

The reference has no distributed code; its per-file loop (worker.py:49) carries no state
from one file to the next except the running detection ID, so files are independent units
(SURVEY.md 8(e)).  Plan:
  * longest-processing-time-first assignment of files to ranks by header duration,
  * every rank runs the single-GPU pipeline over its shard (no data-path collective),
  * fixed-width rows (file_index, start, end) are gathered with one collective (all_gather of a
    fixed-capacity buffer whose first row is the rank's row count; RCCL over xGMI when the backend is
    "nccl", gloo in the CPU tests) -- KB-scale, latency-bound,
  * rank 0 sorts by (file_index, start) and numbers the rows in file-list order, which reproduces the
    reference's serial ID order (worker.py:107-124).

A single long recording shards by WINDOW RANGES instead (second half): windows are independent given
the padded signal (NNDetector.py:55-82), so rank r infers a contiguous range of them; the overlap averaging needs
each bin's up to five windows (NNDetector.py:168-186), so the per-window logits (1 KB per window) are gathered -- one
collective again -- and the recording's owner runs the tail of the path (averaging, threshold, regions) on all of
them: `detect_recording_sharded`.  Every rank decodes the whole file (1 % of the work; the samples a range needs
could be cut out instead when recordings are hours long).
"""
from __future__ import annotations

import heapq
from os.path import basename, dirname

import numpy as np


def shard_files(durations, world_size: int):
    """LPT: -> list (len world_size) of lists of file indexes; deterministic (ties by index)."""
    order = sorted(range(len(durations)), key=lambda i: (-float(durations[i]), i))
    heap = [(0.0, r) for r in range(world_size)]
    heapq.heapify(heap)
    shards = [[] for _ in range(world_size)]
    for i in order:
        load, r = heapq.heappop(heap)
        shards[r].append(i)
        heapq.heappush(heap, (load + float(durations[i]), r))
    for s in shards:
        s.sort()
    return shards


def rows_to_array(rows):
    """[(file_index, start, end)] or an (n, 3) array -> float64 (n, 3) (file indexes are exact in a double)."""
    if isinstance(rows, np.ndarray):
        return np.ascontiguousarray(rows, dtype=np.float64).reshape(-1, 3)
    a = np.zeros((len(rows), 3), dtype=np.float64)
    for k, (fi, s, e) in enumerate(rows):
        a[k] = (fi, s, e)
    return a


# rows a rank's gather buffer holds; every rank keeps the same value (it only changes on what all ranks see: the gathered counts)
_capacity = [1024]


def gather_rows(local_rows, group=None, device=None):
    """All ranks call this; returns the merged, sorted (n, 3) array on every rank.

    One collective in the common case: every rank sends a fixed-capacity buffer whose first row carries its row count.  When
    some rank has more rows than the capacity, every rank sees that in the gathered counts, grows the capacity to the same
    power of two and the exchange is repeated once."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = device if device is not None else torch.device("cpu")
    local = rows_to_array(local_rows)
    n = local.shape[0]
    while True:
        cap = _capacity[0]
        buf = np.zeros((cap + 1, 3), dtype=np.float64)
        buf[0, 0] = n
        buf[1:1 + min(n, cap)] = local[:cap]
        mine = torch.from_numpy(buf).to(dev)
        everyone = torch.empty((world * (cap + 1), 3), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(everyone, mine, group=group)
        got = everyone.cpu().numpy().reshape(world, cap + 1, 3)
        counts = got[:, 0, 0].astype(np.int64)
        if int(counts.max()) <= cap:
            break
        while _capacity[0] < int(counts.max()):
            _capacity[0] *= 2
    parts = [got[r, 1:1 + int(counts[r])] for r in range(world)]
    merged = np.concatenate(parts, axis=0) if parts else np.zeros((0, 3))
    if len(merged):
        order = np.lexsort((merged[:, 1], merged[:, 0]))
        merged = merged[order]
    return merged


def number_rows(merged, files, first_id: int = 1):
    """(n,3) sorted rows -> list of dict rows in the reference's schema (worker.py:113-123)."""
    out = []
    for k, (fi, s, e) in enumerate(merged):
        f = files[int(fi)]
        out.append({'ID': first_id + k, 'file_path': dirname(f), 'file_name': basename(f),
                    'start_time': float(s), 'end_time': float(e), 'erase': 0, 'user_comment': '',
                    'review_datetime': ''})
    return out


def run_sharded(files, durations, detect_fn, group=None, device=None):
    """detect_fn(list_of_paths) -> {path: [(start, end)]} on this rank's GPU.
    Returns numbered rows on rank 0 (None elsewhere)."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    mine = shard_files(durations, world)[rank]
    found = detect_fn([files[i] for i in mine]) if mine else {}
    rows = []
    for i in mine:
        for (s, e) in found.get(files[i], []):
            rows.append((i, s, e))
    merged = gather_rows(rows, group, device)
    return number_rows(merged, files) if rank == 0 else None


# ---- one long recording across ranks: contiguous window ranges, logits gathered to the owner --------------------------
def split_windows(n_windows: int, world_size: int):
    """-> [(lo, hi)] per rank: contiguous, in order, sizes differing by at most one."""
    base, extra = divmod(int(n_windows), world_size)
    out, lo = [], 0
    for r in range(world_size):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def gather_window_logits(local_logits, n_windows: int, group=None, device=None):
    """All ranks call this with their range's logits [hi - lo, 256] (ranges as split_windows gives them); returns the
    recording's [n_windows, 256] float32 on every rank.  One all_gather of equal-size buffers (the largest range)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = device if device is not None else torch.device("cpu")
    ranges = split_windows(n_windows, world)
    cap = max(1, max(hi - lo for lo, hi in ranges))
    local = np.ascontiguousarray(local_logits, dtype=np.float32).reshape(-1, 256)
    lo, hi = ranges[rank]
    if local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} owns windows [{lo}, {hi}) but holds {local.shape[0]} rows of logits")
    buf = np.zeros((cap, 256), dtype=np.float32)
    buf[: hi - lo] = local
    everyone = torch.empty((world * cap, 256), dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(everyone, torch.from_numpy(buf).to(dev), group=group)
    got = everyone.cpu().numpy().reshape(world, cap, 256)
    return np.concatenate([got[r, : ranges[r][1] - ranges[r][0]] for r in range(world)], axis=0)


def detect_windows_sharded(n_windows: int, infer_fn, finish_fn, group=None, device=None):
    """infer_fn(lo, hi) -> float32 [hi - lo, 256]: this rank's range on its GPU; finish_fn(logits [n_windows, 256]) -> result,
    run on rank 0 (the owner) only.  Returns finish_fn's result on rank 0, None elsewhere."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = split_windows(n_windows, world)[rank]
    local = infer_fn(lo, hi) if hi > lo else np.zeros((0, 256), dtype=np.float32)
    full = gather_window_logits(local, n_windows, group, device)
    return finish_fn(full) if rank == 0 else None


def detect_recording_sharded(ctx, pcm, fmt: int, sample_rate: int, channels: int, frames: int, threshold: float = 0.1,
                             break_s: float = 0.5, group=None, device=None):
    """One recording on all ranks' GPUs (`ctx`: this rank's native.Context).  -> [(start_s, end_s)] on rank 0 (the worker's
    "-3 s" applied, as Context.regions gives them), None elsewhere: the table a one-GPU ss_run of the recording gives, bit
    for bit (same kernels per window, same averaging and region code over the gathered logits)."""
    from . import native
    ctx.reset()
    fid = ctx.add_pcm(pcm, fmt, sample_rate, channels, frames)
    starts = native.plan_windows(frames / sample_rate)
    # the plan comes from the header duration, the data from the resampler: clamp to what fits, as ss_run does
    n_padded = ctx.signal_length(fid, padded=True)
    while len(starts) and starts[-1] + native.WINDOW_SAMPLES > n_padded:
        starts = starts[:-1]

    def infer(lo, hi):
        return ctx.infer_windows(fid, starts[lo:hi])[1].reshape(-1, 256)

    def finish(logits):
        ctx.run_from_logits(logits, threshold, break_s)
        return ctx.regions(fid)

    return detect_windows_sharded(len(starts), infer, finish, group, device)
