"""Multi-GPU jobs: one process per GPU, files sharded across ranks, ONE exchange at job end.

The reference has no distributed code; its per-file loop (worker.py:49) carries no state
from one file to the next except the running detection ID, so files are independent units
(SURVEY.md 8(e)).  Plan:
  * longest-processing-time-first assignment of files to ranks by header duration,
  * every rank runs the single-GPU pipeline over its shard (no data-path collective),
  * fixed-width rows (file_index, start, end) are gathered to rank 0 with two collectives
    (all_gather of row counts, all_gather of padded row buffers; RCCL over xGMI when the backend is
    "nccl", gloo in the CPU tests) -- KB-scale, latency-bound,
  * rank 0 sorts by (file_index, start) and numbers the rows in file-list order, which reproduces the
    reference's serial ID order (worker.py:107-124).
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
    """[(file_index, start, end)] -> float64 (n, 3) (file indexes are exact in a double)."""
    a = np.zeros((len(rows), 3), dtype=np.float64)
    for k, (fi, s, e) in enumerate(rows):
        a[k] = (fi, s, e)
    return a


def gather_rows(local_rows, group=None, device=None):
    """All ranks call this; returns the merged, sorted (n, 3) array on every rank.
    Two collectives: counts, then padded rows."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = device if device is not None else torch.device("cpu")
    local = torch.from_numpy(rows_to_array(local_rows)).to(dev)
    count = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, count, group=group)
    counts = [int(c.item()) for c in counts]
    width = max(max(counts), 1)
    padded = torch.zeros((width, 3), dtype=torch.float64, device=dev)
    padded[: local.shape[0]] = local
    bufs = [torch.zeros((width, 3), dtype=torch.float64, device=dev) for _ in range(world)]
    dist.all_gather(bufs, padded, group=group)
    parts = [b[:n].cpu().numpy() for b, n in zip(bufs, counts)]
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
