"""The N>1 path on CPU: world_size-2 gloo, files sharded by duration, one gather of detection rows,
rank 0 numbers them in file-list order.  The per-rank detector is a stub here (the
GPU pipeline itself is covered by the -m gpu tests); what is tested is sharding + exchange + ordering."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _fake_detect(paths):
    out = {}
    for p in paths:
        k = int(os.path.basename(p).split("_")[1].split(".")[0])
        out[p] = [(float(k) + 0.25 * j, float(k) + 0.25 * j + 0.1) for j in range(k % 4)]   # 0..3 regions
    return out


def _worker(rank, world, port, files, durs, q):
    import torch.distributed as dist
    from softspoken_amd.parallel import run_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows = run_sharded(files, durs, _fake_detect)
    # the exchange itself: more rows than the gather buffer's capacity on one rank only (every rank grows it and repeats), array
    # input, empty input
    from softspoken_amd import parallel
    parallel._capacity[0] = 4
    mine = np.array([[rank, 0.5 * j, 0.5 * j + 0.25] for j in range(3 if rank else 37)])
    m = parallel.gather_rows(mine)
    assert m.shape == (40, 3) and parallel._capacity[0] == 64
    assert np.array_equal(m[:37, 0], np.zeros(37)) and np.array_equal(m[37:, 0], np.ones(3)) and np.all(np.diff(m[:37, 1]) > 0)
    assert parallel.gather_rows([]).shape == (0, 3)
    assert parallel.gather_rows([(rank, 1.0, 2.0)] if rank else []).tolist() == [[1.0, 1.0, 2.0]]
    if rank == 0:
        q.put(rows)
    else:
        assert rows is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_reproduces_serial_order():
    files = [f"/data/rec_{k}.wav" for k in range(11)]
    durs = [600.0, 3.0, 60.0, 600.0, 10.0, 0.5, 300.0, 300.0, 1.0, 59.0, 600.0]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, files, durs, q)) for r in range(2)]
    for p in procs:
        p.start()
    rows = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # serial reference: files in list order, IDs continue across files (worker.py:107-124)
    want = []
    nid = 1
    serial = _fake_detect(files)
    for f in files:
        for (s, e) in serial[f]:
            want.append((nid, os.path.dirname(f), os.path.basename(f), s, e)); nid += 1
    got = [(r["ID"], r["file_path"], r["file_name"], r["start_time"], r["end_time"]) for r in rows]
    assert got == want
    assert all(r["erase"] == 0 and r["user_comment"] == "" for r in rows)
