"""Where one bench step spends its wall time (C2 workload, bf16): host-side pieces around ss_run.
usage: python tools/step_breakdown.py [chunk]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from softspoken_amd import synth, native, checkpoint
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else None
blob = checkpoint.pack_state_dict(synth.make_state_dict(0))
ctx = native.Context(blob, 0, bf16=True, chunk=chunk)
clips = bench.make_clips(0)
frames = np.array([len(c) for c in clips], dtype=np.int64)
pcm = np.concatenate(clips)
d_pcm = ctx.device_alloc(pcm.nbytes); ctx.device_upload(d_pcm, pcm)
T = {"reset": 0.0, "add": 0.0, "run": 0.0, "regions": 0.0}
def step(acc):
    t = time.perf_counter(); ctx.reset(); t1 = time.perf_counter(); acc["reset"] += t1 - t
    first = ctx.add_pcm_batch_device(d_pcm, native.PCM_S16, bench.CLIP_SR, 1, frames); t2 = time.perf_counter(); acc["add"] += t2 - t1
    ctx.run(0.1, 0.5); t3 = time.perf_counter(); acc["run"] += t3 - t2
    counts, reg = ctx.regions_batch(first, len(clips))
    rows = np.column_stack([np.repeat(np.arange(len(clips), dtype=np.float64), counts), reg[:, 0], reg[:, 1]]); acc["regions"] += time.perf_counter() - t3
    return rows
for _ in range(3): step({k: 0.0 for k in T})
n = 10
t0 = time.perf_counter()
for _ in range(n): step(T)
ctx.sync()
tot = (time.perf_counter() - t0) / n * 1e3
print("ms/step", round(tot, 3), {k: round(v / n * 1e3, 3) for k, v in T.items()}, "device_ms_last_run", round(ctx.last_run_device_ms(), 3))
try:
    import torch
    free, total = torch.cuda.mem_get_info(0)
    print("HBM in use after the run: %.1f GB of %.0f GB" % ((total - free) / 1e9, total / 1e9))
except Exception as e:
    print("mem info unavailable:", e)
