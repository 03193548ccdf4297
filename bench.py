#!/usr/bin/env python
"""Throughput bench of the voice-detector hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json configs[1], "C2"): per GPU a batch of 256 x 3 s 16 kHz mono PCM16 synthetic
clips, bf16 inference.  One step = one pass of the whole path over that batch with the PCM already
resident in HBM when the clock starts:
    PCM16 -> float, resample 16 k -> 22.05 k, 3 s pad           (decode_mono_batch, resample_batch)
    2 560 windows -> fused STFT/mel/log front-end               (frontend)
    SpecUNet_2D conv stack + mask head, bf16 MFMA               (conv_first, conv3x3_*, flatten, mask_head)
    overlap averaging on the device, averaged logits to host    (average + D2H)
    threshold / gap-merge -> detection rows                     (host)
    N > 1: gather of detection rows to rank 0 over RCCL         (two small collectives per step)
metric = audio-seconds processed per wall-second, whole job (sum over ranks), weak scaling.

One JSON line on stdout (rank 0).  Extra objects:
  roofline     dominant kernel instantiation (largest share of device time in a profiled pass of the same step, HIP
               events on the library's stream; names are rocprofv3's): algorithmic bytes and FLOPs per launch /
               measured duration.  The bound is HBM when the launch's FLOP/byte is below the ridge (2.5 PFLOP/s / 8 TB/s),
               MFMA otherwise; both fractions are reported.  `traffic` = HBM bytes per launch from the committed PMC pass.
  stft_stage   the front-end kernel's algorithmic bytes / duration vs HBM peak (north-star sub-target).
  cpu_baseline the torch-CPU oracle (the reference's own torch ops restated; oracle/oracle_np.py) timed on
               this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}
FRONTEND_BYTES_PER_WINDOW = 66150 * 4 + 128 * 256 * 4       # 395 672 B
N_CLIPS, CLIP_S, CLIP_SR = 256, 3.0, 16000


def make_clips(rank):
    from softspoken_amd import synth
    clips = [synth.to_pcm16(synth.synth_audio(2000 + 1000 * rank + k, CLIP_S, CLIP_SR, 1, with_silence=False))
             for k in range(N_CLIPS)]
    return clips


def cpu_baseline(sd_np, clips, n_sample=16):
    """The CPU restatement on a bounded sample: resample + pad + batches of 32 windows + averaging + regions."""
    import torch
    from softspoken_amd import synth
    from oracle import oracle_np as O
    cores = os.cpu_count() or 1
    threads = max(1, cores // 2)                  # the reference's rule (settings.py:32, NNDetector.py:25)
    torch.set_num_threads(threads)
    torch.set_grad_enabled(False)
    sd = synth.to_torch_state_dict(sd_np)
    x0 = (clips[0].astype(np.float32) / np.float32(32768.0))
    O.detect_signal(sd, O.resample(x0, CLIP_SR), CLIP_S)          # warm-up (thread pools, oneDNN primitives)
    t0 = time.perf_counter()
    nwin = 0
    for k in range(n_sample):
        x = clips[k].astype(np.float32) / np.float32(32768.0)
        r = O.detect_signal(sd, O.resample(x, CLIP_SR), CLIP_S)
        nwin += len(r["starts"])
    dt = time.perf_counter() - t0
    return {"value": round(n_sample * CLIP_S / dt, 3), "unit": "audio-seconds/s", "cores": threads, "kind": "port",
            "sample": f"{n_sample} of the {N_CLIPS} clips ({nwin} windows), fp32, torch CPU ops, {threads} of {cores} host threads, {dt:.1f} s",
            "windows_per_s": round(nwin / dt, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--contexts", type=int, default=1, choices=[1, 2],
                    help="1: the results of an ended job are read after the next job has been submitted (its host half runs behind the next "
                         "device half; kernels of different jobs never overlap).  2: two library contexts alternate, so the tail of one job "
                         "also overlaps the head of the next (+2-3 %%; per-kernel durations of such a run include shared time)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=16)
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # rehearsal on a box with fewer GPUs than ranks (development only): SOFTSPOKEN_DIST_BACKEND=gloo shares the cards round-robin
        backend = os.environ.get("SOFTSPOKEN_DIST_BACKEND", "nccl")
        if backend != "nccl":
            local_rank %= max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from softspoken_amd import synth, native, checkpoint, parallel, pipeline
    sd_np = synth.make_state_dict(0)
    blob = checkpoint.pack_state_dict(sd_np)
    bf16 = a.precision == "bf16"
    t_c0 = time.perf_counter()
    ctx = native.Context(blob, local_rank, bf16=bf16, chunk=a.chunk or None)     # fold + pack + upload of the checkpoint
    t_create = time.perf_counter() - t_c0

    clips = make_clips(rank)
    frames = np.array([len(c) for c in clips], dtype=np.int64)
    pcm = np.concatenate(clips)
    d_pcm = ctx.device_alloc(pcm.nbytes)
    ctx.device_upload(d_pcm, pcm)                       # inputs resident in HBM before the clock starts
    files = [f"/synthetic/rank{rank}/clip_{k:04d}.wav" for k in range(N_CLIPS)]
    dev = torch.device("cuda", local_rank) if (dist is None or dist.get_backend() == "nccl") else torch.device("cpu")

    def submit(c, _job=None):                           # device half of a job: decode + resample + windows + averaging, enqueued
        c.reset()
        first = c.add_pcm_batch_device(d_pcm, native.PCM_S16, CLIP_SR, 1, frames)
        c.run_begin(0.1, 0.5)
        return first

    def end(c, _job, first):                            # wait for the device half
        c.run_end()

    def results(c, _job, first):                        # host half: regions, rows (+ the gather across ranks)
        counts, reg = c.regions_batch(first, N_CLIPS)          # detection rows (file index, start, end) of the whole job
        fidx = np.repeat(np.arange(N_CLIPS, dtype=np.int64) + rank * N_CLIPS, counts)
        rows = np.column_stack([fidx.astype(np.float64), reg[:, 0], reg[:, 1]]) if len(fidx) else np.zeros((0, 3))
        if world > 1:
            return parallel.gather_rows(rows, device=dev)
        return rows

    def step(c):
        first = submit(c)
        end(c, None, first)
        return results(c, None, first)

    ctxs = [ctx]
    if a.contexts == 2:
        ctxs.append(native.Context(blob, local_rank, bf16=bf16, chunk=a.chunk or None))

    def run_steps(k_steps, cs=None):                    # jobs in flight: one per context (softspoken_amd/pipeline.py)
        rows = None
        for rows in pipeline.run_jobs(cs or ctxs, range(k_steps), submit, end, results):
            pass
        return rows

    def fence():
        if world > 1:
            dist.barrier()
        for c in ctxs:
            c.sync()
        torch.cuda.synchronize()

    # "reference-style" clock (silencer_ui.py:222-225 starts it before the detector is built): context creation +
    # the first, cold step (workspace allocation, first-use kernel loads); reported beside the steady-state value, never as it
    t_first = None
    if a.warmup > 0:                                    # the cold step is the first of the W warm-up steps
        t_f0 = time.perf_counter()
        rows = step(ctx)
        ctx.sync()
        t_first = time.perf_counter() - t_f0
    if a.warmup > 1:
        rows = run_steps(a.warmup - 1)
    if len(ctxs) > 1 and a.warmup < 3:                  # every context has run once before the clock starts
        rows = step(ctxs[1])
    fence()
    t0 = time.perf_counter()
    rows = run_steps(a.steps)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    n_windows = sum(ctx.num_windows(k) for k in range(N_CLIPS))
    # the strictly sequential variant (one context: every job's host half with the device idle), reported beside `value`
    dt_seq = None
    if len(ctxs) > 1:
        fence()
        t0s = time.perf_counter()
        run_steps(a.steps, [ctx])
        fence()
        dt_seq = time.perf_counter() - t0s
        if world > 1:
            t = torch.tensor([dt_seq], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_seq = float(t.item())
    device_ms = ctx.last_run_device_ms()               # (of that sequential pass when there are two contexts: not overlapped)

    # PCIe-inclusive variant (noted in DESIGN.md, never `value`): host PCM handed over each step
    pcie = None
    if rank == 0:
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(max(1, a.steps // 4)):
            ctx.device_upload(d_pcm, pcm)
            step(ctx) if world == 1 else None
        ctx.sync()
        if world == 1:
            pcie = N_CLIPS * CLIP_S * max(1, a.steps // 4) / (time.perf_counter() - t1)

    # ---- roofline: profiled pass of the same step (HIP events around every launch on the library's stream) ----
    roof = stft = None
    kernels = []
    if rank == 0:
        prof = native.Context(blob, local_rank, bf16=bf16, profile=True, chunk=a.chunk or None)
        saved_world = world
        for _ in range(2):
            prof.reset(); prof.add_pcm_batch_device(d_pcm, native.PCM_S16, CLIP_SR, 1, frames); prof.run(0.1, 0.5)
        prof.reset_stats()
        nprof = 3
        for _ in range(nprof):
            prof.reset(); prof.add_pcm_batch_device(d_pcm, native.PCM_S16, CLIP_SR, 1, frames); prof.run(0.1, 0.5)
        raw = [s for s in prof.kernel_stats() if s["launches"]]
        layers = [dict(s) for s in raw if "/" in s["name"]]
        merged = {}
        for s in raw:                                   # "<kernel>/<layer>" -> per-kernel totals
            k = s["name"].split("/")[0]
            m = merged.setdefault(k, dict(name=k, launches=0, total_ms=0.0, flops=0.0, bytes=0.0))
            for f in ("launches", "total_ms", "flops", "bytes"):
                m[f] += s[f]
        stats = list(merged.values())
        tot = sum(s["total_ms"] for s in stats)
        for s in sorted(stats, key=lambda s: -s["total_ms"]):
            kernels.append({"name": s["name"], "launches_per_step": s["launches"] // nprof,
                            "ms_per_step": round(s["total_ms"] / nprof, 4), "share": round(s["total_ms"] / tot, 4),
                            "avg_us": round(1e3 * s["total_ms"] / s["launches"], 2),
                            "tflops": round(s["flops"] / max(s["total_ms"], 1e-9) / 1e9, 2),
                            "gbs": round(s["bytes"] / max(s["total_ms"], 1e-9) / 1e6, 1)})
        dom = max(stats, key=lambda s: s["total_ms"])
        peak = MFMA_PEAK_TFLOPS[a.precision]
        ach_tf = dom["flops"] / dom["total_ms"] / 1e9
        ach_gbs = dom["bytes"] / dom["total_ms"] / 1e6
        intensity = dom["flops"] / max(dom["bytes"], 1.0)              # algorithmic FLOP per algorithmic byte
        ridge = peak * 1e12 / (HBM_PEAK_GBS * 1e9)
        # HBM bytes per launch of that instantiation from the committed rocprofv3 PMC pass (same launch shapes), or null
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            if a.precision == "bf16" and dom["name"] in tj["kernels"]:
                # measured per window on the same instantiation (tile-based kernels: bytes scale with the windows of a launch)
                traffic = tj["kernels"][dom["name"]]["hbm_bytes_per_window"] * n_windows / (dom["launches"] / nprof)
        except Exception:
            traffic = None
        common = {"kernel": dom["name"], "traffic": traffic, "avg_launch_us": round(1e3 * dom["total_ms"] / dom["launches"], 2),
                  "flops_per_launch": dom["flops"] / dom["launches"], "bytes_per_launch": dom["bytes"] / dom["launches"],
                  "flop_per_byte": round(intensity, 1), "ridge_flop_per_byte": round(ridge, 1),
                  "mfma": {"achieved_tflops": round(ach_tf, 2), "peak": peak, "frac": round(ach_tf / peak, 4)},
                  "hbm": {"achieved_gbs": round(ach_gbs, 1), "peak": HBM_PEAK_GBS, "frac": round(ach_gbs / HBM_PEAK_GBS, 4)},
                  "measured": "HIP events around each launch on the library's stream, separate profiled pass of the same step"}
        if intensity < ridge:   # below the ridge the kernel's roof is bandwidth
            roof = dict(bound="hbm", achieved=round(ach_gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach_gbs / HBM_PEAK_GBS, 4), **common)
        else:
            roof = dict(bound="mfma", achieved=round(ach_tf, 2), peak=peak, unit="TFLOP/s", frac=round(ach_tf / peak, 4), **common)
        conv_ms = sum(s["total_ms"] for s in stats if s["name"].startswith(("conv3x3", "resblock32")))
        conv_fl = sum(s["flops"] for s in stats if s["name"].startswith(("conv3x3", "resblock32")))
        roof["all_conv3x3_tflops"] = round(conv_fl / conv_ms / 1e9, 2)
        roof["all_conv3x3_frac"] = round(conv_fl / conv_ms / 1e9 / peak, 4)
        fe = next(s for s in stats if s["name"] == "frontend")
        fe_gbs = fe["bytes"] / fe["total_ms"] / 1e6
        stft = {"kernel": "frontend", "bound": "hbm", "achieved": round(fe_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(fe_gbs / HBM_PEAK_GBS, 4), "bytes_per_window": FRONTEND_BYTES_PER_WINDOW,
                "windows_per_s": round(fe["bytes"] / FRONTEND_BYTES_PER_WINDOW / (fe["total_ms"] / 1e3), 0), "traffic": None}
        try:                                            # measured HBM bytes per launch (FETCH_SIZE / WRITE_SIZE passes, profiles/)
            fk = tj["kernels"]["frontend_kernel"]
            stft["traffic"] = fk["hbm_bytes_per_window"] * n_windows / (fe["launches"] / nprof)
            stft["traffic_bytes_per_window"] = round(fk["hbm_bytes_per_window"], 1)
        except Exception:
            pass
        prof.close()
        layer_table = [{"layer": s["name"], "us": round(1e3 * s["total_ms"] / s["launches"], 1),
                        "tflops": round(s["flops"] / max(s["total_ms"], 1e-9) / 1e9, 1)} for s in layers]

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:       # the CPU baseline is a 1-GPU-run item (other ranks would wait on it)
        cpu = cpu_baseline(sd_np, clips, a.cpu_sample)

    if rank == 0:
        total_audio = world * N_CLIPS * CLIP_S * a.steps
        out = {
            "metric": "audio-seconds processed/sec (whole node), 16 kHz mono",
            "value": round(total_audio / dt, 2), "unit": "audio-seconds/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.precision, "data": "synthetic",
            "config": {"workload": f"C2: {N_CLIPS} x {CLIP_S:g} s {CLIP_SR} Hz mono PCM16 clips per GPU, {a.precision} inference, "
                                   "PCM resident in HBM; decode+resample+front-end+U-Net+averaging+regions"
                                   + ("; two contexts alternate (host half and tail of job k overlap the head of job k+1)" if len(ctxs) > 1
                                      else "; results of job k are read after job k+1 has been submitted")
                                   + ("+RCCL row gather" if world > 1 else ""),
                       "windows_per_step_per_gpu": int(n_windows), "graph": "mask-only (spec head skipped, 6.360 GFLOP/window)",
                       "weights": "synthetic checkpoint, reference state_dict layout", "parallelism": f"file-sharded dp{world}"},
            "windows_per_s": round(world * n_windows * a.steps / dt, 1),
            "device_ms_last_run": round(device_ms, 3),
            "rows_last_step": int(len(rows)),
            "roofline": roof, "stft_stage": stft, "cpu_baseline": cpu, "kernels": kernels, "layers": layer_table,
        }
        if dt_seq:
            out["value_one_context"] = round(total_audio / dt_seq, 2)
        if pcie:
            out["value_pcie_inclusive"] = round(pcie, 2)
        if t_first is not None:
            out["cold_start"] = {"create_ms": round(1e3 * t_create, 1), "first_step_ms": round(1e3 * t_first, 1),
                                 "value_reference_style": round(N_CLIPS * CLIP_S / (t_create + t_first), 1),
                                 "note": "one job of 256 clips on a fresh process: context creation + first step (rank 0's clock)"}
        print(json.dumps(out), flush=True)
    ctx.device_free(d_pcm)
    for c in ctxs:
        c.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
