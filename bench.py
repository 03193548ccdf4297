#!/usr/bin/env python
"""Throughput bench of the voice-detector hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: this script starts its own N ranks, one per GPU, over RCCL)

Headline workload (BASELINE.json configs[2], "C3", the largest single-GPU configuration): per GPU 100 x 10 min 16 kHz mono PCM16
recordings = 100 500 windows, at the reference's precision (scores within 1e-4 of its fp32 CPU path).  One step = one pass of the
whole path over those recordings, FROM THE WAV FILE IMAGES IN (page-locked) HOST MEMORY (the reference's loop starts at the file
too: worker.py:57 -> voice_activity.py:37, NNDetector.py:90):
    RIFF/WAVE header walk of every file, samples host -> HBM    (ss_upload_wav_batch_async: the copies of job k + 1 run on the
                                                                 library's copy stream beside the kernels of job k; one of them
                                                                 is inside every timed step)
    PCM16 -> float, resample 16 k -> 22.05 k, 3 s pad           (decode_mono_batch, resample_batch)
    windows -> fused STFT/mel/log front-end                     (frontend)
    SpecUNet_2D conv stack + mask head                          (conv3x3_*, mask_head_parts)
    overlap averaging + threshold bits on the device, to host   (average, bin_masks + D2H)
    run-length / gap-merge -> detection rows                    (host)
    N > 1: gather of detection rows to rank 0 over RCCL         (one small collective per step)
metric = audio-seconds processed per wall-second, whole job (sum over ranks), weak scaling (every rank has its own 100 recordings).

`--precision` picks the arithmetic of the conv stack for the headline:
    f16x2 (default)  fp32-accurate on the f16 matrix cores: every operand is two f16 halves, three products per term, fp32 accumulate
    fp32             fp32 operands on the fp32 matrix instructions (exact fp32 FMA chains)
    bf16             throughput mode (scores differ from the reference by up to ~0.1: NOT the parity mode)
Both parity modes are timed in every 1-GPU run (`value` is the chosen one, the other is `secondary.c3_<mode>`).

One JSON line on stdout (rank 0).  Extra objects:
  roofline     dominant kernel instantiation of the headline step (largest share of device time in a profiled pass, HIP events on the
               library's stream; names are rocprofv3's): `achieved` / `frac` = ALGORITHMIC FLOPs per launch (2 x the layers'
               multiply-adds) / measured duration against the dense f16 peak; `frac_issued` beside it counts the
               three matrix products the f16x2 mode issues per multiply-add; `traffic` = measured HBM bytes per launch of that
               kernel from the committed PMC passes (profiles/r04_traffic_f16x2.json).
  stft_stage   the front-end kernel: what bounds it (vector issue), its fp32-vector fraction, and its algorithmic bytes / duration
               vs the HBM peak (north-star sub-target).
  secondary    c3_resident (the same job with the PCM already in HBM: round 2's headline); C3 in fp32 on all 100 recordings; C2
               (256 x 3 s clips) in bf16; C5 (48 kHz stereo) decode + mixdown + resample + front-end; worker_dropin: 10 x 10-min WAV
               files on tmpfs through root.code.backend.worker.ProcessWorker.run with its progress signals, CSV written.
  cpu_baseline the torch-CPU oracle (the reference's own torch ops restated; oracle/oracle_np.py) on this box's host cores: batches
               of 32 windows, cores // 2 threads (the reference's rule) and all cores, mask + spec graph and mask-only.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3, "f16x2": 2500.0}
# What the matrix pipe sustains on this part under its power cap (tools/probes/mfma_power.hip beside rocm-smi, round 2): a loop of nothing
# but v_mfma_f32_32x32x16_f16 on register operands settles at 1.78 GHz / ~1300 W and 1705 TFLOP/s, with two ds_read_b128 per
# product (this kernel's operand traffic) at 1.72 GHz and 1518 TFLOP/s.  The conv stack itself runs at ~1.88 GHz / ~1320 W in
# both modes: the 2.5 PFLOP/s in `peak` is priced at a 2.4 GHz the part does not hold under matrix load.
MFMA_SUSTAINED_TFLOPS = {"bf16": 1705.0, "f16x2": 1705.0}
MFMA_PRODUCTS = {"bf16": 1, "fp32": 1, "f16x2": 3}           # matrix-instruction products per algorithmic multiply-add
FRONTEND_BYTES_PER_WINDOW = 66150 * 4 + 128 * 256 * 4       # 395 672 B
FRONTEND_FLOPS_PER_WINDOW = 256 * 2.5 * 2048 * 11 + 2 * 1469 * 256 + 2 * 32768   # FFT + sparse mel + log/sqrt = 15.2 MFLOP
VALU_FP32_PEAK_TFLOPS = 157.3                               # MI355X_MICROARCH.md: peak fp32 vector rate
FRONTEND_VALU_PER_WINDOW = 124432                           # SQ_INSTS_VALU of frontend_kernel / windows (profiles/r04_pmc.md: 125 053 995 per 1005 windows)
TRAFFIC_JSON = os.path.join("profiles", "r04_traffic_f16x2.json")   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/profile_r04.sh)
TRAFFIC_MAX_DRIFT = 0.10         # a traffic figure is printed only while the profiled launch's duration is within 10 % of this run's
C5_BYTES_PER_WINDOW = 576000 + 128 * 256 * 4                # 707 072 B (48 kHz stereo PCM16 source)
REC_S, REC_SR = 600.0, 16000
N_DISTINCT = 4                   # distinct synthetic recordings; the job's files repeat them (tools/scale_check.py c4 does the same)


def traffic_of(kernel, avg_us, windows_per_launch, algorithmic_bytes_per_launch, prefix=False):
    """Measured HBM bytes per launch of `kernel` from the committed PMC passes (FETCH_SIZE x 2 -- gfx950 counts wide streaming reads at
    half --, WRITE_SIZE; separate rocprofv3 --pmc runs, MI355X_MICROARCH.md 'HBM'), scaled to this run's windows per launch --
    but ONLY when the file names this kernel instantiation and the launch it profiled took within TRAFFIC_MAX_DRIFT of what the
    launch takes in THIS run: a kernel that changed since the profile was taken prints null and the reason, never a stale ratio.
    -> (traffic or None, traffic_source)."""
    src = {"file": TRAFFIC_JSON}
    try:
        tj = json.load(open(os.path.join(ROOT, TRAFFIC_JSON)))
        src["profiled_at_head"] = tj.get("head")
        src["kernel_sources_unchanged_since"] = tj.get("tree_sha16") == csrc_sha16()     # (same hash as tools/traffic_summary.py records)
        ks = tj["kernels"]
        tk = next((v for k, v in ks.items() if k.startswith(kernel)), None) if prefix else ks.get(kernel)
        if tk is None:
            src["traffic_null_because"] = f"the file has no kernel named {kernel!r} (profiled: {len(ks)} kernels)"
            return None, src
        there = tk["avg_us"] * windows_per_launch / tj["windows_per_launch"]
        drift = abs(there - avg_us) / max(avg_us, 1e-9)
        src.update({"windows_per_launch_there": tj["windows_per_launch"], "avg_us_there": tk["avg_us"], "avg_us_here": round(avg_us, 2), "drift": round(drift, 4)})
        if drift > TRAFFIC_MAX_DRIFT:
            src["traffic_null_because"] = f"the profiled launch took {there:.1f} us at this size, this run's takes {avg_us:.1f} us: more than {TRAFFIC_MAX_DRIFT:.0%} apart -- re-run tools/profile_r04.sh"
            return None, src
        per_window = tk["hbm_bytes_per_launch"] / tj["windows_per_launch"]
        src.update({"fetch_bytes": tk["fetch_bytes_per_launch"], "write_bytes": tk["write_bytes_per_launch"],
                    "measured_over_algorithmic": round(per_window * windows_per_launch / max(algorithmic_bytes_per_launch, 1.0), 3)})
        return round(per_window * windows_per_launch), src
    except Exception as e:                               # (no committed traffic file for this precision: traffic stays null)
        src["traffic_null_because"] = f"{type(e).__name__}: {e}"
        return None, src


def csrc_sha16():
    """Hash over softspoken_amd/csrc/*.hip, *.h -- what tools/traffic_summary.py writes into the traffic JSON as tree_sha16."""
    import glob
    import hashlib
    h = hashlib.sha256()
    root = os.path.join(ROOT, "softspoken_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def launch_ranks(a, argv):
    """`python bench.py --gpus N` without a launcher around it: start N ranks as children -- before this process has touched
    torch or HIP -- and pass rank 0's JSON line through."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def make_recordings(n_files):
    import numpy as np
    from softspoken_amd import synth
    base = [synth.to_pcm16(synth.synth_audio(3000 + k, REC_S, REC_SR, 1)) for k in range(min(N_DISTINCT, n_files))]
    files = [base[k % len(base)] for k in range(n_files)]
    frames = np.array([len(f) for f in files], dtype=np.int64)
    return base, files, frames


def cpu_baseline(sd_np, recordings):
    """the CPU restatement on a bounded sample -- batches of settings.prediction_batch_size = 32 windows drawn from
    two recordings, the reference's thread rule (cores // 2, settings.py:32) and all cores, the mask + spec graph the reference
    executes (pytorch_neural_nets.py:184-185) and mask-only.  One timed batch per case after a 2-window warm-up."""
    import numpy as np
    import torch
    from softspoken_amd import synth
    from oracle import oracle_np as O
    cores = os.cpu_count() or 1
    torch.set_grad_enabled(False)
    sd = synth.to_torch_state_dict(sd_np)
    wins = []
    for rec in recordings[:2]:
        x = rec[: REC_SR * 30].astype(np.float32) / np.float32(32768.0)          # 30 s of each: 16 windows from the middle
        padded = O.pad_3s(O.resample(x, REC_SR))
        starts = O.plan_windows(30.0)[10:26]
        wins += [padded[s: s + 66150] for s in starts]
    batch = torch.from_numpy(np.stack(wins))                                         # (32, 66150)
    cases = {}
    cap = min(32, cores)
    plan = [("half", max(1, cores // 2)), ("all", cores)] + ([("capped", cap)] if cap < max(1, cores // 2) else [])
    t_all = time.perf_counter()
    for tag, threads in plan:
        torch.set_num_threads(threads)
        # (all cores: the mask + spec graph only -- on a 256-thread host a batch takes half a minute there, and the sample is bounded)
        for graph, spec in ((("mask+spec", True),) if tag == "all" and cores > 64 else (("mask+spec", True), ("mask-only", False))):
            O.model_forward(sd, batch[:2], want_spec=spec)                           # warm-up (thread pool, oneDNN primitives)
            t0 = time.perf_counter()
            O.model_forward(sd, batch, want_spec=spec)
            dt = time.perf_counter() - t0
            cases[f"{tag}/{graph}"] = {"threads": threads, "windows_per_s": round(32 / dt, 2), "audio_s_per_s": round(32 * 0.6 / dt, 3),
                                       "s_per_batch_of_32": round(dt, 2)}
    ref = cases["half/mask+spec"]
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    best = max(cases.items(), key=lambda kv: kv[1]["windows_per_s"])
    return {"value": ref["audio_s_per_s"], "unit": "audio-seconds/s", "cores": ref["threads"], "kind": "port",
            "sample": "one batch of 32 windows (16 from each of two recordings) per case, fp32 torch CPU ops, model forward incl. the mel "
                      f"front-end; audio-seconds = windows x 0.6 s step; {time.perf_counter() - t_all:.0f} s of CPU work in all",
            "host_threads": cores, "cpu_model": model, "cases": cases,
            "fastest_case": {"case": best[0], **best[1]},
            "note": "value = the reference's configuration (cores // 2 threads, settings.py:32; mask + spec graph, batch 32, settings.py:12); "
                    "on a many-core host more threads than ~32 only slow a batch of 32 windows down (see cases)"}


def worker_dropin(base, precision, n_files=10):
    """The number a GUI user gets: `n_files` 10-minute WAV files on tmpfs through the drop-in's own entry point --
    root.code.frontend.NNDetector.NNDetector + root.code.backend.worker.ProcessWorker.run (the reference: silencer_ui.py:225-243,
    worker.py:38-139) -- with every signal connected (progress after each batch of 32 windows), rows appended to the detections
    frame and the CSV rewritten after every file, as the reference does.  One job to warm up (context, workspace), one timed."""
    import shutil
    import tempfile
    import numpy as np
    from softspoken_amd import synth
    from softspoken_amd.detections import DetectionProject
    from root.code.frontend.NNDetector import NNDetector
    from root.code.backend.worker import ProcessWorker
    tmp = tempfile.mkdtemp(prefix="ss_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        files = []
        for k in range(n_files):
            path = os.path.join(tmp, f"rec_{k:03d}.wav")
            synth.write_wav(path, base[k % len(base)], REC_SR)
            files.append(path)
        ck = os.path.join(tmp, "model_checkpoint.pth")
        synth.save_checkpoint(ck, 0, epoch=0)

        class PM:
            def __init__(self, fl, csv):
                self.files, self.current_project = fl, {"detections_file": csv}

            def get_unprocessed_list(self):
                return list(self.files)
        det = None
        out = {}
        for rep in ("warm-up", "timed"):
            csv = os.path.join(tmp, f"{rep}_detections.csv")
            pm = PM(files, csv)
            t0 = time.perf_counter()
            if det is None:                              # (a GUI session builds the detector once per "Begin Processing" click; its cost is `detector_s`)
                det = NNDetector(pm, checkpoint_path=ck)
                det.model.precision = precision
            t_det = time.perf_counter() - t0
            plan = det.plan_detection_job()
            n_prog = [0]
            w = ProcessWorker(det, DetectionProject(pm), plan)
            w.signals.fileProgressChanged.connect(lambda p: n_prog.__setitem__(0, n_prog[0] + 1))
            t1 = time.perf_counter()
            w.run()
            dt = time.perf_counter() - t1
            rows = sum(1 for _ in open(csv)) - 1
            out[rep] = (dt, t_det, rows, n_prog[0])
        dt, _, rows, n_prog = out["timed"]
        audio = n_files * REC_S
        return {"workload": f"{n_files} x 10 min 16 kHz mono PCM16 WAV files on tmpfs through NNDetector + ProcessWorker.run: progress signals "
                            "(one per 32 windows), rows appended, CSV rewritten after every file",
                "value": round(audio / dt, 1), "unit": "audio-seconds/s", "s_per_job": round(dt, 3), "dtype": det.model.effective_precision(),
                "detection_rows": rows, "progress_signals": n_prog, "first_job_s": round(out["warm-up"][0], 3), "detector_s": round(out["warm-up"][1], 3),
                "value_first_job_incl_detector": round(audio / (out["warm-up"][0] + out["warm-up"][1]), 1)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def rehearse(a, rank, world):
    """The N > 1 control path without a GPU: rendezvous (gloo), one row gather per step, max-over-ranks timing, rank 0's line."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from softspoken_amd import parallel
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    rows = np.array([[rank * 10 + k, 0.5 * k, 0.5 * k + 0.25] for k in range(3)], dtype=np.float64)
    t0 = time.perf_counter()
    merged = rows
    for _ in range(a.warmup + a.steps):
        merged = parallel.gather_rows(rows) if world > 1 else rows
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({"metric": "audio-seconds processed/sec (whole node), 16 kHz mono", "value": None, "unit": "audio-seconds/s",
                          "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "rehearsal": True, "rows_last_step": int(len(merged)),
                          "config": {"rccl_world_size": dist.get_world_size() if world > 1 else 1, "backend": "gloo"}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--precision", default="f16x2", choices=["f16x2", "fp32", "bf16"])
    ap.add_argument("--files", type=int, default=100, help="10-minute recordings per GPU (C3: 100)")
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--rehearse", action="store_true",
                    help="launcher / rendezvous / gather path only, on the CPU with gloo (tests/test_parallel.py): no device work, value null")
    a = ap.parse_args()

    world_env = os.environ.get("WORLD_SIZE")
    if a.gpus > 1 and world_env is None:                 # the documented `python bench.py --gpus N`: start the ranks ourselves
        sys.exit(launch_ranks(a, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node equal to --gpus (or let bench.py start the ranks)")

    import numpy as np
    import torch
    if a.rehearse:
        return rehearse(a, rank, world)
    dist = None
    backend = None
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
        assert dist.get_world_size() == a.gpus

    from softspoken_amd import synth, native, checkpoint, parallel, pipeline
    sd_np = synth.make_state_dict(0)
    blob = checkpoint.pack_state_dict(sd_np)
    t_c0 = time.perf_counter()
    ctx = native.Context(blob, local_rank, precision=a.precision, chunk=a.chunk or None)     # fold + pack + upload of the checkpoint
    t_create = time.perf_counter() - t_c0

    base, files, frames = make_recordings(a.files)
    n_files = len(files)
    pcm = np.concatenate(files)
    dev = torch.device("cuda", local_rank) if (dist is None or backend == "nccl") else torch.device("cpu")
    audio_s_per_step = float(frames.sum()) / REC_SR
    # ---- the job's files as RIFF/WAVE images in page-locked host memory (what a loader thread would have read from disk), and two
    # staging buffers in HBM: job k + 1's samples arrive in one while job k's decode kernels read the other ----
    images = [synth.wav_bytes(f, REC_SR) for f in base]
    wavs = []
    for k in range(n_files):
        img = images[k % len(images)]
        h = ctx.host_alloc(len(img))
        h[:] = np.frombuffer(img, dtype=np.uint8)
        wavs.append(h)
    stage_cap = int(pcm.nbytes) + 4096
    stage = [ctx.device_alloc(stage_cap), ctx.device_alloc(stage_cap)]
    d_pcm = stage[0]
    uploaded = {"turn": 0, "infos": None}                # the upload in flight (or landed) and the slot it went to

    def upload_next(c):                                 # header walk of every file + async H2D of its samples (copy stream)
        slot = uploaded["turn"] & 1
        uploaded["turn"] += 1
        infos = c.upload_wav_batch_async(wavs, stage[slot], stage_cap)
        uploaded["infos"] = (slot, infos)

    def submit(c, _job=None):                           # device half of a job: decode + resample + windows + averaging, enqueued
        if uploaded["infos"] is None:
            upload_next(c)
        slot, infos = uploaded["infos"]
        fr = np.fromiter((i.frames for i in infos), dtype=np.int64, count=n_files)
        i0 = infos[0]
        c.reset()
        first = c.add_pcm_batch_device(stage[slot], i0.format, i0.sample_rate, i0.channels, fr)   # (waits for the copies on the device)
        c.run_begin(0.1, 0.5)
        upload_next(c)                                  # the next job's files cross PCIe while this one computes
        return first

    def submit_resident(c, _job=None):                  # round 2's step: the PCM already is in HBM (stage[0] holds a landed upload)
        c.reset()
        first = c.add_pcm_batch_device(d_pcm, native.PCM_S16, REC_SR, 1, frames)
        c.run_begin(0.1, 0.5)
        return first

    def end(c, _job, first):                            # wait for the device half
        c.run_end()

    def results(c, _job, first):                        # host half: regions, rows (+ the gather across ranks)
        counts, reg = c.regions_batch(first, n_files)          # detection rows (file index, start, end) of the whole job
        fidx = np.repeat(np.arange(n_files, dtype=np.int64) + rank * n_files, counts)
        rows = np.column_stack([fidx.astype(np.float64), reg[:, 0], reg[:, 1]]) if len(fidx) else np.zeros((0, 3))
        if world > 1:
            return parallel.gather_rows(rows, device=dev)
        return rows

    def run_steps(c, k_steps, sub=None):                # one context: results of job k are read after job k + 1 has been submitted
        rows = None
        for rows in pipeline.run_jobs([c], range(k_steps), sub or submit, end, results):
            pass
        return rows

    def fence(cs):
        if world > 1:
            dist.barrier()
        for c in cs:
            c.sync()
        torch.cuda.synchronize()

    def timed(c, warmup, steps, sub=None):
        if warmup > 0:
            run_steps(c, warmup, sub)
        fence([c])
        t0 = time.perf_counter()
        rows = run_steps(c, steps, sub)
        fence([c])
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, rows

    # "reference-style" clock (silencer_ui.py:222-225 starts it before the detector is built): context creation + the
    # first, cold step (workspace allocation, first-use kernel loads); reported beside the steady-state value, never as it
    t_f0 = time.perf_counter()
    rows = run_steps(ctx, 1) if a.warmup > 0 else None
    ctx.sync()
    t_first = time.perf_counter() - t_f0 if a.warmup > 0 else None
    dt, rows = timed(ctx, max(a.warmup - 1, 0), a.steps)
    n_windows = sum(ctx.num_windows(k) for k in range(n_files))
    device_ms = ctx.last_run_device_ms()
    mem_in_use = None
    try:
        free_b, total_b = torch.cuda.mem_get_info(local_rank)
        mem_in_use = round((total_b - free_b) / 2 ** 30, 1)
    except Exception:
        pass

    # ---- roofline: profiled pass of the same path (HIP events around every launch on the library's stream) ----
    roof = stft = None
    kernels, layer_table = [], []
    ctx.upload_wait()                                    # (stage[0] holds a landed upload from here on: the blocks below read it as resident PCM)
    if rank == 0:
        prof = native.Context(blob, local_rank, precision=a.precision, profile=True, chunk=a.chunk or None)
        pf = min(n_files, 10)                           # 10 recordings: 10 050 windows in 10 passes of 1005 (the job's passes are 1015-1016)

        def prof_pass():
            prof.reset(); prof.add_pcm_batch_device(d_pcm, native.PCM_S16, REC_SR, 1, frames[:pf]); prof.run(0.1, 0.5)
        prof_pass()
        prof.reset_stats()
        nprof = 2
        for _ in range(nprof):
            prof_pass()
        pw = sum(prof.num_windows(k) for k in range(pf))
        raw = [s for s in prof.kernel_stats() if s["launches"]]
        layers = [dict(s) for s in raw if "/" in s["name"]]
        merged = {}
        for s in raw:                                   # "<kernel>/<layer>" -> per-kernel totals
            k = s["name"].split("/")[0]
            m = merged.setdefault(k, dict(name=k, launches=0, total_ms=0.0, flops=0.0, bytes=0.0, issued_flops=0.0))
            for f in ("launches", "total_ms", "flops", "bytes", "issued_flops"):
                m[f] += s[f]
        stats = list(merged.values())
        tot = sum(s["total_ms"] for s in stats)
        for s in sorted(stats, key=lambda s: -s["total_ms"]):
            kernels.append({"name": s["name"], "launches_per_pass": s["launches"] // nprof,
                            "ms_per_pass": round(s["total_ms"] / nprof, 4), "share": round(s["total_ms"] / tot, 4),
                            "avg_us": round(1e3 * s["total_ms"] / s["launches"], 2),
                            "tflops": round(s["flops"] / max(s["total_ms"], 1e-9) / 1e9, 2),
                            "gbs": round(s["bytes"] / max(s["total_ms"], 1e-9) / 1e6, 1)})
        dom = max(stats, key=lambda s: s["total_ms"])
        peak = MFMA_PEAK_TFLOPS[a.precision]
        prods = MFMA_PRODUCTS[a.precision]
        ach_tf = dom["flops"] / dom["total_ms"] / 1e9                   # algorithmic: 2 x multiply-adds of the layer
        iss_tf = dom["issued_flops"] / dom["total_ms"] / 1e9            # what the matrix pipe was given: products x the multiply-adds of the form that ran
        ach_gbs = dom["bytes"] / dom["total_ms"] / 1e6
        intensity = prods * dom["flops"] / max(dom["bytes"], 1.0)      # issued matrix FLOP per algorithmic byte
        ridge = peak * 1e12 / (HBM_PEAK_GBS * 1e9)
        wpl = round(pw * nprof / max(1, next(s for s in stats if s["name"] == "frontend")["launches"]), 1)
        # measured HBM bytes per launch of this instantiation: the committed PMC passes (FETCH_SIZE x 2 -- gfx950 counts wide streaming
        # reads at half --, WRITE_SIZE; separate rocprofv3 --pmc runs, MI355X_MICROARCH.md 'HBM'), scaled to this pass's windows
        avg_us = 1e3 * dom["total_ms"] / dom["launches"]
        traffic, traffic_src = traffic_of(dom["name"], avg_us, wpl, dom["bytes"] / dom["launches"])
        common = {"kernel": dom["name"], "traffic": traffic, "traffic_source": traffic_src, "avg_launch_us": round(1e3 * dom["total_ms"] / dom["launches"], 2),
                  "windows_per_launch": wpl,
                  "layers_of_this_instantiation": [l["name"].split("/", 1)[1] for l in layers if l["name"].startswith(dom["name"] + "/")],
                  "flops_per_launch": dom["flops"] / dom["launches"], "bytes_per_launch": dom["bytes"] / dom["launches"],
                  "matrix_products_per_multiply_add": prods,
                  "flop_per_byte_issued": round(intensity, 1), "ridge_flop_per_byte": round(ridge, 1),
                  "frac_issued": round(iss_tf / peak, 4),
                  "mfma": {"achieved_tflops_algorithmic": round(ach_tf, 2), "issued_tflops": round(iss_tf, 2), "peak": peak,
                           "frac_algorithmic": round(ach_tf / peak, 4), "frac_issued": round(iss_tf / peak, 4),
                           "issued_over_algorithmic": round(iss_tf / max(ach_tf, 1e-9), 3),
                           "ceiling_of_algorithmic_frac": round(ach_tf / max(iss_tf, 1e-9), 4),
                           "note": "issued = the products the form that ran gives the matrix pipe (ss_kernel_stat.issued_flops): 3 per multiply-add in "
                                   "f16x2, and the sub-pixel launches run 4 taps instead of 9 on their upsampled input half -- comparable with the "
                                   "mfma_busy counter of profiles/r04_pmc.md"},
                  "hbm": {"achieved_gbs": round(ach_gbs, 1), "peak": HBM_PEAK_GBS, "frac": round(ach_gbs / HBM_PEAK_GBS, 4)},
                  "mfma_sustained": ({"tflops": MFMA_SUSTAINED_TFLOPS[a.precision], "frac_issued": round(iss_tf / MFMA_SUSTAINED_TFLOPS[a.precision], 4),
                                      "note": "a loop of only this matrix instruction under the part's power cap (1.78 GHz, ~1300 W): tools/probes/mfma_power.hip, DESIGN.md"}
                                     if a.precision in MFMA_SUSTAINED_TFLOPS else None),
                  "measured": "HIP events around each launch on the library's stream, profiled passes of the same path over "
                              f"{pf} of the recordings ({pw} windows per pass of the job)"}
        if intensity < ridge:   # below the ridge the kernel's roof is bandwidth
            roof = dict(bound="hbm", achieved=round(ach_gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach_gbs / HBM_PEAK_GBS, 4), **common)
        else:                   # achieved / peak in ALGORITHMIC FLOP/s (f16x2 issues `prods` matrix products per multiply-add: frac_issued)
            roof = dict(bound="mfma", achieved=round(ach_tf, 2), peak=peak, unit="TFLOP/s", frac=round(ach_tf / peak, 4), **common)
        conv_ms = sum(s["total_ms"] for s in stats if s["name"].startswith("conv3x3"))
        conv_fl = sum(s["flops"] for s in stats if s["name"].startswith("conv3x3"))
        conv_is = sum(s["issued_flops"] for s in stats if s["name"].startswith("conv3x3"))
        roof["all_conv3x3_tflops_algorithmic"] = round(conv_fl / conv_ms / 1e9, 2)
        roof["all_conv3x3_frac_algorithmic"] = round(conv_fl / conv_ms / 1e9 / peak, 4)
        roof["all_conv3x3_frac_issued"] = round(conv_is / conv_ms / 1e9 / peak, 4)
        fe = next(s for s in stats if s["name"] == "frontend")
        fe_gbs = fe["bytes"] / fe["total_ms"] / 1e6
        fe_win = fe["bytes"] / FRONTEND_BYTES_PER_WINDOW
        fe_tf = fe_win * FRONTEND_FLOPS_PER_WINDOW / (fe["total_ms"] / 1e3) / 1e12
        # the stage sits above the fp32 ridge (38 FLOP/B against 157.3 TFLOP/s / 8 TB/s = 20): what bounds it is vector issue, not HBM
        stft = {"kernel": "frontend_kernel", "bound": "valu", "achieved": round(fe_tf, 2), "peak": VALU_FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(fe_tf / VALU_FP32_PEAK_TFLOPS, 4), "flops_per_window": FRONTEND_FLOPS_PER_WINDOW,
                "hbm": {"achieved_gbs": round(fe_gbs, 1), "peak": HBM_PEAK_GBS, "frac": round(fe_gbs / HBM_PEAK_GBS, 4), "bytes_per_window": FRONTEND_BYTES_PER_WINDOW,
                        "note": "the north star's >= 60 % of HBM for this stage: algorithmic bytes / duration"},
                "flop_per_byte": round(FRONTEND_FLOPS_PER_WINDOW / FRONTEND_BYTES_PER_WINDOW, 1), "ridge_flop_per_byte_fp32_vector": round(VALU_FP32_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBS, 1),
                "windows_per_s": round(fe_win / (fe["total_ms"] / 1e3), 0),
                "avg_launch_us": round(1e3 * fe["total_ms"] / fe["launches"], 2), "traffic": None}
        # what this instruction stream could reach: the kernel issues FRONTEND_VALU_PER_WINDOW vector instructions per window (SQ_INSTS_VALU,
        # profiles/r04_pmc.md), 4 cycles of a SIMD each at best (gfx950 issues packed fp32 at the scalar rate): with every SIMD busy every
        # cycle that is the stream's floor, and the algorithmic bytes over it the most of the HBM roof this kernel can show
        floor_us = FRONTEND_VALU_PER_WINDOW * 4.0 / (1024 * 2.4e3)
        stft["instruction_stream"] = {"valu_instructions_per_window": FRONTEND_VALU_PER_WINDOW, "floor_us_per_window": round(floor_us, 4),
                                      "attainable_hbm_frac": round(FRONTEND_BYTES_PER_WINDOW / (floor_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                      "achieved_over_attainable": round(floor_us * fe_win / (fe["total_ms"] * 1e3), 4),
                                      "note": "the stage sits above the fp32 ridge: its ceiling is vector issue, not HBM; the north star's 60 % of HBM is out of "
                                              "this stream's reach at fp32 accuracy (DESIGN.md section 6)"}
        fe_name = "frontend_kernel"
        stft["traffic"], stft["traffic_source"] = traffic_of(fe_name, stft["avg_launch_us"], fe_win / fe["launches"], fe["bytes"] / fe["launches"], prefix=True)
        prof.close()
        layer_table = [{"layer": s["name"], "us": round(1e3 * s["total_ms"] / s["launches"], 1),
                        "tflops": round(s["flops"] / max(s["total_ms"], 1e-9) / 1e9, 1)} for s in layers]

    # ---- secondary blocks (1-GPU runs only: other ranks would wait) ----
    secondary = {}
    if rank == 0 and world == 1 and not a.no_secondary:
        # the headline's job with the PCM already resident in HBM (no header walk, no H2D in the step): round 2's headline
        ctx.upload_wait()
        ctx.device_upload(d_pcm, pcm)
        nres = min(a.steps, 5)
        dtr, _ = timed(ctx, 1, nres, submit_resident)
        secondary["c3_resident"] = {"workload": f"C3, PCM resident in HBM before the clock starts (round 2's headline): {n_files} x 10 min, {a.precision}",
                                    "value": round(audio_s_per_step * nres / dtr, 1), "unit": "audio-seconds/s",
                                    "ms_per_step": round(1e3 * dtr / nres, 3), "steps": nres, "dtype": a.precision}
        uploaded["infos"] = None
        # C3 in the other parity mode, on all the recordings of the job (BASELINE config 3: 100)
        other = "fp32" if a.precision != "fp32" else "f16x2"
        oc = native.Context(blob, local_rank, precision=other, chunk=a.chunk or None)
        nf2 = n_files
        fr2 = frames[:nf2]

        def submit2(c, _j=None):
            c.reset(); first = c.add_pcm_batch_device(d_pcm, native.PCM_S16, REC_SR, 1, fr2); c.run_begin(0.1, 0.5); return first
        for _ in pipeline.run_jobs([oc], range(1), submit2, end, lambda c, j, f: None):
            pass
        n2 = 2
        oc.sync(); t0 = time.perf_counter()
        for _ in pipeline.run_jobs([oc], range(n2), submit2, end, lambda c, j, f: c.regions_batch(f, nf2)):
            pass
        oc.sync(); dt2 = time.perf_counter() - t0
        w2 = sum(oc.num_windows(k) for k in range(nf2))
        secondary[f"c3_{other}"] = {"workload": f"C3: {nf2} x 10 min 16 kHz mono, whole path from PCM resident in HBM, {other}", "value": round(n2 * float(fr2.sum()) / REC_SR / dt2, 1),
                                    "unit": "audio-seconds/s", "windows_per_s": round(n2 * w2 / dt2, 1), "ms_per_step": round(1e3 * dt2 / n2, 1), "steps": n2, "dtype": other}
        oc.close()
        # C2: 256 x 3 s clips, bf16 (BASELINE configs[1]; the throughput mode, scores NOT within 1e-4)
        clips = [synth.to_pcm16(synth.synth_audio(2000 + k, 3.0, 16000, 1, with_silence=False)) for k in range(256)]
        cfr = np.array([len(c_) for c_ in clips], dtype=np.int64)
        cp = np.concatenate(clips)
        bc = native.Context(blob, local_rank, precision="bf16")
        d_c = bc.device_alloc(cp.nbytes); bc.device_upload(d_c, cp)

        def submit3(c, _j=None):
            c.reset(); first = c.add_pcm_batch_device(d_c, native.PCM_S16, 16000, 1, cfr); c.run_begin(0.1, 0.5); return first
        for _ in pipeline.run_jobs([bc], range(3), submit3, end, lambda c, j, f: c.regions_batch(f, 256)):
            pass
        bc.sync(); t0 = time.perf_counter()
        for _ in pipeline.run_jobs([bc], range(20), submit3, end, lambda c, j, f: c.regions_batch(f, 256)):
            pass
        bc.sync(); dt3 = time.perf_counter() - t0
        secondary["c2_bf16"] = {"workload": "C2: 256 x 3 s 16 kHz mono clips, whole path, bf16 (throughput mode: scores differ from the reference by up to ~0.1)",
                                "value": round(20 * 768 / dt3, 1), "unit": "audio-seconds/s", "windows_per_s": round(20 * 2560 / dt3, 1),
                                "ms_per_step": round(1e3 * dt3 / 20, 3), "steps": 20, "dtype": "bf16"}
        bc.device_free(d_c); bc.close()
        # C5: 48 kHz stereo PCM16 -> decode + mixdown + resample (147/320) + mel front-end only (BASELINE configs[4], the HBM roofline run)
        x2 = synth.to_pcm16(synth.synth_audio(5000, 120.0, 48000, 2, with_silence=False))       # (frames, 2) int16, 2 min ...
        x5 = np.concatenate([x2] * 5)                                                            # ... tiled to 10 min
        nf5 = 8
        fr5 = np.array([x5.shape[0]] * nf5, dtype=np.int64)
        fc = native.Context(blob, local_rank, precision="bf16", profile=True)
        p5 = np.concatenate([x5] * nf5)
        d5 = fc.device_alloc(p5.nbytes); fc.device_upload(d5, p5)
        st5 = native.plan_windows(600.0)
        dt5 = None
        for rep in range(3):
            fc.sync(); t0 = time.perf_counter()
            fc.reset()
            first = fc.add_pcm_batch_device(d5, native.PCM_S16, 48000, 2, fr5)
            for k in range(nf5):
                fc.features(first + k, st5, discard=True)
            fc.sync(); dt5 = time.perf_counter() - t0
            if rep == 0:
                fc.reset_stats()
        w5 = nf5 * len(st5)
        ks = {s["name"]: s for s in fc.kernel_stats() if s["launches"]}
        dev_ms = sum(s["total_ms"] for s in ks.values()) / 2
        secondary["c5_frontend"] = {"workload": f"C5: {nf5} x 10 min 48 kHz stereo PCM16: decode + mixdown + resample 147/320 + STFT/mel front-end (no conv stack)",
                                    "value": round(nf5 * 600 / dt5, 1), "unit": "audio-seconds/s", "windows_per_s": round(w5 / dt5, 1),
                                    "bytes_per_window": C5_BYTES_PER_WINDOW,
                                    "roofline": {"bound": "hbm", "achieved": round(w5 * C5_BYTES_PER_WINDOW / (dev_ms / 1e3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                                                 "unit": "GB/s", "frac": round(w5 * C5_BYTES_PER_WINDOW / (dev_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 4),
                                                 "device_ms_per_pass": round(dev_ms, 3),
                                                 "note": "algorithmic bytes of the stage / summed kernel time (HIP events); wall-clock value includes host launch gaps"},
                                    "kernels_ms_per_pass": {k: round(v["total_ms"] / 2, 3) for k, v in ks.items()}}
        fc.device_free(d5); fc.close()

    if rank == 0 and world == 1 and not a.no_secondary:
        secondary["worker_dropin"] = worker_dropin(base, a.precision)

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:       # the CPU baseline is a 1-GPU-run item (other ranks would wait on it)
        cpu = cpu_baseline(sd_np, base)

    if rank == 0:
        total_audio = world * audio_s_per_step * a.steps
        dtype_note = {"f16x2": "f16x2 (fp32-accurate: fp32 operands as two f16 halves, three f16 matrix products per term, fp32 accumulate)",
                      "fp32": "fp32 (fp32 matrix instructions)", "bf16": "bf16 (throughput mode, not the parity mode)"}[a.precision]
        out = {
            "metric": "audio-seconds processed/sec (whole node), 16 kHz mono",
            "value": round(total_audio / dt, 2), "unit": "audio-seconds/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.precision, "data": "synthetic",
            "config": {"workload": f"C3: {n_files} x {REC_S / 60:g} min {REC_SR} Hz mono PCM16 recordings per GPU ({N_DISTINCT} distinct ones, repeated), "
                                   f"{dtype_note}, from host WAV bytes, H2D in the step: header walk + samples host->HBM (the copies of job k+1 beside "
                                   "the kernels of job k) + decode+resample+front-end+U-Net+averaging+regions"
                                   "; results of job k are read after job k+1 has been submitted"
                                   + ("+RCCL row gather" if world > 1 else ""),
                       "windows_per_step_per_gpu": int(n_windows), "graph": "mask-only (spec head skipped, 6.360 GFLOP/window)",
                       "precision_contract": "scores within 1e-4 of the reference's fp32 CPU path (tests/test_gpu_parity.py, tests/test_gpu_c3.py)"
                                             if a.precision != "bf16" else "throughput mode: scores within ~0.1",
                       "weights": "synthetic checkpoint, reference state_dict layout", "parallelism": f"file-sharded dp{world}",
                       "rccl_world_size": (dist.get_world_size() if world > 1 else 1), "backend": backend or "none"},
            "windows_per_s": round(world * n_windows * a.steps / dt, 1),
            "device_ms_last_run": round(device_ms, 3),
            "hbm_in_use_gib": mem_in_use,
            "rows_last_step": int(len(rows)) if rows is not None else 0,
            "roofline": roof, "stft_stage": stft, "cpu_baseline": cpu, "secondary": secondary, "kernels": kernels, "layers": layer_table,
        }
        if t_first is not None:
            out["cold_start"] = {"create_ms": round(1e3 * t_create, 1), "first_step_ms": round(1e3 * t_first, 1),
                                 "value_reference_style": round(audio_s_per_step / (t_create + t_first), 1),
                                 "note": "one job on a fresh process: context creation + first step (rank 0's clock)"}
        print(json.dumps(out), flush=True)
    ctx.upload_wait()
    for p in stage:
        ctx.device_free(p)
    for h in wavs:
        ctx.host_free(h)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
