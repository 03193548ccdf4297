"""Scale / configuration checks on a GPU box (development aid; numbers quoted in DESIGN.md section 5):
  c5     48 kHz stereo PCM16, 10-min files: decode + mixdown + resample(147/320) + front-end only (BASELINE config 5)
  c3     10-min 16 kHz mono files through the whole path in fp32 (BASELINE config 3 shape, fewer files)
  c4     10-min 16 kHz mono files through the whole path in bf16, PCM resident in HBM (one GPU's share of BASELINE config 4)
  nccl   the row gather of softspoken_amd.parallel over RCCL with world_size 1
usage: python tools/scale_check.py c5|c3|c4|nccl [n_files]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softspoken_amd import synth, native, checkpoint

what = sys.argv[1]
nfiles = int(sys.argv[2]) if len(sys.argv) > 2 else 4
blob = checkpoint.pack_state_dict(synth.make_state_dict(0))

if what == "c5":
    ctx = native.Context(blob, 0, bf16=True, profile=True)
    x = synth.to_pcm16(synth.synth_audio(5000, 600.0, 48000, 2, with_silence=False))       # (frames, 2) int16
    frames = np.array([x.shape[0]] * nfiles, dtype=np.int64)
    pcm = np.concatenate([x] * nfiles)
    d = ctx.device_alloc(pcm.nbytes); ctx.device_upload(d, pcm)
    starts = native.plan_windows(600.0)
    for rep in range(3):
        ctx.sync(); t0 = time.perf_counter()
        ctx.reset()
        first = ctx.add_pcm_batch_device(d, native.PCM_S16, 48000, 2, frames)
        for k in range(nfiles):
            ctx.features(first + k, starts, discard=True)
        ctx.sync(); dt = time.perf_counter() - t0
        if rep == 0: ctx.reset_stats()
    nwin = nfiles * len(starts)
    print(f"C5: {nfiles} x 600 s 48 kHz stereo: {nfiles * 600 / dt:.0f} audio-s/s, {nwin / dt:.0f} windows/s, "
          f"{nwin * 707072 / dt / 1e9:.1f} GB/s algorithmic (707 072 B/window)")
    for s in ctx.kernel_stats():
        if s["launches"]:
            print("   %-22s n=%4d %9.3f ms  %8.1f GB/s" % (s["name"], s["launches"], s["total_ms"], s["bytes"] / max(s["total_ms"], 1e-9) / 1e6))
elif what == "c3":
    ctx = native.Context(blob, 0, bf16=False)
    files = [synth.to_pcm16(synth.synth_audio(3000 + k, 600.0, 16000, 1)) for k in range(nfiles)]
    t0 = time.perf_counter()
    ids = [ctx.add_pcm(f, native.PCM_S16, 16000, 1, len(f)) for f in files]
    t1 = time.perf_counter()
    assert ctx.run()
    t2 = time.perf_counter()
    nreg = sum(len(ctx.regions(i)) for i in ids)
    print(f"C3 shape: {nfiles} x 600 s fp32: upload+decode {t1 - t0:.2f} s, run {t2 - t1:.2f} s -> {nfiles * 600 / (t2 - t0):.0f} audio-s/s "
          f"({nfiles * 1005 / (t2 - t1):.0f} windows/s), {nreg} regions, device {ctx.last_run_device_ms():.0f} ms")
elif what == "c4":
    ctx = native.Context(blob, 0, bf16=True)
    base = [synth.to_pcm16(synth.synth_audio(4000 + k, 600.0, 16000, 1)) for k in range(4)]      # four distinct recordings, repeated
    files = [base[k % 4] for k in range(nfiles)]
    frames = np.array([len(f) for f in files], dtype=np.int64)
    pcm = np.concatenate(files)
    d = ctx.device_alloc(pcm.nbytes); ctx.device_upload(d, pcm)
    for rep in range(2):
        ctx.sync(); t0 = time.perf_counter()
        ctx.reset()
        first = ctx.add_pcm_batch_device(d, native.PCM_S16, 16000, 1, frames)
        assert ctx.run()
        counts, reg = ctx.regions_batch(first, nfiles)
        dt = time.perf_counter() - t0
    nwin = sum(ctx.num_windows(first + k) for k in range(nfiles))
    same = all(counts[k] == counts[k % 4] for k in range(nfiles))
    print(f"C4 share: {nfiles} x 600 s bf16: {nfiles * 600 / dt:.0f} audio-s/s ({nwin / dt:.0f} windows/s, {nwin} windows), {int(counts.sum())} regions, "
          f"repeated recordings give repeated tables: {same}, device {ctx.last_run_device_ms():.0f} ms")
    assert same
elif what == "nccl":
    import torch, torch.distributed as dist
    from softspoken_amd import parallel
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    rows = [(3, 1.5, 2.5), (0, 0.25, 0.5), (3, 0.1, 0.2)]
    m = parallel.gather_rows(rows, device=torch.device("cuda", 0))
    print("nccl gather ok:", m.tolist())
    assert m.tolist() == [[0.0, 0.25, 0.5], [3.0, 0.1, 0.2], [3.0, 1.5, 2.5]]
    dist.barrier(); dist.destroy_process_group()
