"""Windows per pass (SOFTSPOKEN_CHUNK) against throughput on 20 x 10 min (development aid): python tools/chunk_sweep.py f16x2 512 1024 2048"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softspoken_amd import synth, native, checkpoint
prec = sys.argv[1]
blob = checkpoint.pack_state_dict(synth.make_state_dict(0))
x = synth.to_pcm16(synth.synth_audio(3000, 600.0, 16000, 1))
nf = 20
for chunk in [int(v) for v in sys.argv[2:]]:
    c = native.Context(blob, 0, precision=prec, chunk=chunk)
    frames = np.array([len(x)] * nf, dtype=np.int64)
    pcm = np.concatenate([x] * nf)
    d = c.device_alloc(pcm.nbytes); c.device_upload(d, pcm)
    best = 1e9
    for rep in range(3):
        c.reset(); first = c.add_pcm_batch_device(d, native.PCM_S16, 16000, 1, frames)
        c.sync(); t0 = time.perf_counter()
        assert c.run()
        best = min(best, time.perf_counter() - t0)
    print(f"{prec} chunk {chunk}: {nf * 600 / best:.0f} audio-s/s ({nf * 1005 / best:.0f} windows/s), workspace {c.workspace_bytes() / 2**30:.1f} GiB", flush=True)
    c.device_free(d); c.close()
