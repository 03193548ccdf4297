import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def sd_np():
    from softspoken_amd import synth
    return synth.make_state_dict(0)


@pytest.fixture(scope="session")
def sd_torch(sd_np):
    from softspoken_amd import synth
    return synth.to_torch_state_dict(sd_np)


@pytest.fixture(scope="session")
def blob(sd_np):
    from softspoken_amd import checkpoint
    return checkpoint.pack_state_dict(sd_np)


@pytest.fixture(scope="session")
def c1():
    """The C1 case (60 s, 16 kHz mono, seed 1001): wav bytes, oracle-resampled signal, plan."""
    from softspoken_amd import synth
    from oracle import oracle_np as O
    pcm = synth.to_pcm16(synth.synth_audio(1001, 60.0, 16000, 1))
    wav = synth.wav_bytes(pcm, 16000)
    sig, _, info = O.load_audio_from_bytes(wav)
    return dict(wav=wav, pcm=pcm, sig=sig, info=info, padded=O.pad_3s(sig), starts=O.plan_windows(60.0))


@pytest.fixture(scope="session")
def gold():
    return {name: np.load(os.path.join(GOLDEN, name + ".npz")) for name in
            ("c1_logits", "c1_features", "c1_layers", "c1_signal_check", "mel_tables")}


@pytest.fixture(scope="session")
def build_all():
    import __graft_entry__ as ge
    ge.build()
    return True
