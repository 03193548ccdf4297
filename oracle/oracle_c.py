"""ORACLE -- TEST INFRASTRUCTURE ONLY.  ctypes loader for the plain-C restatement (oracle/ss_oracle.c).
Only tests/, __graft_entry__ and bench.py's cpu_baseline leg use it."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "ss_oracle.c")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "libss_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    os.makedirs(OUT_DIR, exist_ok=True)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        # no -march=native: the .so travels to the GPU box, whose host CPU may differ
        cmd = ["gcc", "-O3", "-fopenmp", "-shared", "-fPIC", "-ffp-contract=off", SRC, "-o", LIB, "-lm"]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("gcc failed:\n" + r.stdout)
    return LIB


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        P = C.c_void_p
        L.so_decode_mono.argtypes = [P, C.c_int, C.c_int, C.c_int64, P]
        L.so_resampled_length.restype = C.c_int64
        L.so_resampled_length.argtypes = [C.c_int64, C.c_int]
        L.so_resample.argtypes = [P, C.c_int64, C.c_int, P, C.c_int64]
        L.so_mel_features.argtypes = [P, C.c_int, P, P, P]
        L.so_unet_forward.argtypes = [P, P, C.c_int, P, P, P]
        L.so_plan_windows.restype = C.c_int64
        L.so_plan_windows.argtypes = [C.c_double]
        L.so_average.restype = C.c_int64
        L.so_average.argtypes = [P, C.c_int64, C.c_int64, P, P]
        L.so_regions.restype = C.c_int64
        L.so_regions.argtypes = [P, P, C.c_int64, C.c_double, C.c_double, P, C.c_int64]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def decode_resample(pcm_bytes: np.ndarray, fmt: int, ch: int, frames: int, sr: int) -> np.ndarray:
    raw = np.ascontiguousarray(pcm_bytes)
    mono = np.empty(frames, np.float32)
    lib().so_decode_mono(_p(raw), fmt, ch, frames, _p(mono))
    n = lib().so_resampled_length(frames, sr)
    out = np.empty(n, np.float32)
    lib().so_resample(_p(mono), frames, sr, _p(out), n)
    return out


def mel_features(x: np.ndarray, window: np.ndarray, fb: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty((x.shape[0], 128, 256), np.float32)
    lib().so_mel_features(_p(x), x.shape[0], _p(np.ascontiguousarray(window, np.float32)),
                          _p(np.ascontiguousarray(fb, np.float32)), _p(out))
    return out


def unet_forward(blob: bytes, feats: np.ndarray, want_spec: bool = False):
    b = np.frombuffer(blob, np.uint8)
    feats = np.ascontiguousarray(feats, np.float32)
    n = feats.shape[0]
    mask = np.empty((n, 1, 256), np.float32)
    spec = np.empty((n, 2, 128, 256), np.float32) if want_spec else None
    flat = np.empty((n, 4, 256), np.float32)
    lib().so_unet_forward(_p(b), _p(feats), n, _p(mask), _p(spec), _p(flat))
    return spec, mask, flat


def plan_windows(duration_s: float) -> int:
    return lib().so_plan_windows(float(duration_s))


def average(logits: np.ndarray, n_padded: int):
    lg = np.ascontiguousarray(logits, np.float32).reshape(-1, 256)
    nb = int(round(n_padded / 22050 * 256 / 3)) + 8
    avg = np.empty(nb, np.float64)
    idx = np.empty(nb, np.int64)
    k = lib().so_average(_p(lg), lg.shape[0], n_padded, _p(avg), _p(idx))
    return avg[:k].copy(), idx[:k].copy()


def regions(avg, idx, thr=0.1, brk=0.5):
    avg = np.ascontiguousarray(avg, np.float64)
    idx = np.ascontiguousarray(idx, np.int64)
    out = np.empty((len(avg) // 2 + 2, 2), np.float64)
    m = lib().so_regions(_p(avg), _p(idx), len(avg), thr, brk, _p(out), out.shape[0])
    return [(float(out[i, 0]), float(out[i, 1])) for i in range(m)]
