"""Build libsoftspoken_hip.so (gfx950) in-tree with hipcc.  hipcc cross-compiles without a GPU.

    python -m softspoken_amd.build [--force] [--dev]

Two libraries come out of the same sources:
  libsoftspoken_hip.so       the product: every kernel choice fixed at build time; reads SOFTSPOKEN_CHUNK / SOFTSPOKEN_PRECISION only
  libsoftspoken_hip_dev.so   -DSS_DEVBUILD: the development switches (alternate kernel forms, timing-only ablation bits, sleeps at
                             the conv kernels' synchronisation points) are compiled in and read from the environment.  Loaded by
                             tests/ and tools/ through SOFTSPOKEN_LIB, never by the drop-in.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libsoftspoken_hip.so")
DEV_LIB = os.path.join(HERE, "libsoftspoken_hip_dev.so")
SOURCES = ["conv2.hip", "conv2_ups.hip", "conv4.hip", "conv4_ups.hip", "conv1s.hip", "frontend.hip", "heads.hip", "weights.hip", "engine.hip", "host.hip", "abi.hip"]
HEADERS = [os.path.join(CSRC, "kernels.h"), os.path.join(CSRC, "engine.h"), os.path.join(os.path.dirname(HERE), "include", "softspoken.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wno-unused-value"]
# per-file additions.  frontend.hip: the SLP vectoriser turns the complex butterflies into v_pk_*_f32 and then spends a quarter of
# the FFT loop's instructions on v_mov to pair registers for them; hand-packed code is shorter (measured on the GPU).
# conv4.hip: the same flag keeps v_pk_add_f32 out of the residual adds (packed fp32 beside MFMAs costs more than it saves): +0.6 %.
EXTRA_FLAGS = {"frontend.hip": ["-fno-slp-vectorize"], "conv4.hip": ["-fno-slp-vectorize"], "conv4_ups.hip": ["-fno-slp-vectorize"],
               "conv1s.hip": ["-fno-slp-vectorize"]}
# sources whose dev build differs from the product build (the others are shared between the two libraries)
DEV_SOURCES = ("conv2.hip", "conv2_ups.hip", "conv4.hip", "conv4_ups.hip", "conv1s.hip", "frontend.hip", "engine.hip", "weights.hip", "abi.hip")
JITTER_LIB = DEV_LIB          # (the sleeps at synchronisation points are one of the dev build's switches)


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, dev: bool = False, jitter: bool = False) -> str:
    """The product library, or (dev=True) the development build of the same sources."""
    dev = dev or jitter
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    objs, procs = [], []
    lib = DEV_LIB if dev else LIB
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        d = dev and src in DEV_SOURCES
        o = os.path.join(OBJ, src.replace(".hip", "_dev.o" if d else ".o"))
        objs.append(o)
        if force or _stale(o, [s] + HEADERS):
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + (["-DSS_DEVBUILD"] if d else []) + ["-c", s, "-o", o]
            if verbose:
                print("[build]", " ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out))
    if force or procs or _stale(lib, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, dev="--dev" in sys.argv or "--jitter" in sys.argv))
