"""ctypes binding of libsoftspoken_hip.so (C ABI: include/softspoken.h).

There is no CPU fallback: if the library is missing or there is no gfx950 device, the calls
raise.  ctypes releases the GIL for the duration of every foreign call, so the reference's worker
thread (root/code/backend/worker.py, run from a QThreadPool, silencer_ui.py:243) does not block
the GUI thread while the device works.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SOFTSPOKEN_LIB: another build of the same library (tests: the -DSS_JITTER build); never a different implementation
LIB_PATH = os.environ.get("SOFTSPOKEN_LIB") or os.path.join(_HERE, "libsoftspoken_hip.so")

SS_OK = 0
SS_ERR_STOPPED = 5
SS_ERR_NOMEM = 6
SS_ERR_CAPACITY = 7
SS_ERR_RANGE = 8          # f16x2: a weight or an activation has no f16 representation -> run the checkpoint in the fp32 mode
ABI_VERSION = 3
FLAG_BF16 = 1
FLAG_PROFILE = 2
FLAG_F16X2 = 4
PRECISIONS = ("fp32", "f16x2", "bf16")
PCM_U8, PCM_S16, PCM_S24, PCM_S32, PCM_F32, PCM_F64 = 1, 2, 3, 4, 5, 6
PCM_S8, PCM_S16BE, PCM_S24BE, PCM_S32BE, PCM_F32BE, PCM_F64BE = 7, 8, 9, 10, 11, 12    # AIFF / AIFF-C: big endian, 8-bit samples signed

_BPS = {PCM_U8: 1, PCM_S16: 2, PCM_S24: 3, PCM_S32: 4, PCM_F32: 4, PCM_F64: 8,
        PCM_S8: 1, PCM_S16BE: 2, PCM_S24BE: 3, PCM_S32BE: 4, PCM_F32BE: 4, PCM_F64BE: 8}     # bytes per sample of enum ss_pcm_format
SAMPLE_RATE = 22050
WINDOW_SAMPLES = 66150
STEP_SAMPLES = 13230


class WavInfo(C.Structure):
    _fields_ = [("format", C.c_int32), ("channels", C.c_int32), ("sample_rate", C.c_int32), ("bits", C.c_int32),
                ("frames", C.c_int64), ("data_offset", C.c_int64), ("data_bytes", C.c_int64)]


class Region(C.Structure):
    _fields_ = [("start", C.c_double), ("end", C.c_double)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("launches", C.c_int64), ("total_ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double), ("issued_flops", C.c_double)]


PROGRESS_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int64, C.c_int64)

# every symbol include/softspoken.h declares: (restype, argtypes)
_P = C.c_void_p
_SIGS = {
    "ss_abi_version": (C.c_int, []),
    "ss_last_error": (C.c_char_p, [_P]),
    "ss_wav_parse": (C.c_int, [_P, C.c_size_t, C.POINTER(WavInfo)]),
    "ss_resampled_length": (C.c_int64, [C.c_int64, C.c_int]),
    "ss_plan_windows": (C.c_int64, [C.c_double, _P, C.c_int64]),
    "ss_find_regions": (C.c_int, [_P, _P, C.c_int64, C.c_double, C.c_double, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "ss_format_csv_rows": (C.c_int64, [C.c_char_p, C.c_char_p, _P, C.c_int64, C.c_int64, _P, C.c_int64]),
    "ss_create": (C.c_int, [C.c_int, _P, C.c_size_t, C.c_uint32, C.POINTER(_P)]),
    "ss_destroy": (None, [_P]),
    "ss_set_chunk_windows": (C.c_int, [_P, C.c_int]),
    "ss_reset": (C.c_int, [_P]),
    "ss_reset_generation": (C.c_uint64, [_P]),
    "ss_add_pcm": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_int)]),
    "ss_add_pcm_device": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_int)]),
    "ss_add_pcm_batch_device": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.POINTER(C.c_int)]),
    "ss_add_f32_22k": (C.c_int, [_P, _P, C.c_int64, C.POINTER(C.c_int)]),
    "ss_add_padded_f32_22k": (C.c_int, [_P, _P, C.c_int64, C.POINTER(C.c_int)]),
    "ss_signal_length": (C.c_int64, [_P, C.c_int, C.c_int]),
    "ss_read_signal": (C.c_int, [_P, C.c_int, C.c_int, C.c_int64, C.c_int64, _P]),
    "ss_silence_pcm": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int64, _P, C.c_int64, _P]),
    "ss_wav_header_pcm16": (C.c_int, [C.c_int, C.c_int, C.c_int64, _P]),
    "ss_stft512_frames": (C.c_int64, [C.c_int64]),
    "ss_stft512_magnitude": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64]),
    "ss_device_alloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "ss_device_free": (C.c_int, [_P, _P]),
    "ss_device_upload": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "ss_host_alloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "ss_host_free": (C.c_int, [_P, _P]),
    "ss_upload_wav_batch_async": (C.c_int, [_P, _P, _P, C.c_int, _P, C.c_size_t, _P]),
    "ss_device_upload_async": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "ss_upload_wait": (C.c_int, [_P]),
    "ss_features": (C.c_int, [_P, C.c_int, _P, C.c_int, _P]),
    "ss_infer_windows": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, _P]),
    "ss_run": (C.c_int, [_P, C.c_double, C.c_double, _P, _P, _P]),
    "ss_run_begin": (C.c_int, [_P, C.c_double, C.c_double]),
    "ss_run_end": (C.c_int, [_P]),
    "ss_run_begin_tracked": (C.c_int, [_P, C.c_double, C.c_double]),
    "ss_run_poll": (C.c_int, [_P, _P, _P, C.c_int]),
    "ss_run_from_logits": (C.c_int, [_P, _P, C.c_int64, C.c_double, C.c_double]),
    "ss_num_windows": (C.c_int64, [_P, C.c_int]),
    "ss_get_window_logits": (C.c_int, [_P, C.c_int, _P, C.c_int64]),
    "ss_get_avg": (C.c_int, [_P, C.c_int, _P, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "ss_get_regions": (C.c_int, [_P, C.c_int, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "ss_get_regions_batch": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "ss_sync": (C.c_int, [_P]),
    "ss_reset_kernel_stats": (C.c_int, [_P]),
    "ss_get_kernel_stats": (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_int)]),
    "ss_last_run_device_ms": (C.c_double, [_P]),
    "ss_workspace_bytes": (C.c_int64, [_P]),
}
EXPORTS = tuple(_SIGS)
# exported by the development build only (libsoftspoken_hip_dev.so, loaded through SOFTSPOKEN_LIB by tests and tools)
_DEV_SIGS = {
    "ss_debug_fail_workspace_alloc": (C.c_int, [_P, C.c_int]),
}

_lib = None


class NativeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libsoftspoken_hip: status {code}: {msg}")
        self.code = code


def lib():
    """Load the shared library (once).  Raises if it has not been built: no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found - build it with `python -m softspoken_amd.build` "
                              "(the voice-detector path has no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        for name, (res, args) in _DEV_SIGS.items():
            fn = getattr(L, name, None)
            if fn is not None:
                fn.restype, fn.argtypes = res, args
        if L.ss_abi_version() != ABI_VERSION:
            raise ImportError(f"{LIB_PATH}: ABI version {L.ss_abi_version()}, this binding is for {ABI_VERSION} -- a stale build "
                              "(python -m softspoken_amd.build --force)")
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _check(rc, ctx=None):
    if rc != SS_OK:
        msg = lib().ss_last_error(ctx)
        raise NativeError(rc, msg.decode("utf-8", "replace") if msg else "")


# ---- host-only helpers ----------------------------------------------------------------------------
def wav_parse(buf) -> WavInfo:
    a = np.frombuffer(buf, dtype=np.uint8)
    info = WavInfo()
    _check(lib().ss_wav_parse(_ptr(a), a.size, C.byref(info)))
    return info


def plan_windows(duration_s: float) -> np.ndarray:
    n = lib().ss_plan_windows(float(duration_s), None, 0)
    out = np.zeros(max(n, 0), dtype=np.int64)
    if n > 0:
        lib().ss_plan_windows(float(duration_s), _ptr(out), n)
    return out


def find_regions(avg, bin_idx, threshold=0.1, break_s=0.5):
    avg = np.ascontiguousarray(avg, dtype=np.float64)
    bin_idx = np.ascontiguousarray(bin_idx, dtype=np.int64)
    cap = len(avg) // 2 + 1
    out = (Region * cap)()
    n = C.c_int64(0)
    _check(lib().ss_find_regions(_ptr(avg), _ptr(bin_idx), len(avg), threshold, break_s, out, cap, C.byref(n)))
    return [(out[i].start, out[i].end) for i in range(n.value)]


def format_csv_rows(file_path: str, file_name: str, regions, first_id: int = 1) -> str:
    arr = (Region * max(1, len(regions)))()
    for i, (s, e) in enumerate(regions):
        arr[i].start, arr[i].end = float(s), float(e)
    fp, fn = file_path.encode(), file_name.encode()
    need = lib().ss_format_csv_rows(fp, fn, arr, len(regions), first_id, None, 0)
    buf = C.create_string_buffer(need + 1)
    lib().ss_format_csv_rows(fp, fn, arr, len(regions), first_id, buf, need + 1)
    return buf.value.decode()


def wav_header_pcm16(sr: int, channels: int, frames: int) -> bytes:
    buf = C.create_string_buffer(44)
    _check(lib().ss_wav_header_pcm16(int(sr), int(channels), int(frames), buf))
    return buf.raw


# ---- context --------------------------------------------------------------------------------------
class Context:
    """One detector context on one GPU (not thread-safe; one per device)."""

    def __init__(self, blob: bytes | np.ndarray | None, device: int = 0, bf16: bool = False, profile: bool = False,
                 chunk: int | None = None, precision: str | None = None):
        """blob None -> audio-only context (decode / mixdown / resample; model calls raise).
        precision: "fp32" (fp32 matrix instructions, exact fp32 FMA chains), "f16x2" (fp32-accurate on the f16 matrix cores: operands
        split into two f16 halves, three products per term) or "bf16" (throughput mode, scores differ by up to ~0.1); bf16=True is
        the older spelling of precision="bf16"."""
        L = lib()
        b = np.frombuffer(blob, dtype=np.uint8) if blob is not None else None
        self._h = C.c_void_p()
        precision = precision or ("bf16" if bf16 else "fp32")
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {PRECISIONS}, not {precision!r}")
        bf16 = precision == "bf16"
        self.precision = precision
        flags = (FLAG_BF16 if bf16 else 0) | (FLAG_F16X2 if precision == "f16x2" else 0) | (FLAG_PROFILE if profile else 0)
        rc = L.ss_create(int(device), _ptr(b), b.size if b is not None else 0, flags, C.byref(self._h))
        if rc != SS_OK:
            self._h = C.c_void_p()
            _check(rc)
        self.bf16 = bf16
        self._cb_keep = None
        if chunk:
            self.set_chunk(chunk)

    @property
    def alive(self) -> bool:
        """False once close() has destroyed the native context."""
        return bool(getattr(self, "_h", None) and self._h.value)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().ss_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def _ck(self, rc):
        _check(rc, self._h)

    def set_chunk(self, n):
        self._ck(lib().ss_set_chunk_windows(self._h, int(n)))

    def reset(self):
        self._ck(lib().ss_reset(self._h))

    def reset_generation(self) -> int:
        """Number of reset() calls so far: file ids handed out before the last one no longer name the caller's files."""
        return int(lib().ss_reset_generation(self._h))

    @staticmethod
    def _need_bytes(pcm: np.ndarray, fmt: int, channels: int, frames) -> None:
        """The C ABI copies frames * channels * bytes_per_sample from the pointer: refuse a shorter buffer here (ValueError),
        where the library would read past its end."""
        if fmt not in _BPS:
            raise ValueError(f"unknown PCM format code {fmt}")
        need = int(np.sum(np.asarray(frames, dtype=np.int64))) * int(channels) * _BPS[fmt]
        if int(channels) < 1 or need < 0 or pcm.nbytes < need:
            raise ValueError(f"PCM buffer holds {pcm.nbytes} bytes, {need} needed for {frames} frames x {channels} channels")

    def add_pcm(self, pcm: np.ndarray, fmt: int, sr: int, channels: int, frames: int) -> int:
        pcm = np.ascontiguousarray(pcm)
        self._need_bytes(pcm, fmt, channels, frames)
        fid = C.c_int(-1)
        self._ck(lib().ss_add_pcm(self._h, _ptr(pcm), fmt, sr, channels, frames, C.byref(fid)))
        return fid.value

    def silence_pcm(self, pcm: np.ndarray, fmt: int, sr: int, channels: int, frames: int, regions) -> np.ndarray:
        """Interleaved int16 (frames, channels) with the (start_s, end_s) regions zeroed."""
        pcm = np.ascontiguousarray(pcm)
        self._need_bytes(pcm, fmt, channels, frames)
        arr = (Region * max(1, len(regions)))()
        for i, (s, e) in enumerate(regions):
            arr[i].start, arr[i].end = float(s), float(e)
        out = np.empty((frames, channels), dtype=np.int16)
        self._ck(lib().ss_silence_pcm(self._h, _ptr(pcm), fmt, sr, channels, frames, arr, len(regions), _ptr(out)))
        return out

    def stft512_magnitude(self, samples) -> np.ndarray:
        """|STFT| (n_fft = win = 512, hop 256, centred, zero padded) of a mono float32 signal -> float32 [257, 1 + n // 256]."""
        x = np.ascontiguousarray(samples, dtype=np.float32).reshape(-1)
        nf = int(lib().ss_stft512_frames(x.size))
        out = np.empty((257, nf), dtype=np.float32)
        self._ck(lib().ss_stft512_magnitude(self._h, _ptr(x), x.size, _ptr(out), nf))
        return out

    def add_pcm_device(self, dev_ptr: int, fmt: int, sr: int, channels: int, frames: int) -> int:
        fid = C.c_int(-1)
        self._ck(lib().ss_add_pcm_device(self._h, C.c_void_p(dev_ptr), fmt, sr, channels, frames, C.byref(fid)))
        return fid.value

    def add_pcm_batch_device(self, dev_ptr: int, fmt: int, sr: int, channels: int, frames, host_copy: np.ndarray | None = None) -> int:
        """frames[i] frames per file, files back to back in the device buffer (host_copy: the array that was uploaded there, if
        the caller still has it -- its size is then checked against the frame counts)."""
        fr = np.ascontiguousarray(frames, dtype=np.int64)
        if host_copy is not None:
            self._need_bytes(np.asarray(host_copy), fmt, channels, fr)
        fid = C.c_int(-1)
        self._ck(lib().ss_add_pcm_batch_device(self._h, C.c_void_p(dev_ptr), fmt, sr, channels, _ptr(fr), len(fr), C.byref(fid)))
        return fid.value

    def add_wav_bytes(self, buf) -> tuple[int, WavInfo]:
        info = wav_parse(buf)
        a = np.frombuffer(buf, dtype=np.uint8, count=info.data_bytes, offset=info.data_offset)
        return self.add_pcm(a, info.format, info.sample_rate, info.channels, info.frames), info

    def add_f32_22k(self, x: np.ndarray, padded: bool = False) -> int:
        x = np.ascontiguousarray(x, dtype=np.float32)
        fid = C.c_int(-1)
        fn = lib().ss_add_padded_f32_22k if padded else lib().ss_add_f32_22k
        self._ck(fn(self._h, _ptr(x), x.size, C.byref(fid)))
        return fid.value

    def signal_length(self, fid: int, padded: bool = False) -> int:
        return lib().ss_signal_length(self._h, fid, int(padded))

    def read_signal(self, fid: int, padded: bool = False) -> np.ndarray:
        n = self.signal_length(fid, padded)
        out = np.empty(n, dtype=np.float32)
        self._ck(lib().ss_read_signal(self._h, fid, int(padded), 0, n, _ptr(out)))
        return out

    def device_alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self._ck(lib().ss_device_alloc(self._h, nbytes, C.byref(p)))
        return p.value

    def device_free(self, p: int):
        self._ck(lib().ss_device_free(self._h, C.c_void_p(p)))

    def device_upload(self, dst: int, src: np.ndarray):
        src = np.ascontiguousarray(src)
        self._ck(lib().ss_device_upload(self._h, C.c_void_p(dst), _ptr(src), src.nbytes))

    # ---- ingest: the next job's files cross PCIe on the copy stream while the job in flight computes -------------------
    def host_alloc(self, nbytes: int) -> np.ndarray:
        """Page-locked host memory as a uint8 array (free it with host_free; it does not free itself)."""
        p = C.c_void_p()
        self._ck(lib().ss_host_alloc(self._h, int(nbytes), C.byref(p)))
        buf = (C.c_uint8 * int(nbytes)).from_address(p.value)
        a = np.frombuffer(buf, dtype=np.uint8)
        return a

    def host_free(self, a: np.ndarray):
        self._ck(lib().ss_host_free(self._h, C.c_void_p(a.ctypes.data)))

    def upload_wav_batch_async(self, files, dev_dst: int, cap: int):
        """files: uint8 arrays holding RIFF/WAVE images (page-locked ones copy asynchronously).  Header walk of each + one
        asynchronous copy per file of its samples, back to back from dev_dst, on the copy stream -> list of WavInfo.  The arrays
        must stay alive and untouched until upload_wait() / sync(); the next add_pcm*_device waits for the copies on the device."""
        n = len(files)
        ptrs = (C.c_void_p * n)(*[f.ctypes.data for f in files])
        sizes = (C.c_size_t * n)(*[f.nbytes for f in files])
        infos = (WavInfo * n)()
        self._ck(lib().ss_upload_wav_batch_async(self._h, ptrs, sizes, n, C.c_void_p(dev_dst), int(cap), infos))
        self._upload_keep = (files, ptrs, sizes)
        return infos

    def device_upload_async(self, dst: int, src: np.ndarray):
        src = np.ascontiguousarray(src)
        self._upload_keep = src
        self._ck(lib().ss_device_upload_async(self._h, C.c_void_p(dst), _ptr(src), src.nbytes))

    def upload_wait(self):
        self._ck(lib().ss_upload_wait(self._h))
        self._upload_keep = None

    def features(self, fid: int, starts, discard: bool = False):
        s = np.ascontiguousarray(starts, dtype=np.int64)
        out = None if discard else np.empty((len(s), 128, 256), dtype=np.float32)
        self._ck(lib().ss_features(self._h, fid, _ptr(s), len(s), _ptr(out)))
        return out

    def infer_windows(self, fid: int, starts, want_spec: bool = False):
        s = np.ascontiguousarray(starts, dtype=np.int64)
        mask = np.empty((len(s), 1, 256), dtype=np.float32)
        spec = np.empty((len(s), 2, 128, 256), dtype=np.float32) if want_spec else None
        self._ck(lib().ss_infer_windows(self._h, fid, _ptr(s), len(s), _ptr(mask), _ptr(spec)))
        return spec, mask

    def run(self, threshold: float = 0.1, break_s: float = 0.5, progress=None, stop_flag=None):
        """progress(done, total) is called between chunks; stop_flag: ctypes.c_int polled between chunks."""
        cb = None
        if progress is not None:
            cb = PROGRESS_FN(lambda _u, d, t: progress(d, t))
        self._cb_keep = cb
        rc = lib().ss_run(self._h, threshold, break_s, C.cast(cb, C.c_void_p) if cb else None, None,
                          C.cast(C.byref(stop_flag), C.c_void_p) if stop_flag is not None else None)
        self._cb_keep = None
        if rc == SS_ERR_STOPPED:
            return False
        self._ck(rc)
        return True

    def run_begin(self, threshold: float = 0.1, break_s: float = 0.5, track: bool = False):
        """First half of run(): plan and enqueue; returns while the device works.  The context takes no other work until run_end().
        track: an event behind every pass, for run_poll()."""
        fn = lib().ss_run_begin_tracked if track else lib().ss_run_begin
        self._ck(fn(self._h, threshold, break_s))

    def run_poll(self, progress=None, block: bool = True):
        """Progress of the run started with run_begin(track=True): progress(done, total) for every value of the reference's sequence
        (32, 64, ..., total windows) that has completed since the last call; block: wait for all of them."""
        cb = PROGRESS_FN(lambda _u, d, t: progress(d, t)) if progress is not None else None
        self._ck(lib().ss_run_poll(self._h, C.cast(cb, C.c_void_p) if cb else None, None, int(bool(block))))

    def run_end(self):
        """Second half of run(): wait for the device, find the regions."""
        self._ck(lib().ss_run_end(self._h))

    def run_from_logits(self, logits: np.ndarray, threshold: float = 0.1, break_s: float = 0.5):
        """The tail of run() on per-window logits computed elsewhere (window ranges of one recording on several GPUs):
        logits [total windows of the files added since reset()][256]."""
        lg = np.ascontiguousarray(logits, dtype=np.float32).reshape(-1, 256)
        self._ck(lib().ss_run_from_logits(self._h, _ptr(lg), lg.shape[0], threshold, break_s))

    def debug_fail_workspace_alloc(self, nth: int):
        """Development build only (SOFTSPOKEN_LIB=libsoftspoken_hip_dev.so)."""
        self._ck(lib().ss_debug_fail_workspace_alloc(self._h, int(nth)))

    def workspace_bytes(self) -> int:
        return int(lib().ss_workspace_bytes(self._h))

    def num_windows(self, fid: int) -> int:
        return lib().ss_num_windows(self._h, fid)

    def window_logits(self, fid: int) -> np.ndarray:
        w = self.num_windows(fid)
        out = np.empty((w, 1, 256), dtype=np.float32)
        self._ck(lib().ss_get_window_logits(self._h, fid, _ptr(out), w))
        return out

    def avg(self, fid: int):
        n = C.c_int64(0)
        self._ck(lib().ss_get_avg(self._h, fid, None, None, 0, C.byref(n)))
        a = np.empty(n.value, dtype=np.float64)
        idx = np.empty(n.value, dtype=np.int64)
        self._ck(lib().ss_get_avg(self._h, fid, _ptr(a), _ptr(idx), n.value, C.byref(n)))
        return a, idx

    def regions(self, fid: int):
        n = C.c_int64(0)
        self._ck(lib().ss_get_regions(self._h, fid, None, 0, C.byref(n)))
        arr = (Region * max(1, n.value))()
        self._ck(lib().ss_get_regions(self._h, fid, arr, n.value, C.byref(n)))
        return [(arr[i].start, arr[i].end) for i in range(n.value)]

    def regions_batch(self, first: int, n_files: int):
        """-> (counts int64[n_files], regions float64[total, 2]) for files first .. first + n_files - 1, one foreign call each way."""
        n = C.c_int64(0)
        self._ck(lib().ss_get_regions_batch(self._h, first, n_files, None, None, 0, C.byref(n)))
        counts = np.zeros(max(n_files, 1), dtype=np.int64)
        out = np.zeros((max(n.value, 1), 2), dtype=np.float64)
        self._ck(lib().ss_get_regions_batch(self._h, first, n_files, _ptr(counts), _ptr(out), n.value, C.byref(n)))
        return counts[:n_files], out[:n.value]

    def sync(self):
        self._ck(lib().ss_sync(self._h))

    def reset_stats(self):
        self._ck(lib().ss_reset_kernel_stats(self._h))

    def kernel_stats(self):
        n = C.c_int(0)
        self._ck(lib().ss_get_kernel_stats(self._h, None, 0, C.byref(n)))
        arr = (KernelStat * max(1, n.value))()
        self._ck(lib().ss_get_kernel_stats(self._h, arr, n.value, C.byref(n)))
        return [dict(name=arr[i].name.decode(), launches=arr[i].launches, total_ms=arr[i].total_ms, flops=arr[i].flops,
                     bytes=arr[i].bytes, issued_flops=arr[i].issued_flops) for i in range(n.value)]

    def last_run_device_ms(self) -> float:
        return lib().ss_last_run_device_ms(self._h)
