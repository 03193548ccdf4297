"""Constants of the voice-detector path.

Same names and values as the reference's root/code/backend/settings.py:1-32 (every module of the
path reads them by these names); only the two directory defaults are spelled with os.path.join
so that they also resolve on Linux (the reference hard-codes Windows back-slashes, which makes its
checkpoint lookup silently fail there - SURVEY.md section 5).
"""
import os

# ---- STFT (the model's front-end uses n_fft * 4 = 2048 with a 512-sample window, hop 256) ----
n_fft = 512
win_length = n_fft
hop_length = win_length // 2

# ---- sliding 3 s windows advance by this many seconds ----
step_size = 0.6

# ---- inference batching and decision threshold (applied to overlap-averaged raw logits) ----
prediction_batch_size = 32
threshold = 0.1

# ---- internal sample rate of the whole application ----
vad_resample = 22050

# ---- model checkpoint ----
model_dir = os.path.join('.', 'root', 'models', 'spec_unet_2d_pytorch')
model_name = 'model_checkpoint.pth'

# ---- project files ----
project_dir = os.path.join('.', 'projects')

# ---- review screen hides detections shorter than this (seconds) ----
minimum_detection_len = 0.1

user_guide_url = 'https://github.com/AVianEco/Softspoken'

# ---- host threads: the reference uses half the cores for torch CPU inference; the MI355X path only
#      uses this for host-side bookkeeping, the name is kept for compatibility ----
cpu_threads = max(1, (os.cpu_count() or 2) // 2)

# ---- knobs of the MI355X build (not in the reference) ----
# windows per pass through the conv stack; 0 = library default
hip_chunk_windows = int(os.environ.get('SOFTSPOKEN_CHUNK', '0') or 0)
# conv stack arithmetic.  Both parity modes keep the scores within 1e-4 of the reference's fp32 CPU path:
#   'f16x2'  fp32 operands as two f16 halves on the f16 matrix cores, three products per term, fp32 accumulation (default: ~2x 'fp32')
#   'fp32'   fp32 operands on the fp32 matrix instructions (exact fp32 FMA chains; also for activations beyond the f16 range, 65504)
#   'bf16'   throughput mode: scores differ from the reference by up to ~0.1, region boundaries by a bin or two
hip_precision = os.environ.get('SOFTSPOKEN_PRECISION', 'f16x2')
# device contexts ProcessWorker.run alternates its files between (2: the next file is queued on the device before the current one ends;
# each context holds its own activation workspace -- 44 GB for a 10-minute file in the parity modes --; 1: one context, as round 2)
hip_file_contexts = 2
# load-time check of the f16x2 mode on the checkpoint actually loaded: a handful of fixed windows through f16x2 and through fp32;
# beyond 5e-5 the detector logs once and keeps fp32 for those weights (SpecUNet_2D._selfcheck; < 1 s per set of weights)
hip_selfcheck = True
