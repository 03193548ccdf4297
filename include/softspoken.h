/*
 * softspoken.h -- C ABI of libsoftspoken_hip.so: the MI355X (gfx950) implementation of Softspoken's
 * "Run Voice Detector" hot path.  Plain pointers and sizes only; no torch / numpy types.
 *
 * The reference (AVianEco/Softspoken) has no FFI: its boundary is a Python surface.  Each entry
 * point below names the reference code it stands behind (paths relative to the reference root):
 *
 *   ss_wav_parse, ss_resampled_length      root/code/backend/voice_activity.py:23-30   get_audio_data
 *   ss_plan_windows                        root/code/frontend/NNDetector.py:55-82      plan_detection_job
 *   ss_create / ss_destroy                 NNDetector.py:21-34,42-53                   model build + load_checkpoint
 *   ss_upload_wav_batch_async              root/code/backend/worker.py:57 -> voice_activity.py:37 (sf.read of every file of the job)
 *   ss_add_pcm / ss_add_pcm_device /       voice_activity.py:32-69 load_audio  +  root/code/backend/worker.py:58-62 (3 s pad)
 *   ss_add_pcm_batch_device /
 *   ss_add_f32_22k / ss_add_padded_f32_22k
 *   ss_read_signal                         (returns what load_audio returns: float32 @ 22050 Hz)
 *   ss_features                            root/code/backend/pytorch_neural_nets.py:92-99,144-153 (mel front-end)
 *   ss_infer_windows                       NNDetector.py:84-101 process_batch -> SpecUNet_2D.forward (pytorch_neural_nets.py:142-197)
 *   ss_run (ss_run_begin + ss_run_end)     worker.py:49-100 (per-file loop: batches, averaging, regions, -3 s)
 *   ss_get_avg                             NNDetector.py:153-190 average_overlapping_detections
 *   ss_get_regions / ss_find_regions       NNDetector.py:103-143 find_speech_regions + worker.py:100
 *   ss_format_csv_rows                     worker.py:103-125 + root/code/frontend/silencer_ui.py:816-817 (DataFrame.to_csv text)
 *
 * Conventions: every function returns an int status (SS_OK == 0) unless stated; the caller owns all
 * host buffers passed in and they only need to stay alive for the duration of the call; the library
 * owns all device memory.  One context per GPU; a context is NOT thread-safe, different contexts
 * are independent.  Calls may come from any host thread (the reference calls from a QThreadPool
 * thread, silencer_ui.py:243).  Unlike the reference, decode / IO / device failures are reported,
 * never swallowed (voice_activity.py:39-41 returns (None, None) and the worker then crashes).
 */
#ifndef SOFTSPOKEN_H
#define SOFTSPOKEN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SS_ABI_VERSION 3   /* 2: ss_kernel_stat.name[128], SS_ERR_RANGE, ingest (ss_host_alloc, ss_upload_wav_batch_async, ...); 3: ss_kernel_stat.issued_flops */

/* fixed properties of the path (reference settings.py:4-16, NNDetector.py:69-75) */
#define SS_SAMPLE_RATE 22050
#define SS_WINDOW_SAMPLES 66150   /* 3 s */
#define SS_STEP_SAMPLES 13230     /* floor(22050 * 0.6) */
#define SS_N_MELS 128
#define SS_N_FRAMES 256           /* time bins per window */

typedef struct ss_ctx ss_ctx;

enum ss_status {
    SS_OK = 0,
    SS_ERR_ARG = 1,      /* bad argument */
    SS_ERR_HIP = 2,      /* HIP runtime / kernel failure (message has the HIP error string) */
    SS_ERR_FORMAT = 3,   /* malformed weights blob or WAV */
    SS_ERR_STATE = 4,    /* call out of order (e.g. ss_run before any ss_add_*) */
    SS_ERR_STOPPED = 5,  /* stop flag observed; partial results of the current run are discarded */
    SS_ERR_NOMEM = 6,
    SS_ERR_CAPACITY = 7, /* caller's output buffer too small; required size is reported */
    SS_ERR_RANGE = 8     /* SS_FLAG_F16X2 only: a weight or an activation left the f16 range (|x| > 65504 after the build's power-of-two
                            channel normalisation) or was not finite; the results of the call are not to be used -- run this checkpoint
                            in the fp32 mode (flags without a precision bit).  The drop-in does that by itself (NNDetector.detect_files) */
};

enum ss_pcm_format {     /* sample encodings, interleaved: 1-6 = the WAV data chunk's (little endian); 7-12 = the AIFF / AIFF-C sound chunk's
                          * (big endian, 8-bit samples signed), round 4: libsndfile, which the reference reads files with
                          * (voice_activity.py:37), converts all of them to float the same way, x / 2^(bits-1) */
    SS_PCM_U8 = 1, SS_PCM_S16 = 2, SS_PCM_S24 = 3, SS_PCM_S32 = 4, SS_PCM_F32 = 5, SS_PCM_F64 = 6,
    SS_PCM_S8 = 7, SS_PCM_S16BE = 8, SS_PCM_S24BE = 9, SS_PCM_S32BE = 10, SS_PCM_F32BE = 11, SS_PCM_F64BE = 12
};

enum ss_flags {
    SS_FLAG_BF16 = 1u,       /* conv stack in bf16 with fp32 accumulation (throughput mode: scores differ from the reference by up to ~0.1) */
    SS_FLAG_PROFILE = 2u,    /* time every kernel launch with HIP events (ss_get_kernel_stats) */
    SS_FLAG_F16X2 = 4u       /* conv stack on the f16 matrix cores with every fp32 operand split into two f16 halves (x = hi + lo; three
                                products per term: w_hi x_hi + w_hi x_lo + w_lo x_hi, fp32 accumulation): scores within 1e-4 of the
                                reference's fp32 like the default, at 2.8 x its speed.  Every tensor is kept near 1 by an exact power-of-two
                                channel normalisation chosen from the weights at ss_create, so a checkpoint's BatchNorm gains do not decide
                                whether it fits; a value that still has no f16 representation (not finite, or beyond 65504 in normalised
                                units) is reported as SS_ERR_RANGE.  Default (no precision flag): fp32 operands on the fp32 matrix
                                instructions, an exact fp32 FMA chain */
};

typedef struct ss_wav_info {
    int32_t format;        /* enum ss_pcm_format */
    int32_t channels;
    int32_t sample_rate;
    int32_t bits;
    int64_t frames;
    int64_t data_offset;   /* byte offset of the first sample in the file */
    int64_t data_bytes;
} ss_wav_info;

typedef struct ss_region {  /* seconds relative to the start of the file (the reference's "-3" already applied) */
    double start;
    double end;
} ss_region;

typedef struct ss_kernel_stat {
    char name[128];        /* "<kernel instantiation as rocprofv3 prints it>/<layer>" */
    int64_t launches;
    double total_ms;        /* sum of HIP-event durations (SS_FLAG_PROFILE only) */
    double flops;           /* algorithmic FLOPs summed over launches (0 for byte-bound kernels) */
    double bytes;           /* algorithmic bytes summed over launches */
    double issued_flops;    /* FLOPs the matrix pipe was actually given: 2 x (products issued per multiply-add: 3 in f16x2, 1 otherwise) x the
                             * multiply-adds of the form that ran (the sub-pixel launches run 4 taps instead of 9 on their upsampled half) */
} ss_kernel_stat;

/* progress callback: done/total windows of the current run, called on the calling thread with the reference's sequence of values
 * -- done = 32, 64, ... (settings.prediction_batch_size; it emits after each batch, worker.py:82-84), then the total --.  The
 * passes through the network keep their full size (ss_set_chunk_windows): the values that fall into a pass are reported, in
 * order, when that pass has completed on the device. */
typedef void (*ss_progress_fn)(void* user, int64_t windows_done, int64_t windows_total);

/* ---- host-only helpers (no GPU needed) ----------------------------------------------------- */
int ss_abi_version(void);
/* last error message of the calling thread for calls that have no context (or ctx == NULL) */
const char* ss_last_error(const ss_ctx* ctx);

/* Walk a RIFF/WAVE image (PCM 8/16/24/32, IEEE float 32/64, WAVE_FORMAT_EXTENSIBLE) or, since round 4, an AIFF / AIFF-C one ("FORM",
 * COMM + SSND chunks; big-endian PCM 8/16/24/32, AIFF-C "NONE" / "sowt" (little-endian PCM) / "fl32" / "fl64"): the containers
 * soundfile.read hands the reference as float32 (voice_activity.py:37).  Other containers (FLAC, OGG) are reported as SS_ERR_FORMAT. */
int ss_wav_parse(const void* file_bytes, size_t nbytes, ss_wav_info* out);
/* ceil(frames * 22050 / sample_rate): length of the resampled signal (librosa.resample's rule). */
int64_t ss_resampled_length(int64_t frames, int sample_rate);
/* Window start table of one file.  Returns the number of windows W (also when starts == NULL);
 * writes min(W, cap) entries.  starts[i] = i * 13230 into the 3 s-padded signal. */
int64_t ss_plan_windows(double duration_s, int64_t* starts, int64_t cap);
/* Threshold + run-length + gap merge on averaged logits (double), exactly as the reference does it
 * through "%.4f" time strings; bin_idx[i] is the bin number of avg[i].  Returns regions in seconds
 * minus 3.  *n_out receives the number found; SS_ERR_CAPACITY if it exceeds cap. */
int ss_find_regions(const double* avg, const int64_t* bin_idx, int64_t n, double threshold, double break_s,
                    ss_region* out, int64_t cap, int64_t* n_out);
/* CSV body lines (no header) for `n` regions of one file, first ID = first_id, DataFrame.to_csv text.
 * Returns bytes needed (excluding NUL); writes at most cap bytes. */
int64_t ss_format_csv_rows(const char* file_path, const char* file_name, const ss_region* regions, int64_t n,
                           int64_t first_id, char* out, int64_t cap);

/* ---- context -------------------------------------------------------------------------------- */
/* weights_blob == NULL creates an audio-only context (ss_add_pcm / ss_read_signal work, model calls
 * return SS_ERR_STATE).  weights_blob: "SSWBLOB1" container of the checkpoint's state_dict tensors (packed by
 * softspoken_amd.checkpoint.pack_state_dict).  BatchNorm folding and MFMA fragment packing happen here. */
int ss_create(int device_id, const void* weights_blob, size_t nbytes, uint32_t flags, ss_ctx** out);
void ss_destroy(ss_ctx* ctx);
/* windows processed per pass through the conv stack (activation workspace is sized for this) */
int ss_set_chunk_windows(ss_ctx* ctx, int chunk);

/* ---- signal arena: files of the current job, resident in HBM ------------------------------- */
int ss_reset(ss_ctx* ctx);
/* counts the ss_reset calls of this context: a caller that caches file ids compares it to know whether they still name its files */
uint64_t ss_reset_generation(ss_ctx* ctx);
/* Decode + mixdown + resample + 3 s pad on the device.  pcm: interleaved samples (host memory). */
int ss_add_pcm(ss_ctx* ctx, const void* pcm, int format, int sample_rate, int channels, int64_t frames, int* file_id);
/* Same, but `pcm_dev` already is device memory of this GPU (bench: inputs resident in HBM). */
int ss_add_pcm_device(ss_ctx* ctx, const void* pcm_dev, int format, int sample_rate, int channels, int64_t frames,
                      int* file_id);
/* `n_files` recordings of one format stored back to back in one device buffer: decode + resample for
 * the whole batch in two launches.  File ids are first_file_id .. first_file_id + n_files - 1. */
int ss_add_pcm_batch_device(ss_ctx* ctx, const void* pcm_dev, int format, int sample_rate, int channels,
                            const int64_t* frames, int n_files, int* first_file_id);
/* A signal that already is mono float32 at 22 050 Hz (the parity boundary). Pads 3 s each side. */
int ss_add_f32_22k(ss_ctx* ctx, const float* samples, int64_t n, int* file_id);
/* A signal stored as is, no padding added (what worker.py hands to process_batch; also any buffer of
 * back-to-back windows).  Its header duration is taken as (n - 6 s) / 22050, clamped at 0. */
int ss_add_padded_f32_22k(ss_ctx* ctx, const float* padded, int64_t n, int* file_id);
int64_t ss_signal_length(ss_ctx* ctx, int file_id, int padded);
int ss_read_signal(ss_ctx* ctx, int file_id, int padded, int64_t offset, int64_t n, float* out);
/* Silencer (silencer_ui.py:974-998 SilenceWorker.run): decode the interleaved samples
 * to float32 as the loader does, zero frames [round(start*sr), round(end*sr)) of every region (Python
 * round, clamped to the file; regions may overlap, be unsorted or lie outside the file), and return
 * interleaved 16-bit PCM -- lrintf(x * 32767), no clipping: what the reference's sf.write(path, data, sr)
 * stores, soundfile's default WAV subtype being PCM_16.  `out` holds frames*channels int16.  Any context
 * will do (audio-only included). */
int ss_silence_pcm(ss_ctx* ctx, const void* pcm, int format, int sample_rate, int channels, int64_t frames,
                   const ss_region* regions, int64_t n_regions, int16_t* out);
/* Review-screen spectrogram (voice_activity.py:148-154 wav_to_spec): magnitude of the STFT with
 * n_fft = win_length = 512 (settings.py:4-6), hop 256, periodic Hann, centred frames over zero padding (librosa.stft's
 * defaults).  out is [257][frames] float32, frames = ss_stft512_frames(n) = 1 + n / 256.  Any context will do. */
int64_t ss_stft512_frames(int64_t n_samples);
int ss_stft512_magnitude(ss_ctx* ctx, const float* samples, int64_t n_samples, float* out, int64_t cap_frames);
/* The 44-byte RIFF/WAVE header that goes in front of ss_silence_pcm's output.  Host only. */
int ss_wav_header_pcm16(int sample_rate, int channels, int64_t frames, void* out44);
/* device allocation helpers so a host without its own HIP binding can stage inputs in HBM (what is still allocated when the context
 * is destroyed is freed with it) */
int ss_device_alloc(ss_ctx* ctx, size_t nbytes, void** dev_ptr);
int ss_device_free(ss_ctx* ctx, void* dev_ptr);
int ss_device_upload(ss_ctx* ctx, void* dev_dst, const void* host_src, size_t nbytes);
/* ---- ingest: the job's WAV files from host memory into HBM, beside the kernels of the job before ----------------------
 * (worker.py:57 reads and decodes one file at a time in front of its batches; here the bytes of job k + 1 cross PCIe on the context's
 * copy stream while job k's kernels run on its compute stream.)
 * Page-locked host memory (hipHostMalloc): copies from it are asynchronous and run at the link's rate. */
int ss_host_alloc(ss_ctx* ctx, size_t nbytes, void** host_ptr);
int ss_host_free(ss_ctx* ctx, void* host_ptr);
/* n_files RIFF/WAVE images in host memory: the header walk of every file (ss_wav_parse -> infos[i]) and one asynchronous
 * host -> device copy per file of exactly frames * channels * bytes-per-sample bytes of its data chunk, back to back from
 * dev_dst (cap bytes; SS_ERR_CAPACITY when they do not fit), on the copy stream.  Returns when the copies are enqueued: the
 * file images must stay untouched until ss_upload_wait or the next ss_sync.  Allowed while a run is in flight (it touches
 * neither the signal arena nor the workspace).  A following ss_add_pcm_device / ss_add_pcm_batch_device on this context waits
 * for the copies ON THE DEVICE (an event between the two streams), not on the host.  The caller alternates two staging buffers:
 * the decode kernels of the job in flight read the other one. */
int ss_upload_wav_batch_async(ss_ctx* ctx, const void* const* files, const size_t* nbytes, int n_files, void* dev_dst, size_t cap,
                              ss_wav_info* infos);
/* the same for raw bytes (no header walk) */
int ss_device_upload_async(ss_ctx* ctx, void* dev_dst, const void* host_src, size_t nbytes);
/* host-side wait for every copy enqueued so far on the copy stream */
int ss_upload_wait(ss_ctx* ctx);

/* ---- compute -------------------------------------------------------------------------------- */
/* Mel front-end only: feat_out[n][128][256] float32 for windows starting at starts[i] (padded-signal index).
 * feat_out == NULL runs the kernels and discards the result (front-end timing). */
int ss_features(ss_ctx* ctx, int file_id, const int64_t* starts, int n, float* feat_out);
/* process_batch: mask_out[n][256] raw logits; spec_out (nullable) [n][2][128][256]. Any n >= 1. */
int ss_infer_windows(ss_ctx* ctx, int file_id, const int64_t* starts, int n, float* mask_out, float* spec_out);
/* Whole job over every file added since ss_reset: plan windows from each file's header duration,
 * front-end + conv stack over all windows in chunks (across file boundaries), overlap averaging on
 * the device, region finding on the host.  stop_flag (nullable) is polled between chunks. */
int ss_run(ss_ctx* ctx, double threshold, double break_s, ss_progress_fn progress, void* user,
           const volatile int* stop_flag);
/* The same job in two halves (worker.py:49-100 has no counterpart: its loop is synchronous).  ss_run_begin plans and enqueues
 * everything up to the last device -> host copy and returns; ss_run_end waits for it and finds the regions.  Between the two
 * the context accepts no ss_reset / ss_add_* / compute call (SS_ERR_STATE).  A caller with two contexts on one device alternates
 * them, so that one job's host half runs while the other job's kernels do: ss_run == ss_run_begin + ss_run_end. */
int ss_run_begin(ss_ctx* ctx, double threshold, double break_s);
int ss_run_end(ss_ctx* ctx);
/* ss_run_begin with an event behind every pass, and the progress of the run in flight read from them: ss_run_poll calls
 * `progress` with the values of ss_progress_fn's sequence that have completed since the last call -- block != 0: waits for all of
 * them --.  A caller that keeps the device busy with file k + 1 while it files the rows of file k reports k + 1's progress from
 * here when its turn comes (root/code/backend/worker.py).  ss_run(progress) == ss_run_begin_tracked + ss_run_poll(block) + ss_run_end. */
int ss_run_begin_tracked(ss_ctx* ctx, double threshold, double break_s);
int ss_run_poll(ss_ctx* ctx, ss_progress_fn progress, void* user, int block);
/* The tail of ss_run for files whose per-window logits were computed elsewhere -- a long recording whose window ranges ran on
 * several GPUs (windows are independent, NNDetector.py:55-82; averaging needs the neighbours, :168-186, so the
 * logits are gathered to the recording's owner): logits[n_windows][256] for every window of every file added since ss_reset, in
 * file order, as ss_get_window_logits returns them.  Averaging, thresholding and region finding are ss_run's own code, so the table
 * equals that of a one-GPU ss_run bit for bit.  Works on an audio-only context too.  n_windows must equal the plan's total. */
int ss_run_from_logits(ss_ctx* ctx, const float* logits, int64_t n_windows, double threshold, double break_s);
/* Results are those of the last ENDED run; file_id counts that run's files from 0.  The regions (ss_get_regions*, found when first
 * asked for) and ss_num_windows stay readable while the next job is added and in flight, so one context can also overlap the host
 * half of job k with the device half of job k + 1: ss_run_end(k), ss_reset, ss_add_*(k + 1), ss_run_begin(k + 1), then the getters
 * for k.  ss_get_window_logits / ss_get_avg read device buffers that the next ss_run_begin reuses: after it, or after ss_reset,
 * they return SS_ERR_STATE. */
int64_t ss_num_windows(ss_ctx* ctx, int file_id);
int ss_get_window_logits(ss_ctx* ctx, int file_id, float* out, int64_t cap_windows);   /* [W][256] */
/* averaged logits (double) and their bin numbers; returns count via *n_out */
int ss_get_avg(ss_ctx* ctx, int file_id, double* avg, int64_t* bin_idx, int64_t cap, int64_t* n_out);
int ss_get_regions(ss_ctx* ctx, int file_id, ss_region* out, int64_t cap, int64_t* n_out);
/* The same for files [first_file, first_file + n_files) in one call: counts[i] regions of file first_file + i, back to
 * back in out (either may be NULL; *n_out = total). */
int ss_get_regions_batch(ss_ctx* ctx, int first_file, int n_files, int64_t* counts, ss_region* out, int64_t cap, int64_t* n_out);

/* ---- measurement ---------------------------------------------------------------------------- */
/* Enqueue-only variant of ss_run used by bench.py: same work, no host readback until ss_sync. */
int ss_sync(ss_ctx* ctx);
int ss_reset_kernel_stats(ss_ctx* ctx);
int ss_get_kernel_stats(ss_ctx* ctx, ss_kernel_stat* out, int cap, int* n_out);
/* elapsed device time of the last ss_run between its first and last kernel (HIP events on the
 * context's stream), milliseconds */
double ss_last_run_device_ms(ss_ctx* ctx);
/* bytes of device memory the context's activation workspace holds (0 after a failed growth) */
int64_t ss_workspace_bytes(ss_ctx* ctx);

#ifdef SS_DEVBUILD
/* ---- development build only (libsoftspoken_hip_dev.so, -DSS_DEVBUILD): not exported by the product library ----------- */
/* Fault injection for the tests: the nth (0-based) allocation of the next activation-workspace growth fails with an out-of-memory
 * error; nth < 0 switches it off.  The context must come out of such a failure without a workspace (and allocate one afresh on
 * the next call), never with dangling tensors. */
int ss_debug_fail_workspace_alloc(ss_ctx* ctx, int nth);
#endif

#ifdef __cplusplus
}
#endif
#endif /* SOFTSPOKEN_H */
