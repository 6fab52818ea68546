/* pcodec.h -- C ABI of libpcodec.so, the MI355X-native encode/decode hot path of
 * EIDOSLAB/ProgressiveCodec (ChannelProgresssiveWACNN.compress()/decompress()).
 *
 * Plain C: pointers, sizes and integer status codes only; no C++/torch types cross the
 * boundary.  Device pointers are HIP device pointers, `stream` arguments are hipStream_t
 * passed as void* (NULL = default stream).  Every function returns PC_OK (0) or a negative
 * PC_ERR_* code and never aborts.  Functions are re-entrant for distinct objects/buffers.
 *
 * Each entry point names the reference interface it replaces (paths under
 * /root/reference/src/compress).  INTEGRATION.md shows the ctypes binding a maintainer of
 * the reference would add.
 */
#ifndef PCODEC_H
#define PCODEC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define PC_API __attribute__((visibility("default")))
#else
#define PC_API
#endif

enum {
    PC_OK = 0,
    PC_ERR_ARG = -1,        /* invalid argument / unsupported shape (reference: ValueError, entropy_models.py:214-224) */
    PC_ERR_INDEX = -2,      /* CDF index out of range (reference: assert, rans_interface.cpp:110-111) */
    PC_ERR_BUFFER = -3,     /* output buffer too small */
    PC_ERR_TRUNCATED = -4,  /* bitstream shorter than the symbols it should hold */
    PC_ERR_CDF = -5,        /* malformed CDF / pmf (reference: assert, ops.cpp:45; rans_interface.cpp:48-57) */
    PC_ERR_HIP = -6,        /* HIP runtime error (see pc_last_hip_error) */
    PC_ERR_NOMEM = -7,
    PC_ERR_STATE = -8,      /* object not ready (e.g. tables not set: "Uninitialized CDFs. Run update() first"), or busy: a codec object
                               serves one compress / decompress / forward call at a time (use one object per concurrent caller) */
    PC_ERR_MISSING = -9     /* state_dict tensor missing or wrong shape */
};

PC_API const char* pc_version(void);
/* Id of the numeric contract this build codes under (include/pc_math.h: PC_NUMERIC_CONTRACT_ID).  Byte strings are only decodable by a
 * build with the same id -- not by the reference's PyTorch path, nor by a build of another contract revision: the decoder re-derives
 * mu / scale / mask from decoded data, and one differently rounded float desynchronises rANS.  No reference counterpart (the reference
 * has no on-wire format and assumes encoder == decoder process). */
PC_API uint32_t pc_contract_id(void);
PC_API const char* pc_strerror(int code);
PC_API int pc_last_hip_error(void);

/* ---------------------------------------------------------------------------------------
 * Entropy coder (host).  rANS: 64-bit state, 32-bit words, 16-bit precision, 4-bit bypass.
 * ------------------------------------------------------------------------------------- */

/* Upper bound in bytes of one encoded stream of n symbols (worst case all-bypass). */
PC_API size_t pc_rans_bound(size_t n);

/* Replaces compressai.ans.RansEncoder.encode_with_indexes (cpp_exts/rans/rans_interface.cpp:193-204,
 * i.e. BufferedRansEncoder::encode_with_indexes :99-164 + flush :166-191).
 * cdfs is a dense [n_cdf][cdf_stride] int32 table (the module buffer `_quantized_cdf`),
 * cdf_sizes = `_cdf_length`, offsets = `_offset`.  Output bytes are identical to the reference's. */
PC_API int pc_rans_encode_with_indexes(const int32_t* symbols, const int32_t* indexes, size_t n,
                                       const int32_t* cdfs, int n_cdf, int cdf_stride,
                                       const int32_t* cdf_sizes, const int32_t* offsets,
                                       uint8_t* out, size_t out_cap, size_t* out_len);

/* Replaces compressai.ans.RansDecoder.decode_with_indexes (rans_interface.cpp:206-275). */
PC_API int pc_rans_decode_with_indexes(const uint8_t* encoded, size_t encoded_len,
                                       const int32_t* indexes, size_t n,
                                       const int32_t* cdfs, int n_cdf, int cdf_stride,
                                       const int32_t* cdf_sizes, const int32_t* offsets,
                                       int32_t* symbols_out);

/* Replaces compressai.ans.RansDecoder.set_stream + decode_stream (rans_interface.cpp:277-350): one stream decoded in several calls
 * (each with its own indexes / tables).  `state` is two uint64 owned by the caller: zero both before the first call on a stream (that
 * call reads the stream's initial state, as set_stream does); every successful call advances them.  The reference keeps this state
 * inside the decoder object; here it stays with the caller so the function is re-entrant.
 * compressai.ans.BufferedRansEncoder (encode_with_indexes ... flush, rans_interface.cpp:99-191) needs no entry point of its own:
 * its output is by construction that of ONE pc_rans_encode_with_indexes call over the concatenated symbols / indexes. */
PC_API int pc_rans_decode_stream(const uint8_t* encoded, size_t encoded_len, uint64_t* state,
                                 const int32_t* indexes, size_t n,
                                 const int32_t* cdfs, int n_cdf, int cdf_stride,
                                 const int32_t* cdf_sizes, const int32_t* offsets,
                                 int32_t* symbols_out);

/* Batched forms: n_streams independent streams of n symbols each (one per image, as the loop at
 * entropy_models.py:227-235 / :276-286 produces), coded on a host thread pool.
 * symbols/indexes: [n_streams][n].  Encode writes stream s at out + s*out_stride. */
PC_API int pc_rans_encode_batch(const int32_t* symbols, const int32_t* indexes, size_t n_streams, size_t n,
                                const int32_t* cdfs, int n_cdf, int cdf_stride,
                                const int32_t* cdf_sizes, const int32_t* offsets,
                                uint8_t* out, size_t out_stride, size_t* out_lens, int n_threads);
PC_API int pc_rans_decode_batch(const uint8_t* const* encoded, const size_t* encoded_lens, size_t n_streams,
                                const int32_t* indexes, size_t n,
                                const int32_t* cdfs, int n_cdf, int cdf_stride,
                                const int32_t* cdf_sizes, const int32_t* offsets,
                                int32_t* symbols_out, int n_threads);

/* The decoder's fast form of pc_rans_decode_batch: CDF indexes as bytes (n_cdf <= 256; the GaussianConditional has 64 rows), a per-row
 * start table instead of the reference's linear scan (rans_interface.cpp:238-241), two streams per host thread in lock step.  Same
 * symbols, same error codes. */
PC_API int pc_rans_decode_batch_u8(const uint8_t* const* encoded, const size_t* encoded_lens, size_t n_streams,
                                   const uint8_t* indexes, size_t n,
                                   const int32_t* cdfs, int n_cdf, int cdf_stride,
                                   const int32_t* cdf_sizes, const int32_t* offsets,
                                   int32_t* symbols_out, int n_threads);

/* Replaces compressai._CXX.pmf_to_quantized_cdf (cpp_exts/ops/ops.cpp:10-67).  cdf_out has n+1 entries. */
PC_API int pc_pmf_to_quantized_cdf(const float* pmf, int n, int precision, uint32_t* cdf_out);

/* ---------------------------------------------------------------------------------------
 * Device stages (asynchronous on `stream`).  Activations are NHWC float32.
 * ------------------------------------------------------------------------------------- */

/* Generic convolution / transposed convolution / linear layer with the reference's weight layout.
 *   kind 0: nn.Conv2d weight [Cout][Cin][k][k], stride s, padding k/2   (models/utils.py:186, layers/layers.py:15,27)
 *   kind 1: nn.ConvTranspose2d(k=5, s=2, p=2, output_padding=1) weight [Cin][Cout][5][5]   (models/utils.py:196)
 * `w_packed` must come from pc_pack_conv_weight (tap-major; the layout -- [taps][Cin][Cout] or [taps][Cout][Cin] -- is a
 * function of (kind, Cin, Cout, k) only).
 * act: 0 none, 1 GELU.  x: [B][H][W][Cin], out: [B][Ho][Wo][Cout]. */
PC_API int pc_pack_conv_weight(const float* w_host, int kind, int Cout, int Cin, int k, float* w_packed_host);
PC_API int pc_conv2d_nhwc(const float* x, int B, int H, int W, int Cin,
                          const float* w_packed, const float* bias, int kind, int Cout, int k, int stride,
                          int act, int tile_cfg, float* out, void* stream);

/* GDN / IGDN (layers/gdn.py:50-63) with already re-parametrised beta [C] and gamma [C_out][C_in] (the module's own layout). */
PC_API int pc_gdn_nhwc(const float* x, int B, int H, int W, int C, const float* beta, const float* gamma,
                       int inverse, float* out, void* stream);

/* Shifted-window attention core between the qkv and proj Linears (layers/win_attention.py:84-115,153-207).
 * qkv: [B][H][W][3C]; bias: dense [heads][T][T]; out: [B][H][W][C]. */
PC_API int pc_win_attention_nhwc(const float* qkv, const float* bias, int B, int H, int W, int C, int heads,
                                 int window, int shift, float* out, void* stream);

/* ChannelMask "point-based-std" threshold (layers/masking.py:205-223): thr[b] = torch.quantile(scale[b].ravel(), q).
 * scale: [B][HW][C] with pixel stride ld. */
PC_API int pc_mask_quantile_threshold(const float* scale, int ld, int B, int HW, int C, float q, float* thr, void* stream);

/* Fused mask -> index -> quantise -> dequantise of one 32-channel slice (encoder side):
 *   mask = scale >= thr[b]                      (masking.py:219; mask_mode 1; 0 = no mask, 2 = ones, 3 = zeros)
 *   idx  = build_indexes(scale * mask)          (entropy_models.py:661-666; CHProg_cnn.py:751,828)
 *   sym  = int(round((y [- ybase] - mu) [* mask]))   (entropy_models.py:137-150; CHProg_cnn.py:752,780-781,830)
 *   yhat = float(sym) + mu                      (CHProg_cnn.py:754-755,833-834)
 * scale/mu/y/ybase/yhat NHWC with their own pixel strides; sym/idx/mask out in [B][32][HW] (rANS order). */
PC_API int pc_gc_prep_encode(const float* scale, int ld_scale, const float* mu, int ld_mu,
                             const float* y, int ld_y, const float* ybase, int ld_ybase,
                             const float* thr, int mask_mode, int B, int HW,
                             const float* scale_table, int n_table, float scale_bound,
                             int32_t* sym, int32_t* idx, float* mask, float* yhat, int ld_yhat, void* stream);
PC_API int pc_gc_prep_decode_index(const float* scale, int ld_scale, const float* thr, int mask_mode, int B, int HW,
                                   const float* scale_table, int n_table, float scale_bound,
                                   int32_t* idx, float* mask, void* stream);
PC_API int pc_gc_dequantize(const int32_t* sym, const float* mu, int ld_mu, int B, int HW,
                            float* yhat, int ld_yhat, void* stream);

/* ---------------------------------------------------------------------------------------
 * The codec object: ChannelProgresssiveWACNN (models/CHProg_cnn.py:30) in the canonical
 * configuration, weights resident in HBM, compress()/decompress() as native launch sequences.
 * ------------------------------------------------------------------------------------- */
typedef struct pc_codec pc_codec;

enum { PC_MASK_POINT_BASED_STD = 0, PC_MASK_TWO_LEVELS = 1, PC_MASK_THREE_LEVELS_STD = 2 };   /* layers/masking.py:205-247 */
enum { PC_F32 = 0, PC_I32 = 1, PC_I64 = 2 };

PC_API int pc_codec_create(pc_codec** out, int device);
PC_API void pc_codec_destroy(pc_codec* c);

/* load_state_dict (models/cnn.py:195-202): one call per reference state_dict key, host pointers.
 * Unknown keys are ignored (returns PC_OK); shape mismatches return PC_ERR_MISSING. */
PC_API int pc_codec_set_tensor(pc_codec* c, const char* name, const void* data, int dtype,
                               const int64_t* shape, int ndim);
/* Entropy tables: the module buffers _quantized_cdf/_cdf_length/_offset after update() (models/cnn.py:137-142). */
PC_API int pc_codec_set_tables(pc_codec* c, int which /*0 = gaussian_conditional, 1 = entropy_bottleneck*/,
                               const int32_t* cdf, int n_cdf, int cdf_stride, const int32_t* cdf_sizes,
                               const int32_t* offsets);
/* update(scale_table=...) (models/cnn.py:137-142 -> GaussianConditional.update_scale_table, entropy_models.py:588-597): replace the
 * table of scales build_indexes searches (2..64 ascending floats, host pointer); set the matching CDFs with pc_codec_set_tables.
 * Only after pc_codec_finalize. */
PC_API int pc_codec_set_scale_table(pc_codec* c, const float* table, int n);
/* Validate that every tensor is present, fold the GDN re-parametrisation, pack weights into HBM. */
PC_API int pc_codec_finalize(pc_codec* c);
PC_API int pc_codec_set_threads(pc_codec* c, int n_threads);
/* Schedule options of one codec object -- how the launch sequence is laid over streams and host threads, never what it computes (every
 * byte string and x_hat is identical under every setting).  No reference counterpart (the reference is one stream, one thread).
 *   "serial_schedule" 0 / 1   1: the whole chain on the caller's stream, one batch lane, no base || enhancement pipelining -- the form a
 *                             profiler wants (a launch's duration is its own); default 0
 *   "lanes_enc", "lanes_dec"  0 (default: 1 / 2) .. 8: sub-batches of compress / decompress that run on streams of their own
 *   "host_threads"            as pc_codec_set_threads
 *   "profile_in_schedule" 0/1 1: pc_codec_profile_begin only brackets the launches and leaves the schedule alone (default 0: profiling
 *                             forces the serial schedule so that a launch's duration is its own)
 * Unknown names and out-of-range values return PC_ERR_ARG.  The tuning switches of the profiling rounds (PC_CONV_*, PC_LANES ...) exist
 * only in the -DPC_TUNING build of the library (csrc/Makefile `tuning`), not here. */
PC_API int pc_codec_set_option(pc_codec* c, const char* name, int value);
/* How the host entropy-coding pool of this process is laid out: threads = min(16, CPUs allowed by affinity and cgroup quota / local ranks),
 * pinned -- when there are several local ranks (LOCAL_WORLD_SIZE / LOCAL_RANK) -- to this rank's contiguous slice of the allowed CPUs
 * starting at first_cpu.  No reference counterpart (the reference codes on one thread under the GIL, entropy_models.py:226-235). */
PC_API int pc_host_pool_plan(int* n_threads, int* first_cpu, int* n_allowed);

/* compress (models/CHProg_cnn.py:686-847).  x: device, NCHW [B][3][H][W], H and W multiples of 64.
 * On success the codec holds n_slices*B + B byte strings (n_slices = 10 for quality <= 0, else 20) until
 * the next call; masks_out (device, [10][B][32][H/16][W/16], may be NULL) receives the "masks" entry. */
PC_API int pc_codec_compress(pc_codec* c, const float* x, int B, int H, int W, double quality, int mask_pol,
                             float* masks_out, void* stream);
/* cust_map of compress() / decompress() (CHProg_cnn.py:686,849 -> layers/masking.py:171-194): device tensor NCHW
 * [B][320][H/16][W/16]; when set, the NEXT compress / compress_levels / decompress / decompress_levels call thresholds this map
 * (top quality*10 % per image and slice; quality >= 10 all, 0 none, whatever the mask policy) instead of the scale, then the
 * pointer is cleared.  NULL clears it. */
PC_API int pc_codec_set_cust_map(pc_codec* c, const float* cust_map);
/* The REM model family -- PostRateProcessedNetwork (models/CHProgREM.py:205; compress :673, decompress :896): a frozen base codec whose
 * predicted scale of every enhancement slice is refined by a LatentRateReduction CNN (:12-86, apply_latent_enhancement :375-428) chosen
 * by the range [check_levels[k], check_levels[k+1]) the quality falls in.  The CNN weights are state-dict tensors named
 * "post_latent.<level>.<slice>.<subnet>.<block>.{conv1,conv2,skip}.{weight,bias}" set with pc_codec_set_tensor before
 * pc_codec_finalize.  n_levels in 1..3 switches the refinement on for the following compress / decompress / forward calls, 0 switches it
 * off (plain ChannelProgresssiveWACNN).  mu_std (:30,42,397-416) and dimension ("big" / "middle", :23-43) are properties of the loaded
 * post_latent tensors (their names and shapes); checkpoint_rep is handed over with pc_codec_set_rem_checkpoint below. */
PC_API int pc_codec_set_rem(pc_codec* c, const double* check_levels, int n_levels);
/* checkpoint_rep of PostRateProcessedNetwork.compress / decompress (models/CHProgREM.py:676,773 / :901,989): a device tensor NCHW
 * [B][320][H/16][W/16] that replaces the decoded base slices as the x_base input of the LatentRateReduction nets in the NEXT compress /
 * decompress call, then the pointer is cleared (what the escalation mode chains from check level to check level, :335-373).  NULL clears. */
PC_API int pc_codec_set_rem_checkpoint(pc_codec* c, const float* checkpoint_rep);
PC_API int pc_codec_num_slices(const pc_codec* c);
/* string of y slice `slice` (0..n_slices-1) or of z (slice = -1) for image b */
PC_API int pc_codec_get_string(const pc_codec* c, int slice, int b, const uint8_t** data, size_t* len);

/* decompress (models/CHProg_cnn.py:849-999).  y_strings: [n_slices][B] pointers (slice-major), z_strings: [B];
 * zh, zw = "shape"; x_hat: device NCHW [B][3][64*zh][64*zw]. */
PC_API int pc_codec_decompress(pc_codec* c, const uint8_t* const* y_strings, const size_t* y_lens, int n_slices,
                               const uint8_t* const* z_strings, const size_t* z_lens, int B, int zh, int zw,
                               double quality, int mask_pol, float* x_hat, void* stream);

/* Multi-level ("progressive") coding of one batch -- SURVEY.md section 8(f) rank 1.  The reference's harness calls
 * compress() / decompress() once per mask level (training/step.py:322-337), recomputing g_a, h_a, the hyper-latent strings,
 * h_s and the ten base slices (CHProg_cnn.py:692-767 / :855-904) although none of them depends on the level.  These entry
 * points compute that shared part once and run only the enhancement chain (:775-845 / :930-983) and the synthesis transform
 * per level.  Every string, mask and x_hat is identical, bit for bit, to what n_levels separate pc_codec_compress /
 * pc_codec_decompress calls return.
 *   string slots: slice 0..9 = base (shared), slice 10..19 of level l = slot 10 + 10*l + (slice-10); a level with
 *   quality <= 0 has no enhancement strings.  masks_out: NULL or n_levels device pointers ([10][B][32][H/16][W/16] each, NULL
 *   allowed per level; untouched for quality <= 0).  x_hat: device, [n_levels][B][3][64*zh][64*zw]. */
PC_API int pc_codec_compress_levels(pc_codec* c, const float* x, int B, int H, int W, const double* qualities, int n_levels,
                                    int mask_pol, float* const* masks_out, void* stream);
/* string of level `level`: slice -1 = z, 0..9 = base (the same for every level), 10..19 = that level's enhancement */
PC_API int pc_codec_get_level_string(const pc_codec* c, int level, int slice, int b, const uint8_t** data, size_t* len);
/* y_strings / y_lens: [(10 + 10*n_levels) * B] in slot order (slot-major, then image); slots of quality-0 levels are ignored */
PC_API int pc_codec_decompress_levels(pc_codec* c, const uint8_t* const* y_strings, const size_t* y_lens,
                                      const uint8_t* const* z_strings, const size_t* z_lens, int B, int zh, int zw,
                                      const double* qualities, int n_levels, int mask_pol, float* x_hat, void* stream);

/* Bulk string transfer (binding overhead: 672 strings per Config-2 batch).  After a compress: _strings_size gives the total byte count
 * and the number of strings (all y slots, images inside a slot, then the z strings); _copy_strings concatenates them into dst (cap bytes)
 * and writes their lengths (lens_cap = entries available in lens; PC_ERR_BUFFER if either buffer is too small).  _decompress_packed takes the same layout: y strings of slots 0 .. 10+10*n_levels-1 (B each; empty strings
 * for the slots of quality-0 levels), then B z strings. */
PC_API int pc_codec_strings_size(const pc_codec* c, size_t* total_bytes, int* n_strings);
PC_API int pc_codec_copy_strings(const pc_codec* c, uint8_t* dst, size_t cap, size_t* lens, size_t lens_cap);
PC_API int pc_codec_decompress_packed(pc_codec* c, const uint8_t* data, const size_t* lens, int B, int zh, int zw,
                                      const double* qualities, int n_levels, int mask_pol, float* x_hat, void* stream);

/* forward_single_quality in eval mode (models/CHProg_cnn.py:1002-1198) -- SURVEY.md section 8(f) rank 2: the rate-estimation path
 * behind test_epoch / valid_epoch (training/step.py:215-267).  Runs the encoder chain without entropy coding and returns the
 * likelihood tensors: y_lik device [B][320 or 640][H/16][W/16] (640 when quality != 0), z_lik device [B][192][H/64][W/64],
 * x_hat device [B][3][H][W] (identical to decompress(compress(x))), masks_out as in pc_codec_compress (may be NULL).
 * force_enhanced != 0 with quality == 0 (:1006,1022,1064): both hyper-priors and the enhancement chain run with all-zero masks
 * (every enhancement symbol 0), y_lik has 640 channels and x_hat comes from the enhancement synthesis g_s[1].
 * Needs the entropy_bottleneck._matrix/_bias/_factor tensors in the state dict (PC_ERR_STATE otherwise). */
PC_API int pc_codec_forward(pc_codec* c, const float* x, int B, int H, int W, double quality, int mask_pol, float* x_hat,
                            float* y_lik, float* z_lik, float* masks_out, int force_enhanced, void* stream);

/* Measurement aid (bench.py roofline leg): while profiling is on, every launch of the MFMA convolution kernel made by
 * compress()/decompress() is bracketed by HIP events on the call's stream; _end returns the launch count, the summed
 * event time and the algorithmic FLOPs (2*M*N*K per launch, no padding counted). */
PC_API int pc_codec_profile_begin(pc_codec* c);
PC_API int pc_codec_profile_end(pc_codec* c, int64_t* n_launches, double* total_ms, double* total_flops);
/* algorithmic HBM bytes (every operand of a launch once: input, weights, bias, output, aux tensors) summed over the launches recorded
 * since pc_codec_profile_begin; read it after pc_codec_profile_end */
PC_API int pc_codec_profile_bytes(const pc_codec* c, double* total_algorithmic_bytes);
/* In-schedule profile (option "profile_in_schedule"): the bracketed launches of one or several codec objects -- e.g. the encoder and the
 * decoder object of progressivecodec_amd.CodecPipeline, which keep ~2.5 conv kernels in flight -- on ONE timeline.  pc_profile_set_epoch
 * drains `device` and records the epoch; after pc_codec_profile_end, pc_codec_profile_intervals returns every recorded launch's start / end
 * (ms since the epoch) and algorithmic FLOPs (cap entries available; all three arrays NULL: only *n is written).  bench.py folds the
 * intervals of both objects: FLOPs / (time during which at least one conv kernel runs) = the in-situ MFMA fraction of the schedule it
 * timed.  Measurement aids; no reference counterpart. */
PC_API int pc_profile_set_epoch(int device);
PC_API int pc_codec_profile_intervals(const pc_codec* c, double* t0_ms, double* t1_ms, double* flops, size_t cap, size_t* n);
/* Host entropy-coding figures of the object's last compress / decompress call ("rANS: report Msym/s per stream
 * and streams in flight"; no reference counterpart -- the reference times decompress() as a whole, training/step.py:332-340).
 * out[0] wall ms of the last compress call, out[1] host ms spent in rANS encoding during it (mostly hidden behind the GPU chain),
 * out[2] of those, the ms of the last pass that nothing hides ("exposed"), out[3] host ms of rANS decoding in the last decompress call
 * summed over the decoder's lanes (each slice's decode sits between two GPU steps of its chain), out[4] / out[5] symbols encoded /
 * decoded by those calls.  n >= 6. */
PC_API int pc_codec_host_stats(const pc_codec* c, double* out, int n);

/* Test aid: the conv epilogue evaluates GELU (layers/layers.py:45, nn.GELU) on two elements per lane with packed f32 instructions; this
 * compares that form with the contract's scalar pc_geluf (include/pc_math.h) over ALL 2^32 float arguments on the current device.
 * *n_mismatch must be 0; *n_nan_payload counts arguments for which both forms return NaN with different payload bits.  Synchronous. */
PC_API int pc_selftest_packed_gelu(uint64_t* n_mismatch, uint64_t* n_nan_payload);

/* Debug/test taps: copy an internal device tensor of the last call to host ("y", "z", "latent_means", ...). */
PC_API int pc_codec_read_tap(pc_codec* c, const char* name, float* host_out, size_t cap_floats, size_t* n_floats);
PC_API int pc_codec_read_tap_i32(pc_codec* c, const char* name, int32_t* host_out, size_t cap, size_t* n);

#ifdef __cplusplus
}
#endif
#endif /* PCODEC_H */
