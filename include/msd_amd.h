/*
 * msd_amd.h -- C ABI of the MI355X-native DDPM spectrogram synthesizer.
 *
 * Drop-in boundary for ONE path of magenta/music-spectrogram-diffusion: the
 * denoising loop reached through InferenceModel.predict
 * (music_spectrogram_diffusion/inference.py:200-203) ->
 * {Diffusion,ContextDiffusion}Model.predict_batch_with_aux
 * (models/diffusion/models.py:149-205, 340-400) ->
 * diffusion_utils.eval_scan (models/diffusion/diffusion_utils.py:456-476) over
 * network.{Transformer,ContinuousContextTransformer}.{encode,decode}
 * (models/diffusion/network.py:460-606).
 *
 * The reference is pure Python/JAX: it has no FFI.  These entry points are what
 * a maintainer would bind (ctypes stub in INTEGRATION.md) to replace the jitted
 * `predict_fn(params, batch, rng)` of inference.py:183-198 -- one call group per
 * stage of predict_batch_with_aux.
 *
 * Conventions
 *   - plain C types only; every function returns an msd_status (0 = ok) and never
 *     throws; msd_last_error() gives the message for the last failure on a handle.
 *   - one handle <-> one device <-> one caller thread at a time (not re-entrant).  Handles are independent of each
 *     other: several may be created, loaded and run concurrently from different threads, also on ONE device (the
 *     library's own synchronous copies use a non-blocking stream of the handle, never the legacy stream).  This
 *     covers every msd_* entry point that takes a handle; the stand-alone msd_op_* building blocks (unit-test entry
 *     points: they allocate, clear and copy scratch through the legacy stream) are NOT part of that guarantee -- do
 *     not call them while another thread captures or runs a model.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); all
 *     device work is enqueued on it; calls return without synchronising unless
 *     stated.
 *   - "dev" pointers are device pointers owned by the caller (e.g. torch
 *     tensor.data_ptr()); "host" pointers are host memory.  Weights, caches,
 *     tables and graph objects are owned by the library.
 *   - tensors are dense row-major; float = IEEE binary32.
 */
#ifndef MSD_AMD_H_
#define MSD_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSD_AMD_ABI_VERSION 6   /* 6: cross_merge_in_launch, cross_q_fold, mlp_in_persistent appended to msd_config.
                                   5: dedup_layer0, cross_key_split, keep_raw_weights, kv_touch_ahead appended to msd_config.
                                   4: every caller-selectable knob is a msd_config field (attn_q_planes / attn_p_planes
                                      replace ABI 3's attn_query_planes; graph_steps; weight_prefetch): the library reads
                                      NO environment variable.  3: MSD_ERR_RANGE; distinct bfloat16-plane precisions */

typedef struct msd_model msd_model; /* opaque */

typedef enum msd_status {
  MSD_OK = 0,
  MSD_ERR_INVALID_ARGUMENT = 1, /* -> ValueError (unknown sampler/schedule/...:
                                   diffusion_utils.py:202,320,450; network.py:106,237,336) */
  MSD_ERR_UNKNOWN_WEIGHT = 2,   /* -> KeyError */
  MSD_ERR_SHAPE_MISMATCH = 3,   /* -> ValueError */
  MSD_ERR_BAD_STATE = 4,        /* call order violated -> RuntimeError */
  MSD_ERR_HIP = 5,              /* HIP runtime failure -> RuntimeError */
  MSD_ERR_UNSUPPORTED = 6,      /* valid in the reference, not built here -> NotImplementedError */
  MSD_ERR_RANGE = 7             /* an ACTIVATION left the range of the IEEE-half operand planes (|x| > 65504) during
                                   this call: its result is invalid.  The reference computes in float32
                                   (gin/models/diffusion/context/t5_base.gin:72) and has no such limit; the way out is
                                   the bfloat16-plane build (MSD_PREC_BF16X3, libmsd_amd_bf16.so) -> msd RangeError
                                   (an ArithmeticError) in the Python layer */
} msd_status;

/* Arithmetic of the transformer GEMMs / attention.  Residual stream, RMSNorm
 * statistics, softmax, FiLM, input/output projections and the sampler are always
 * fp32 (network.py:454; diffusion_utils.py:461). */
typedef enum msd_precision {
  MSD_PREC_F16 = 0,    /* one IEEE-half plane per operand (v_mfma_f32_*_f16), fp32 accumulate: fast, not parity-grade */
  MSD_PREC_F16X3 = 1,  /* operands split hi + lo half planes (22 significand bits), 3 MFMAs per product
                          (hi.hi + hi.lo + lo.hi): float32-class results -- the parity mode and the default.
                          The query side of the decoder's attentions may run on one plane: msd_config.attn_q_planes /
                          attn_p_planes.
                          Weights must satisfy |w| < 128 (packed times 2^9; checked by msd_finalize_weights ->
                          MSD_ERR_UNSUPPORTED); activations |x| <= 65504 (checked on every conversion ->
                          MSD_ERR_RANGE from the call that saw it). */
  MSD_PREC_BF16 = 2,   /* one bfloat16 plane per operand */
  MSD_PREC_BF16X3 = 3  /* hi + lo bfloat16 planes (16 significand bits, float32's exponent range: no range limit) */
  /* The plane FORMAT is a property of the library build: libmsd_amd.so implements the two half precisions,
   * libmsd_amd_bf16.so (same sources, same ABI) the two bfloat16 ones; msd_create returns MSD_ERR_UNSUPPORTED for a
   * precision of the other build (ABI <= 2 aliased the names and silently ran whatever planes the library had). */
} msd_precision;

typedef enum msd_sampler_kind {
  MSD_SAMPLER_DDPM = 0, /* diffusion_utils.py:382-395 */
  MSD_SAMPLER_DDIM = 1  /* diffusion_utils.py:369-379 */
} msd_sampler_kind;

typedef enum msd_schedule_kind {
  MSD_SCHEDULE_COSINE = 0, /* diffusion_utils.py:181-187 */
  MSD_SCHEDULE_LINEAR = 1  /* diffusion_utils.py:189-199: betas linspace(start, stop, num_steps) */
} msd_schedule_kind;

typedef enum msd_model_output {
  MSD_OUTPUT_EPS = 0, /* diffusion_utils.py:296-300 */
  MSD_OUTPUT_X0 = 1,  /* :301-305 */
  MSD_OUTPUT_V = 2    /* :312-317 (x0_and_eps needs a 2n-channel network output, which the
                         reference's own Decoder (network.py:451-456) cannot produce: rejected) */
} msd_model_output;

typedef enum msd_logvar_kind {
  MSD_LOGVAR_LARGE = 0, /* diffusion_utils.py:146-149 */
  MSD_LOGVAR_SMALL = 1, /* :142-145 */
  MSD_LOGVAR_MEDIUM = 2 /* :150-157, "medium:<frac>" -> logvar_frac */
} msd_logvar_kind;

/* Hyper-parameters: network.T5Config (network.py:54-72), DiffusionConfig & co
 * (diffusion_utils.py:25-59), %TASK_FEATURE_LENGTHS (inference.py:97-101) and the
 * codec range (audio_codecs.py:207-213).  Fixed by construction on this path:
 * mlp_activations=("gelu","linear"), head_dim=64; anything else is rejected by the Python layer / msd_create. */
typedef struct msd_config {
  int32_t struct_size;            /* sizeof(msd_config), ABI check */
  int32_t has_context;            /* 0 DiffusionModel, 1 ContextDiffusionModel */
  int32_t vocab_size;
  int32_t emb_dim;
  int32_t num_heads;
  int32_t head_dim;
  int32_t mlp_dim;
  int32_t num_encoder_layers;
  int32_t num_decoder_layers;
  int32_t inputs_length;          /* L  */
  int32_t targets_length;         /* T  */
  int32_t context_length;         /* C (0 without context) */
  int32_t n_dims;                 /* mel bins */
  int32_t num_steps;              /* sampler schedule num_steps */
  int32_t sampler;                /* msd_sampler_kind */
  int32_t clip_x0;
  int32_t context_terminal_relative; /* T5Config.context_positions */
  int32_t precision;              /* msd_precision */
  int32_t max_batch;              /* largest `batch` accepted by encode/sample */
  float max_decoder_noise_time;
  float cfg_weight;               /* eval_condition_weight; 1.0 = single pass */
  float feature_min;              /* codec min_value */
  float feature_max;              /* codec max_value */
  /* ABI 2: the sampler / schedule branches of diffusion_utils.py (all step-indexed table work) */
  int32_t model_output;           /* msd_model_output: DiffusionConfig.model_output */
  int32_t logvar_type;            /* msd_logvar_kind: SamplerConfig.logvar_type */
  float logvar_frac;              /* the <frac> of "medium:<frac>", in [0, 1] */
  int32_t sampler_schedule;       /* msd_schedule_kind of SamplerConfig.schedule (num_steps above) */
  float sampler_schedule_start;   /* linear only */
  float sampler_schedule_stop;
  int32_t train_schedule;         /* msd_schedule_kind of DiffusionConfig.train_schedule: the log-SNR
                                     at which the model output is converted (diffusion_utils.py:294) */
  float train_schedule_start;     /* linear only */
  float train_schedule_stop;
  int32_t train_schedule_num_steps;
  int32_t cross_attend_sum;       /* T5Config.decoder_cross_attend_style: 0 = "concat_encodings" (every shipped
                                     gin), 1 = "sum_cross_attends" (the dataclass default, network.py:199-216:
                                     one cross-attention module per encoding, outputs summed) */
  /* ABI 4: the query side of the decoder's attentions in the two-plane modes (the memory side -- K, V -- always keeps
   * hi + lo).  0 = the library's choice (DESIGN.md 3 says which and why), 1 = one 16-bit plane, 2 = hi + lo. */
  int32_t attn_q_planes;          /* Q in q.k^T: one plane perturbs a logit by ~|s| 2^-12 -- fine for O(1) logits,
                                     not for sharp (trained) attention */
  int32_t attn_p_planes;          /* the softmax weights in P.V: one plane = 2^-12 relative on each weight, whatever
                                     the logits */
  int32_t graph_steps;            /* DDPM steps captured per hipGraph (0 = the library's choice, 8) */
  int32_t weight_prefetch;        /* producers warm the next GEMM's weights: 0 = the library decides from the model's
                                     size (on when a step streams more than the 256 MB Infinity Cache holds), 1 = on,
                                     2 = off */
  /* ABI 5 */
  int32_t dedup_layer0;           /* a CFG step computes decoder layer 0's self-attention block once for both passes
                                     (they are bit-identical up to the first cross-attention: models/diffusion/models.py:
                                     373-386): 0 = the library's choice (on), 1 = on, 2 = off (A/B and bitwise tests) */
  int32_t cross_key_split;        /* blocks that share the key axis of one (head, query tile) of the decoder's
                                     cross-attention: 0 = the library chooses per msd_encode from the segment's key
                                     count and the batch, else 1, 2, 4 or 8 */
  int32_t keep_raw_weights;       /* 0 = msd_finalize_weights frees the float32 staging copy of every matrix it has
                                     packed into operand planes (1.5 GB of 1.65 at base_with_context); 1 = keep them
                                     (they have no reader; for memory-accounting A/Bs) */
  int32_t kv_touch_ahead;         /* the cross-attention launches' prefetch wave touches the cached K / V^T lines this
                                     many 128-key ring stages ahead of their LDS-DMA (they are HBM-cold at every
                                     step): 0 = the library's choice (2, with one song per handle; batched launches
                                     are bandwidth-bound and never touch), -1 = off, 1 .. 16.  The touches ride on the
                                     launch's weight-prefetch wave: a cross-attention launch WITHOUT a weight target
                                     (weight_prefetch off -- the library's choice for models whose step fits the 256 MB
                                     cache, e.g. `small` / tiny presets -- or a module other than a layer's last) never
                                     touches, whatever this field says */
  /* ABI 6 */
  int32_t cross_merge_in_launch;  /* a key-split cross-attention finishes INSIDE its launch: every block publishes its
                                     partial write-through, the last block of a (query tile, head) group to arrive merges
                                     them -- no separate merge launch (one kernel boundary less per decoder layer);
                                     bit-identical to the merge launch.  0 = the library's choice (on), 1 = on, 2 = off */
  int32_t cross_q_fold;           /* the cross-attention's query projection has no launch of its own (one kernel boundary
                                     less per decoder layer).  Exact algebra on network.py:174-198: with x1 = x0 + ao . Wo,
                                     (x1 (.) gamma) . Wq = (x0 (.) gamma) . Wq + ao . (Wo diag(gamma) Wq) -- the first term
                                     rides on the QKV launch's idle CUs, the second runs beside the self-attention output
                                     projection (same A operand), and the 1/rms of the norm moves onto the logits inside
                                     the attention kernel.  Same float32-class result, NOT bit-identical to the unfolded
                                     order (3e-7 relative on a decoder pass).  Two-plane precisions, up to 3 songs per
                                     call.  0 = the library's choice (on), 1 = on, 2 = off */
  int32_t mlp_in_persistent;      /* batched songs (>= 4 per call: 128 x 128 tiles, several per CU): the decoder's gated-MLP
                                     input projection runs as ONE resident block per CU that walks its tiles, with the
                                     epilogue on the accumulator registers and the next tile's operands landing under
                                     it.  Same float32-class result, not bit-identical to the per-tile launch (another
                                     contraction of the epilogue's multiply-adds).  0 = the library's choice (on), 1 = on,
                                     2 = off */
} msd_config;

const char* msd_version(void);

/* Number of visible HIP devices (<0 on failure).  */
int msd_device_count(void);

/* Create a model on the CURRENT HIP device.  Allocates weights/caches/tables. */
int msd_create(const msd_config* cfg, msd_model** out);
void msd_destroy(msd_model* m);
const char* msd_last_error(const msd_model* m);

/* Parameter tree.  Names are the Flax names of the reference modules, '/'-joined
 * (e.g. "decoder/layers_3/FiLMLayer_0/DenseGeneral_0/kernel"); shapes as stored
 * by the reference ([in, out] kernels, layers.py:430-431).  `data` may be host or
 * device memory (hipMemcpyDefault; the copy runs on the handle's own stream and the call has no stream argument, so
 * a DEVICE-side `data` must be complete -- its producer stream synchronised -- before the call).  Weights are loaded
 * ONCE per handle: after msd_finalize_weights every msd_set_weight returns MSD_ERR_BAD_STATE.  Replaces the params
 * pytree handed to predict_fn (inference.py:197-203). */
int msd_num_weights(const msd_model* m);
int msd_weight_info(const msd_model* m, int index, const char** name, int64_t shape[2], int* ndim);
int msd_set_weight(msd_model* m, const char* name, const float* data,
                   const int64_t* shape, int ndim);
/* Pack weights for the kernels and build every step-indexed table (log-SNR and
 * sampler coefficients, time-embedding MLP, FiLM scale/bias).  Synchronises. */
int msd_finalize_weights(msd_model* m, void* stream);

/* module.encode of predict_batch_with_aux (models.py:365-371; network.py:537-559 /
 * 470-482): runs the token encoder (and context encoder: clip+scale to [-1,1],
 * models.py:361-363) once and caches the decoder's cross-attention K/V.
 *   tokens   int32 [batch, L]        (host or device)
 *   ctx      float [batch, C, n]     mel units (device), NULL without context
 *   ctx_mask int32 [batch, C]        (host or device), NULL without context
 * Synchronises `stream` at entry (device-side tokens / ctx_mask written on it are staged through the host) and before
 * it returns (half-plane range flag, like msd_sample). */
int msd_encode(msd_model* m, int batch, const int32_t* tokens, const float* ctx_dev,
               const int32_t* ctx_mask, void* stream);

/* eval_scan (diffusion_utils.py:456-476) + scale_to_features (models.py:395).  SYNCHRONISES `stream` before it
 * returns (ABI 3): behind that one wait it reads the handle's half-plane range flag, so that a run whose
 * activations left the plane range fails THIS call with MSD_ERR_RANGE instead of handing back a wrong spectrogram.
 *   init_z_dev float [batch,T,n] or NULL  -> generated (Philox, see msd_fill_normal)
 *   noise_dev  float [N,batch,T,n] or NULL -> generated; noise_dev[i] is the draw
 *              used at scan index i (diffusion_utils.py:389-390).  NULL: the sampler kernel draws step i's noise
 *              itself (sub-sequence 1 + i of msd_fill_normal's generator: no [N,batch,T,n] buffer exists; the
 *              values are those msd_fill_normal(seed, stream_id, 1 + i, ...) writes, bit for bit)
 *   stream     NULL = the legacy stream: the call waits for it (hipStreamSynchronize(NULL)) and runs on the handle's
 *              own stream (the legacy stream cannot be captured); other handles' streams are not waited for
 *   seed/stream_id key the generator when a pointer is NULL (stream_id = segment)
 *   out_dev    float [batch,T,n] mel units                                    */
int msd_sample(msd_model* m, int batch, uint64_t seed, uint64_t stream_id,
               const float* init_z_dev, const float* noise_dev, float* out_dev,
               void* stream);

/* Drop the captured hipGraph of the DDPM step; the next msd_sample captures it again. */
int msd_reset_graph(msd_model* m);

/* One decoder call of the scan body: pred_fn(z, time=(i+1)/N, include_conditioning)
 * (models.py:373-386 -> network.py:561-573).  For parity tests and profiling.
 *   z_dev float [batch,T,n]; eps_out_dev float [batch,T,n]                     */
int msd_decoder_pass(msd_model* m, int batch, int step_index, const float* z_dev,
                     int include_conditioning, float* eps_out_dev, void* stream);

/* The library's counter-based normal generator: Philox4x32-10, key
 * (seed_lo, seed_hi), counter (elem/4, stream_id_lo, stream_id_hi|..., subseq),
 * Box-Muller; documented in DESIGN.md and restated in oracle/philox.py.
 * subseq: 0 = init_z, 1 + i = step-i noise.                                   */
int msd_fill_normal(uint64_t seed, uint64_t stream_id, uint32_t subseq,
                    float* out_dev, int64_t n, void* stream);

/* Step-indexed tables, for parity tests: copies [num_steps, 8] floats to host:
 * {logsnr_t, logsnr_s, x0_scale, x0_eps_coef, mean_z_coef, mean_x0_coef, std,
 *  logsnr of the TRAIN schedule at t (model-output conversion)}. */
int msd_get_schedule(const msd_model* m, float* host_out);

/* Read an internal buffer as float (bf16 widened) into host memory, for tests.
 * Returns the element count in *n_out; copies min(count, max_elems).  Synchronises. */
int msd_debug_read(msd_model* m, const char* buffer, float* host_out, int64_t max_elems,
                   int64_t* n_out);

/* Per-kernel-class timing of `n_steps` eagerly launched DDPM steps, measured with
 * hipEvents on `stream` around every launch (bench.py roofline leg).
 *   names_out  receives a pointer to a static NULL-terminated array of class names
 *   ms_out / launches_out  [MSD_MAX_KERNEL_CLASSES] totals over the run          */
#define MSD_MAX_KERNEL_CLASSES 16
int msd_profile_steps(msd_model* m, int batch, int n_steps, const char* const** names_out,
                      double* ms_out, int64_t* launches_out, void* stream);

/* Standalone ops (the building blocks, for unit parity tests). All device ptrs.  They synchronise and fail like the
 * model does: weights beyond the half-plane range -> MSD_ERR_UNSUPPORTED, activations beyond it -> MSD_ERR_RANGE,
 * a `precision` of the other library build -> MSD_ERR_UNSUPPORTED. */
int msd_op_gemm_h16(int precision, const float* a_dev, const float* w_dev, float* c_dev,
                    int m, int n, int k, void* stream); /* C = A[m,k] @ W[k,n] on 16-bit operand planes */
/* deprecated ABI <= 2 name of msd_op_gemm_h16 (the planes were bfloat16 then) */
int msd_op_gemm_bf16(int precision, const float* a_dev, const float* w_dev, float* c_dev,
                     int m, int n, int k, void* stream);
int msd_op_gemm_f32(const float* a_dev, const float* w_dev, float* c_dev,
                    int m, int n, int k, void* stream);
int msd_op_attention(int precision, const float* q_dev, const float* k_dev,
                     const float* v_dev, float* o_dev, int n_q, int n_keys, int n_keys_valid,
                     int heads, void* stream); /* q [n_q, heads*64] (n_q % 64 == 0), k/v [n_keys, heads*64] (n_keys % 32 == 0) */
/* The same with the query-side single-plane switches of the two-plane modes: qp bit 0 = Q enters q.k^T as ONE
 * 16-bit plane (msd_config.attn_q_planes = 1), bit 1 = the softmax weights enter P.V as one plane
 * (attn_p_planes = 1); K and V always keep hi + lo.  msd_op_attention is qp = 0. */
int msd_op_attention_qp(int precision, int qp, const float* q_dev, const float* k_dev,
                        const float* v_dev, float* o_dev, int n_q, int n_keys, int n_keys_valid,
                        int heads, void* stream);
/* ABI 6: the same with the key axis split over `ksplit` blocks per (head, query tile) (a power of two is used: 3 runs as 2)
 * and the partials merged by the separate merge launch (merge_in_launch = 0) or inside the attention launch by the last
 * block of each group to arrive (1; attention.h attention_inlaunch_merge); launched `repeats` times back to back. */
int msd_op_attention_split(int precision, int qp, int ksplit, int merge_in_launch, int repeats, const float* q_dev,
                           const float* k_dev, const float* v_dev, float* o_dev, int n_q, int n_keys,
                           int n_keys_valid, int heads, void* stream);


/* Standalone forms of the FUSED kernels of the step (each restates one reference function and has its
 * own parity test against the oracle, tests/test_gpu_fused_ops.py).  All device pointers, fp32. */

/* eval_step.body after the decoder calls (diffusion_utils.py:416-452): model-output conversion, CFG
 * combine, x0 / clip / eps, ddpm_step (:382-395, diffusion_reverse :120-163) or ddim_step (:369-379).
 * Uses cfg->{num_steps, sampler, clip_x0, cfg_weight, model_output, logvar_*, *_schedule*}.
 *   out_uncond_dev may be NULL iff cfg_weight == 1; noise_dev (this step's draw) may be NULL (zeros). */
int msd_op_sampler_step(const msd_config* cfg, int step_index, const float* z_dev,
                        const float* out_cond_dev, const float* out_uncond_dev,
                        const float* noise_dev, float* z_out_dev, int64_t n, void* stream);

/* x_out = x_in + a.w1 ; h_out = (RMSNorm(x_out; gamma) (.) (film_scale+1) + film_bias) . w2
 * (layers.py:632-666 + the Dense that follows).  folded=1: the decoder's folded-norm epilogues
 * (EpiResidualNorm producer, row-scale + tabulated bias.W consumer); folded=2: the same with the producer as the
 * 4-way split-K launch of the experiments build (tools/ubench/exp; MSD_ERR_UNSUPPORTED in the product library);
 * folded=3: the producer on the 32 x 48 tiles the decoder uses where they give one tile per CU (d % 48 == 0);
 * folded=0: separate norm kernel.
 * film_scale_dev / film_bias_dev [D] may both be NULL (plain RMSNorm).
 *   x [m,d]  a [m,k]  w1 [k,d]  gamma [d]  w2 [d,n]  x_out [m,d]  h_out [m,n]; m,k,d,n % 64 == 0 */
int msd_op_residual_norm_gemm(int folded, const float* x_in_dev, const float* a_dev,
                              const float* w1_dev, const float* gamma_dev,
                              const float* film_scale_dev, const float* film_bias_dev,
                              const float* w2_dev, float* x_out_dev, float* h_out_dev,
                              int m, int k, int d, int n, void* stream);

/* MlpBlock's gated input (layers.py:483-497): out [m,f] = gelu_tanh(a.wi0) * (a.wi1) */
int msd_op_geglu(const float* a_dev, const float* wi0_dev, const float* wi1_dev, float* out_dev,
                 int m, int k, int f, void* stream);

/* Fused q|k|v projection (layers.py:262-264) through the attention kernel's operand layouts (V^T with
 * the per-16 key permutation, per segment of seg_len rows), returned un-permuted: q,k,v [m,j]. */
int msd_op_qkv(const float* a_dev, const float* wq_dev, const float* wk_dev, const float* wv_dev,
               float* q_out_dev, float* k_out_dev, float* v_out_dev, int m, int k, int j,
               int seg_len, void* stream);

/* decoder_norm + spec_out_dense in exact fp32 (network.py:445-456): out [m,n] = RMSNorm(x; gamma).w */
int msd_op_final_proj(const float* x_dev, const float* gamma_dev, const float* w_dev, float* out_dev,
                      int m, int d, int n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MSD_AMD_H_ */
