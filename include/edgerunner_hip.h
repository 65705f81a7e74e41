/*
 * edgerunner_hip.h - C ABI of the MI355X-native (gfx950) ArAE decode path.
 *
 * The reference (NVlabs/EdgeRunner) has no plugin/FFI layer on this path; the
 * seam this library replaces is the call
 *     output_ids = self.mesh_decoder.generate(**kwargs)      core/models.py:303
 * (kwargs built at core/models.py:286-301) together with the arithmetic above
 * and below it inside LMM.generate:
 *     encode_cond          core/models.py:101-144  (+ core/transformer/point.py:172-206)
 *     embd(input_ids)      core/models.py:228      (core/transformer/modeling_opt.py:313)
 *     ShapeOPT.forward     core/transformer/modeling_opt.py:464-517 (prefill + cached steps)
 *     logits processors    core/utils.py:118-141, core/models.py:236-275 (grammar)
 *     greedy / top-k=10 sampling, EOS/pad bookkeeping  (transformers 4.46.2 _sample)
 *
 * Conventions
 *   - plain C types only; every "dev" pointer is a device (HBM) pointer owned by
 *     the caller (e.g. torch-ROCm tensor.data_ptr()), every "host" pointer is
 *     ordinary host memory;
 *   - every function returns 0 on success, a negative er_status otherwise; the
 *     message is retrievable with er_last_error() (thread-local);
 *   - no C++ exception crosses the ABI;
 *   - all device work is enqueued on the caller-supplied hipStream_t (passed as
 *     void*; NULL = the default stream).  A context is bound to one device and
 *     is not thread-safe; different contexts are independent;
 *   - weights, KV cache and workspaces are context-owned (hipMalloc).
 */
#ifndef EDGERUNNER_HIP_H
#define EDGERUNNER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ER_ABI_VERSION 1

typedef enum {
    ER_OK = 0,
    ER_ERR_INVALID = -1,      /* bad argument / shape / state        */
    ER_ERR_HIP = -2,          /* a HIP runtime call failed           */
    ER_ERR_MISSING = -3,      /* a required tensor was never loaded  */
    ER_ERR_CAPACITY = -4,     /* KV cache / position table too small */
    ER_ERR_UNSUPPORTED = -5   /* configuration not built             */
} er_status;

/* Element types.  As the dtype of a CHECKPOINT tensor handed to er_load_tensor / er_dit_load_tensor all three are accepted
 * (converted on the device).  As a context's STORAGE precision (er_config.weight_dtype / kv_dtype) two combinations are
 * built: ER_F32 + ER_F32 (exact mode) and ER_F16 + ER_F16 (fast mode, the reference's GPU dtype); ER_BF16 storage is
 * rejected by er_create with ER_ERR_UNSUPPORTED (the reference never runs this path in bf16). */
typedef enum { ER_F32 = 0, ER_F16 = 1, ER_BF16 = 2 } er_dtype;
typedef enum { ER_COND_NONE = 0, ER_COND_POINT = 1, ER_COND_POINT_LATENT = 2 } er_cond_mode;
typedef enum { ER_GREEDY = 0, ER_SAMPLE = 1 } er_gen_mode;
/* prefix_allowed_tokens_fn variants LMM.generate can build (core/models.py:236-275) */
typedef enum {
    ER_GRAMMAR_NONE = 0,      /* prefix_allowed_tokens_fn = None                            */
    ER_GRAMMAR_NAIVE9 = 1,    /* tokenizer is None: ids >= 3, EOS iff len % 9 == 1  (:237-242) */
    ER_GRAMMAR_LR_ABSCO = 2   /* meto LR / LR_ABSCO counter automaton               (:246-271) */
} er_grammar;

/* Model description: mirrors ShapeOPTConfig (core/transformer/modeling_opt.py:86-134)
 * and the Options fields LMM.__init__ reads (core/models.py:32-99). */
typedef struct {
    int32_t hidden_dim, num_heads, num_layers, intermediate_dim;
    int32_t vocab_size, max_positions, num_cond_tokens;
    int32_t point_hidden_dim, point_num_heads, point_latent_size, point_latent_dim;
    int32_t point_freq_dim;      /* columns of point_embed.basis (24)          */
    int32_t num_face_buckets;    /* rows of embed_num_face (10); 0 = no face cond */
    int32_t cond_mode;           /* er_cond_mode                               */
    int32_t pad_token_id, bos_token_id, eos_token_id;
    int32_t weight_dtype;        /* er_dtype the decoder weights are streamed in */
    int32_t kv_dtype;            /* er_dtype of the KV cache                    */
    float   ln_eps;              /* 1e-5 (nn.LayerNorm default)                 */
} er_config;

/* kwargs of mesh_decoder.generate (core/models.py:286-301) that are not tensors. */
typedef struct {
    int32_t mode;                /* er_gen_mode: num_beams=1 | do_sample=True     */
    int32_t top_k;               /* 10 in the reference (sample mode only)         */
    int32_t grammar;             /* er_grammar                                    */
    int32_t max_new_tokens;
    int32_t min_new_tokens;      /* HF MinNewTokensLength semantics; 0 = off      */
    uint64_t seed;               /* Philox key of the device sampler              */
} er_decode_params;

typedef struct er_ctx er_ctx;

int         er_abi_version(void);
const char* er_last_error(void);

int er_create(const er_config* cfg, int device, er_ctx** out);
int er_destroy(er_ctx* ctx);

/* Checkpoint loading: one call per state_dict entry, keyed by the reference's own
 * key names (replaces model.load_state_dict, infer.py:44-50).
 * `data` is host (on_device=0) or device (on_device=1) memory, contiguous,
 * row-major.  Unknown keys are ignored (strict=False) and reported via return
 * value 1.  The tensor is converted to the context's storage dtype and repacked
 * (q/k/v fused per layer). */
int er_load_tensor(er_ctx* ctx, const char* key, const void* data, int dtype,
                   int ndim, const int64_t* shape, int on_device);
/* Verifies every required tensor was provided (ER_ERR_MISSING names the first gap). */
int er_finalize_weights(er_ctx* ctx);

/* Pre-allocates the context-owned KV cache for `batch` sequences of up to
 * `max_len` positions (prefix + generated).  Replaces the per-step torch.cat
 * growth of core/transformer/modeling_opt.py:191-192. */
int er_kv_reserve(er_ctx* ctx, int batch, int max_len);

/* encode_cond (core/models.py:101-144), eval mode.
 *   cond_mode POINT:        conds_dev = float[B, N, 3] point clouds
 *   cond_mode POINT_LATENT: conds_dev = float[B, point_latent_size, point_latent_dim], n_points ignored
 *   cond_mode NONE:         conds_dev ignored
 * face_bucket_host = int[B] = quantize_num_faces(num_faces) (core/utils.py:89-116), ignored when
 * num_face_buckets == 0.  cond_out_dev = float[B, num_cond_tokens, hidden_dim]. */
int er_encode_cond(er_ctx* ctx, const float* conds_dev, int batch, int n_points,
                   const int32_t* face_bucket_host, float* cond_out_dev, void* stream);

/* mesh_decoder.model.embd(input_ids) (core/models.py:228): ids_host int[B*R] -> float[B, R, hidden]. */
int er_embed_tokens(er_ctx* ctx, const int32_t* ids_host, int batch, int n_tokens,
                    float* out_dev, void* stream);

/* First generation step of ShapeOPT (prepare_inputs_for_generation step 0,
 * core/transformer/modeling_opt.py:536-538): runs all layers over
 * inputs_embeds float[B, S, hidden], fills KV positions [0, S) and leaves the
 * hidden state of the last position ready for er_logits / er_decode. */
int er_prefill(er_ctx* ctx, const float* embeds_dev, int batch, int seq_len, void* stream);

/* logits[:, -1, :].float() of the most recent forward (prefill or er_feed): float[B, vocab]. */
int er_logits(er_ctx* ctx, float* logits_out_dev, void* stream);

/* One cached decode step with caller-chosen ids (teacher forcing / host-side
 * logits processors): ShapeOPT(input_ids=ids[:, None], past_key_values=...). */
int er_feed(er_ctx* ctx, const int32_t* ids_host, void* stream);

/* The whole generation loop on device (no host round trip per token): replaces
 * GenerationMixin._sample for the reference's kwargs.  out_ids_dev = int64[B, max_new_tokens]
 * (rows padded with pad_token_id after EOS, as HF does); *n_steps_host = number of
 * columns HF would have returned (stops when every row has emitted EOS).
 * Blocks until the result is complete. */
int er_decode(er_ctx* ctx, const er_decode_params* p, int64_t* out_ids_dev,
              int32_t* n_steps_host, void* stream);
/* Sample mode draws are Philox4x32-10(key = seed, counter = (step, stream id of the row)).  The stream id of row b is b until this
 * call sets it (ids_host: n = reserved batch entries; NULL restores the identity).  A caller that shards or batches the reference's
 * serial loop (infer.py:99-101: file x test_repeat x test_num_face) passes the GLOBAL job index of every row, so a job's tokens do
 * not depend on the world size or on which jobs share its batch (torch.multinomial's global generator gives the reference the
 * same property for its serial loop). */
int er_set_row_streams(er_ctx* ctx, const uint32_t* ids_host, int n);

/* ---- mesh tokenizer (host code, no device work): the reference's pybind11 module meto (meto/src/bindings.cpp,
 * meto.Engine(discrete_bins, verbose, backend), meto/meto/__init__.py:21-54) for the backends Options.meto_backend
 * admits (core/options.py:26). */
enum er_meto_backend { ER_METO_LR_ABSCO = 0, ER_METO_LR = 1 };

/* detokenise - the step right after the decode loop.
 * Engine_LR_ABSCO::decode (meto/include/meto/engine_lr_absco.h:223-295) / Engine_LR::decode
 * (meto/include/meto/engine_lr.h:171-253) + Vertex::undiscrete (meto/include/meto/mesh.h:36-42), reached from
 * save_mesh/detokenize_mesh (core/provider.py:39-66,112-147).  tokens_host = meto ids (model ids - 3, cut at EOS).
 * Capacities: vertices_out float[3*(n/3+3)], faces_out int32[3*(n/4+2)], face_type_out int32[n/4+3]. */
int er_meto_decode(const int32_t* tokens_host, int n_tokens, int discrete_bins, int backend, float* vertices_out,
                   int32_t* faces_out, int32_t* face_type_out, int32_t* n_vertices, int32_t* n_faces,
                   int32_t* n_face_types);

/* Engine_LR_ABSCO::encode (meto/include/meto/engine_lr_absco.h:66-220) / Engine_LR::encode
 * (meto/include/meto/engine_lr.h:59-168) over Mesh::Mesh (meto/include/meto/mesh.h:153-262):
 * vertices float[3*n_vertices] in [-1,1], faces int32[3*n_faces].  Capacities: tokens_out >= 10*n_faces and
 * face_order_out / face_type_out >= n_faces for LR_ABSCO; twice that for LR (it can emit a face twice). */
int er_meto_encode(const float* vertices_host, int n_vertices, const int32_t* faces_host, int n_faces,
                   int discrete_bins, int backend, int32_t* tokens_out, int32_t* n_tokens, int32_t* face_order_out,
                   int32_t* face_type_out, int32_t* n_faces_out);

/* ---- DiT image-conditioned front-end (scope row f3): core/transformer/dit.py + core/models_dit.py::MDiT ----
 * Produces the latents [B, latent_size, latent_dim] that LMM.generate consumes in cond_mode 'point_latent'
 * (infer_dit.py:55,111-113).  fp32, built from the prefill kernels (MFMA GEMM, row softmax) plus k_dit.h. */
typedef struct {
    int32_t hidden_dim, num_heads, num_layers;   /* Options.dit_hidden_dim / dit_num_heads / dit_num_layers */
    int32_t latent_size, latent_dim;             /* point_latent_size (2048), point_latent_dim (64)          */
    int32_t clip_dim;                            /* width of the image encoder's last_hidden_state (1280)   */
    int32_t clip_layers, clip_heads, clip_mlp_dim; /* CLIP ViT-H/14: 32 / 16 / 5120; clip_layers = 0: encoder not loaded */
    int32_t clip_image_size, clip_patch;         /* 224 / 14                                                  */
    int32_t weight_dtype;                        /* ER_F32 (exact) or ER_F16: every Linear on fp16-input MFMA, fp32 accumulate */
} er_dit_config;
typedef struct er_dit_ctx er_dit_ctx;
int er_dit_create(const er_dit_config* cfg, int device, er_dit_ctx** out);
int er_dit_destroy(er_dit_ctx* ctx);
/* MDiT checkpoint keys: "dit.*", "proj_cond.*", "norm_cond.*" (others - image_encoder.*, point_encoder.* - are ignored: returns 1) */
int er_dit_load_tensor(er_dit_ctx* ctx, const char* key, const void* data, int dtype, int ndim,
                       const int64_t* shape, int on_device);
int er_dit_finalize_weights(er_dit_ctx* ctx);
/* MDiT.get_cond after the image encoder: cond = norm_cond(proj_cond(clip_hidden))   core/models_dit.py:113
 * clip_hidden_dev float[B, M, clip_dim] -> cond_out_dev float[B, M, hidden_dim] */
int er_dit_project_cond(er_dit_ctx* ctx, const float* clip_hidden_dev, int batch, int m_tokens,
                        float* cond_out_dev, void* stream);
/* The frozen image encoder of MDiT.get_cond (core/models_dit.py:104-111): normalize + bilinear resize to 224 +
 * CLIPVisionModel(...).last_hidden_state.  images_dev float[B,3,H,W] in [0,1] -> clip_hidden_out_dev float[B, 257, clip_dim].
 * Checkpoint keys "image_encoder.vision_model.*" (transformers 4.46.2 names). */
int er_dit_encode_image(er_dit_ctx* ctx, const float* images_dev, int batch, int height, int width,
                        float* clip_hidden_out_dev, void* stream);
/* DiT.forward(x, c, t)   core/transformer/dit.py:168-196: x float[B,N,latent_dim], c float[B,M,hidden], t_host float[B] */
int er_dit_forward(er_dit_ctx* ctx, const float* x_dev, const float* c_dev, const float* t_host, int batch,
                   int m_tokens, float* out_dev, void* stream);
/* MDiT.run's denoise loop (core/models_dit.py:184-229; num_repeat is a repeat_interleave of cond on the caller's
 * side): DDIM (v-prediction, scaled-linear betas 0.00085..0.012, leading spacing, steps_offset 1, eta 0) with
 * classifier-free guidance over [zeros | cond], over scheduler.timesteps[init_step:] (init_step = 0: from pure
 * noise; > 0: the img2img branch, :207-209, the caller has already added noise at timesteps[init_step]).
 * latents_dev float[B, N, latent_dim] holds the starting latents on entry and the result on exit. */
int er_dit_sample(er_dit_ctx* ctx, const float* cond_dev, int batch, int m_tokens, float* latents_dev,
                  int num_inference_steps, float guidance_scale, int init_step, void* stream);

/* ---- which kernels a decode context of this shape runs (pure host logic: callable without a device) ----
 * The rules live in ONE function that er_kv_reserve applies and this entry point reports; the environment knobs of
 * er_create (ER_DECODE_V, ER_ATTN_V_BATCHED, ER_FORCE_BATCHED) are honoured, l_cap is rounded up to 32 like er_kv_reserve. */
typedef enum { ER_ATTN_SPLIT1 = 1, ER_ATTN_SPLIT2 = 2, ER_ATTN_BALANCED = 3, ER_ATTN_STREAM = 4 } er_attn_kernel;
typedef struct {
    int32_t batched;            /* 1: B > 4 path (matrix-core projections, weights once per 32 rows) */
    int32_t decode_version;     /* single-row path: 3 = balanced chunks + merge fused into out_proj, 2 = fixed chunks + merge kernel */
    int32_t attn_kernel;        /* er_attn_kernel */
    int32_t attn_chunks;        /* chunks per head of the balanced kernel */
    int32_t merge_launch;       /* 1: the attention partials are merged by their own launch */
    int32_t launches_per_layer; /* single-row path only (0 when batched: the count depends on the row passes); a token is
                                   layers x this + lm_head + sample_head launches */
} er_decode_plan;
int er_plan_decode(int batch, int heads, int head_dim, int hidden, int l_cap, er_decode_plan* out);
/* the plan of a live context (knobs as read by er_create, cache as reserved by the last er_kv_reserve) */
int er_ctx_plan(er_ctx* ctx, er_decode_plan* out);
/* tile shape launch_gemm* picks for an [m, n] output in `batch` slices: 1 = 128x128, 2 = 64x128, 3 = 64x64 (ER_GEMM_TILE forces) */
int er_plan_gemm_tile(int m, int n, int batch);

/* ---- measurement ---- */
#define ER_NUM_KERNEL_KINDS 8
/* kinds: 0 qkv_gemv 1 attn_decode 2 attn_combine 3 out_proj_gemv 4 fc1_gemv 5 fc2_gemv 6 lm_head_gemv 7 sample_head */
const char* er_kernel_kind_name(int kind);
/* Times each decode kernel kind in place (HIP events on `stream`, eager launches
 * sweeping all layers so weights are not cache-resident): avg_us_out[kind] =
 * average duration of ONE launch, bytes_out[kind] = algorithmic HBM bytes of one
 * launch at the current context length.  Does not advance the generation state. */
int er_profile_decode_kernels(er_ctx* ctx, int repeats, float* avg_us_out, double* bytes_out, void* stream);
/* Same sweep with the attention kernels run at `context_len` keys (<= the reserved capacity; 0 = the current context
 * length): lets bench.py time the dominant kernel at the MEAN context length of the run it timed, which is what a
 * rocprofv3 --stats average over that run reports.  The launches are captured into a hipGraph and replayed, like the
 * generation loop does (`use_graph` != 0), or issued eagerly. */
int er_profile_decode_kernels_at(er_ctx* ctx, int repeats, int context_len, int use_graph, float* avg_us_out,
                                 double* bytes_out, void* stream);
/* Milliseconds spent inside the last er_decode between its first and last step (HIP events). */
int er_last_decode_ms(er_ctx* ctx, float* ms_out);

/* ---- single-kernel entry points (unit tests call these through the ABI) ---- */
/* y[b,n] = act(sum_k W[n,k] x[b,k] + bias[n]) (+resid) ; ln_w != NULL -> x = LayerNorm(x) first.
 * batch <= 4: the single-row decode kernels; batch > 4: the batched ones exactly as the decode step picks them
 * (matrix-core kernels on a tiled copy of W for the fc1- and fc2-shaped cases, VALU kernels for the narrow ones;
 * env ER_BATCHED_VALU=1 forces the VALU kernels) */
int er_k_gemv(const float* w_dev, const float* bias_dev, const float* x_dev, const float* ln_w_dev,
              const float* ln_b_dev, const float* resid_dev, float* y_dev, float* xnorm_out_dev,
              int batch, int n, int k, int relu, float eps, void* stream);
/* softmax(q K^T / sqrt(D)) V for one new token over a [B,H,Lcap,D] cache holding len[b] keys;
 * steps in {2,4,8}: one workgroup per chunk of 32*steps keys (split kernels); variant = ER_ATTN_SPLIT2 (the single-row fallback:
 * per-wave softmax + one-round-trip merge kernel), ER_ATTN_SPLIT1 (the leaner split kernel of mid-size batches + the same merge)
 * or ER_ATTN_STREAM (one workgroup per (row, head) walks the whole key range, head_dim 96 only) */
int er_k_attn_decode(const float* q_dev, const void* k_dev, const void* v_dev, const int32_t* len_host,
                     float* out_dev, int batch, int heads, int head_dim, int l_cap, int steps, int kv_half,
                     int variant, void* stream);
/* Version 3 of the single-row decode attention (env ER_DECODE_V=3), one row, 16 heads of 96:
 * y[1536] = Wo . softmax(q K^T / sqrt(D)) V + bo + resid over a [16,Lcap,96] cache holding len keys (Lcap <= 8192):
 * balanced chunks (16 per head) + the partial merge fused into the out_proj GEMV; w_half: Wo is fp16 */
int er_k_attn_outproj3(const float* q_dev, const void* k_dev, const void* v_dev, int len, const void* wo_dev,
                       const float* bo_dev, const float* resid_dev, float* y_dev, int l_cap, int kv_half, int w_half,
                       void* stream);
/* C[M,N] = A[M,K] op(B) (+bias)(relu)(+resid); b_is_kn=0: B is [N,K] (Linear weight), 1: B is [K,N] */
int er_k_gemm(const float* a_dev, const float* b_dev, const float* bias_dev, const float* resid_dev,
              float* c_dev, int m, int n, int k, int lda, int ldb, int ldc, int b_is_kn, int relu,
              float div, void* stream);
/* C[M,N] = relu?(fp16(A fp32 [M,K]) . W fp16 [N,K]^T + bias) (+resid): the fp16-input MFMA GEMM (k % 32 == 0) */
int er_k_gemm_f16(const float* a_dev, const void* w_half_dev, const float* bias_dev, const float* resid_dev,
                  float* c_dev, int m, int n, int k, int lda, int ldb, int ldc, int relu, void* stream);
/* Same shape with the fp32 activations split into fp16 hi + lo parts on their way to the matrix cores (fp32-grade operand,
 * two fp16 MFMAs per fragment): the fast-mode prefill Linears (fp16-stored weights x fp32 activations, fp32 accumulate). */
/* the same product with BOTH operands in fp16 brought in by LDS-DMA (k % 64 == 0): a is rounded to an fp16 copy first (in the
 * DiT path the producing kernel writes that copy); c16_out_dev (optional) receives the result rounded to fp16 [m][n] */
int er_k_gemm_hh(const float* a_dev, const void* w_half_dev, const float* bias_dev, const float* resid_dev, float* c_dev,
                 void* c16_out_dev, int m, int n, int k, int lda, int ldb, int ldc, int relu, void* stream);
/* the fused q/k/v projection of the DiT self-attention (core/transformer/dit.py:100-126 -> attention.py:176-178) as the LDS-DMA
 * kernel runs it: n = 3 * heads * 64 output columns, the first two thirds (q, k) rounded to fp16 into qk16_out_dev [m][n] (its V
 * columns are left untouched), the V third written by the GEMM epilogue as V^T per head into vt_out_dev
 * [m / rows_per_batch][heads][64][rows_per_batch], keys in the order flash_attn_hh_kernel reads them (position of key k inside its
 * group of 16: {0-3, 8-11, 4-7, 12-15}).  m, rows_per_batch multiples of 64; force_tile 0 = the product rule, 1 / 2 / 3 = 128x128 /
 * 64x128 / 64x64 tiles (4 waves), 4 = 256x256 (8 waves). */
int er_k_gemm_hh_qkv(const float* a_dev, const void* w_half_dev, const float* bias_dev, void* qk16_out_dev, void* vt_out_dev, int m,
                     int n, int k, int rows_per_batch, int force_tile, void* stream);
/* the GEGLU feed-forward-in of the DiT block (core/transformer/dit.py FeedForward: x, gate = Linear(h).chunk(2); x * gelu(gate),
 * erf form) as the LDS-DMA kernels run it: out16[m][f] = fp16((fp16(a) . Wx^T + bx) * gelu(fp16(a) . Wg^T + bg)) with w_half_dev the
 * [2 f][k] fp16 weight in checkpoint order; the [m][2 f] pre-activation never reaches HBM.  force_tile 0 = the product rule,
 * 1 / 2 = 128x128 / 64x128 tiles (4 waves), 4 = 256x256 (8 waves, f % 128 == 0): all forms are bit-identical. */
int er_k_gemm_hh_geglu(const float* a_dev, const void* w_half_dev, const float* bias_dev, void* out16_dev, int m, int f, int k,
                       int force_tile, void* stream);
int er_k_gemm_f16s(const float* a, const void* w_half, const float* bias, const float* resid, float* c, int m, int n, int k,
                   int lda, int ldb, int ldc, int relu, void* stream);
/* softmax(q k^T / 8) v, head_dim 64, non-causal, fp16 operands / fp32 accumulate; q,o [B,N,H*64], k,v [B,M,H*64] fp32 */
int er_k_flash_attn_f16(const float* q_dev, const float* k_dev, const float* v_dev, float* o_dev, int batch,
                        int heads, int n_queries, int m_keys, void* stream);
/* softmax(q k^T / sqrt(D) [+ causal mask]) v in exact fp32 on the f32-input matrix cores, no score matrix in HBM
 * (csrc/k_flash_attn_f32.h; replaces attention() of core/transformer/attention.py:27-62 for the prefill, N == M causal,
 * head_dim 96, and for the point encoder's cross-attention, head_dim 64).  q/o: [B, N, H*D], k/v: [B, M, H*D]. */
int er_k_flash_attn_f32(const float* q, const float* k, const float* v, float* o, int batch, int heads, int n, int m,
                        int head_dim, int causal, void* stream);
/* the same attention with q / k / v in fp16 brought in by LDS-DMA (V transposed per head): the unit entry converts and transposes
 * the fp32 inputs first and widens the fp16 output */
int er_k_flash_attn_hh(const float* q_dev, const float* k_dev, const float* v_dev, float* o_dev, int batch, int heads, int n, int m,
                       void* stream);
/* the attention of er_k_flash_attn_f32 for head_dim 96 on the fp16 matrix cores with hi/lo-split q and p (the fast-mode prefill's
 * prefix attention, csrc/k_flash_attn_f16s.h); k / v must hold fp16-representable values (the fast-mode prefill's scratch does) */
int er_k_flash_attn_f16s(const float* q_dev, const float* k_dev, const float* v_dev, float* o_dev, int batch, int heads,
                         int n_queries, int m_keys, int causal, void* stream);
int er_k_layernorm(const float* x_dev, const float* w_dev, const float* b_dev, float* y_dev,
                   int rows, int cols, float eps, void* stream);
/* rows of scores[rows, ld]: softmax over the first n_valid(row) columns (causal: row+1+causal_offset), zeros after */
int er_k_softmax(float* s_dev, int rows, int cols, int ld, int causal, void* stream);
/* one sampling-head step on given logits float[B,V]; state arrays are int[B] on device */
int er_k_sample_head(const float* logits_dev, const er_decode_params* p, int vocab, int eos, int pad,
                     int batch, int step, const int32_t* last_tok_host, const int32_t* counter_host,
                     const int32_t* unfinished_host, int32_t* next_tok_host, int32_t* counter_out_host,
                     int32_t* unfinished_out_host, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EDGERUNNER_HIP_H */
