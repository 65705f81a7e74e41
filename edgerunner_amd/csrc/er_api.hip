// C-ABI implementation (see include/edgerunner_hip.h): context, checkpoint loading,
// KV cache, prefill, point encoder, device-side generation loop (hipGraph replay).
// Single translation unit: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC.
#include "../../include/edgerunner_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "er_common.h"
#include "k_attn_decode.h"
#include "k_outproj_merge.h"
#include "k_flash_attn_f16s.h"
#include "k_gemm.h"
#include "k_gemm_stream.h"
#include "k_gemv.h"
#include "k_gemv_mfma.h"
#include "k_head.h"
#include "k_rowops.h"
#include "k_flash_attn.h"
#include "k_flash_attn_f32.h"
#include "meto_decode.h"
#include "meto_encode.h"

using namespace er;

// ------------------------------------------------------------------------------------ errors
static thread_local char g_err[1024] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(ER_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
    } while (0)
#define ERCHK(expr)                 \
    do {                            \
        int r_ = (expr);            \
        if (r_ < 0) return r_;      \
    } while (0)

// ------------------------------------------------------------------------------------ context
struct LayerW {
    // fast mode (fp16 storage): *_h hold the streamed fp16 matrices; the fp32 copies then hold the SAME
    // fp16-rounded values (used by the prefill GEMMs), so prefill and decode see one model
    _Float16 *wqkv_h = nullptr, *wo_h = nullptr, *w1_h = nullptr, *w2_h = nullptr;
    void *wqkv_t = nullptr, *wo_t = nullptr, *w1_t = nullptr, *w2_t = nullptr;   // tiled copies for the matrix-core batched kernels (k_gemv_mfma.h)
    float *wqkv = nullptr, *bqkv = nullptr;   // fused [3*hidden][hidden] in q,k,v order
    float *wo = nullptr, *bo = nullptr, *ln1w = nullptr, *ln1b = nullptr;
    float *w1 = nullptr, *b1 = nullptr, *w2 = nullptr, *b2 = nullptr, *ln2w = nullptr, *ln2b = nullptr;
};

struct Buf {   // grow-only device scratch
    float* p = nullptr;
    size_t n = 0;
};

struct er_ctx {
    er_config cfg{};
    int device = 0;
    int D = 0;               // head_dim
    hipStream_t own_stream = nullptr;
    // weights
    std::vector<LayerW> layers;
    float *embd = nullptr, *posemb = nullptr, *lm_head = nullptr, *embed_num_face = nullptr;
    _Float16* lm_head_h = nullptr;
    bool fast = false;       // fp16 weights + fp16 KV cache, fp32 accumulate
    float *proj_w = nullptr, *proj_b = nullptr, *normc_w = nullptr, *normc_b = nullptr;
    // point encoder
    float *pe_query = nullptr, *pe_basis = nullptr, *pe_mlp_w = nullptr, *pe_mlp_b = nullptr, *pe_ln_w = nullptr,
          *pe_ln_b = nullptr;
    float *ca_ln1_w = nullptr, *ca_ln1_b = nullptr, *ca_ln2_w = nullptr, *ca_ln2_b = nullptr;
    float *ca_q_w = nullptr, *ca_q_b = nullptr, *ca_k_w = nullptr, *ca_k_b = nullptr, *ca_v_w = nullptr, *ca_v_b = nullptr,
          *ca_o_w = nullptr, *ca_o_b = nullptr;
    float *ff0_w = nullptr, *ff0_b = nullptr, *ff2_w = nullptr, *ff2_b = nullptr, *lin_w = nullptr, *lin_b = nullptr;
    int pe_kpad = 0;         // padded input width of point_embed.mlp (51 -> 64)
    std::map<std::string, bool> need;   // required keys -> loaded?
    std::vector<void*> owned;           // every hipMalloc'd weight block
    // KV cache
    int B = 0, Lcap = 0, S_splits = 0;
    void *kc = nullptr, *vc = nullptr;        // [layers][B][H][Lcap][D], fp32 or fp16 (fast)
    int kv_esz = 4;
    long long kv_bstride = 0, kv_lstride = 0;
    // decode workspace ([B][...])
    float *ypre = nullptr, *hbuf = nullptr, *ypre1 = nullptr, *h1buf = nullptr, *qbuf = nullptr, *abuf = nullptr,
          *fbuf = nullptr, *logits = nullptr, *part = nullptr;
    int* state_block = nullptr;   // backing store of GenState
    GenState st{};
    DecodeParamsDev* d_params = nullptr;
    int* d_ids_tmp = nullptr;
    unsigned int* d_row_stream = nullptr;   // [B] Philox stream id per row (identity until er_set_row_streams)
    long long* d_out_ids = nullptr;   // [B][Lcap] generated ids (graph writes here; copied to the caller at the end)
    int* h_pinned = nullptr;      // small pinned host buffer
    int base_pos = 0;             // prefill length of the current generation
    bool have_hidden = false;     // ypre holds a valid last-position state
    // hipGraph of one step
    hipGraphExec_t step_exec = nullptr;
    bool use_graph = true;
    bool batched = false;     // B > 4 (or ER_FORCE_BATCHED=1): weights streamed once per pass of 32 rows (matrix cores)
    bool batched_valu = false;   // ER_BATCHED_VALU=1: the older VALU kernels (one pass per 16 rows), kept for A/B runs
    float* skpart = nullptr;  // split-K partials of the batched projections
    size_t skpart_floats = 0; // ... and how many floats the block holds (checked by gemv_mfma_groups)
    // fast-mode batches on the matrix cores: activations in the tiled hi | lo operand layout (k_gemv.h xt_entry), one image per producer
    void *xt_h = nullptr, *xt_att = nullptr, *xt_f = nullptr;     // LayerNorm rows (qkv / fc1 input), attention output, fc1 output
    bool xt = false;          // ER_XT=0 keeps the row-major fp32 inputs (A/B + parity matrix)
    bool tiled_valid = false; // LayerW::*_t match the loaded weights
    // waves per workgroup of the qkv / fc1 GEMVs (env ER_NW_QKV: 4, 6 or 9; ER_NW_FC1: 4 or 12).  Exact mode: qkv 6 waves x 1 row
    // = 768 workgroups (3 per CU), fc1 4 waves x 2 rows = 768; fast mode: one fat workgroup per CU (9 / 12 waves x 2 rows).  The other
    // shapes are fixed (out_proj 3 waves x 1 row, fc2 4 K-slices x 2 rows, 128-key chunks for the fixed-chunk attention): their
    // round-1/2 knobs (ER_RW_*, ER_NW_OUT, ER_ATTN_STEPS, ER_ATTN_V, ER_COMBINE_V, ER_ATTN_GRID_HS, ER_OUT_VALU) are settled and gone
    int nw_qkv = 6, nw_fc1 = 4;
    int prefill_attn_f16s = -1;   // fast-mode prefix attention on the fp16 matrix cores with hi/lo-split q and p (k_flash_attn_f16s.h): on unless
                                  // ER_PREFILL_ATTN_F16S=0 (the fp32-matrix-core kernel; kept for the parity matrix)
    int prof_len = 0;         // > 0: attention kernels run at this fixed length (er_profile_decode_kernels_at)
    int attn_v_batched = 0;   // attention kernel at B > 4 (env ER_ATTN_V_BATCHED): 0 = auto (streaming when B*H >= 256, else split + merge), 1 = split kernel + merge, 3 = one streaming workgroup per (row, head), no merge
    bool stream_attn = false; // batched, D == 96 and (forced or B*H >= 256: at least one streaming workgroup per CU)
    int decode_v = 3;         // single-row decode: 3 = balanced-chunk attention + merge fused into out_proj (one row, D = 96, 16 heads, Lcap <= 8192); ER_DECODE_V=2 = fixed 128-key chunks + merge kernel (also the fallback when the cache does not qualify)
    bool v3 = false;          // decode_v == 3 and the reserved cache qualifies
    int nch3 = 0;             // chunks per head of the balanced attention kernel
    float* part_ml = nullptr; // version 3: {m, l} of the partials
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_decode_ms = 0.f;
    // scratch for prefill / encoder
    Buf p_hi, p_lo;           // fast-mode prefill: hi / lo fp16 halves of the activation a Linear is about to read (LDS-DMA GEMM, split form)
    Buf p_h, p_q, p_a, p_y, p_f, p_sc, p_qkv, p_ap, p_aml, e_a0, e_x, e_k, e_v, e_qln, e_q, e_sc, e_att, e_l, e_ln, e_u, e_g, e_lat, e_tmp, e_ids, e_stage;
};

constexpr int ER_MAX_BATCH = 1023;   // h_pinned holds B ints + one flag
constexpr int NBM = 32;   // batch rows per pass of the matrix-core decode projections (k_gemv_mfma.h)

static int ensure(Buf& b, size_t n) {
    if (b.n >= n) return 0;
    if (b.p) hipFree(b.p);
    b.p = nullptr;
    b.n = 0;
    HIPCHK(hipMalloc(&b.p, n * sizeof(float)));
    b.n = n;
    return 0;
}

static hipStream_t pick(er_ctx* c, void* s) { return s ? (hipStream_t)s : c->own_stream; }

extern "C" int er_abi_version(void) { return ER_ABI_VERSION; }
extern "C" const char* er_last_error(void) { return g_err; }

static const char* kKindNames[ER_NUM_KERNEL_KINDS] = {"qkv_gemv", "attn_decode", "attn_combine", "out_proj_gemv",
                                                      "fc1_gemv", "fc2_gemv", "lm_head_gemv", "sample_head"};
extern "C" const char* er_kernel_kind_name(int k) { return (k >= 0 && k < ER_NUM_KERNEL_KINDS) ? kKindNames[k] : "?"; }

static void register_keys(er_ctx* c) {
    auto& n = c->need;
    const er_config& g = c->cfg;
    auto lin = [&](const std::string& p, bool bias = true) {
        n[p + ".weight"] = false;
        if (bias) n[p + ".bias"] = false;
    };
    if (g.cond_mode == ER_COND_POINT) {
        const std::string pe = "point_encoder";
        n[pe + ".query_embed"] = false;
        n[pe + ".point_embed.basis"] = false;
        lin(pe + ".point_embed.mlp");
        lin(pe + ".ln");
        lin(pe + ".cross_att.ln1");
        lin(pe + ".cross_att.ln2");
        for (const char* p : {"q_proj", "k_proj", "v_proj", "out_proj"}) lin(pe + ".cross_att.att." + p);
        lin(pe + ".cross_att.mlp.net.0");
        lin(pe + ".cross_att.mlp.net.2");
        lin(pe + ".linear");
    }
    if (g.cond_mode != ER_COND_NONE) {
        lin("proj_cond");
        lin("norm_cond");
    }
    if (g.num_face_buckets > 0) n["embed_num_face.weight"] = false;
    n["mesh_decoder.model.embd.weight"] = false;
    n["mesh_decoder.model.embed_positions.weight"] = false;
    for (int i = 0; i < g.num_layers; ++i) {
        const std::string L = "mesh_decoder.model.layers." + std::to_string(i);
        for (const char* p : {"k_proj", "v_proj", "q_proj", "out_proj"}) lin(L + ".self_attn." + p);
        lin(L + ".self_attn_layer_norm");
        lin(L + ".fc1");
        lin(L + ".fc2");
        lin(L + ".final_layer_norm");
    }
    n["mesh_decoder.lm_head.weight"] = false;
}

static int dev_alloc(er_ctx* c, float** p, size_t n) {
    HIPCHK(hipMalloc(p, n * sizeof(float)));
    c->owned.push_back(*p);
    return 0;
}

extern "C" int er_create(const er_config* cfg, int device, er_ctx** out) {
    if (!cfg || !out) return fail(ER_ERR_INVALID, "er_create: null argument");
    if (cfg->hidden_dim != 1536 || cfg->intermediate_dim != 6144)
        return fail(ER_ERR_UNSUPPORTED, "this build streams hidden_dim=1536 / intermediate_dim=6144 (ArAE, DiT presets); got %d/%d",
                    cfg->hidden_dim, cfg->intermediate_dim);
    if (cfg->num_heads <= 0 || cfg->hidden_dim % cfg->num_heads) return fail(ER_ERR_INVALID, "hidden_dim %% num_heads != 0");
    const int D = cfg->hidden_dim / cfg->num_heads;
    if (D != 96 && D != 64) return fail(ER_ERR_UNSUPPORTED, "head_dim %d not built (96, 64)", D);
    const bool exact = cfg->weight_dtype == ER_F32 && cfg->kv_dtype == ER_F32;
    const bool fast = cfg->weight_dtype == ER_F16 && cfg->kv_dtype == ER_F16;
    if (!exact && !fast)
        return fail(ER_ERR_UNSUPPORTED, "built modes: fp32 weights + fp32 KV (exact) or fp16 weights + fp16 KV (fast)");
    if (cfg->vocab_size > ER_HEAD_MAX_VOCAB) return fail(ER_ERR_UNSUPPORTED, "vocab_size > %d", ER_HEAD_MAX_VOCAB);
    if (cfg->cond_mode == ER_COND_POINT && (cfg->point_hidden_dim != 1024 || cfg->point_hidden_dim % cfg->point_num_heads))
        return fail(ER_ERR_UNSUPPORTED, "point encoder width %d not built (1024)", cfg->point_hidden_dim);
    HIPCHK(hipSetDevice(device));
    er_ctx* c = new er_ctx();
    c->cfg = *cfg;
    c->device = device;
    c->D = D;
    c->fast = fast;
    c->kv_esz = fast ? 2 : 4;
    c->layers.resize(cfg->num_layers);
    const char* ng = getenv("ER_NO_GRAPH");
    c->use_graph = !(ng && ng[0] == '1');
    auto env_int = [](const char* name, int dflt) { const char* v = getenv(name); return (v && v[0]) ? atoi(v) : dflt; };
    // one fat workgroup per CU (qkv 9 waves x 2 rows, fc1 12 waves x 2 rows) pays with fp16 weights only: the LayerNorm prologue and
    // its 18 KB of x / affine reads are then once per CU instead of three times (profiles/r03_fat_workgroups.log: fp16 +2.7 %, fp32 +-0)
    c->nw_qkv = env_int("ER_NW_QKV", fast ? 9 : 6);
    if (c->nw_qkv != 4 && c->nw_qkv != 9) c->nw_qkv = 6;
    c->nw_fc1 = env_int("ER_NW_FC1", fast ? 12 : 4) == 12 ? 12 : 4;
    c->prefill_attn_f16s = env_int("ER_PREFILL_ATTN_F16S", -1);
    if (c->prefill_attn_f16s != 0 && c->prefill_attn_f16s != 1) c->prefill_attn_f16s = -1;
    c->attn_v_batched = env_int("ER_ATTN_V_BATCHED", 0);
    if (c->attn_v_batched != 1 && c->attn_v_batched != 3) c->attn_v_batched = 0;
    c->decode_v = env_int("ER_DECODE_V", 3) == 2 ? 2 : 3;
    HIPCHK(hipStreamCreateWithFlags(&c->own_stream, hipStreamDefault));
    HIPCHK(hipEventCreate(&c->ev0));
    HIPCHK(hipEventCreate(&c->ev1));
    HIPCHK(hipHostMalloc((void**)&c->h_pinned, 4096, hipHostMallocDefault));
    register_keys(c);
    *out = c;
    return ER_OK;
}

static void free_kv(er_ctx* c) {
    for (void* p : {c->kc, c->vc, (void*)c->ypre, (void*)c->hbuf, (void*)c->ypre1, (void*)c->h1buf, (void*)c->qbuf,
                    (void*)c->abuf, (void*)c->fbuf, (void*)c->logits, (void*)c->part})
        if (p) hipFree(p);
    c->kc = c->vc = c->ypre = c->hbuf = c->ypre1 = c->h1buf = c->qbuf = c->abuf = c->fbuf = c->logits = c->part = nullptr;
    if (c->skpart) hipFree(c->skpart);
    c->skpart = nullptr;
    for (void** p : {&c->xt_h, &c->xt_att, &c->xt_f}) {
        if (*p) hipFree(*p);
        *p = nullptr;
    }
    if (c->part_ml) hipFree(c->part_ml);
    c->part_ml = nullptr;
    if (c->state_block) hipFree(c->state_block);
    if (c->d_params) hipFree(c->d_params);
    if (c->d_ids_tmp) hipFree(c->d_ids_tmp);
    if (c->d_row_stream) hipFree(c->d_row_stream);
    c->d_row_stream = nullptr;
    if (c->d_out_ids) hipFree(c->d_out_ids);
    c->d_out_ids = nullptr;
    c->state_block = nullptr; c->d_params = nullptr; c->d_ids_tmp = nullptr;
    if (c->step_exec) hipGraphExecDestroy(c->step_exec);
    c->step_exec = nullptr;
    c->B = 0; c->Lcap = 0;
}

extern "C" int er_destroy(er_ctx* c) {
    if (!c) return ER_OK;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    free_kv(c);
    for (void* p : c->owned) hipFree(p);
    for (Buf* b : {&c->p_hi, &c->p_lo, &c->p_h, &c->p_q, &c->p_a, &c->p_y, &c->p_f, &c->p_sc, &c->p_qkv, &c->p_ap, &c->p_aml, &c->e_a0, &c->e_x, &c->e_k, &c->e_v, &c->e_qln,
                   &c->e_q, &c->e_sc, &c->e_att, &c->e_l, &c->e_ln, &c->e_u, &c->e_g, &c->e_lat, &c->e_tmp, &c->e_ids, &c->e_stage})
        if (b->p) hipFree(b->p);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (c->h_pinned) hipHostFree(c->h_pinned);
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    delete c;
    return ER_OK;
}

// ------------------------------------------------------------------------------------ weights
static std::vector<float> to_f32_host(const void* data, int dtype, size_t n, int on_device, int* err) {
    std::vector<float> out(n);
    const size_t esz = (dtype == ER_F32) ? 4 : 2;
    std::vector<unsigned char> raw;
    const unsigned char* src = (const unsigned char*)data;
    if (on_device) {
        raw.resize(n * esz);
        if (hipMemcpy(raw.data(), data, n * esz, hipMemcpyDeviceToHost) != hipSuccess) { *err = 1; return out; }
        src = raw.data();
    }
    if (dtype == ER_F32) {
        memcpy(out.data(), src, n * 4);
    } else if (dtype == ER_BF16) {
        const uint16_t* h = (const uint16_t*)src;
        for (size_t i = 0; i < n; ++i) { uint32_t u = (uint32_t)h[i] << 16; memcpy(&out[i], &u, 4); }
    } else {  // IEEE fp16
        const uint16_t* h = (const uint16_t*)src;
        for (size_t i = 0; i < n; ++i) {
            const uint32_t s = (h[i] >> 15) & 1, e = (h[i] >> 10) & 31, m = h[i] & 1023;
            uint32_t u;
            if (e == 0) {
                if (m == 0) u = s << 31;
                else { int sh = 0; uint32_t mm = m; while (!(mm & 1024)) { mm <<= 1; ++sh; } u = (s << 31) | ((127 - 15 - sh + 1) << 23) | ((mm & 1023) << 13); }
            } else if (e == 31) u = (s << 31) | 0x7f800000u | (m << 13);
            else u = (s << 31) | ((e - 15 + 127) << 23) | (m << 13);
            memcpy(&out[i], &u, 4);
        }
    }
    return out;
}

static bool ends_with(const std::string& s, const char* suf) {
    const size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

// dtype conversion on the device (round 1 converted every tensor through a host fp32 vector: 7.4 s for the 2.7 GB
// checkpoint, most of it scalar fp16 loops)
__device__ __forceinline__ float raw_to_f32(const void* src, int dtype, size_t i) {
    if (dtype == ER_F32) return reinterpret_cast<const float*>(src)[i];
    if (dtype == ER_F16) return (float)reinterpret_cast<const _Float16*>(src)[i];
    const unsigned int u = (unsigned int)reinterpret_cast<const unsigned short*>(src)[i] << 16;      // bf16
    return __uint_as_float(u);
}
__global__ void cvt_f32_kernel(const void* src, int dtype, float* dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = raw_to_f32(src, dtype, i);
}
// streamed decoder matrix in fast mode: the fp16 copy (round to nearest even) and an fp32 copy of the SAME rounded values
__global__ void cvt_streamed_kernel(const void* src, int dtype, float* dst32, _Float16* dst16, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const _Float16 hv = (_Float16)raw_to_f32(src, dtype, i);
        dst16[i] = hv;
        dst32[i] = (float)hv;
    }
}

extern "C" int er_load_tensor(er_ctx* c, const char* key_c, const void* data, int dtype, int ndim, const int64_t* shape,
                              int on_device) {
    if (!c || !key_c || !data || ndim < 1 || ndim > 4) return fail(ER_ERR_INVALID, "er_load_tensor: bad argument");
    if (dtype != ER_F32 && dtype != ER_F16 && dtype != ER_BF16) return fail(ER_ERR_INVALID, "er_load_tensor: dtype %d", dtype);
    HIPCHK(hipSetDevice(c->device));
    const std::string key(key_c);
    auto it = c->need.find(key);
    if (it == c->need.end()) return 1;   // strict=False: unknown keys are ignored
    c->tiled_valid = false;              // any reload invalidates the tiled copies of the batched path
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
    const er_config& g = c->cfg;
    const int H = g.hidden_dim, I = g.intermediate_dim, PH = g.point_hidden_dim;
    const size_t esz = (dtype == ER_F32) ? 4 : 2;
    // the tensor's raw bytes on the device: the caller's buffer, or one upload into the grow-only staging block
    const void* src = data;
    if (!on_device) {
        ERCHK(ensure(c->e_stage, (n * esz + 3) / 4));
        HIPCHK(hipMemcpy(c->e_stage.p, data, n * esz, hipMemcpyHostToDevice));
        src = c->e_stage.p;
    }
    const unsigned cgrid = (unsigned)std::min<size_t>((n + 255) / 256, 4096);

    auto expect = [&](size_t want) -> int {
        if (n != want) return fail(ER_ERR_INVALID, "er_load_tensor(%s): %zu elements, expected %zu", key_c, n, want);
        return 0;
    };
    auto finish = [&]() -> int {          // the staging block / caller buffer may be reused as soon as we return
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(c->own_stream));
        return 0;
    };
    auto put = [&](float** dst, size_t want) -> int {   // plain copy into a fresh block
        ERCHK(expect(want));
        if (!*dst) ERCHK(dev_alloc(c, dst, want));
        hipLaunchKernelGGL(cvt_f32_kernel, dim3(cgrid), dim3(256), 0, c->own_stream, src, dtype, *dst, want);
        return finish();
    };
    // streamed decoder matrix: fp32 copy (+ fp16 copy and fp16-rounded fp32 values in fast mode)
    auto put_w = [&](float** dst, _Float16** dst_h, size_t total, size_t off, size_t want) -> int {
        ERCHK(expect(want));
        if (!*dst) { ERCHK(dev_alloc(c, dst, total)); HIPCHK(hipMemset(*dst, 0, total * 4)); }
        if (c->fast) {
            if (!*dst_h) {
                HIPCHK(hipMalloc((void**)dst_h, total * 2));
                c->owned.push_back(*dst_h);
                HIPCHK(hipMemset(*dst_h, 0, total * 2));
            }
            hipLaunchKernelGGL(cvt_streamed_kernel, dim3(cgrid), dim3(256), 0, c->own_stream, src, dtype, *dst + off, *dst_h + off, want);
        } else {
            hipLaunchKernelGGL(cvt_f32_kernel, dim3(cgrid), dim3(256), 0, c->own_stream, src, dtype, *dst + off, want);
        }
        return finish();
    };
    auto put_at = [&](float** dst, size_t total, size_t off, size_t want) -> int {   // slice of a fused block
        ERCHK(expect(want));
        if (!*dst) { ERCHK(dev_alloc(c, dst, total)); HIPCHK(hipMemset(*dst, 0, total * 4)); }
        hipLaunchKernelGGL(cvt_f32_kernel, dim3(cgrid), dim3(256), 0, c->own_stream, src, dtype, *dst + off, want);
        return finish();
    };

    int rc = 0;
    const std::string dec = "mesh_decoder.model.layers.";
    if (key.rfind(dec, 0) == 0) {
        const size_t dot = key.find('.', dec.size());
        const int li = atoi(key.substr(dec.size(), dot - dec.size()).c_str());
        if (li < 0 || li >= g.num_layers) return fail(ER_ERR_INVALID, "layer index out of range in %s", key_c);
        LayerW& L = c->layers[li];
        const std::string rest = key.substr(dot + 1);
        const size_t HH = (size_t)H * H;
        if (rest == "self_attn.q_proj.weight") rc = put_w(&L.wqkv, &L.wqkv_h, 3 * HH, 0, HH);
        else if (rest == "self_attn.k_proj.weight") rc = put_w(&L.wqkv, &L.wqkv_h, 3 * HH, HH, HH);
        else if (rest == "self_attn.v_proj.weight") rc = put_w(&L.wqkv, &L.wqkv_h, 3 * HH, 2 * HH, HH);
        else if (rest == "self_attn.q_proj.bias") rc = put_at(&L.bqkv, 3 * H, 0, H);
        else if (rest == "self_attn.k_proj.bias") rc = put_at(&L.bqkv, 3 * H, H, H);
        else if (rest == "self_attn.v_proj.bias") rc = put_at(&L.bqkv, 3 * H, 2 * H, H);
        else if (rest == "self_attn.out_proj.weight") rc = put_w(&L.wo, &L.wo_h, HH, 0, HH);
        else if (rest == "self_attn.out_proj.bias") rc = put(&L.bo, H);
        else if (rest == "self_attn_layer_norm.weight") rc = put(&L.ln1w, H);
        else if (rest == "self_attn_layer_norm.bias") rc = put(&L.ln1b, H);
        else if (rest == "fc1.weight") rc = put_w(&L.w1, &L.w1_h, (size_t)I * H, 0, (size_t)I * H);
        else if (rest == "fc1.bias") rc = put(&L.b1, I);
        else if (rest == "fc2.weight") rc = put_w(&L.w2, &L.w2_h, (size_t)H * I, 0, (size_t)H * I);
        else if (rest == "fc2.bias") rc = put(&L.b2, H);
        else if (rest == "final_layer_norm.weight") rc = put(&L.ln2w, H);
        else if (rest == "final_layer_norm.bias") rc = put(&L.ln2b, H);
        else return 1;
    } else if (key == "mesh_decoder.model.embd.weight") rc = put(&c->embd, (size_t)g.vocab_size * H);
    else if (key == "mesh_decoder.model.embed_positions.weight") rc = put(&c->posemb, (size_t)g.max_positions * H);
    else if (key == "mesh_decoder.lm_head.weight") rc = put_w(&c->lm_head, &c->lm_head_h, (size_t)g.vocab_size * H, 0, (size_t)g.vocab_size * H);
    else if (key == "embed_num_face.weight") rc = put(&c->embed_num_face, (size_t)g.num_face_buckets * H);
    else if (key == "proj_cond.weight") rc = put(&c->proj_w, (size_t)H * g.point_latent_dim);
    else if (key == "proj_cond.bias") rc = put(&c->proj_b, H);
    else if (key == "norm_cond.weight") rc = put(&c->normc_w, H);
    else if (key == "norm_cond.bias") rc = put(&c->normc_b, H);
    else if (key == "point_encoder.query_embed") rc = put(&c->pe_query, (size_t)g.point_latent_size * PH);
    else if (key == "point_encoder.point_embed.basis") rc = put(&c->pe_basis, (size_t)3 * g.point_freq_dim);
    else if (key == "point_encoder.point_embed.mlp.weight") {
        // [PH][2F+3] -> zero-padded [PH][kpad] so the GEMM K is a multiple of 16
        const int kin = 2 * g.point_freq_dim + 3;
        c->pe_kpad = (kin + 15) / 16 * 16;
        ERCHK(expect((size_t)PH * kin));
        int err = 0;
        std::vector<float> h = to_f32_host(data, dtype, n, on_device, &err);     // small tensor: padded on the host
        if (err) return fail(ER_ERR_HIP, "er_load_tensor(%s): device read failed", key_c);
        std::vector<float> padded((size_t)PH * c->pe_kpad, 0.f);
        for (int r = 0; r < PH; ++r) memcpy(&padded[(size_t)r * c->pe_kpad], &h[(size_t)r * kin], kin * 4);
        if (!c->pe_mlp_w) ERCHK(dev_alloc(c, &c->pe_mlp_w, padded.size()));
        HIPCHK(hipMemcpy(c->pe_mlp_w, padded.data(), padded.size() * 4, hipMemcpyHostToDevice));
    } else if (key == "point_encoder.point_embed.mlp.bias") rc = put(&c->pe_mlp_b, PH);
    else if (key == "point_encoder.ln.weight") rc = put(&c->pe_ln_w, PH);
    else if (key == "point_encoder.ln.bias") rc = put(&c->pe_ln_b, PH);
    else if (key == "point_encoder.cross_att.ln1.weight") rc = put(&c->ca_ln1_w, PH);
    else if (key == "point_encoder.cross_att.ln1.bias") rc = put(&c->ca_ln1_b, PH);
    else if (key == "point_encoder.cross_att.ln2.weight") rc = put(&c->ca_ln2_w, PH);
    else if (key == "point_encoder.cross_att.ln2.bias") rc = put(&c->ca_ln2_b, PH);
    else if (key == "point_encoder.cross_att.att.q_proj.weight") rc = put(&c->ca_q_w, (size_t)PH * PH);
    else if (key == "point_encoder.cross_att.att.q_proj.bias") rc = put(&c->ca_q_b, PH);
    else if (key == "point_encoder.cross_att.att.k_proj.weight") rc = put(&c->ca_k_w, (size_t)PH * PH);
    else if (key == "point_encoder.cross_att.att.k_proj.bias") rc = put(&c->ca_k_b, PH);
    else if (key == "point_encoder.cross_att.att.v_proj.weight") rc = put(&c->ca_v_w, (size_t)PH * PH);
    else if (key == "point_encoder.cross_att.att.v_proj.bias") rc = put(&c->ca_v_b, PH);
    else if (key == "point_encoder.cross_att.att.out_proj.weight") rc = put(&c->ca_o_w, (size_t)PH * PH);
    else if (key == "point_encoder.cross_att.att.out_proj.bias") rc = put(&c->ca_o_b, PH);
    else if (key == "point_encoder.cross_att.mlp.net.0.weight") rc = put(&c->ff0_w, (size_t)8 * PH * PH);
    else if (key == "point_encoder.cross_att.mlp.net.0.bias") rc = put(&c->ff0_b, (size_t)8 * PH);
    else if (key == "point_encoder.cross_att.mlp.net.2.weight") rc = put(&c->ff2_w, (size_t)PH * 4 * PH);
    else if (key == "point_encoder.cross_att.mlp.net.2.bias") rc = put(&c->ff2_b, PH);
    else if (key == "point_encoder.linear.weight") rc = put(&c->lin_w, (size_t)g.point_latent_dim * PH);
    else if (key == "point_encoder.linear.bias") rc = put(&c->lin_b, g.point_latent_dim);
    else return 1;
    if (rc < 0) return rc;
    it->second = true;
    (void)ends_with;
    return ER_OK;
}

extern "C" int er_finalize_weights(er_ctx* c) {
    if (!c) return fail(ER_ERR_INVALID, "null ctx");
    for (auto& kv : c->need)
        if (!kv.second) return fail(ER_ERR_MISSING, "tensor '%s' was never loaded", kv.first.c_str());
    if (c->e_stage.p) {                   // checkpoint complete: the upload staging block (up to one tensor) is not needed any more
        hipFree(c->e_stage.p);
        c->e_stage.p = nullptr;
        c->e_stage.n = 0;
    }
    return ER_OK;
}

// second copy of the qkv / fc1 / fc2 matrices in the layout the matrix-core batched kernels stream (made the first
// time a batch > 4 is reserved: 2.5 GB fp32 / 1.3 GB fp16 of the 288 GB)
template <typename WT>
static int make_tiled(er_ctx* c, const void* src, void** dst, int N, int K) {
    if (!*dst) {
        HIPCHK(hipMalloc(dst, tiled_weight_bytes<WT>(N, K)));
        c->owned.push_back(*dst);
    }
    hipLaunchKernelGGL((tile_weights_kernel<WT>), dim3(2048), dim3(ER_WG), 0, c->own_stream, reinterpret_cast<const WT*>(src),
                       reinterpret_cast<f32x4*>(*dst), N, K);
    HIPCHK(hipGetLastError());
    return 0;
}
static int make_tiled_weights(er_ctx* c) {
    if (c->tiled_valid) return 0;
    for (auto& kv : c->need)
        if (!kv.second) return 0;          // weights still loading: er_prefill comes back here
    const int H = c->cfg.hidden_dim, I = c->cfg.intermediate_dim;
    for (LayerW& L : c->layers) {
        if (c->fast) {
            ERCHK(make_tiled<_Float16>(c, L.wqkv_h, &L.wqkv_t, 3 * H, H));
            ERCHK(make_tiled<_Float16>(c, L.wo_h, &L.wo_t, H, H));
            ERCHK(make_tiled<_Float16>(c, L.w1_h, &L.w1_t, I, H));
            ERCHK(make_tiled<_Float16>(c, L.w2_h, &L.w2_t, H, I));
        } else {
            ERCHK(make_tiled<float>(c, L.wqkv, &L.wqkv_t, 3 * H, H));
            ERCHK(make_tiled<float>(c, L.wo, &L.wo_t, H, H));
            ERCHK(make_tiled<float>(c, L.w1, &L.w1_t, I, H));
            ERCHK(make_tiled<float>(c, L.w2, &L.w2_t, H, I));
        }
    }
    HIPCHK(hipStreamSynchronize(c->own_stream));
    c->tiled_valid = true;
    return 0;
}

// ------------------------------------------------------------------------------------ which decode kernels a cache shape gets
// ONE place for the selection rules (kv_alloc applies them, er_plan_decode reports them; pure host logic):
//   batched     : B > 4 (or forced) - weights streamed once per pass of 32 rows on the matrix cores
//   version 3   : one row, 16 heads of 96, hidden 1536, reserved cache <= 16 chunks x 512 keys; else version 2
//   attention B>4: streaming kernel when forced or (auto and B * heads >= 256: at least one workgroup per CU - at B = 16 it
//                  ties the split kernel and saves the merge launch, at B = 8 it is 1.5x slower), else split + merge
static void plan_decode(int decode_v, int attn_v_batched, bool force_batched, int batch, int H, int D, int hid, int Lcap,
                        er_decode_plan* p) {
    p->batched = (batch > 4 || force_batched) ? 1 : 0;
    p->attn_chunks = attn3_num_chunks(H);
    const bool v3 = decode_v == 3 && batch == 1 && !p->batched && D == 96 && H == 16 && hid == 1536 && attn3_fits(Lcap, H);
    p->decode_version = v3 ? 3 : 2;
    const bool stream = p->batched && D == 96 && (attn_v_batched == 3 || (attn_v_batched == 0 && batch * H >= 256));
    p->attn_kernel = !p->batched ? (v3 ? ER_ATTN_BALANCED : ER_ATTN_SPLIT2)
                                 : (stream ? ER_ATTN_STREAM : ER_ATTN_SPLIT1);
    p->merge_launch = (p->attn_kernel == ER_ATTN_SPLIT1 || p->attn_kernel == ER_ATTN_SPLIT2) ? 1 : 0;
    p->launches_per_layer = p->batched ? 0 : 5 + p->merge_launch;       // qkv, attention, (merge,) out_proj, fc1, fc2
}

static int env_int_(const char* name, int dflt) { const char* v = getenv(name); return (v && v[0]) ? atoi(v) : dflt; }

extern "C" int er_plan_decode(int batch, int heads, int head_dim, int hidden, int l_cap, er_decode_plan* out) {
    if (!out || batch <= 0 || heads <= 0 || head_dim <= 0 || l_cap <= 0) return fail(ER_ERR_INVALID, "er_plan_decode: bad argument");
    int dv = env_int_("ER_DECODE_V", 3) == 2 ? 2 : 3;
    int avb = env_int_("ER_ATTN_V_BATCHED", 0);
    if (avb != 1 && avb != 3) avb = 0;
    const char* fb = getenv("ER_FORCE_BATCHED");
    plan_decode(dv, avb, fb && fb[0] == '1', batch, heads, head_dim, hidden, (l_cap + 31) / 32 * 32, out);
    return ER_OK;
}

// the plan of a LIVE context: its knobs were fixed at er_create and its cache by the last er_kv_reserve (er_plan_decode above
// evaluates the same rules for hypothetical shapes with the environment as it is at call time)
extern "C" int er_ctx_plan(er_ctx* c, er_decode_plan* out) {
    if (!c || !out) return fail(ER_ERR_INVALID, "er_ctx_plan: null argument");
    if (c->B <= 0) return fail(ER_ERR_INVALID, "er_ctx_plan: no cache reserved (call er_kv_reserve)");
    plan_decode(c->decode_v, c->attn_v_batched, /*force_batched=*/c->batched && c->B <= 4, c->B, c->cfg.num_heads, c->D,
                c->cfg.hidden_dim, c->Lcap, out);
    return ER_OK;
}

extern "C" int er_plan_gemm_tile(int m, int n, int batch) {
    if (m <= 0 || n <= 0 || batch <= 0) return fail(ER_ERR_INVALID, "er_plan_gemm_tile: bad argument");
    return gemm_pick_tile(m, n, batch);
}

// ------------------------------------------------------------------------------------ KV cache / workspace
static int kv_alloc(er_ctx* c, int batch, int Lcap);

extern "C" int er_kv_reserve(er_ctx* c, int batch, int max_len) {
    if (!c || batch <= 0 || max_len <= 0) return fail(ER_ERR_INVALID, "er_kv_reserve: bad argument");
    if (batch > ER_MAX_BATCH) return fail(ER_ERR_UNSUPPORTED, "er_kv_reserve: batch %d > %d (host staging buffers are sized for %d rows)", batch, ER_MAX_BATCH, ER_MAX_BATCH);
    HIPCHK(hipSetDevice(c->device));
    const er_config& g = c->cfg;
    if (max_len > g.max_positions)
        return fail(ER_ERR_CAPACITY, "max_len %d exceeds the position table (%d)", max_len, g.max_positions);
    const int Lcap = (max_len + 31) / 32 * 32;
    if (c->B == batch && c->Lcap == Lcap) return ER_OK;
    HIPCHK(hipDeviceSynchronize());
    free_kv(c);
    const int rc_alloc = kv_alloc(c, batch, Lcap);
    if (rc_alloc < 0) { free_kv(c); return rc_alloc; }     // a failed hipMalloc half way leaves nothing behind
    return ER_OK;
}

static int kv_alloc(er_ctx* c, int batch, int Lcap) {
    const er_config& g = c->cfg;
    const int H = g.num_heads, D = c->D, hid = g.hidden_dim;
    c->kv_bstride = (long long)H * Lcap * D;
    c->kv_lstride = c->kv_bstride * batch;
    const size_t kv_elems = (size_t)c->kv_lstride * g.num_layers;
    HIPCHK(hipMalloc(&c->kc, kv_elems * c->kv_esz));
    HIPCHK(hipMalloc(&c->vc, kv_elems * c->kv_esz));
    // decode attention: one workgroup per (row, head, chunk of 32*steps keys)
    const int S = attn_num_chunks(Lcap, attn_chunk(ATTN_STEPS_DEFAULT, c->fast));
    c->S_splits = S;
    const size_t b = (size_t)batch;
    HIPCHK(hipMalloc(&c->ypre, b * hid * 4));
    HIPCHK(hipMalloc(&c->hbuf, b * hid * 4));
    HIPCHK(hipMalloc(&c->ypre1, b * hid * 4));
    HIPCHK(hipMalloc(&c->h1buf, b * hid * 4));
    HIPCHK(hipMalloc(&c->qbuf, b * hid * 4));
    HIPCHK(hipMalloc(&c->abuf, b * hid * 4));
    HIPCHK(hipMalloc(&c->fbuf, b * g.intermediate_dim * 4));
    HIPCHK(hipMalloc(&c->logits, b * g.vocab_size * 4));
    c->nch3 = attn3_num_chunks(H);
    HIPCHK(hipMalloc(&c->part, b * H * (size_t)std::max(S * (D + 2), c->nch3 * D) * 4));
    HIPCHK(hipMalloc(&c->part_ml, b * H * (size_t)c->nch3 * 2 * 4));
    HIPCHK(hipMalloc(&c->state_block, (7 * b + 8) * sizeof(int)));
    int* sb = c->state_block;
    c->st.tok = sb; c->st.pos = sb + b; c->st.counter = sb + 2 * b; c->st.ngen = sb + 3 * b;
    c->st.unfinished = sb + 4 * b; c->st.eos_step = sb + 5 * b; c->st.base_pos = sb + 6 * b; c->st.n_unfinished = sb + 7 * b; c->st.error = sb + 7 * b + 1;
    HIPCHK(hipMalloc(&c->d_params, sizeof(DecodeParamsDev)));
    HIPCHK(hipMalloc(&c->d_ids_tmp, b * sizeof(int)));
    HIPCHK(hipMalloc(&c->d_row_stream, b * sizeof(unsigned int)));
    {
        std::vector<unsigned int> ident(b);
        for (size_t i = 0; i < b; ++i) ident[i] = (unsigned int)i;
        HIPCHK(hipMemcpy(c->d_row_stream, ident.data(), b * sizeof(unsigned int), hipMemcpyHostToDevice));
    }
    c->st.row_stream = c->d_row_stream;
    HIPCHK(hipMalloc(&c->d_out_ids, b * (size_t)Lcap * sizeof(long long)));
    c->B = batch;
    c->Lcap = Lcap;
    c->have_hidden = false;
    const char* fb = getenv("ER_FORCE_BATCHED");
    er_decode_plan plan{};
    plan_decode(c->decode_v, c->attn_v_batched, fb && fb[0] == '1', batch, H, D, hid, Lcap, &plan);
    c->batched = plan.batched != 0;
    const char* bv = getenv("ER_BATCHED_VALU");
    c->batched_valu = bv && bv[0] == '1';
    if (c->batched && !c->batched_valu) ERCHK(make_tiled_weights(c));
    c->stream_attn = plan.attn_kernel == ER_ATTN_STREAM;
    c->v3 = plan.decode_version == 3;
    // fast mode, matrix-core projections: the activations travel in the tiled hi | lo operand layout (k_gemv.h xt_entry).  One image
    // of K x 128 bytes per group of 32 rows; zeroed once so that the rows of a last, partial group never hold NaN patterns.
    const char* xe = getenv("ER_XT");
    c->xt = c->fast && c->batched && !c->batched_valu && !(xe && xe[0] == '0');
    // split-K partials of the batched projections.  A finish launched right behind its producer re-uses ONE [4][32][N] block; only the
    // tiled path defers finishes to a later launch (prep_rows_kernel, sk_part) and keeps a [16][32][hidden] block per group of 32 rows
    c->skpart_floats = c->xt ? ((b + NBM - 1) / NBM) * (size_t)16 * NBM * (size_t)hid : (size_t)4 * NBM * (size_t)std::max(hid, g.vocab_size);
    HIPCHK(hipMalloc(&c->skpart, c->skpart_floats * 4));
    if (c->xt) {
        const size_t groups = (size_t)(batch + NBM - 1) / NBM;
        const size_t b_h = groups * (size_t)hid * 128, b_f = groups * (size_t)g.intermediate_dim * 128;
        HIPCHK(hipMalloc(&c->xt_h, b_h));
        HIPCHK(hipMalloc(&c->xt_att, b_h));
        HIPCHK(hipMalloc(&c->xt_f, b_f));
        HIPCHK(hipMemset(c->xt_h, 0, b_h));
        HIPCHK(hipMemset(c->xt_att, 0, b_h));
        HIPCHK(hipMemset(c->xt_f, 0, b_f));
    }
    return ER_OK;
}

// ------------------------------------------------------------------------------------ decode step
struct StepPlan {   // which kinds to launch (profiling launches one kind at a time)
    bool head = true, sample = true, layers = true;
    int only_kind = -1;   // >= 0: launch just this kind
    int only_layer = -1;
};

template <typename WT, int KS, int RW, int PRO, int EPI, int NW = ER_NWAVES>
static hipError_t gemv_groups(GemvArgs a, int B, int K, hipStream_t st) {
    // rows are processed in groups of up to 4 (one weight stream, 1..4 accumulators per row)
    int b = 0;
    while (b < B) {
        int nb = B - b;
        nb = nb >= 4 ? 4 : nb;
        GemvArgs g = a;
        if (g.xin) g.xin += (long long)b * K;
        if (g.hout) g.hout += (long long)b * K;
        if (g.tok) g.tok += b;
        if (g.pos) g.pos += b;
        if (g.out) g.out += (long long)b * a.N;
        if (g.resid) g.resid += (long long)b * a.N;
        if (g.q) g.q += (long long)b * a.hidden;
        const long long kvb = (long long)b * a.kv_bstride * (a.kv_half ? 2 : 4);
        if (g.kcache) g.kcache = (char*)g.kcache + kvb;
        if (g.vcache) g.vcache = (char*)g.vcache + kvb;
        hipError_t e;
        if (nb == 4) e = launch_gemv<WT, KS, 4, RW, PRO, EPI, NW>(g, st);
        else if (nb == 3) e = launch_gemv<WT, KS, 3, RW, PRO, EPI, NW>(g, st);
        else if (nb == 2) e = launch_gemv<WT, KS, 2, RW, PRO, EPI, NW>(g, st);
        else e = launch_gemv<WT, KS, 1, RW, PRO, EPI, NW>(g, st);
        if (e != hipSuccess) return e;
        b += nb;
    }
    return hipSuccess;
}

// single-row GEMV with a LayerNorm / embedding prologue by waves per workgroup: 4 or 6 waves x 1 row (qkv), 4 waves x 2 rows (fc1),
// or ONE fat workgroup per CU - 9 waves x 2 rows = 4608 qkv rows / 256, 12 waves x 2 rows = 6144 fc1 rows / 256
template <typename WT, int PRO, int EPI>
static hipError_t gemv_nw(int nw, GemvArgs a, int B, int K, hipStream_t st) {
    if (nw == 6) return gemv_groups<WT, 1, 1, PRO, EPI, 6>(a, B, K, st);
    if (nw == 9) return gemv_groups<WT, 1, 2, PRO, EPI, 9>(a, B, K, st);
    if (nw == 12) return gemv_groups<WT, 1, 2, PRO, EPI, 12>(a, B, K, st);
    if (EPI == EPI_QKV) return gemv_groups<WT, 1, 1, PRO, EPI>(a, B, K, st);
    return gemv_groups<WT, 1, 2, PRO, EPI>(a, B, K, st);
}

// out_proj of 5..8 rows: ONE pass of the VALU kernel with 5..8 accumulators per weight row.  The matrix-core kernel has only 48
// row tiles for this 1536-row matrix (48 of 256 CUs: 9.3 us at B = 5 against 4.9 us for the four-row VALU launch), which left the
// aggregate rate of five rows below that of four (profiles/r03_batch_table_v2.log); same per-(row, batch row) arithmetic as every
// other path.
template <typename WT>
static hipError_t gemv_outproj_rows8(const GemvArgs& a, int B, hipStream_t st) {
    switch (B) {
        case 5: return launch_gemv<WT, 1, 5, 1, PRO_NONE, EPI_RESID, 3>(a, st);
        case 6: return launch_gemv<WT, 1, 6, 1, PRO_NONE, EPI_RESID, 3>(a, st);
        case 7: return launch_gemv<WT, 1, 7, 1, PRO_NONE, EPI_RESID, 3>(a, st);
        case 8: return launch_gemv<WT, 1, 8, 1, PRO_NONE, EPI_RESID, 3>(a, st);
    }
    return hipErrorInvalidValue;
}

// B > 4: weights streamed once per pass of up to 16 rows (gemv_batched_kernel)
constexpr int NBB = 16;
template <typename WT, int PH, int RW, int EPI>
static hipError_t gemv_batched_groups(GemvArgs a, int B, int K, hipStream_t st) {
    for (int b = 0; b < B; b += NBB) {
        const int nb = (B - b) < NBB ? (B - b) : NBB;
        GemvArgs g = a;
        g.xin += (long long)b * K;
        if (g.pos) g.pos += b;
        if (g.out) g.out += (long long)b * a.N;
        if (g.resid) g.resid += (long long)b * a.N;
        if (g.q) g.q += (long long)b * a.hidden;
        const long long kvb = (long long)b * a.kv_bstride * (a.kv_half ? 2 : 4);
        if (g.kcache) g.kcache = (char*)g.kcache + kvb;
        if (g.vcache) g.vcache = (char*)g.vcache + kvb;
        hipError_t e = launch_gemv_batched<WT, PH, NBB, RW, EPI>(g, nb, st);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
// XT: a.xin is the tiled image of the input (a group of 32 rows is K * 32 floats there as well, so the group offsets coincide)
struct SkPart { float* p; size_t floats; };       // the context's split-K partial block and its size
template <typename WT, int EPI, bool XT = false>
static hipError_t gemv_mfma_groups(GemvArgs a, int B, int K, SkPart part, hipStream_t st, bool defer_finish = false, bool narrow = false) {
    {   // the block was sized in er_kv_reserve for K <= 6144 (4 wide / 16 narrow slices) and, when the finish is not deferred, for ONE
        // group at a time on an in-order stream: refuse anything that would overrun it instead of writing past the end
        const size_t slices = narrow ? K / (4 * GM_KW) : K / (GM_WAVES * GM_KW), groups = (size_t)(B + NBM - 1) / NBM;
        const size_t need = (defer_finish ? groups : 1) * slices * NBM * (size_t)a.N;
        if (!part.p || need > part.floats) return hipErrorInvalidValue;
    }
    for (int b = 0; b < B; b += NBM) {
        const int nb = (B - b) < NBM ? (B - b) : NBM;
        GemvArgs g = a;
        if (g.xin) g.xin += (long long)b * K;
        if (g.xt_out) g.xt_out = (char*)g.xt_out + (long long)b * a.N * 4;     // xt_entry() indexes inside a group
        if (g.hout) g.hout += (long long)b * K;
        if (g.tok) g.tok += b;
        if (g.pos) g.pos += b;
        if (g.out) g.out += (long long)b * a.N;
        if (g.resid) g.resid += (long long)b * a.N;
        if (g.q) g.q += (long long)b * a.hidden;
        const long long kvb = (long long)b * a.kv_bstride * (a.kv_half ? 2 : 4);
        if (g.kcache) g.kcache = (char*)g.kcache + kvb;
        if (g.vcache) g.vcache = (char*)g.vcache + kvb;
        // deferred: every group keeps its own partial block (prep_rows_kernel: g * 4 * 32 * K floats)
        const int slices = narrow ? K / (4 * GM_KW) : K / (GM_WAVES * GM_KW);
        hipError_t e = launch_gemv_mfma<WT, EPI, XT>(g, nb, K, defer_finish ? part.p + (long long)(b / NBM) * slices * NBM * a.N : part.p, st, defer_finish, narrow);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
template <int PRO>
static hipError_t prep_rows(const GemvArgs& a, int B, hipStream_t st) {
    hipLaunchKernelGGL((prep_rows_kernel<PRO>), dim3(B), dim3(ER_WG), 0, st, a);
    return hipGetLastError();
}

static AttnDecArgs attn_args(er_ctx* c, int layer) {
    AttnDecArgs a{};
    a.q = c->qbuf;
    a.kcache = (char*)c->kc + (long long)layer * c->kv_lstride * c->kv_esz;
    a.vcache = (char*)c->vc + (long long)layer * c->kv_lstride * c->kv_esz;
    a.chunk = attn_chunk(ATTN_STEPS_DEFAULT, c->fast);
    a.pos = c->st.pos;
    a.fixed_len = c->prof_len;
    a.len_dev = nullptr;
    a.part = c->part;
    a.part_ml = c->part_ml;
    a.out = c->abuf;
    a.H = c->cfg.num_heads;
    a.l_cap = c->Lcap;
    a.S = c->S_splits;
    a.hidden = c->cfg.hidden_dim;
    a.kv_bstride = c->kv_bstride;
    a.sqrt_d = sqrtf((float)c->D);
    a.out_xt = (c->xt && c->stream_attn && c->B > 8) ? c->xt_att : nullptr;      // out_proj then reads the tiled image (launch_kind_t case 3)
    return a;
}

static hipError_t launch_attn_partial(const AttnDecArgs& a, int D, int steps, bool kv_half, int B, hipStream_t st, int ver = 2) {
    return D == 96 ? launch_attn_partial_d<96>(a, steps, kv_half, B, st, ver) : launch_attn_partial_d<64>(a, steps, kv_half, B, st, ver);
}
static hipError_t launch_attn_combine(const AttnDecArgs& a, int D, int B, hipStream_t st) {
    return D == 96 ? launch_attn_combine_d<96>(a, B, st) : launch_attn_combine_d<64>(a, B, st);
}

template <typename WT>
static hipError_t launch_kind_t(er_ctx* c, int kind, int layer, hipStream_t st, long long* out_ids, int out_ld) {
    constexpr bool HALF = sizeof(WT) == 2;
    const er_config& g = c->cfg;
    const int H = g.hidden_dim, I = g.intermediate_dim, B = c->B;
    const int nl = g.num_layers;
    GemvArgs a{};
    a.eps = g.ln_eps;
    a.hidden = H; a.head_dim = c->D; a.l_cap = c->Lcap; a.kv_bstride = c->kv_bstride; a.kv_half = HALF ? 1 : 0;
    switch (kind) {
        case 0: {   // qkv
            const LayerW& L = c->layers[layer];
            a.W = HALF ? (const void*)L.wqkv_h : (const void*)L.wqkv; a.bias = L.bqkv; a.N = 3 * H;
            a.hout = c->hbuf; a.pos = c->st.pos;
            a.q = c->qbuf;
            a.kcache = (char*)c->kc + (long long)layer * c->kv_lstride * c->kv_esz;
            a.vcache = (char*)c->vc + (long long)layer * c->kv_lstride * c->kv_esz;
            if (layer == 0) {
                a.embd = c->embd; a.posemb = c->posemb; a.tok = c->st.tok;
            } else {
                a.xin = c->ypre; a.ln_w = c->layers[layer - 1].ln2w; a.ln_b = c->layers[layer - 1].ln2b;
            }
            if (c->batched) {
                if constexpr (HALF) {
                    a.xt_out = c->xt ? c->xt_h : nullptr;
                    if (c->xt && layer > 0) {      // the previous layer's fc2 deferred its split-K finish to this LayerNorm (case 5)
                        a.sk_part = c->skpart; a.sk_bias = c->layers[layer - 1].b2; a.sk_resid = c->h1buf; a.sk_batch = B; a.sk_slices = 16;
                    }
                }
                hipError_t e = layer == 0 ? prep_rows<PRO_EMBED>(a, B, st) : prep_rows<PRO_LN>(a, B, st);
                a.sk_part = nullptr;
                if (e != hipSuccess) return e;
                a.xin = c->hbuf;
                a.xt_out = nullptr;
                if constexpr (HALF) {
                    if (c->xt) { a.W = L.wqkv_t; a.xin = (const float*)c->xt_h; return gemv_mfma_groups<WT, EPI_QKV, true>(a, B, H, SkPart{c->skpart, c->skpart_floats}, st); }
                }
                if (!c->batched_valu) { a.W = L.wqkv_t; return gemv_mfma_groups<WT, EPI_QKV>(a, B, H, SkPart{c->skpart, c->skpart_floats}, st); }   // 144 tiles of 32 rows
                return gemv_batched_groups<WT, 1, 3, EPI_QKV>(a, B, H, st);   // 4608 rows = 192 workgroups x 24: one round
            }
            if (layer == 0) return gemv_nw<WT, PRO_EMBED, EPI_QKV>(c->nw_qkv, a, B, H, st);
            return gemv_nw<WT, PRO_LN, EPI_QKV>(c->nw_qkv, a, B, H, st);
        }
        // v2 holds a wave's whole K/V slice in flight (latency-bound single rows); with hundreds of workgroups per CU's
        // worth of work (B > 4) the leaner v1 (66-74 VGPRs, 6-7 waves per SIMD) streams faster: 570 vs 636 us at B = 32, L = 18050
        case 1:
            if (c->v3) return launch_attn_partial3_d<96>(attn_args(c, layer), HALF, c->nch3, B, st);
            if (c->stream_attn) return launch_attn_stream_d<96>(attn_args(c, layer), HALF, B, st);
            return launch_attn_partial(attn_args(c, layer), c->D, ATTN_STEPS_DEFAULT, HALF, B, st, c->batched ? 1 : 2);
        case 2:
            if (c->v3 || c->stream_attn) return hipSuccess;      // the merge runs inside the out_proj kernel / there are no partials
            return launch_attn_combine(attn_args(c, layer), c->D, B, st);
        case 3: {   // out_proj + bias + residual(h) -> ypre1
            const LayerW& L = c->layers[layer];
            if (c->v3) {
                OutMergeArgs m{};
                m.W = HALF ? (const void*)L.wo_h : (const void*)L.wo; m.bias = L.bo; m.resid = c->hbuf; m.out = c->ypre1;
                m.part_o = c->part; m.part_ml = c->part_ml; m.N = H;
                return launch_outproj_merge<WT, 96>(m, c->nch3, st);
            }
            a.W = HALF ? (const void*)L.wo_h : (const void*)L.wo; a.bias = L.bo; a.N = H; a.xin = c->abuf; a.out = c->ypre1; a.resid = c->hbuf;
            // 48 row tiles of 32: the matrix-core kernel runs on 48 CUs only, but streams the matrix ONCE for 32 rows where the
            // VALU kernel needs a pass per 16
            if (c->batched && !c->batched_valu && B >= 5 && B <= 8) return gemv_outproj_rows8<WT>(a, B, st);
            if constexpr (HALF) {
                // 4-wave workgroups (48 row tiles x 4 K-ranges of 384 instead of 48 x one of 1536); + bias + residual happen in fc1's
                // LayerNorm-rows launch (case 4), which reads the four partials
                if (c->xt && c->stream_attn && B > 8) { a.W = L.wo_t; a.xin = (const float*)c->xt_att; return gemv_mfma_groups<WT, EPI_RESID, true>(a, B, H, SkPart{c->skpart, c->skpart_floats}, st, true, true); }
            }
            if (c->batched && !c->batched_valu) { a.W = L.wo_t; return gemv_mfma_groups<WT, EPI_RESID>(a, B, H, SkPart{c->skpart, c->skpart_floats}, st); }
            if (c->batched) return gemv_batched_groups<WT, 1, 1, EPI_RESID>(a, B, H, st);
            return gemv_groups<WT, 1, 1, PRO_NONE, EPI_RESID, 3>(a, B, H, st);     // 3 waves x 1 row: 512 workgroups = 2 per CU
        }
        case 4: {   // h1 = LN1(ypre1); f = relu(fc1 h1 + b)
            const LayerW& L = c->layers[layer];
            a.W = HALF ? (const void*)L.w1_h : (const void*)L.w1; a.bias = L.b1; a.N = I; a.xin = c->ypre1; a.ln_w = L.ln1w; a.ln_b = L.ln1b;
            a.hout = c->h1buf; a.out = c->fbuf;
            if (c->batched) {
                if constexpr (HALF) {
                    a.xt_out = c->xt ? c->xt_h : nullptr;
                    if (c->xt && c->stream_attn && B > 8) {      // out_proj (case 3) left four K-range partials: ypre1 = ((sum) + bo) + h
                        a.sk_part = c->skpart; a.sk_bias = L.bo; a.sk_resid = c->hbuf; a.sk_batch = B; a.sk_slices = 4;
                    }
                }
                hipError_t e = prep_rows<PRO_LN>(a, B, st);
                a.sk_part = nullptr;
                if (e != hipSuccess) return e;
                a.xin = c->h1buf;
                a.xt_out = nullptr;
                if constexpr (HALF) {
                    if (c->xt) {      // input and output both tiled: fc2 below reads xt_f
                        a.W = L.w1_t; a.xin = (const float*)c->xt_h; a.xt_out = c->xt_f;
                        return gemv_mfma_groups<WT, EPI_RELU, true>(a, B, H, SkPart{c->skpart, c->skpart_floats}, st);
                    }
                }
                if (!c->batched_valu) { a.W = L.w1_t; return gemv_mfma_groups<WT, EPI_RELU>(a, B, H, SkPart{c->skpart, c->skpart_floats}, st); }   // 192 tiles of 32 rows
                return gemv_batched_groups<WT, 1, 3, EPI_RELU>(a, B, H, st);   // 6144 rows = 256 workgroups x 24
            }
            return gemv_nw<WT, PRO_LN, EPI_RELU>(c->nw_fc1, a, B, H, st);
        }
        case 5: {   // ypre = fc2 f + b + h1
            const LayerW& L = c->layers[layer];
            a.W = HALF ? (const void*)L.w2_h : (const void*)L.w2; a.bias = L.b2; a.N = H; a.xin = c->fbuf; a.out = c->ypre; a.resid = c->h1buf;
            if constexpr (HALF) {
                // layers 0 .. nl-2 leave the four K-range partials to the next layer's LayerNorm launch (case 0); the last layer finishes
                // into ypre, which the lm_head reads (after a prefill ypre comes from the GEMM path, so case 6 always reads ypre)
                // 4-wave workgroups: 48 row tiles x 16 K-ranges of 384 (768 workgroups = 3 per CU instead of 192 on 192 CUs)
                if (c->xt) { a.W = L.w2_t; a.xin = (const float*)c->xt_f; return gemv_mfma_groups<WT, EPI_RESID, true>(a, B, I, SkPart{c->skpart, c->skpart_floats}, st, layer + 1 < nl, true); }
            }
            if (c->batched && !c->batched_valu) { a.W = L.w2_t; return gemv_mfma_groups<WT, EPI_RESID>(a, B, I, SkPart{c->skpart, c->skpart_floats}, st); }   // 48 tiles x 4 K-ranges
            if (c->batched) return gemv_batched_groups<WT, 4, 1, EPI_RESID>(a, B, I, st);
            if constexpr (HALF) {
                // fast mode, one row: FAT workgroups like qkv's and fc1's - 4 or 6 rows per workgroup instead of 2 (384 / 256 workgroups
                // instead of 768), so that a CU fetches the 24 KB input vector once or twice instead of three times beside its 72 KB of
                // fp16 weights: fc2 5.60 -> 5.37 us at 6 rows, 5.68 at 4, ids unchanged (profiles/r05_ab_fc2_rows.log; ER_RW_FC2 = 2 / 4 / 6)
                // (read ONCE per process, like every A/B knob of the step graph: the graph is captured with the first value; 2 / 4 / 6 only)
                static const int rw = [] {
                    const char* v = getenv("ER_RW_FC2");
                    const int r = v ? atoi(v) : 6;
                    if (r != 2 && r != 4 && r != 6) fprintf(stderr, "[edgerunner_hip] ER_RW_FC2=%s is not 2 / 4 / 6: using 2 rows per fc2 workgroup\n", v);
                    return r;
                }();
                if (B == 1 && rw == 6) return launch_gemv<WT, 4, 1, 6, PRO_NONE, EPI_RESID>(a, st);
                if (B == 1 && rw == 4) return launch_gemv<WT, 4, 1, 4, PRO_NONE, EPI_RESID>(a, st);
            }
            return gemv_groups<WT, 4, 2, PRO_NONE, EPI_RESID>(a, B, I, st);
        }
        case 6: {   // logits = lm_head LN2_last(ypre)
            a.W = HALF ? (const void*)c->lm_head_h : (const void*)c->lm_head; a.bias = nullptr; a.N = g.vocab_size; a.xin = c->ypre;
            a.ln_w = c->layers[nl - 1].ln2w; a.ln_b = c->layers[nl - 1].ln2b; a.hout = nullptr; a.out = c->logits;
            if (c->batched) {
                a.hout = c->hbuf;
                hipError_t e = prep_rows<PRO_LN>(a, B, st);
                if (e != hipSuccess) return e;
                a.xin = c->hbuf;
                return gemv_batched_groups<WT, 1, 1, EPI_STORE>(a, B, H, st);
            }
            return gemv_groups<WT, 1, 1, PRO_LN, EPI_STORE>(a, B, H, st);
        }
        case 7:
            hipLaunchKernelGGL(sample_head_kernel, dim3(B), dim3(ER_WG), sample_head_lds(g.vocab_size), st, c->logits,
                               c->d_params, c->st, out_ids, out_ld);
            return hipGetLastError();
    }
    return hipErrorInvalidValue;
}

static hipError_t launch_kind(er_ctx* c, int kind, int layer, hipStream_t st, long long* out_ids, int out_ld) {
    return c->fast ? launch_kind_t<_Float16>(c, kind, layer, st, out_ids, out_ld)
                   : launch_kind_t<float>(c, kind, layer, st, out_ids, out_ld);
}

static hipError_t enqueue_layers(er_ctx* c, hipStream_t st) {
    for (int l = 0; l < c->cfg.num_layers; ++l)
        for (int k = 0; k <= 5; ++k) {
            hipError_t e = launch_kind(c, k, l, st, nullptr, 0);
            if (e != hipSuccess) return e;
        }
    return hipSuccess;
}

static hipError_t enqueue_step(er_ctx* c, hipStream_t st, long long* out_ids, int out_ld) {
    hipError_t e = launch_kind(c, 6, 0, st, nullptr, 0);
    if (e != hipSuccess) return e;
    e = launch_kind(c, 7, 0, st, out_ids, out_ld);
    if (e != hipSuccess) return e;
    return enqueue_layers(c, st);
}

// ------------------------------------------------------------------------------------ GEMM helpers (prefill / encoder)
static hipError_t linear(const float* A, int lda, const float* W, const float* bias, float* C, int ldc, int M, int N, int K,
                         bool relu, const float* resid, int ldr, hipStream_t st) {
    GemmArgs g = gemm_args_default();
    g.A = A; g.B = W; g.C = C; g.bias = bias; g.resid = resid;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = K; g.ldc = ldc; g.ldr = ldr;
    g.relu = relu ? 1 : 0;
    return launch_gemm(g, 1, st);
}

// fast mode: C = epilogue((A_hi + A_lo) . W_fp16^T) on the fp16 matrix cores (k_gemm.h, gemm_f16s_mfma_kernel)
static hipError_t linear_h(const float* A, int lda, const _Float16* W, const float* bias, float* C, int ldc, int M, int N, int K,
                           bool relu, const float* resid, int ldr, hipStream_t st) {
    GemmArgs g = gemm_args_default();
    g.A = A; g.B = reinterpret_cast<const float*>(W); g.C = C; g.bias = bias; g.resid = resid;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = K; g.ldc = ldc; g.ldr = ldr;
    g.relu = relu ? 1 : 0;
    return launch_gemm_f16s(g, st);
}

// the same product through the LDS-DMA kernel (k_gemm.h, split form): the fp32 activation is split into hi / lo fp16 arrays by one
// streaming pass, then both A images and the weight tile reach LDS by DMA.  K % 64 == 0 (1536 / 6144 on the prefill path);
// same split and MFMA order as linear_h (fast-mode logits unchanged to the last digit); used for small M only (see the body)
static int linear_hs(er_ctx* c, const float* A, int lda, const _Float16* W, const float* bias, float* C, int ldc, int M, int N, int K,
                     bool relu, const float* resid, int ldr, hipStream_t st);

#define HIPRET(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(ER_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
    } while (0)

static int linear_hs(er_ctx* c, const float* A, int lda, const _Float16* W, const float* bias, float* C, int ldc, int M, int N, int K,
                     bool relu, const float* resid, int ldr, hipStream_t st) {
    // measured (profiles/r03_prefill_fast_gemm.log): with its 64-row tiles (two A images per stage) and the extra split pass the
    // LDS-DMA form wins at 2050 rows (one prefix: encode + prefill 24.2 -> 23.1 ms) and loses to the register-staged 128 x 128 kernel
    // at 16400 rows and beyond (8 prefixes: 13.4 -> 13.8 ms per sample, 32: 12.65 -> 13.1).  The rule is on ROWS - what decides is how
    // many 64-row tiles a CU has to walk, whoever owns the rows - so one long resumed prefix (core/models.py:225-226; 14050 rows in the
    // long-context tests) takes the register-staged kernel exactly as seven short ones would.  Both forms give the same bits
    // (tests/test_gpu_kernels.py::test_gemm_f16s_forms_agree).
    constexpr int HS_MAX_ROWS = 4608;          // two 2050-token prefixes + slack
    if (K % XBK != 0 || M > HS_MAX_ROWS) { HIPRET(linear_h(A, lda, W, bias, C, ldc, M, N, K, relu, resid, ldr, st)); return 0; }
    ERCHK(ensure(c->p_hi, (size_t)M * K / 2 + 8));
    ERCHK(ensure(c->p_lo, (size_t)M * K / 2 + 8));
    _Float16* hi = reinterpret_cast<_Float16*>(c->p_hi.p);
    _Float16* lo = reinterpret_cast<_Float16*>(c->p_lo.p);
    hipLaunchKernelGGL(split_rows_f16_kernel, split_rows_grid(M, K), dim3(ER_WG), 0, st, A, hi, lo, (long long)M, K, lda);
    HIPRET(hipGetLastError());
    GemmArgs g = gemm_args_default();
    g.A = reinterpret_cast<const float*>(hi); g.a_lo = lo; g.B = reinterpret_cast<const float*>(W); g.C = C; g.bias = bias; g.resid = resid;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = ldc; g.ldr = ldr;
    g.relu = relu ? 1 : 0;
    HIPRET(launch_gemm_hh_split(g, st));
    return 0;
}

// softmax(Q K^T / sqrt(D)) V for one sample, all heads; scores live in `sc` ([H][N][ldS]).
static int attention_full(const float* Q, int ldq, const float* Kp, int ldk, long long k_hstride, const float* Vp, int ldv,
                          long long v_hstride, float* out, int ldo, float* sc, int H, int D, int N, int M, bool causal,
                          hipStream_t st) {
    const int ldS = (M + 15) / 16 * 16;
    GemmArgs g = gemm_args_default();
    g.A = Q; g.lda = ldq; g.sA2 = D;
    g.B = Kp; g.ldb = ldk; g.sB2 = k_hstride;
    g.C = sc; g.ldc = ldS; g.sC2 = (long long)N * ldS;
    g.Z2 = H; g.M = N; g.N = M; g.K = D; g.div = sqrtf((float)D);
    g.causal = causal ? 1 : 0; g.causal_off = M - N;
    HIPRET(launch_gemm(g, H, st));
    HIPRET(launch_softmax_rows(sc, N, M, (long long)ldS, ldS, (long long)N * ldS, H, causal ? 1 : 0, M - N, st));
    GemmArgs p = gemm_args_default();
    p.A = sc; p.lda = ldS; p.sA2 = (long long)N * ldS;
    p.B = Vp; p.ldb = ldv; p.sB2 = v_hstride; p.b_is_kn = 1; p.kb_valid = M;
    p.C = out; p.ldc = ldo; p.sC2 = D;
    p.Z2 = H; p.M = N; p.N = D; p.K = ldS;
    p.causal = causal ? 1 : 0; p.causal_off = M - N;
    HIPRET(launch_gemm(p, H, st));
    return 0;
}

// ------------------------------------------------------------------------------------ encode_cond
extern "C" int er_encode_cond(er_ctx* c, const float* conds, int B, int n_points, const int32_t* face_bucket,
                              float* cond_out, void* stream) {
    if (!c || !cond_out || B <= 0) return fail(ER_ERR_INVALID, "er_encode_cond: bad argument");
    ERCHK(er_finalize_weights(c));
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = pick(c, stream);
    const er_config& g = c->cfg;
    const int H = g.hidden_dim, C = g.num_cond_tokens, PH = g.point_hidden_dim, Lq = g.point_latent_size, LD = g.point_latent_dim;
    const int n_lat = (g.cond_mode == ER_COND_NONE) ? 0 : Lq;
    const int n_face = g.num_face_buckets > 0 ? 1 : 0;
    if (n_lat + n_face != C) return fail(ER_ERR_INVALID, "num_cond_tokens %d != latent tokens %d + face token %d", C, n_lat, n_face);
    if (g.cond_mode != ER_COND_NONE && !conds) return fail(ER_ERR_INVALID, "er_encode_cond: conds is null");
    if (LD % 16) return fail(ER_ERR_UNSUPPORTED, "point_latent_dim must be a multiple of 16");

    const int PHh = g.point_num_heads > 0 ? g.point_num_heads : 1, PD = PH / PHh;
    // samples are encoded in chunks of up to 32 (scratch for one chunk at N = 4096: ~5 GB); every GEMM / LayerNorm /
    // attention launch of a chunk covers all of its samples
    constexpr int ENC_CHUNK = 32;
    for (int b0 = 0; b0 < B; b0 += ENC_CHUNK) {
        const int nb = std::min(ENC_CHUNK, B - b0);
        const float* lat = nullptr;   // [nb][Lq][LD]
        if (g.cond_mode == ER_COND_POINT) {
            const int N = n_points;
            if (N <= 0) return fail(ER_ERR_INVALID, "n_points must be > 0");
            const size_t R = (size_t)nb * N, RQ = (size_t)nb * Lq;
            ERCHK(ensure(c->e_a0, R * c->pe_kpad));
            ERCHK(ensure(c->e_x, R * PH));
            ERCHK(ensure(c->e_k, R * PH));
            ERCHK(ensure(c->e_v, R * PH));
            ERCHK(ensure(c->e_qln, (size_t)Lq * PH));
            ERCHK(ensure(c->e_q, (size_t)Lq * PH));
            ERCHK(ensure(c->e_att, RQ * PH));
            ERCHK(ensure(c->e_l, RQ * PH));
            ERCHK(ensure(c->e_ln, RQ * PH));
            ERCHK(ensure(c->e_u, RQ * 8 * PH));
            ERCHK(ensure(c->e_g, RQ * 4 * PH));
            ERCHK(ensure(c->e_lat, RQ * LD));
            const float* pts = conds + (size_t)b0 * N * 3;
            // x = ln(point_embed(pts))                                          point.py:194
            hipLaunchKernelGGL(point_embed_kernel, dim3(ew_grid((long long)R * c->pe_kpad)), dim3(ER_WG), 0, st, pts,
                               c->pe_basis, c->e_a0.p, (long long)R, g.point_freq_dim, c->pe_kpad);
            HIPRET(hipGetLastError());
            HIPRET(linear(c->e_a0.p, c->pe_kpad, c->pe_mlp_w, c->pe_mlp_b, c->e_x.p, PH, (int)R, PH, c->pe_kpad, false, nullptr, 0, st));
            HIPRET(launch_layernorm(c->e_x.p, c->pe_ln_w, c->pe_ln_b, c->e_x.p, (int)R, PH, PH, PH, g.ln_eps, st));
            // cross attention: l = q + out_proj(attn(q_proj(ln1(q)), k_proj(x), v_proj(x)))   point.py:123-124
            // (the learned queries and their projection are the same for every sample: computed once)
            HIPRET(launch_layernorm(c->pe_query, c->ca_ln1_w, c->ca_ln1_b, c->e_qln.p, Lq, PH, PH, PH, g.ln_eps, st));
            HIPRET(linear(c->e_qln.p, PH, c->ca_q_w, c->ca_q_b, c->e_q.p, PH, Lq, PH, PH, false, nullptr, 0, st));
            HIPRET(linear(c->e_x.p, PH, c->ca_k_w, c->ca_k_b, c->e_k.p, PH, (int)R, PH, PH, false, nullptr, 0, st));
            HIPRET(linear(c->e_x.p, PH, c->ca_v_w, c->ca_v_b, c->e_v.p, PH, (int)R, PH, PH, false, nullptr, 0, st));
            if (PD == 64 || PD == 96) {
                Flash32Args f{};
                f.Q = c->e_q.p; f.ldq = PH; f.qs_b = 0; f.qs_h = PD;                      // queries shared by the batch
                f.K = c->e_k.p; f.ldk = PH; f.ks_b = (long long)N * PH; f.ks_h = PD;
                f.V = c->e_v.p; f.ldv = PH; f.vs_b = (long long)N * PH; f.vs_h = PD;
                f.O = c->e_att.p; f.ldo = PH; f.os_b = (long long)Lq * PH; f.os_h = PD;
                f.N = Lq; f.M = N; f.sqrt_d = sqrtf((float)PD); f.causal_off = 0;
                HIPRET(launch_flash_attn_f32(f, PD, false, PHh, nb, st));
            } else {
                const int ldS = (N + 15) / 16 * 16;
                ERCHK(ensure(c->e_sc, (size_t)PHh * Lq * ldS));
                for (int b = 0; b < nb; ++b)
                    ERCHK(attention_full(c->e_q.p, PH, c->e_k.p + (size_t)b * N * PH, PH, PD, c->e_v.p + (size_t)b * N * PH, PH, PD,
                                         c->e_att.p + (size_t)b * Lq * PH, PH, c->e_sc.p, PHh, PD, Lq, N, false, st));
            }
            {   // l = query_embed + out_proj(att): the residual table has Lq rows shared by every sample
                GemmArgs ga = gemm_args_default();
                ga.A = c->e_att.p; ga.B = c->ca_o_w; ga.C = c->e_l.p; ga.bias = c->ca_o_b; ga.resid = c->pe_query; ga.resid_mod = Lq;
                ga.M = (int)RQ; ga.N = PH; ga.K = PH; ga.lda = PH; ga.ldb = PH; ga.ldc = PH; ga.ldr = PH;
                HIPRET(launch_gemm(ga, 1, st));
            }
            // l = l + net2(GEGLU(net0(ln2(l))))                                   point.py:125, 68-84
            HIPRET(launch_layernorm(c->e_l.p, c->ca_ln2_w, c->ca_ln2_b, c->e_ln.p, (int)RQ, PH, PH, PH, g.ln_eps, st));
            HIPRET(linear(c->e_ln.p, PH, c->ff0_w, c->ff0_b, c->e_u.p, 8 * PH, (int)RQ, 8 * PH, PH, false, nullptr, 0, st));
            hipLaunchKernelGGL(geglu_kernel, dim3(ew_grid((long long)RQ * 4 * PH)), dim3(ER_WG), 0, st, c->e_u.p, c->e_g.p,
                               (long long)RQ, 4 * PH);
            HIPRET(hipGetLastError());
            HIPRET(linear(c->e_g.p, 4 * PH, c->ff2_w, c->ff2_b, c->e_l.p, PH, (int)RQ, PH, 4 * PH, false, c->e_l.p, PH, st));
            // latent mean = linear(l)                                              point.py:201
            HIPRET(linear(c->e_l.p, PH, c->lin_w, c->lin_b, c->e_lat.p, LD, (int)RQ, LD, PH, false, nullptr, 0, st));
            lat = c->e_lat.p;
        } else if (g.cond_mode == ER_COND_POINT_LATENT) {
            lat = conds + (size_t)b0 * Lq * LD;
        }
        if (lat) {   // norm_cond(proj_cond(latent))                               core/models.py:124 / 128-129
            ERCHK(ensure(c->e_tmp, (size_t)nb * Lq * H));
            HIPRET(linear(lat, LD, c->proj_w, c->proj_b, c->e_tmp.p, H, nb * Lq, H, LD, false, nullptr, 0, st));
        }
        for (int b = 0; b < nb; ++b) {
            float* out_b = cond_out + (size_t)(b0 + b) * C * H;
            if (lat) HIPRET(launch_layernorm(c->e_tmp.p + (size_t)b * Lq * H, c->normc_w, c->normc_b, out_b, Lq, H, H, H, g.ln_eps, st));
            if (n_face) {   // embed_num_face(quantize_num_faces(n))                     core/models.py:135-139
                const int bucket = face_bucket ? face_bucket[b0 + b] : 0;
                if (bucket < 0 || bucket >= g.num_face_buckets) return fail(ER_ERR_INVALID, "face bucket %d out of range", bucket);
                HIPCHK(hipMemcpyAsync(out_b + (size_t)n_lat * H, c->embed_num_face + (size_t)bucket * H, (size_t)H * 4,
                                      hipMemcpyDeviceToDevice, st));
            }
        }
    }
    return ER_OK;
}

extern "C" int er_embed_tokens(er_ctx* c, const int32_t* ids, int B, int R, float* out, void* stream) {
    if (!c || !ids || !out || B <= 0 || R <= 0) return fail(ER_ERR_INVALID, "er_embed_tokens: bad argument");
    ERCHK(er_finalize_weights(c));
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = pick(c, stream);
    const int H = c->cfg.hidden_dim;
    const int n = B * R;
    for (int i = 0; i < n; ++i)
        if (ids[i] < 0 || ids[i] >= c->cfg.vocab_size) return fail(ER_ERR_INVALID, "token id %d out of range", ids[i]);
    // one gather launch (round 1 issued one hipMemcpyAsync per token: a 2000-token resume prefix was 2000 copies)
    ERCHK(ensure(c->e_ids, (size_t)n));                     // grow-only int scratch (4-byte slots)
    int* d_ids = reinterpret_cast<int*>(c->e_ids.p);
    HIPCHK(hipMemcpyAsync(d_ids, ids, (size_t)n * sizeof(int), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(gather_rows_kernel, dim3(n), dim3(ER_WG), 0, st, c->embd, d_ids, out, n, H, (long long)H);
    HIPRET(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));     // `ids` is caller-owned host memory
    return ER_OK;
}

// ------------------------------------------------------------------------------------ prefill
// Ragged last row tile of the exact prefill (ER_PREFILL_TAIL=0 switches it off): 2050 prefix rows are 32 row tiles of 64
// plus TWO rows, and those two rows cost out_proj / fc2 a fourth round of tiles on 24 CUs (792 tiles of 64 x 64 = 3.09 per CU) and
// fc1 a seventh half round (1584 tiles of 64 x 128 = 6.19 per CU): encode + prefill 42.2 -> 38.2 ms on the same box
// (profiles/r04_prefill_tail.log).  The out_proj / fc1 / fc2 Linears run on the
// first M - M % 64 rows and the 1..8 left-over rows go through the decode step's own fp32 GEMV kernels (one pass over the matrix,
// 5..6 us), whenever that saves a round of 64-row tiles.  Rows are independent in a Linear, so the split needs no sample boundary.
static int prefill_tail_rows(const er_ctx* c, int M) {
    const char* v = getenv("ER_PREFILL_TAIL");
    if ((v && atoi(v) == 0) || c->fast) return 0;
    if (c->cfg.hidden_dim != 1536 || c->cfg.intermediate_dim != 6144) return 0;      // the GEMV kernels are built for these K
    const int tail = M % 64;
    if (tail < 1 || tail > 8 || M < 1024) return 0;
    const long long nt = c->cfg.hidden_dim / 64;
    const long long r_all = ((long long)(M / 64 + 1) * nt + 255) / 256, r_main = ((long long)(M / 64) * nt + 255) / 256;
    return r_all > r_main ? tail : 0;
}
// C[tail rows] = epilogue(A . W^T): dense rows (lda == K, ldc == ldr == N) behind the GEMM's main part
template <int KS, int RW, int EPI>
static hipError_t linear_tail(const float* A, const float* W, const float* bias, float* C, const float* resid, int rows, int N, int K,
                              hipStream_t st) {
    GemvArgs t{};
    t.W = W; t.bias = bias; t.N = N; t.xin = A; t.out = C; t.resid = resid;
    return gemv_groups<float, KS, RW, PRO_NONE, EPI>(t, rows, K, st);
}

extern "C" int er_prefill(er_ctx* c, const float* embeds, int B, int S, void* stream) {
    if (!c || !embeds || B <= 0 || S <= 0) return fail(ER_ERR_INVALID, "er_prefill: bad argument");
    ERCHK(er_finalize_weights(c));
    if (c->B != B) return fail(ER_ERR_INVALID, "er_prefill: batch %d but KV cache reserved for %d (call er_kv_reserve)", B, c->B);
    if (S >= c->Lcap) return fail(ER_ERR_CAPACITY, "prefix length %d does not fit the reserved KV cache (%d)", S, c->Lcap);
    HIPCHK(hipSetDevice(c->device));
    if (c->batched && !c->batched_valu) ERCHK(make_tiled_weights(c));
    hipStream_t st = pick(c, stream);
    const er_config& g = c->cfg;
    const int H = g.hidden_dim, I = g.intermediate_dim, NH = g.num_heads, D = c->D;
    const int M = B * S;
    ERCHK(ensure(c->p_h, (size_t)M * H));
    ERCHK(ensure(c->p_q, (size_t)M * H));
    ERCHK(ensure(c->p_a, (size_t)M * H));
    ERCHK(ensure(c->p_y, (size_t)M * H));
    ERCHK(ensure(c->p_f, (size_t)M * I));
    float *h = c->p_h.p, *q = c->p_q.p, *a = c->p_a.p, *y = c->p_y.p, *f = c->p_f.p;
    const int tail = prefill_tail_rows(c, M), Mm = M - tail;
    // the causal attention of a single prefix is split over two key ranges per query tile (k_flash_attn_f32.h, KSP; same rule as the launcher)
    bool attn_ksplit = false;
    if (!c->fast && flash32_ksplit(S, NH, B, D, true)) {
        ERCHK(ensure(c->p_ap, flash32_part_o_floats(B, NH, S, D)));
        ERCHK(ensure(c->p_aml, flash32_part_ml_floats(B, NH, S)));
        attn_ksplit = true;
    }

    // hidden = inputs_embeds + pos_embeds(0..S)                       modeling_opt.py:355-357
    hipLaunchKernelGGL(add_pos_kernel, dim3(ew_grid((long long)M * H / 4)), dim3(ER_WG), 0, st, embeds, c->posemb, h, B, S, H, 0);
    HIPRET(hipGetLastError());
    for (int l = 0; l < g.num_layers; ++l) {
        const LayerW& L = c->layers[l];
        char* kc = (char*)c->kc + (long long)l * c->kv_lstride * c->kv_esz;
        char* vc = (char*)c->vc + (long long)l * c->kv_lstride * c->kv_esz;
        if (!c->fast) {
            // q,k,v projections; k,v go straight into the cache layout      modeling_opt.py:185-196
            GemmArgs qa = gemm_args_default();
            qa.A = h; qa.lda = H; qa.B = L.wqkv; qa.ldb = H; qa.bias = L.bqkv; qa.C = q; qa.ldc = H;
            qa.M = M; qa.N = 3 * H; qa.K = H; qa.epi = GEPI_QKV;
            qa.q = q; qa.kcache = (float*)kc; qa.vcache = (float*)vc; qa.S = S; qa.hidden = H; qa.head_dim = D; qa.l_cap = c->Lcap;
            qa.kv_bstride = c->kv_bstride;
            HIPRET(launch_gemm(qa, 1, st));
            {       // causal attention over the prefix, all samples and heads in one launch   modeling_opt.py:229
                Flash32Args f{};
                f.Q = q; f.ldq = H; f.qs_b = (long long)S * H; f.qs_h = D;
                f.K = (float*)kc; f.ldk = D; f.ks_b = c->kv_bstride; f.ks_h = (long long)c->Lcap * D;
                f.V = (float*)vc; f.ldv = D; f.vs_b = c->kv_bstride; f.vs_h = (long long)c->Lcap * D;
                f.O = a; f.ldo = H; f.os_b = (long long)S * H; f.os_h = D;
                f.N = S; f.M = S; f.sqrt_d = sqrtf((float)D); f.causal_off = 0;
                if (attn_ksplit) { f.part_o = c->p_ap.p; f.part_ml = c->p_aml.p; }
                HIPRET(launch_flash_attn_f32(f, D, true, NH, B, st));
            }
        } else {
            // fast mode: fused projection into fp32 scratch [M][3H]; K/V rounded to the cache dtype (fp16) both in
            // the cache and in the scratch the prefix attention reads
            ERCHK(ensure(c->p_qkv, (size_t)M * 3 * H));
            float* qkv = c->p_qkv.p;
            ERCHK(linear_hs(c, h, H, L.wqkv_h, L.bqkv, qkv, 3 * H, M, 3 * H, H, false, nullptr, 0, st));
            hipLaunchKernelGGL(kv_scatter_half_kernel, kv_scatter_grid(M, H), dim3(ER_WG), 0, st, qkv,
                               (_Float16*)kc, (_Float16*)vc, M, S, H, D, c->Lcap, c->kv_bstride);
            HIPRET(hipGetLastError());
            {
                Flash32Args f{};
                f.Q = qkv; f.ldq = 3 * H; f.qs_b = (long long)S * 3 * H; f.qs_h = D;
                f.K = qkv + H; f.ldk = 3 * H; f.ks_b = f.qs_b; f.ks_h = D;
                f.V = qkv + 2 * H; f.ldv = 3 * H; f.vs_b = f.qs_b; f.vs_h = D;
                f.O = a; f.ldo = H; f.os_b = (long long)S * H; f.os_h = D;
                f.N = S; f.M = S; f.sqrt_d = sqrtf((float)D); f.causal_off = 0;
                const bool f16s = D == 96 && c->prefill_attn_f16s != 0;
                if (f16s) HIPRET(launch_flash_attn_f16s(f, D, true, NH, B, st));   // K / V in the scratch are fp16 values already
                else HIPRET(launch_flash_attn_f32(f, D, true, NH, B, st));
            }
        }
        // y = h + out_proj(a); h1 = LN1(y)                               modeling_opt.py:232, 272-274
        const bool hs = c->fast;
        if (hs) ERCHK(linear_hs(c, a, H, L.wo_h, L.bo, y, H, M, H, H, false, h, H, st));
        else {
            HIPRET(linear(a, H, L.wo, L.bo, y, H, Mm, H, H, false, h, H, st));
            if (tail) HIPRET((linear_tail<1, 1, EPI_RESID>(a + (size_t)Mm * H, L.wo, L.bo, y + (size_t)Mm * H, h + (size_t)Mm * H, tail, H, H, st)));
        }
        HIPRET(launch_layernorm(y, L.ln1w, L.ln1b, h, M, H, H, H, g.ln_eps, st));
        // y = h1 + fc2(relu(fc1(h1))); h = LN2(y)                        modeling_opt.py:281-288
        if (hs) {
            ERCHK(linear_hs(c, h, H, L.w1_h, L.b1, f, I, M, I, H, true, nullptr, 0, st));
            ERCHK(linear_hs(c, f, I, L.w2_h, L.b2, y, H, M, H, I, false, h, H, st));
        } else {
            HIPRET(linear(h, H, L.w1, L.b1, f, I, Mm, I, H, true, nullptr, 0, st));
            if (tail) HIPRET((linear_tail<1, 2, EPI_RELU>(h + (size_t)Mm * H, L.w1, L.b1, f + (size_t)Mm * I, nullptr, tail, I, H, st)));
            HIPRET(linear(f, I, L.w2, L.b2, y, H, Mm, H, I, false, h, H, st));
            if (tail) HIPRET((linear_tail<4, 2, EPI_RESID>(f + (size_t)Mm * I, L.w2, L.b2, y + (size_t)Mm * H, h + (size_t)Mm * H, tail, H, I, st)));
        }
        if (l + 1 < g.num_layers) HIPRET(launch_layernorm(y, L.ln2w, L.ln2b, h, M, H, H, H, g.ln_eps, st));
    }
    // keep the last position's pre-LN2 state: the decode head applies LN2 + lm_head to it
    for (int b = 0; b < B; ++b)
        HIPCHK(hipMemcpyAsync(c->ypre + (size_t)b * H, y + ((size_t)b * S + S - 1) * H, (size_t)H * 4, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(init_state_kernel, dim3((B + 63) / 64), dim3(64), 0, st, c->st, B, S);
    HIPRET(hipGetLastError());
    c->base_pos = S;
    c->have_hidden = true;
    return ER_OK;
}

extern "C" int er_logits(er_ctx* c, float* out, void* stream) {
    if (!c || !out) return fail(ER_ERR_INVALID, "er_logits: bad argument");
    if (!c->have_hidden) return fail(ER_ERR_INVALID, "er_logits: no forward pass has run (call er_prefill)");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = pick(c, stream);
    HIPRET(launch_kind(c, 6, 0, st, nullptr, 0));
    HIPCHK(hipMemcpyAsync(out, c->logits, (size_t)c->B * c->cfg.vocab_size * 4, hipMemcpyDeviceToDevice, st));
    return ER_OK;
}

static int check_room(er_ctx* c, int extra) {
    // generated token t is fed at position base_pos + t
    HIPCHK(hipMemcpy(c->h_pinned, c->st.ngen, sizeof(int), hipMemcpyDeviceToHost));
    const int used = c->base_pos + c->h_pinned[0];
    if (used + extra > c->Lcap) return fail(ER_ERR_CAPACITY, "KV cache full: %d + %d > %d", used, extra, c->Lcap);
    if (used + extra > c->cfg.max_positions) return fail(ER_ERR_CAPACITY, "position table exhausted (%d)", c->cfg.max_positions);
    return 0;
}

extern "C" int er_feed(er_ctx* c, const int32_t* ids, void* stream) {
    if (!c || !ids) return fail(ER_ERR_INVALID, "er_feed: bad argument");
    if (!c->have_hidden) return fail(ER_ERR_INVALID, "er_feed: call er_prefill first");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = pick(c, stream);
    HIPCHK(hipStreamSynchronize(st));
    ERCHK(check_room(c, 1));
    for (int b = 0; b < c->B; ++b) {
        if (ids[b] < 0 || ids[b] >= c->cfg.vocab_size) return fail(ER_ERR_INVALID, "token id %d out of range", ids[b]);
        c->h_pinned[b] = ids[b];
    }
    HIPCHK(hipMemcpyAsync(c->d_ids_tmp, c->h_pinned, c->B * sizeof(int), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(force_token_kernel, dim3((c->B + 63) / 64), dim3(64), 0, st, c->d_ids_tmp, c->st, c->B);
    HIPRET(hipGetLastError());
    HIPRET(enqueue_layers(c, st));
    HIPCHK(hipStreamSynchronize(st));   // h_pinned may be reused by the next call
    return ER_OK;
}

// ------------------------------------------------------------------------------------ generation loop
__global__ void fill_i64_kernel(long long* p, long long n, long long v) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void reset_gen_kernel(GenState st, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0) { *st.n_unfinished = B; *st.error = 0; }
    if (b >= B) return;
    st.counter[b] = 0; st.unfinished[b] = 1; st.eos_step[b] = -1;
}

extern "C" int er_decode(er_ctx* c, const er_decode_params* p, int64_t* out_ids, int32_t* n_steps, void* stream) {
    if (!c || !p || !out_ids || !n_steps) return fail(ER_ERR_INVALID, "er_decode: bad argument");
    if (!c->have_hidden) return fail(ER_ERR_INVALID, "er_decode: call er_prefill first");
    if (p->max_new_tokens <= 0) return fail(ER_ERR_INVALID, "max_new_tokens must be > 0");
    if (p->mode != ER_GREEDY && p->mode != ER_SAMPLE) return fail(ER_ERR_INVALID, "bad mode");
    if (p->grammar < 0 || p->grammar > 2) return fail(ER_ERR_INVALID, "bad grammar");
    if (p->mode == ER_SAMPLE && p->top_k <= 0) return fail(ER_ERR_INVALID, "top_k must be > 0 in sample mode");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = pick(c, stream);
    const int B = c->B, T = p->max_new_tokens;
    HIPCHK(hipStreamSynchronize(st));
    ERCHK(check_room(c, T));
    if (c->h_pinned[0] != 0) return fail(ER_ERR_INVALID, "er_decode must directly follow er_prefill (%d tokens already fed)", c->h_pinned[0]);

    DecodeParamsDev dp{};
    dp.mode = p->mode; dp.top_k = p->top_k; dp.grammar = p->grammar; dp.max_new = T; dp.min_new = p->min_new_tokens;
    dp.eos = c->cfg.eos_token_id; dp.pad = c->cfg.pad_token_id; dp.vocab = c->cfg.vocab_size;
    dp.seed_lo = (unsigned int)(p->seed & 0xffffffffu); dp.seed_hi = (unsigned int)(p->seed >> 32);
    HIPCHK(hipMemcpy(c->d_params, &dp, sizeof(dp), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(reset_gen_kernel, dim3((B + 63) / 64), dim3(64), 0, st, c->st, B);
    HIPRET(hipGetLastError());
    hipLaunchKernelGGL(fill_i64_kernel, dim3(ew_grid((long long)B * c->Lcap)), dim3(ER_WG), 0, st, c->d_out_ids,
                       (long long)B * c->Lcap, (long long)dp.pad);
    HIPRET(hipGetLastError());

    // one step = lm_head -> sampling head -> 24 layers on the chosen token; captured once, replayed T times
    if (c->use_graph && !c->step_exec) {
        hipGraph_t graph = nullptr;
        HIPCHK(hipStreamBeginCapture(c->own_stream, hipStreamCaptureModeRelaxed));
        hipError_t e = enqueue_step(c, c->own_stream, c->d_out_ids, c->Lcap);
        hipError_t e2 = hipStreamEndCapture(c->own_stream, &graph);
        if (e != hipSuccess || e2 != hipSuccess) {
            if (graph) hipGraphDestroy(graph);
            return fail(ER_ERR_HIP, "graph capture failed: %s / %s", hipGetErrorString(e), hipGetErrorString(e2));
        }
        HIPCHK(hipGraphInstantiate(&c->step_exec, graph, nullptr, nullptr, 0));
        hipGraphDestroy(graph);
    }

    const int check_every = 32;
    int steps_run = 0;
    bool all_done = false;
    HIPCHK(hipEventRecord(c->ev0, st));
    for (int t = 0; t < T; ++t) {
        if (c->use_graph) HIPCHK(hipGraphLaunch(c->step_exec, st));
        else HIPRET(enqueue_step(c, st, c->d_out_ids, c->Lcap));
        steps_run = t + 1;
        if (steps_run >= p->min_new_tokens && steps_run < T && (steps_run % check_every) == 0) {
            HIPCHK(hipMemcpyAsync(c->h_pinned, c->st.n_unfinished, sizeof(int), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            if (c->h_pinned[0] <= 0) { all_done = true; break; }
        }
    }
    HIPCHK(hipEventRecord(c->ev1, st));
    HIPCHK(hipMemcpy2DAsync(out_ids, (size_t)T * sizeof(long long), c->d_out_ids, (size_t)c->Lcap * sizeof(long long),
                            (size_t)T * sizeof(long long), B, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(c->h_pinned, c->st.eos_step, B * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(c->h_pinned + B, c->st.error, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipEventElapsedTime(&c->last_decode_ms, c->ev0, c->ev1));
    if (c->h_pinned[B] != 0)
        return fail(ER_ERR_INVALID, "er_decode: a row had no finite candidate score (non-finite logits); HF would raise in multinomial/argmax");
    (void)all_done;
    // HF returns as many columns as steps it ran: it stops right after the step in which the last row emits EOS
    int last = -1;
    bool finished = true;
    for (int b = 0; b < B; ++b) {
        if (c->h_pinned[b] < 0) finished = false;
        else if (c->h_pinned[b] > last) last = c->h_pinned[b];
    }
    *n_steps = finished ? last + 1 : T;
    if (!finished && steps_run < T) return fail(ER_ERR_INVALID, "internal: stopped early with unfinished rows");
    return ER_OK;
}

extern "C" int er_set_row_streams(er_ctx* c, const uint32_t* ids, int n) {
    if (!c) return fail(ER_ERR_INVALID, "er_set_row_streams: null context");
    if (c->B <= 0) return fail(ER_ERR_INVALID, "er_set_row_streams: no cache reserved (call er_kv_reserve)");
    if (ids && n != c->B) return fail(ER_ERR_INVALID, "er_set_row_streams: %d ids for a batch of %d rows", n, c->B);
    HIPCHK(hipSetDevice(c->device));
    std::vector<unsigned int> h((size_t)c->B);
    for (int i = 0; i < c->B; ++i) h[i] = ids ? ids[i] : (unsigned int)i;
    HIPCHK(hipDeviceSynchronize());           // a running decode still reads the old ids
    HIPCHK(hipMemcpy(c->d_row_stream, h.data(), h.size() * sizeof(unsigned int), hipMemcpyHostToDevice));
    return ER_OK;
}

extern "C" int er_last_decode_ms(er_ctx* c, float* ms) {
    if (!c || !ms) return fail(ER_ERR_INVALID, "null");
    *ms = c->last_decode_ms;
    return ER_OK;
}

// ------------------------------------------------------------------------------------ per-kernel timing
static int profile_impl(er_ctx* c, int repeats, int use_graph, float* avg_us, double* bytes, void* stream);

extern "C" int er_profile_decode_kernels(er_ctx* c, int repeats, float* avg_us, double* bytes, void* stream) {
    return er_profile_decode_kernels_at(c, repeats, 0, 0, avg_us, bytes, stream);
}

extern "C" int er_profile_decode_kernels_at(er_ctx* c, int repeats, int context_len, int use_graph, float* avg_us, double* bytes,
                                            void* stream) {
    if (!c) return fail(ER_ERR_INVALID, "er_profile_decode_kernels: bad argument");
    if (context_len < 0 || context_len > c->Lcap) return fail(ER_ERR_CAPACITY, "profile: context_len %d outside the reserved cache (%d)", context_len, c->Lcap);
    c->prof_len = context_len;
    const int rc = profile_impl(c, repeats, use_graph, avg_us, bytes, stream);
    c->prof_len = 0;
    return rc;
}

static int profile_impl(er_ctx* c, int repeats, int use_graph, float* avg_us, double* bytes, void* stream) {
    if (!c || !avg_us || !bytes || repeats <= 0) return fail(ER_ERR_INVALID, "er_profile_decode_kernels: bad argument");
    if (!c->have_hidden) return fail(ER_ERR_INVALID, "profile: call er_prefill first");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = pick(c, stream);
    const er_config& g = c->cfg;
    const int B = c->B, H = g.hidden_dim, I = g.intermediate_dim, nl = g.num_layers, V = g.vocab_size;
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipMemcpy(c->h_pinned, c->st.pos, sizeof(int), hipMemcpyDeviceToHost));
    const double len = c->prof_len > 0 ? (double)c->prof_len : (double)c->h_pinned[0] + 1.0;
    // save the state the sweep scribbles on
    std::vector<float> save_y((size_t)B * H);
    HIPCHK(hipMemcpy(save_y.data(), c->ypre, save_y.size() * 4, hipMemcpyDeviceToHost));
    std::vector<int> save_state(7 * (size_t)B + 8);
    HIPCHK(hipMemcpy(save_state.data(), c->state_block, save_state.size() * sizeof(int), hipMemcpyDeviceToHost));
    std::vector<int> head_state = save_state;          // the head sweep runs at step 0 so it writes dummy_ids[b][0]
    for (int b = 0; b < B; ++b) head_state[3 * (size_t)B + b] = 0;
    DecodeParamsDev dp{};
    dp.mode = 0; dp.top_k = 10; dp.grammar = 2; dp.max_new = 1 << 30; dp.min_new = 0;
    dp.eos = g.eos_token_id; dp.pad = g.pad_token_id; dp.vocab = V;
    HIPCHK(hipMemcpy(c->d_params, &dp, sizeof(dp), hipMemcpyHostToDevice));
    long long* dummy_ids = nullptr;
    HIPCHK(hipMalloc(&dummy_ids, (size_t)B * 8 * sizeof(long long)));

    const double w = c->fast ? 2.0 : 4.0;   // bytes per streamed weight / KV element
    bytes[0] = ((double)3 * H * H + 3 * H) * w + (double)B * (H + 3 * H) * w;
    bytes[1] = (double)B * 2.0 * len * H * w;
    bytes[2] = (double)B * g.num_heads * c->S_splits * (c->D + 2) * w + (double)B * H * w;
    bytes[3] = ((double)H * H + H) * w + (double)B * 3 * H * w;
    bytes[4] = ((double)I * H + I) * w + (double)B * (H + I) * w;
    bytes[5] = ((double)I * H + H) * w + (double)B * (I + 2 * H) * w;
    bytes[6] = ((double)V * H) * w + (double)B * (H + V) * w;
    bytes[7] = (double)B * V * w;

    if (c->v3) {   // the merge is part of the out_proj launch: its partial reads are charged there
        bytes[3] += (double)g.num_heads * c->nch3 * (c->D + 2) * 4.0;
        bytes[2] = 0.0;
    }
    for (int kind = 0; kind < ER_NUM_KERNEL_KINDS; ++kind) {
        const bool per_layer = kind <= 5;
        if ((c->v3 || c->stream_attn) && kind == 2) { avg_us[kind] = 0.f; continue; }
        // warm-up + timed sweeps
        hipGraphExec_t gexec = nullptr;
        if (use_graph && per_layer) {     // the nl launches of this kind as one replayable graph (what the generation loop replays)
            hipGraph_t graph = nullptr;
            HIPCHK(hipStreamBeginCapture(c->own_stream, hipStreamCaptureModeRelaxed));
            hipError_t e = hipSuccess;
            for (int l = 0; l < nl && e == hipSuccess; ++l) e = launch_kind(c, kind, l, c->own_stream, dummy_ids, 8);
            hipError_t e2 = hipStreamEndCapture(c->own_stream, &graph);
            if (e != hipSuccess || e2 != hipSuccess) { if (graph) hipGraphDestroy(graph); return fail(ER_ERR_HIP, "profile: graph capture failed"); }
            HIPCHK(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
            hipGraphDestroy(graph);
        }
        for (int pass = 0; pass < 2; ++pass) {
            const int reps = pass == 0 ? 1 : repeats;
            if (pass == 1) HIPCHK(hipEventRecord(c->ev0, st));
            int launches = 0;
            for (int r = 0; r < reps; ++r) {
                if (gexec) {
                    HIPCHK(hipGraphLaunch(gexec, st));
                    launches += nl;
                } else if (per_layer) {
                    for (int l = 0; l < nl; ++l) { HIPRET(launch_kind(c, kind, l, st, dummy_ids, 8)); ++launches; }
                } else {
                    for (int l = 0; l < nl; ++l) {   // same number of back-to-back launches
                        if (kind == 7) {   // keep the head's step counter in range
                            HIPCHK(hipMemcpyAsync(c->state_block, head_state.data(), head_state.size() * sizeof(int), hipMemcpyHostToDevice, st));
                        }
                        HIPRET(launch_kind(c, kind, 0, st, dummy_ids, 8));
                        ++launches;
                    }
                }
            }
            if (pass == 1) {
                HIPCHK(hipEventRecord(c->ev1, st));
                HIPCHK(hipStreamSynchronize(st));
                float ms = 0.f;
                HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
                avg_us[kind] = ms * 1000.0f / (float)launches;
            }
        }
        if (gexec) hipGraphExecDestroy(gexec);
    }
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipMemcpy(c->ypre, save_y.data(), save_y.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->state_block, save_state.data(), save_state.size() * sizeof(int), hipMemcpyHostToDevice));
    hipFree(dummy_ids);
    return ER_OK;
}

// ------------------------------------------------------------------------------------ single-kernel entry points
extern "C" int er_k_gemv(const float* w, const float* bias, const float* x, const float* ln_w, const float* ln_b,
                         const float* resid, float* y, float* xnorm_out, int B, int n, int k, int relu, float eps, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    GemvArgs a{};
    a.W = w; a.bias = bias; a.N = n; a.xin = x; a.ln_w = ln_w; a.ln_b = ln_b; a.eps = eps; a.hout = xnorm_out;
    a.out = y; a.resid = resid;
    hipError_t e;
    if (B > 4) {   // batched kernels: LayerNorm rows first (same arithmetic as the fused prologue), then passes of 32 (VALU: 16) rows
        float* tmp = nullptr;
        if (ln_w) {
            if (k != 1536) return fail(ER_ERR_UNSUPPORTED, "er_k_gemv: LayerNorm prologue needs k=1536");
            if (!xnorm_out) { HIPCHK(hipMalloc(&tmp, (size_t)B * k * 4)); a.hout = tmp; }
            e = prep_rows<PRO_LN>(a, B, st);
            if (e != hipSuccess) { if (tmp) hipFree(tmp); HIPRET(e); }
            a.xin = a.hout;
        }
        const char* bv = getenv("ER_BATCHED_VALU");
        const bool valu = bv && bv[0] == '1';
        float* part = nullptr;
        void* wt = nullptr;              // tiled copy of w for the matrix-core kernels (the decode step keeps one per matrix)
        const bool mfma = !valu && ((k == 1536 && relu && !resid) || (k == 6144 && !relu && resid && !ln_w));
        if (mfma) {
            HIPCHK(hipMalloc(&part, (size_t)4 * NBM * n * 4));
            HIPCHK(hipMalloc(&wt, tiled_weight_bytes<float>(n, k)));
            hipLaunchKernelGGL((tile_weights_kernel<float>), dim3(1024), dim3(ER_WG), 0, st, w, reinterpret_cast<f32x4*>(wt), n, k);
        }
        GemvArgs am = a;
        am.W = wt;
        if (k == 1536) {
            if (relu && !resid) e = valu ? gemv_batched_groups<float, 1, 2, EPI_RELU>(a, B, k, st) : gemv_mfma_groups<float, EPI_RELU>(am, B, k, SkPart{part, (size_t)4 * NBM * n}, st);
            else if (!relu && !resid) e = gemv_batched_groups<float, 1, 1, EPI_STORE>(a, B, k, st);   // narrow: VALU kernel, as in the decode step
            else if (!relu && resid) e = gemv_batched_groups<float, 1, 1, EPI_RESID>(a, B, k, st);
            else e = hipErrorInvalidValue;
        } else if (k == 6144 && !relu && resid && !ln_w) {
            e = valu ? gemv_batched_groups<float, 4, 1, EPI_RESID>(a, B, k, st) : gemv_mfma_groups<float, EPI_RESID>(am, B, k, SkPart{part, (size_t)4 * NBM * n}, st);
        } else {
            e = hipErrorInvalidValue;
        }
        if (wt) { hipStreamSynchronize(st); hipFree(wt); }
        if (part) { hipStreamSynchronize(st); hipFree(part); }
        hipError_t e2 = hipStreamSynchronize(st);
        if (tmp) hipFree(tmp);
        HIPRET(e);
        HIPRET(e2);
        return ER_OK;
    }
    if (k == 1536) {
        if (ln_w && relu && !resid) e = gemv_groups<float, 1, 2, PRO_LN, EPI_RELU>(a, B, k, st);
        else if (ln_w && !relu && !resid) e = gemv_groups<float, 1, 1, PRO_LN, EPI_STORE>(a, B, k, st);
        else if (!ln_w && !relu && resid) e = gemv_groups<float, 1, 1, PRO_NONE, EPI_RESID>(a, B, k, st);
        else return fail(ER_ERR_UNSUPPORTED, "er_k_gemv: combination not instantiated for k=1536");
    } else if (k == 6144) {
        if (!ln_w && !relu && resid) e = gemv_groups<float, 4, 2, PRO_NONE, EPI_RESID>(a, B, k, st);
        else return fail(ER_ERR_UNSUPPORTED, "er_k_gemv: combination not instantiated for k=6144");
    } else {
        return fail(ER_ERR_UNSUPPORTED, "er_k_gemv: k must be 1536 or 6144");
    }
    HIPRET(e);
    return ER_OK;
}

extern "C" int er_k_attn_decode(const float* q, const void* k, const void* v, const int32_t* len_host, float* out, int B,
                                int heads, int head_dim, int l_cap, int steps, int kv_half, int variant, void* stream) {
    if (variant != ER_ATTN_SPLIT1 && variant != ER_ATTN_SPLIT2 && variant != ER_ATTN_STREAM)
        return fail(ER_ERR_INVALID, "er_k_attn_decode: variant must be ER_ATTN_SPLIT1, ER_ATTN_SPLIT2 or ER_ATTN_STREAM");
    if (variant == ER_ATTN_STREAM && head_dim != 96) return fail(ER_ERR_UNSUPPORTED, "the streaming kernel is built for head_dim 96");
    if (head_dim != 96 && head_dim != 64) return fail(ER_ERR_UNSUPPORTED, "head_dim %d", head_dim);
    if (steps != 2 && steps != 4 && steps != 8) return fail(ER_ERR_INVALID, "steps must be 2, 4 or 8 (chunk = 32*steps keys)");
    hipStream_t st = (hipStream_t)stream;
    const int S = attn_num_chunks(l_cap, attn_chunk(steps, kv_half != 0));
    int* len_dev = nullptr;
    float* part = nullptr;
    HIPCHK(hipMalloc(&len_dev, B * sizeof(int)));
    HIPCHK(hipMalloc(&part, (size_t)B * heads * S * (head_dim + 2) * 4));
    HIPCHK(hipMemcpy(len_dev, len_host, B * sizeof(int), hipMemcpyHostToDevice));
    AttnDecArgs a{};
    a.q = q; a.kcache = k; a.vcache = v; a.len_dev = len_dev; a.part = part; a.out = out;
    a.H = heads; a.l_cap = l_cap; a.S = S; a.hidden = heads * head_dim; a.chunk = attn_chunk(steps, kv_half != 0);
    a.kv_bstride = (long long)heads * l_cap * head_dim; a.sqrt_d = sqrtf((float)head_dim);
    hipError_t e;
    if (variant == ER_ATTN_STREAM) {       // the streaming kernel of the batched decode step (no partials)
        e = launch_attn_stream_d<96>(a, kv_half != 0, B, st);
    } else {
        e = launch_attn_partial(a, head_dim, steps, kv_half != 0, B, st, variant == ER_ATTN_SPLIT1 ? 1 : 2);
        if (e == hipSuccess) e = launch_attn_combine(a, head_dim, B, st);
    }
    hipError_t e2 = hipStreamSynchronize(st);
    hipFree(len_dev);
    hipFree(part);
    HIPRET(e);
    HIPRET(e2);
    return ER_OK;
}

extern "C" int er_k_attn_outproj3(const float* q, const void* k, const void* v, int len, const void* wo, const float* bo,
                                  const float* resid, float* y, int l_cap, int kv_half, int w_half, void* stream) {
    // version 3 of the single-row decode attention: balanced chunks (16 heads x 16 chunks) + the merge fused into out_proj
    constexpr int H = 16, D = 96;
    if (len <= 0 || len > l_cap || !attn3_fits(l_cap, H)) return fail(ER_ERR_CAPACITY, "er_k_attn_outproj3: len %d / l_cap %d (<= %d)", len, l_cap, attn3_num_chunks(H) * ATTN3_CAP);
    hipStream_t st = (hipStream_t)stream;
    const int nch = attn3_num_chunks(H);
    float *part = nullptr, *part_ml = nullptr;
    HIPCHK(hipMalloc(&part, (size_t)H * nch * (D + 2) * 4));
    part_ml = part + (size_t)H * nch * D;      // one allocation: nothing to leak on an error path
    AttnDecArgs a{};
    a.q = q; a.kcache = k; a.vcache = v; a.fixed_len = len; a.part = part; a.part_ml = part_ml;
    a.H = H; a.l_cap = l_cap; a.hidden = H * D; a.kv_bstride = (long long)H * l_cap * D; a.sqrt_d = sqrtf((float)D);
    hipError_t e = launch_attn_partial3_d<D>(a, kv_half != 0, nch, 1, st);
    OutMergeArgs m{};
    m.W = wo; m.bias = bo; m.resid = resid; m.out = y; m.part_o = part; m.part_ml = part_ml; m.N = H * D;
    if (e == hipSuccess) e = w_half ? launch_outproj_merge<_Float16, D>(m, nch, st) : launch_outproj_merge<float, D>(m, nch, st);
    hipError_t e2 = hipStreamSynchronize(st);
    hipFree(part);
    HIPRET(e);
    HIPRET(e2);
    return ER_OK;
}

extern "C" int er_k_gemm(const float* a, const float* b, const float* bias, const float* resid, float* cc, int m, int n, int k,
                         int lda, int ldb, int ldc, int b_is_kn, int relu, float div, void* stream) {
    if (k % 16) return fail(ER_ERR_INVALID, "er_k_gemm: k must be a multiple of 16");
    GemmArgs g = gemm_args_default();
    g.A = a; g.B = b; g.C = cc; g.bias = bias; g.resid = resid; g.M = m; g.N = n; g.K = k;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldc; g.b_is_kn = b_is_kn; g.kb_valid = k; g.relu = relu; g.div = div;
    HIPRET(launch_gemm(g, 1, (hipStream_t)stream));
    return ER_OK;
}

extern "C" int er_k_gemm_f16(const float* a, const void* w, const float* bias, const float* resid, float* cc, int m, int n, int k,
                             int lda, int ldb, int ldc, int relu, void* stream) {
    if (k % 32) return fail(ER_ERR_INVALID, "er_k_gemm_f16: k must be a multiple of 32");
    GemmArgs g = gemm_args_default();
    g.A = a; g.B = reinterpret_cast<const float*>(w); g.C = cc; g.bias = bias; g.resid = resid; g.M = m; g.N = n; g.K = k;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldc; g.relu = relu;
    HIPRET(launch_gemm_f16(g, (hipStream_t)stream));
    return ER_OK;
}

// fp16 x fp16 LDS-DMA GEMM (gemm_hh_mfma_kernel): a is converted to an fp16 copy first (in the product the producer writes it),
// c16_out (optional, device fp16 [m][n]) receives the epilogue's fp16 copy of the result
extern "C" int er_k_gemm_hh(const float* a, const void* w, const float* bias, const float* resid, float* cc, void* c16_out, int m,
                            int n, int k, int lda, int ldb, int ldc, int relu, void* stream) {
    if (k % 64 || (ldb & 7)) return fail(ER_ERR_INVALID, "er_k_gemm_hh: k must be a multiple of 64, ldb of 8");
    hipStream_t st = (hipStream_t)stream;
    _Float16* a16 = nullptr;
    HIPCHK(hipMalloc(&a16, (size_t)m * k * sizeof(_Float16)));
    hipLaunchKernelGGL(cvt_rows_f16_kernel, dim3(ew_grid((long long)m * k)), dim3(ER_WG), 0, st, a, a16, (long long)m, k, lda, k);
    GemmArgs g = gemm_args_default();
    g.A = reinterpret_cast<const float*>(a16); g.B = reinterpret_cast<const float*>(w); g.C = cc; g.bias = bias; g.resid = resid;
    g.M = m; g.N = n; g.K = k; g.lda = k; g.ldb = ldb; g.ldc = ldc; g.ldr = ldc; g.relu = relu;
    g.c16 = reinterpret_cast<_Float16*>(c16_out); g.ldc16 = n;
    hipError_t e = launch_gemm_hh(g, st);
    hipError_t e2 = hipStreamSynchronize(st);
    hipFree(a16);
    HIPRET(e);
    HIPRET(e2);
    return ER_OK;
}

extern "C" int er_k_gemm_hh_qkv(const float* a, const void* w, const float* bias, void* qk16_out, void* vt_out, int m, int n, int k,
                                int rows_per_batch, int force_tile, void* stream) {
    if (k % 64 || n % 192 || m % 64 || rows_per_batch <= 0 || rows_per_batch % 64 || m % rows_per_batch)
        return fail(ER_ERR_INVALID, "er_k_gemm_hh_qkv: k, m, rows_per_batch multiples of 64, n of 192");
    hipStream_t st = (hipStream_t)stream;
    _Float16* a16 = nullptr;
    HIPCHK(hipMalloc(&a16, (size_t)m * k * sizeof(_Float16)));
    hipLaunchKernelGGL(cvt_rows_f16_kernel, dim3(ew_grid((long long)m * k)), dim3(ER_WG), 0, st, a, a16, (long long)m, k, k, k);
    GemmArgs g = gemm_args_default();
    g.A = reinterpret_cast<const float*>(a16); g.B = reinterpret_cast<const float*>(w); g.bias = bias;
    g.M = m; g.N = n; g.K = k; g.lda = k; g.ldb = k; g.ldc = n; g.ldr = n;
    g.c16 = reinterpret_cast<_Float16*>(qk16_out); g.ldc16 = n;
    g.vt16 = reinterpret_cast<_Float16*>(vt_out); g.vt_col0 = 2 * (n / 3); g.vt_rows = rows_per_batch; g.vt_ld = rows_per_batch;
    hipError_t e = launch_gemm_hh(g, st, force_tile);
    hipError_t e2 = hipStreamSynchronize(st);
    hipFree(a16);
    HIPRET(e);
    HIPRET(e2);
    return ER_OK;
}

extern "C" int er_k_gemm_hh_geglu(const float* a, const void* w, const float* bias, void* out16, int m, int f, int k, int force_tile,
                                  void* stream) {
    // out16[m][f] = fp16(GEGLU(fp16(a) . w^T + bias)), w = the [2f][k] fp16 weight in the checkpoint's order (value rows, then gate rows):
    // the entry builds the permuted copy the product keeps per layer (geglu_permute_kernel) and runs the fused kernel
    if (!a || !w || !bias || !out16) return fail(ER_ERR_INVALID, "er_k_gemm_hh_geglu: a, w, bias and out16 are required (the permute pass reads the bias)");
    if (k % 64 || f % 64 || m <= 0) return fail(ER_ERR_INVALID, "er_k_gemm_hh_geglu: k and f must be multiples of 64");
    if (force_tile != 0 && force_tile != 1 && force_tile != 2 && force_tile != 4) return fail(ER_ERR_INVALID, "er_k_gemm_hh_geglu: force_tile 0 / 1 / 2 / 4");
    hipStream_t st = (hipStream_t)stream;
    // ONE scratch block, carved: fp16 copy of a | permuted weight | permuted bias (an early return on a failed allocation leaks nothing)
    const size_t na = ((size_t)m * k * sizeof(_Float16) + 255) & ~(size_t)255, nw = ((size_t)2 * f * k * sizeof(_Float16) + 255) & ~(size_t)255;
    char* blk = nullptr;
    HIPCHK(hipMalloc((void**)&blk, na + nw + (size_t)2 * f * sizeof(float)));
    _Float16 *a16 = reinterpret_cast<_Float16*>(blk), *wp = reinterpret_cast<_Float16*>(blk + na);
    float* bp = reinterpret_cast<float*>(blk + na + nw);
    hipLaunchKernelGGL(cvt_rows_f16_kernel, dim3(ew_grid((long long)m * k)), dim3(ER_WG), 0, st, a, a16, (long long)m, k, k, k);
    hipLaunchKernelGGL(geglu_permute_kernel, dim3(2 * f), dim3(ER_WG), 0, st, reinterpret_cast<const _Float16*>(w), bias, wp, bp, f, k);
    GemmArgs g = gemm_args_default();
    g.A = reinterpret_cast<const float*>(a16); g.B = reinterpret_cast<const float*>(wp); g.bias = bp;
    g.M = m; g.N = 2 * f; g.K = k; g.lda = k; g.ldb = k; g.ldc = 2 * f; g.ldr = 2 * f;
    g.c16 = reinterpret_cast<_Float16*>(out16); g.ldc16 = f;
    hipError_t e = launch_gemm_hh_geglu(g, st, force_tile);
    hipError_t e2 = hipStreamSynchronize(st);
    hipFree(blk);
    HIPRET(e);
    HIPRET(e2);
    return ER_OK;
}

extern "C" int er_k_gemm_f16s(const float* a, const void* w, const float* bias, const float* resid, float* cc, int m, int n, int k,
                              int lda, int ldb, int ldc, int relu, void* stream) {
    if (k % 32) return fail(ER_ERR_INVALID, "er_k_gemm_f16s: k must be a multiple of 32");
    hipStream_t st = (hipStream_t)stream;
    GemmArgs g = gemm_args_default();
    g.A = a; g.B = reinterpret_cast<const float*>(w); g.C = cc; g.bias = bias; g.resid = resid; g.M = m; g.N = n; g.K = k;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldc; g.relu = relu;
    // ER_K_GEMM_F16S_FORM = reg / dma pins one of the two forms linear_hs chooses between (unit tests compare them bit for bit)
    const char* form = getenv("ER_K_GEMM_F16S_FORM");
    const bool force_reg = form && form[0] == 'r';
    if (k % XBK == 0 && !(ldb & 7) && !force_reg) {      // the product path of the fast-mode prefill: split pass + LDS-DMA kernel (linear_hs)
        _Float16 *hi = nullptr, *lo = nullptr;
        HIPCHK(hipMalloc(&hi, (size_t)m * k * 2));
        HIPCHK(hipMalloc(&lo, (size_t)m * k * 2));
        hipLaunchKernelGGL(split_rows_f16_kernel, split_rows_grid(m, k), dim3(ER_WG), 0, st, a, hi, lo, (long long)m, k, lda);
        g.A = reinterpret_cast<const float*>(hi); g.a_lo = lo; g.lda = k;
        hipError_t e = launch_gemm_hh_split(g, st);
        hipError_t e2 = hipStreamSynchronize(st);
        hipFree(hi); hipFree(lo);
        HIPRET(e);
        HIPRET(e2);
        return ER_OK;
    }
    HIPRET(launch_gemm_f16s(g, st));      // k % 64 != 0: the register-staged kernel
    return ER_OK;
}

extern "C" int er_k_flash_attn_f16(const float* q, const float* k, const float* v, float* o, int B, int H, int N, int M,
                                   void* stream) {
    // q/k/v/o: [B, rows, H*64] fp32, heads side by side in a row (the layout the projections produce)
    FlashArgs a{};
    a.Q = q; a.K = k; a.V = v; a.O = o; a.N = N; a.M = M;
    a.ldq = a.ldk = a.ldv = a.ldo = H * FA_D;
    a.qs_b = (long long)N * H * FA_D; a.os_b = a.qs_b; a.ks_b = (long long)M * H * FA_D; a.vs_b = a.ks_b;
    a.head_stride = FA_D; a.scale = 1.0f / sqrtf((float)FA_D);
    HIPRET(launch_flash_attn_f16(a, H, B, (hipStream_t)stream));
    return ER_OK;
}

// the LDS-DMA variant (q / k / v in fp16, V transposed per head): converts and transposes the fp32 inputs first - in the DiT
// path the qkv GEMM epilogue and transpose_v_f16_kernel produce these operands - and returns the fp16 output widened to fp32
extern "C" int er_k_flash_attn_hh(const float* q, const float* k, const float* v, float* o, int B, int H, int N, int M, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int C = H * FA_D, Mp = (M + 63) / 64 * 64;
    _Float16 *q16 = nullptr, *k16 = nullptr, *v16 = nullptr, *vt = nullptr, *o16 = nullptr;
    HIPCHK(hipMalloc(&q16, (size_t)B * N * C * 2));
    HIPCHK(hipMalloc(&k16, (size_t)B * M * C * 2));
    HIPCHK(hipMalloc(&v16, (size_t)B * M * C * 2));
    HIPCHK(hipMalloc(&vt, (size_t)B * H * 64 * Mp * 2));
    HIPCHK(hipMalloc(&o16, (size_t)B * N * C * 2));
    hipLaunchKernelGGL(cvt_rows_f16_kernel, dim3(ew_grid((long long)B * N * C)), dim3(ER_WG), 0, st, q, q16, (long long)B * N, C, C, C);
    hipLaunchKernelGGL(cvt_rows_f16_kernel, dim3(ew_grid((long long)B * M * C)), dim3(ER_WG), 0, st, k, k16, (long long)B * M, C, C, C);
    hipLaunchKernelGGL(cvt_rows_f16_kernel, dim3(ew_grid((long long)B * M * C)), dim3(ER_WG), 0, st, v, v16, (long long)B * M, C, C, C);
    hipLaunchKernelGGL(transpose_v_f16_kernel, dim3(Mp / 64, H, B), dim3(ER_WG), 0, st, v16, vt, M, Mp, C, (long long)M * C);
    FlashHArgs a{};
    a.Q = q16; a.K = k16; a.Vt = vt; a.O16 = o16; a.N = N; a.M = M; a.ldq = a.ldk = a.ldo = C; a.ldvt = Mp;
    a.qs_b = a.os_b = (long long)N * C; a.ks_b = (long long)M * C; a.vts_h = 64LL * Mp; a.vts_b = (long long)H * 64 * Mp;
    a.head_stride = FA_D; a.scale = 1.0f / sqrtf((float)FA_D);
    hipError_t e = launch_flash_attn_hh(a, H, B, st);
    hipLaunchKernelGGL(cvt_f16_rows_f32_kernel, dim3(ew_grid((long long)B * N * C)), dim3(ER_WG), 0, st, o16, o, (long long)B * N * C);
    hipError_t e2 = hipStreamSynchronize(st);
    for (void* p : {(void*)q16, (void*)k16, (void*)v16, (void*)vt, (void*)o16}) hipFree(p);
    HIPRET(e);
    HIPRET(e2);
    return ER_OK;
}

extern "C" int er_k_flash_attn_f32(const float* q, const float* k, const float* v, float* o, int B, int H, int N, int M, int D,
                                   int causal, void* stream) {
    // q/o: [B, N, H*D], k/v: [B, M, H*D] fp32, heads side by side in a row; causal: key j visible to query i iff j <= i + (M - N)
    if (D != 64 && D != 96) return fail(ER_ERR_UNSUPPORTED, "er_k_flash_attn_f32: head_dim %d (64, 96)", D);
    if (causal && M < N) return fail(ER_ERR_INVALID, "er_k_flash_attn_f32: causal needs M >= N");
    Flash32Args a{};
    a.Q = q; a.K = k; a.V = v; a.O = o; a.N = N; a.M = M;
    a.ldq = a.ldk = a.ldv = a.ldo = H * D;
    a.qs_b = (long long)N * H * D; a.os_b = a.qs_b; a.ks_b = (long long)M * H * D; a.vs_b = a.ks_b;
    a.qs_h = a.ks_h = a.vs_h = a.os_h = D;
    a.sqrt_d = sqrtf((float)D); a.causal_off = M - N;
    float *po = nullptr, *pml = nullptr;                   // key-range split of the causal prefill shape (k_flash_attn_f32.h, KSP)
    if (flash32_ksplit(N, H, B, D, causal != 0)) {
        HIPCHK(hipMalloc(&po, flash32_part_o_floats(B, H, N, D) * 4));
        HIPCHK(hipMalloc(&pml, flash32_part_ml_floats(B, H, N) * 4));
        a.part_o = po; a.part_ml = pml;
    }
    hipError_t e = launch_flash_attn_f32(a, D, causal != 0, H, B, (hipStream_t)stream);
    if (po) { hipStreamSynchronize((hipStream_t)stream); hipFree(po); hipFree(pml); }
    HIPRET(e);
    return ER_OK;
}

extern "C" int er_k_flash_attn_f16s(const float* q, const float* k, const float* v, float* o, int B, int H, int N, int M, int causal,
                                    void* stream) {
    // STAGED (unmeasured): q/o [B, N, H*96], k/v [B, M, H*96] fp32 holding fp16-representable k / v values
    if (causal && M < N) return fail(ER_ERR_INVALID, "er_k_flash_attn_f16s: causal needs M >= N");
    constexpr int D = 96;
    Flash32Args a{};
    a.Q = q; a.K = k; a.V = v; a.O = o; a.N = N; a.M = M;
    a.ldq = a.ldk = a.ldv = a.ldo = H * D;
    a.qs_b = (long long)N * H * D; a.os_b = a.qs_b; a.ks_b = (long long)M * H * D; a.vs_b = a.ks_b;
    a.qs_h = a.ks_h = a.vs_h = a.os_h = D;
    a.sqrt_d = sqrtf((float)D); a.causal_off = M - N;
    HIPRET(launch_flash_attn_f16s(a, D, causal != 0, H, B, (hipStream_t)stream));
    return ER_OK;
}

extern "C" int er_k_layernorm(const float* x, const float* w, const float* b, float* y, int rows, int cols, float eps, void* stream) {
    HIPRET(launch_layernorm(x, w, b, y, rows, cols, cols, cols, eps, (hipStream_t)stream));
    return ER_OK;
}

extern "C" int er_k_softmax(float* s, int rows, int cols, int ld, int causal, void* stream) {
    HIPRET(launch_softmax_rows(s, rows, cols, (long long)ld, ld, 0LL, 1, causal, 0, (hipStream_t)stream));
    return ER_OK;
}

extern "C" int er_k_sample_head(const float* logits, const er_decode_params* p, int vocab, int eos, int pad, int B, int step,
                                const int32_t* last_tok, const int32_t* counter, const int32_t* unfinished, int32_t* next_tok,
                                int32_t* counter_out, int32_t* unfinished_out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const size_t b = (size_t)B;
    int* sb = nullptr;
    DecodeParamsDev* dpd = nullptr;
    long long* ids = nullptr;
    HIPCHK(hipMalloc(&sb, (7 * b + 8) * sizeof(int)));
    HIPCHK(hipMalloc(&dpd, sizeof(DecodeParamsDev)));
    HIPCHK(hipMalloc(&ids, b * (size_t)(step + 1) * sizeof(long long)));
    std::vector<int> h(7 * b + 8, 0);
    for (size_t i = 0; i < b; ++i) {
        h[i] = last_tok[i]; h[b + i] = 0; h[2 * b + i] = counter[i]; h[3 * b + i] = step;
        h[4 * b + i] = unfinished[i]; h[5 * b + i] = -1; h[6 * b + i] = 0;
    }
    h[7 * b] = B;
    HIPCHK(hipMemcpy(sb, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice));
    GenState s{};
    s.tok = sb; s.pos = sb + b; s.counter = sb + 2 * b; s.ngen = sb + 3 * b; s.unfinished = sb + 4 * b;
    s.eos_step = sb + 5 * b; s.base_pos = sb + 6 * b; s.n_unfinished = sb + 7 * b; s.error = sb + 7 * b + 1;
    DecodeParamsDev dp{};
    dp.mode = p->mode; dp.top_k = p->top_k; dp.grammar = p->grammar; dp.max_new = step + 1; dp.min_new = p->min_new_tokens;
    dp.eos = eos; dp.pad = pad; dp.vocab = vocab;
    dp.seed_lo = (unsigned int)(p->seed & 0xffffffffu); dp.seed_hi = (unsigned int)(p->seed >> 32);
    HIPCHK(hipMemcpy(dpd, &dp, sizeof(dp), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(sample_head_kernel, dim3(B), dim3(ER_WG), sample_head_lds(vocab), st, logits, dpd, s, ids, step + 1);
    hipError_t e = hipGetLastError();
    hipError_t e2 = hipStreamSynchronize(st);
    if (e == hipSuccess && e2 == hipSuccess) {
        hipMemcpy(h.data(), sb, h.size() * sizeof(int), hipMemcpyDeviceToHost);
        for (size_t i = 0; i < b; ++i) { next_tok[i] = h[i]; counter_out[i] = h[2 * b + i]; unfinished_out[i] = h[4 * b + i]; }
    }
    hipFree(sb); hipFree(dpd); hipFree(ids);
    HIPRET(e);
    HIPRET(e2);
    return ER_OK;
}

// ------------------------------------------------------------------------------------ detokenise (host)
extern "C" int er_meto_decode(const int32_t* tokens, int n, int bins, int backend, float* v, int32_t* f, int32_t* t, int32_t* nv,
                              int32_t* nf, int32_t* nt) {
    if (n < 0 || bins <= 0 || (n > 0 && !tokens) || !v || !f || !t || !nv || !nf || !nt)
        return fail(ER_ERR_INVALID, "er_meto_decode: bad argument");
    if (backend != ER_METO_LR_ABSCO && backend != ER_METO_LR) return fail(ER_ERR_UNSUPPORTED, "meto backend %d (LR_ABSCO = 0, LR = 1)", backend);
    const MetoCounts c = backend == ER_METO_LR ? meto_decode_lr(tokens, n, bins, v, f, t) : meto_decode_lr_absco(tokens, n, bins, v, f, t);
    *nv = c.vertices; *nf = c.faces; *nt = c.face_types;
    return ER_OK;
}

extern "C" int er_meto_encode(const float* vertices, int nv, const int32_t* faces, int nf, int bins, int backend, int32_t* tokens,
                              int32_t* n_tokens, int32_t* face_order, int32_t* face_type, int32_t* n_faces_out) {
    if (backend != ER_METO_LR_ABSCO && backend != ER_METO_LR) return fail(ER_ERR_UNSUPPORTED, "meto backend %d (LR_ABSCO = 0, LR = 1)", backend);
    if (nv < 0 || nf < 0 || bins <= 0 || !tokens || !n_tokens || !face_order || !face_type || !n_faces_out ||
        (nv > 0 && !vertices) || (nf > 0 && !faces))
        return fail(ER_ERR_INVALID, "er_meto_encode: bad argument");
    for (int i = 0; i < 3 * nf; ++i)
        if (faces[i] < 0 || faces[i] >= nv) return fail(ER_ERR_INVALID, "er_meto_encode: face index %d out of range", faces[i]);
    const MetoEncodeOut o = backend == ER_METO_LR ? meto_encode<true>(vertices, nv, faces, nf, bins) : meto_encode<false>(vertices, nv, faces, nf, bins);
    const long long per = backend == ER_METO_LR ? 2 : 1;      // LR may open a sub-mesh on an already covered face (see meto_encode.h)
    if ((long long)o.tokens.size() > 10LL * per * nf || (long long)o.face_order.size() > per * nf)
        return fail(ER_ERR_CAPACITY, "er_meto_encode: token bound exceeded");
    memcpy(tokens, o.tokens.data(), o.tokens.size() * sizeof(int32_t));
    memcpy(face_order, o.face_order.data(), o.face_order.size() * sizeof(int32_t));
    memcpy(face_type, o.face_type.data(), o.face_type.size() * sizeof(int32_t));
    *n_tokens = (int32_t)o.tokens.size();
    *n_faces_out = (int32_t)o.face_order.size();
    return ER_OK;
}

// ------------------------------------------------------------------------------------ DiT front-end (f3)
#include "er_dit.h"
