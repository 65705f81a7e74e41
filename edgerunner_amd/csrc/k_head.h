// Device-side sampling head + grammar automaton: what HuggingFace's _sample loop does
// on the host between two forward passes of the reference
// (call site core/models.py:286-303; processors core/utils.py:118-141 and
// core/models.py:236-275; transformers 4.46.2 generation/utils.py::_sample):
//
//   s = logits[:, -1, :].float()
//   [MinNewTokensLength]  s[eos] = -inf while t < min_new_tokens
//   [PrefixConstrained]   s += mask(-inf / 0) from the grammar (integer state machine)
//   greedy: argmax(s) | sample: TopK(10) -> softmax -> one categorical draw
//   next = next*unfinished + pad*(1-unfinished); unfinished &= next != eos
//
// Keeping this on the device removes the per-token host syncs of the reference
// (`0 in attention_mask`, reading input_ids[-1], the EOS check).  One workgroup per
// batch row; all state (token, position, grammar counter, finished flag, output ids)
// lives in HBM so the step can be replayed from a hipGraph.
#pragma once
#include "er_common.h"

namespace er {

struct DecodeParamsDev {       // device copy of er_decode_params + token ids
    int mode, top_k, grammar, max_new, min_new;
    int eos, pad, vocab;
    unsigned int seed_lo, seed_hi;
};

struct GenState {              // per-row arrays, all device
    int* tok;                  // last generated token (input of the next forward)
    int* pos;                  // position the next forward writes its K/V to
    int* counter;              // LR_ABSCO coordinate counter
    int* ngen;                 // tokens generated so far
    int* unfinished;           // 1 while the row has not emitted EOS
    int* eos_step;             // step index at which EOS was emitted (-1)
    int* base_pos;             // prefill length (position of the first generated token's forward)
    int* n_unfinished;         // scalar: rows still running
    int* error;                // scalar: set when a row had no finite candidate score (non-finite logits)
    const unsigned int* row_stream;   // per row: second Philox counter word of the sampler (null: the row index).  A caller that
                               // shards / batches independent jobs sets it to the GLOBAL job index, so a job's draws do not
                               // depend on which rows share its batch (er_set_row_streams)
};

// Philox4x32-10 (Salmon et al. 2011): counter-based, so a draw is a pure function of
// (seed, step, row) and a replayed graph needs no RNG state.
__device__ __host__ inline void philox4x32_10(unsigned int c0, unsigned int c1, unsigned int c2, unsigned int c3,
                                              unsigned int k0, unsigned int k1, unsigned int out[4]) {
    const unsigned int M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)M0 * c0;
        const unsigned long long p1 = (unsigned long long)M1 * c2;
        const unsigned int hi0 = (unsigned int)(p0 >> 32), lo0 = (unsigned int)p0;
        const unsigned int hi1 = (unsigned int)(p1 >> 32), lo1 = (unsigned int)p1;
        const unsigned int n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Grammar step (integer state machine).  Updates the counter from the last generated
// token and reports the allowed set as {lo, hi, extra EOS / control set}.
//   LR_ABSCO (core/models.py:246-268): t==0 -> {5}; last==5 -> c=9; last in {3,4} -> c=3;
//   last>=6 -> c-=1; c>0 -> [6,V) else {3,4,5,eos}.
//   NAIVE9 (core/models.py:237-242): [3,V) plus eos iff t % 9 == 1.
__device__ __forceinline__ bool grammar_allowed(int grammar, int t, int counter, int i, int V, int eos) {
    if (grammar == 2) {
        if (t == 0) return i == 5;
        if (counter > 0) return i >= 6;
        return i == 3 || i == 4 || i == 5 || i == eos;
    }
    if (grammar == 1) return (i >= 3) || (i == eos && (t % 9) == 1);
    return true;
}

__device__ __forceinline__ int grammar_update(int grammar, int t, int counter, int last) {
    if (grammar != 2 || t == 0) return counter;
    if (last == 5) return 9;
    if (last == 3 || last == 4) return 3;
    if (last >= 6) return counter - 1;
    return counter;
}

// Largest vocabulary the sampling head handles: the reference's are 515 (no meto), 518 (LR_ABSCO) and 1030 (LR, 2*512 + 6;
// core/models.py:78-84).
constexpr int ER_HEAD_MAX_VOCAB = 1088;

// grid (B), 256 threads.  Dynamic LDS: vocab floats + 2*vocab ints-ish scratch (see launch).
__global__ __launch_bounds__(ER_WG) void sample_head_kernel(const float* logits, const DecodeParamsDev* pp, GenState st,
                                                            long long* out_ids, int out_ld) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float red[8];
    __shared__ int redi[8];
    __shared__ int chosen_s;
    const DecodeParamsDev P = *pp;
    const int V = P.vocab;
    float* sc = smem;                              // [V] masked scores
    float* cand_e = smem + V;                      // [V] candidate weights (compacted)
    int* cand_i = reinterpret_cast<int*>(smem + 2 * V);   // [V] candidate ids (compacted)
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int t = st.ngen[b];
    if (t >= P.max_new) return;                    // replay past the end: no-op
    const int last = st.tok[b];
    const bool running = st.unfinished[b] != 0;
    int counter = st.counter[b];
    if (running) counter = grammar_update(P.grammar, t, counter, last);

    // masked scores
    const float* lg = logits + (long long)b * V;
    for (int i = tid; i < V; i += ER_WG) {
        bool ok = grammar_allowed(P.grammar, t, counter, i, V, P.eos);
        if (i == P.eos && t < P.min_new) ok = false;
        sc[i] = ok ? lg[i] : -INFINITY;
    }
    __syncthreads();

    // argmax with first-index tie-break (torch.argmax): used by greedy and as softmax max
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < V; i += ER_WG) {
        const float v = sc[i];
        if (v > bv) { bv = v; bi = i; }
    }
    auto take = [&](float ov, int oi) { if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; } };
    take(lane_xor_any<32>(bv, lane), (int)lane_xor_bits<32>((unsigned)bi, lane));
    take(lane_xor_any<16>(bv, lane), (int)lane_xor_bits<16>((unsigned)bi, lane));
    take(lane_xor_any<8>(bv, lane), (int)lane_xor_bits<8>((unsigned)bi, lane));
    take(lane_xor_any<4>(bv, lane), (int)lane_xor_bits<4>((unsigned)bi, lane));
    take(lane_xor_any<2>(bv, lane), (int)lane_xor_bits<2>((unsigned)bi, lane));
    take(lane_xor_any<1>(bv, lane), (int)lane_xor_bits<1>((unsigned)bi, lane));
    if (lane == 0) { red[wid] = bv; redi[wid] = bi; }
    __syncthreads();
    float gmax = red[0];
    int gidx = redi[0];
#pragma unroll
    for (int w = 1; w < ER_NWAVES; ++w)
        if (red[w] > gmax || (red[w] == gmax && redi[w] < gidx)) { gmax = red[w]; gidx = redi[w]; }
    int chosen = gidx;
    // every masked score is NaN or -inf (fp16 overflow upstream, corrupt weights): HF would raise inside
    // torch.multinomial; here the row is closed with EOS and the error flag makes er_decode fail loudly
    const bool no_candidate = gidx < 0 || gidx >= V || !(gmax > -INFINITY);
    if (no_candidate) chosen = P.eos;

    if (P.mode == 1 && !no_candidate) {   // sample: TopK(top_k) -> softmax -> categorical
        // k-th largest value counting duplicates: remove one maximum k-1 times (wave 0, scores in registers)
        if (wid == 0) {
            constexpr int MAXPL = ER_HEAD_MAX_VOCAB / 64;   // scores of the whole vocabulary in wave 0's registers
            float v[MAXPL];
#pragma unroll
            for (int j = 0; j < MAXPL; ++j) { const int i = j * 64 + lane; v[j] = (i < V) ? sc[i] : -INFINITY; }
            const int k = min(P.top_k, V);
            float kth = gmax;
            for (int it = 0; it < k; ++it) {
                float mv = -INFINITY; int mj = 0;
#pragma unroll
                for (int j = 0; j < MAXPL; ++j) if (v[j] > mv) { mv = v[j]; mj = j; }
                const float wmax = wave_max(mv);
                kth = wmax;
                if (wmax == -INFINITY) break;      // fewer than k finite scores: threshold is -inf
                const unsigned long long holders = __ballot(mv == wmax);
                const int first = __ffsll((long long)holders) - 1;
                if (lane == first) {
#pragma unroll
                    for (int j = 0; j < MAXPL; ++j) if (j == mj) v[j] = -INFINITY;
                }
            }
            // candidates: s >= kth and finite; compact in ascending token id
            int base = 0;
            for (int j = 0; j * 64 < V; ++j) {
                const int i = j * 64 + lane;
                const float s = (i < V) ? sc[i] : -INFINITY;
                const bool c = (s >= kth) && (s > -INFINITY);
                const unsigned long long mask = __ballot(c);
                if (c) {
                    const int slot = base + __popcll(mask & ((1ull << lane) - 1ull));
                    cand_i[slot] = i;
                    cand_e[slot] = expf(s - gmax);
                }
                base += __popcll(mask);
            }
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
            if (lane == 0) {
                float total = 0.f;
                for (int c = 0; c < base; ++c) total += cand_e[c];
                unsigned int r[4];
                philox4x32_10((unsigned int)t, st.row_stream ? st.row_stream[b] : (unsigned int)b, 0u, 0u, P.seed_lo, P.seed_hi, r);
                const float u = (float)(r[0] >> 8) * (1.0f / 16777216.0f);
                const float target = u * total;
                float acc = 0.f;
                int pick = base > 0 ? cand_i[base - 1] : P.eos;
                for (int c = 0; c < base; ++c) {
                    acc += cand_e[c];
                    if (acc > target) { pick = cand_i[c]; break; }
                }
                chosen_s = pick;
            }
        }
        __syncthreads();
        chosen = chosen_s;
    }

    if (tid == 0) {
        if (no_candidate && running) *st.error = 1;
        int next = running ? chosen : P.pad;
        out_ids[(long long)b * out_ld + t] = (long long)next;
        st.tok[b] = next;
        st.counter[b] = counter;
        st.pos[b] = st.base_pos[b] + t;
        st.ngen[b] = t + 1;
        if (running && next == P.eos) {
            st.unfinished[b] = 0;
            st.eos_step[b] = t;
            atomicSub(st.n_unfinished, 1);
        }
    }
}

// er_feed: teacher-forced token (host-chosen ids copied to ids_dev beforehand)
__global__ void force_token_kernel(const int* ids_dev, GenState st, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int t = st.ngen[b];
    st.tok[b] = ids_dev[b];
    st.pos[b] = st.base_pos[b] + t;
    st.ngen[b] = t + 1;
}

__global__ void init_state_kernel(GenState st, int B, int base_pos) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0) { *st.n_unfinished = B; *st.error = 0; }
    if (b >= B) return;
    st.tok[b] = 0; st.pos[b] = base_pos; st.counter[b] = 0; st.ngen[b] = 0;
    st.unfinished[b] = 1; st.eos_step[b] = -1; st.base_pos[b] = base_pos;
}

inline size_t sample_head_lds(int vocab) { return (size_t)vocab * 3 * sizeof(float); }

}  // namespace er
