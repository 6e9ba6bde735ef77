// ccr_encoder.hip -- the non-GEMM pieces of a BERT encoder layer as gfx950 kernels (encoder-side fusion).
// The reference encodes with transformers' BertModel under autocast (src/ccrec/models/item_tower.py:122, scripts/al_0_rank.py:92-101);
// measured on one MI355X (profiles/r03_encode_kernel_stats.csv) its projections run at hipBLASLt speed (38 % of the GPU time) and
// the rest is attention at 57 TFLOP/s (22 %) plus separate residual-add / LayerNorm / cast passes (30 %).  Two kernels replace those:
//
//   attention_kernel      softmax(Q K^T * scale + key mask) V for every (sequence, head) of a batch of right-padded or packed
//                         sequences, head width 64, straight from the fused QKV projection's output.  VALU (exp) bound.
//   add_layernorm_kernel  LayerNorm(x_bf16 + residual_f32) -> the fp32 residual stream AND its bf16 copy (the next
//                         projection's operand) in one pass: 12 bytes per element instead of 28 over three kernels.  HBM bound.
//
// Arithmetic = what autocast(bf16) does in the reference's layer: bf16 operands, fp32 scores / softmax / accumulation,
// probabilities rounded to bf16 for the P V product, fp32 residual sum and LayerNorm.
#include <stdlib.h>

#include "ccr_common.h"
#include "ccr_index.h"

namespace ccr {

typedef __bf16 ebf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 ef16x8 __attribute__((ext_vector_type(8)));
typedef float ef32x16 __attribute__((ext_vector_type(16)));

// The layer's 16-bit operand type: bf16 or fp16 -- whichever the caller's autocast context names (the reference's
// torch.cuda.amp.autocast() at scripts/al_0_rank.py:125 is fp16).  Same kernels, same MFMA rate (v_mfma_f32_32x32x16_f16 /
// _bf16), fp32 scores / softmax / residual stream / LayerNorm either way; only the rounding of the 16-bit operands differs.
template <int DT>
struct Half16;
template <>
struct Half16<CCR_DTYPE_BF16> {
    typedef __bf16 elem;
    typedef ebf16x8 vec8;
    static __device__ __forceinline__ ef32x16 mfma(vec8 a, vec8 b, ef32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ float lo(uint32_t w) { return __uint_as_float(w << 16); }
    static __device__ __forceinline__ float hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
};
template <>
struct Half16<CCR_DTYPE_F16> {
    typedef _Float16 elem;
    typedef ef16x8 vec8;
    static __device__ __forceinline__ ef32x16 mfma(vec8 a, vec8 b, ef32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ float lo(uint32_t w) {
        union {
            uint32_t u;
            _Float16 h[2];
        } x;
        x.u = w;
        return (float)x.h[0];
    }
    static __device__ __forceinline__ float hi(uint32_t w) {
        union {
            uint32_t u;
            _Float16 h[2];
        } x;
        x.u = w;
        return (float)x.h[1];
    }
};
// four fp32 values -> four 16-bit values (round to nearest even; a NaN stays a NaN), 8 bytes
template <class E>
__device__ __forceinline__ uint2 round4(float a, float b, float c, float d) {
    union {
        E h[4];
        uint2 u;
    } w;
    w.h[0] = (E)a;
    w.h[1] = (E)b;
    w.h[2] = (E)c;
    w.h[3] = (E)d;
    return w.u;
}

constexpr int ATT_MAX_THREADS = 512;   // up to 8 waves: one workgroup per (sequence, head) stages K / V once, wave w takes query blocks w, w + waves, ..
constexpr int ATT_QW = 32;             // query rows per wave step (the N side of one 32x32 MFMA tile)
constexpr int ATT_KMAX = 8, ATT_VMAX = 4;   // 16-byte key pieces / value-row pairs a thread stages (all in flight at once)
constexpr int ATT_KB = 64;         // keys per loop step (two 32-key score tiles: halves the online-softmax rescaling)
constexpr int ATT_HEAD = 64;       // head width
constexpr int ATT_KROW = 144;      // bytes per key row in LDS: 128 + 16, so the 16 lanes of a ds_read_b128 group hit distinct banks

// LDS: K [lk_pad rows][144 B] row-major | V.  TR = true (r6, the default): V stays ROW-MAJOR, [lk_pad keys][128 B], staged with
// ds_write_b128 as it arrives, the 64-byte halves of a row swapped on rows whose index has bit 1 set; the V^T fragments of the second
// MFMA come from ds_read_b64_tr_b16 (guide T10: a 16-lane group reads a 4-key x 16-column block and receives it column-major), whose
// 32-lane half then touches 4 rows x 64 B on 64 distinct banks.  TR = false (r3-r5, CCR_ATT_TR=0): V TRANSPOSED [64 d][2 * lk_pad + 8 B],
// written two keys per dword (23 % of the LDS-active cycles were bank-conflict cycles of those stores, profiles/r05_attention_pmc.txt).
// lk_pad = longest sequence rounded up to 64.
__host__ __device__ inline size_t attention_lds_bytes(int lk_pad) {
    return (size_t)lk_pad * ATT_KROW + (size_t)ATT_HEAD * (2 * (size_t)lk_pad + 8);   // (the row-major image needs 512 bytes less)
}
typedef short es16x4 __attribute__((ext_vector_type(4)));

// S^T = K Q^T on v_mfma_f32_32x32x16_bf16: A = 32 keys (lane & 31) x 8 head columns (8 * (lane >> 5) + j), B = 32 queries
// likewise -- both operands are 16 contiguous bytes of a row, no transposition.  C layout: lane -> query (lane & 31),
// register e -> key (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5): a lane owns ONE query and 16 of the tile's 32 keys, so the
// softmax row reductions are in-lane plus one exchange with lane ^ 32, and the probabilities, rounded to bf16, ARE the B
// operand of O^T = V^T P^T (contraction index = key; the A operand V^T is read from the transposed LDS image with the
// same key permutation: element j of lane half g <-> key 16 s + 4 g + (j & 3) + 8 (j >> 2)).
template <int DT, bool TR>
__global__ __launch_bounds__(ATT_MAX_THREADS) void attention_kernel(const uint16_t *__restrict__ qkv,
                                                                  const int32_t *__restrict__ seq_start,
                                                                  const int32_t *__restrict__ seq_len,
                                                                  uint16_t *__restrict__ out, int H, int pad_len, int max_len,
                                                                  int lk_pad, float scale_log2e) {
    typedef Half16<DT> HT;
    typedef typename HT::vec8 vec8;
    typedef typename HT::elem elem;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int nthreads = blockDim.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = nthreads >> 6;
    const int wq = (wv + blockIdx.x) % nwaves;   // the wave's first query block rotates with the head: second rounds spread over the SIMDs
    const int b = blockIdx.y, h = blockIdx.x;
    int len = seq_len[b];
    if (len > max_len) len = max_len;   // the LDS image holds max_len keys: a longer entry is cut (the caller's contract: seq_len <= max_len)
    if (len < 0) len = 0;
    const int rows = len > pad_len ? len : pad_len;   // rows the sequence occupies in the token arrays (padding rows get zeros)
    const int HD = H * ATT_HEAD;
    const int64_t stride = 3 * (int64_t)HD;
    const int64_t row0 = seq_start[b];
    const uint16_t *Qg = qkv + row0 * stride + h * ATT_HEAD;
    const uint16_t *Kg = Qg + HD;
    const uint16_t *Vg = Qg + 2 * HD;
    uint16_t *Og = out + row0 * HD + h * ATT_HEAD;
    const int ql = lane & 31, g = lane >> 5;
    if (len == 0) {   // an empty sequence (workgroup-uniform): its padding rows get zeros, nothing is read
        for (int q = tid; q < rows; q += nthreads) {
            uint4 *dst = reinterpret_cast<uint4 *>(Og + (int64_t)q * HD);
#pragma unroll
            for (int i = 0; i < 8; ++i) dst[i] = make_uint4(0u, 0u, 0u, 0u);
        }
        return;
    }

    char *Ks = smem;
    char *Vt = smem + (size_t)lk_pad * ATT_KROW;
    const int VS = 2 * lk_pad + 8;
    const int nkb = (len + ATT_KB - 1) / ATT_KB;
    const int nk = nkb * ATT_KB;

    // ---- stage K (row-major) and V (transposed, two keys per dword) of this (sequence, head); rows beyond len are zeros.
    // Every load of the staging -- and the wave's first query fragment -- is issued before the first LDS store: one memory
    // latency per workgroup instead of one per piece.  The launcher sizes the workgroup so that a thread has at most
    // ATT_KMAX key pieces and ATT_VMAX value pairs.
    uint4 kreg[ATT_KMAX], va[ATT_VMAX], vb[ATT_VMAX];
#pragma unroll
    for (int u = 0; u < ATT_KMAX; ++u) {
        const int i = tid + u * nthreads;
        const int r = i >> 3, c = i & 7;
        kreg[u] = make_uint4(0u, 0u, 0u, 0u);
        if (r < len) kreg[u] = *reinterpret_cast<const uint4 *>(Kg + (int64_t)r * stride + c * 8);
    }
#pragma unroll
    for (int u = 0; u < ATT_VMAX; ++u) {
        const int i = tid + u * nthreads;
        const int p = i >> 3, c = i & 7;
        va[u] = vb[u] = make_uint4(0u, 0u, 0u, 0u);
        if (2 * p < len) va[u] = *reinterpret_cast<const uint4 *>(Vg + (int64_t)(2 * p) * stride + c * 8);
        if (2 * p + 1 < len) vb[u] = *reinterpret_cast<const uint4 *>(Vg + (int64_t)(2 * p + 1) * stride + c * 8);
    }
    vec8 qf[4];
    {
        const int q = wq * ATT_QW + ql;
        const int qr = q < len ? q : len - 1;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const vec8 *>(Qg + (int64_t)qr * stride + 16 * s + 8 * g);
    }
#pragma unroll
    for (int u = 0; u < ATT_KMAX; ++u) {
        const int i = tid + u * nthreads;
        if (i < nk * 8) *reinterpret_cast<uint4 *>(Ks + (i >> 3) * ATT_KROW + (i & 7) * 16) = kreg[u];
    }
#pragma unroll
    for (int u = 0; u < ATT_VMAX; ++u) {
        const int i = tid + u * nthreads;
        if (i < nk * 4) {
            const int p = i >> 3, c = i & 7;
            if constexpr (TR) {   // rows 2p and 2p + 1 as they came: one ds_write_b128 each; 64-byte halves swapped where bit 1 of the row is set
                char *dst = Vt + (size_t)(2 * p) * 128 + ((c ^ ((p & 1) << 2)) << 4);
                *reinterpret_cast<uint4 *>(dst) = va[u];
                *reinterpret_cast<uint4 *>(dst + 128) = vb[u];
            } else {
                const uint32_t a[4] = {va[u].x, va[u].y, va[u].z, va[u].w}, bb[4] = {vb[u].x, vb[u].y, vb[u].z, vb[u].w};
                char *dst = Vt + (size_t)(8 * c) * VS + 4 * p;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    *reinterpret_cast<uint32_t *>(dst + (size_t)(2 * j) * VS) = (a[j] & 0xffffu) | (bb[j] << 16);
                    *reinterpret_cast<uint32_t *>(dst + (size_t)(2 * j + 1) * VS) = (a[j] >> 16) | (bb[j] & 0xffff0000u);
                }
            }
        }
    }
    __syncthreads();
    // TR: this lane's part of every transposed read.  A 16-lane group fetches a block of 4 keys x 16 head columns: lane 4 qq + pp of the
    // group supplies the address of key row (base + qq), columns 4 pp .. 4 pp + 3 (8 bytes) and receives column (lane & 15), keys base .. + 3.
    // Groups 0 / 1 of a 32-lane half take columns 0-15 / 16-31 (the lane's own column d = lane & 31); key base = ... + 4 g (multiple of 4),
    // so bit 1 of the row index is bit 1 of qq: the half swap is a per-lane constant.
    const int tr_qq = (lane & 15) >> 2, tr_pp = lane & 3;
    const int tr_off = (4 * g + tr_qq) * 128 + ((32 * ((lane >> 4) & 1) + 8 * tr_pp) ^ (((tr_qq >> 1) & 1) << 6));
    const int tr_other = (tr_off ^ 64) - tr_off;   // +64 or -64

    for (int q0w = wq * ATT_QW; q0w < rows; q0w += nwaves * ATT_QW) {   // wave-uniform; no barrier below
        const int q = q0w + ql;
        if (q0w >= len) {   // a block of padding rows only: defined output (zeros), no NaNs into the next projection
            if (q < rows) {
                uint2 *dst = reinterpret_cast<uint2 *>(Og + (int64_t)q * HD);
#pragma unroll
                for (int i = 0; i < 8; ++i) dst[i * 2 + g] = make_uint2(0u, 0u);
            }
            continue;
        }
        if (q0w != wq * ATT_QW) {   // the wave's next query block (sequences beyond 32 x waves tokens)
            const int qr = q < len ? q : len - 1;
#pragma unroll
            for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const vec8 *>(Qg + (int64_t)qr * stride + 16 * s + 8 * g);
        }

        ef32x16 o0, o1;
#pragma unroll
        for (int e = 0; e < 16; ++e) o0[e] = o1[e] = 0.f;
        float m = -INFINITY, lsum = 0.f;

        for (int kb = 0; kb < nkb; ++kb) {
            ef32x16 s0, s1;
#pragma unroll
            for (int e = 0; e < 16; ++e) s0[e] = s1[e] = 0.f;
            const char *kp = Ks + (kb * ATT_KB + ql) * ATT_KROW + g * 16;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const vec8 k0 = *reinterpret_cast<const vec8 *>(kp + 32 * s);
                const vec8 k1 = *reinterpret_cast<const vec8 *>(kp + 32 * ATT_KROW + 32 * s);
                s0 = HT::mfma(k0, qf[s], s0);
                s1 = HT::mfma(k1, qf[s], s1);
            }
            float x[32];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                x[e] = s0[e];
                x[16 + e] = s1[e];
            }
            if (kb * ATT_KB + ATT_KB > len) {   // the last step: keys beyond the sequence
#pragma unroll
                for (int e = 0; e < 32; ++e) {
                    const int key = kb * ATT_KB + (e >> 4) * 32 + (e & 3) + 8 * ((e & 15) >> 2) + 4 * g;
                    if (key >= len) x[e] = -INFINITY;
                }
            }
            float mx = x[0];
#pragma unroll
            for (int e = 1; e < 32; ++e) mx = fmaxf(mx, x[e]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mn = fmaxf(m, mx);            // finite: every step holds at least one key < len
            const float alpha = __builtin_amdgcn_exp2f((m - mn) * scale_log2e);   // m = -inf on the first step: 0
            const float bias = -mn * scale_log2e;
            m = mn;
            float ps = 0.f;
#pragma unroll
            for (int e = 0; e < 32; ++e) {
                x[e] = __builtin_amdgcn_exp2f(fmaf(x[e], scale_log2e, bias));
                ps += x[e];
            }
            lsum = fmaf(lsum, alpha, ps);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                o0[e] *= alpha;
                o1[e] *= alpha;
            }
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    vec8 pf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[j] = (elem)x[hb * 16 + s2 * 8 + j];
                    if constexpr (TR) {
                        // keys kbase .. + 3 and kbase + 8 .. + 11 (kbase = kb * 64 + hb * 32 + s2 * 16 + 4 g) of this lane's column d (a0) and d + 32 (a1)
                        typedef __attribute__((address_space(3))) es16x4 *lds_tr_ptr;
                        const char *vp = Vt + (size_t)(kb * ATT_KB + hb * 32 + s2 * 16) * 128 + tr_off;
                        union {
                            es16x4 h[2];
                            vec8 v;
                        } a0, a1;
                        a0.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(vp));
                        a0.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(vp + 8 * 128));
                        const char *vq = vp + tr_other;   // columns d + 32: the other 64-byte half of the same rows
                        a1.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(vq));
                        a1.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(vq + 8 * 128));
                        o0 = HT::mfma(a0.v, pf, o0);
                        o1 = HT::mfma(a1.v, pf, o1);
                    } else {
                        const int kbase = kb * ATT_KB + hb * 32 + s2 * 16 + 4 * g;
                        const char *vp = Vt + (size_t)ql * VS + 2 * kbase;
                        union {
                            uint2 u[2];
                            vec8 v;
                        } a0, a1;
                        a0.u[0] = *reinterpret_cast<const uint2 *>(vp);
                        a0.u[1] = *reinterpret_cast<const uint2 *>(vp + 16);
                        a1.u[0] = *reinterpret_cast<const uint2 *>(vp + (size_t)32 * VS);
                        a1.u[1] = *reinterpret_cast<const uint2 *>(vp + (size_t)32 * VS + 16);
                        o0 = HT::mfma(a0.v, pf, o0);
                        o1 = HT::mfma(a1.v, pf, o1);
                    }
                }
            }
        }

        const float inv = 1.f / (lsum + __shfl_xor(lsum, 32));
        if (q < len) {
            // O^T tile: lane -> query, register e -> head column 32 db + (e & 3) + 8 (e >> 2) + 4 g: four consecutive columns per 8-byte store
            uint16_t *dst = Og + (int64_t)q * HD + 4 * g;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                *reinterpret_cast<uint2 *>(dst + 8 * c4) =
                    round4<elem>(o0[4 * c4] * inv, o0[4 * c4 + 1] * inv, o0[4 * c4 + 2] * inv, o0[4 * c4 + 3] * inv);
                *reinterpret_cast<uint2 *>(dst + 32 + 8 * c4) =
                    round4<elem>(o1[4 * c4] * inv, o1[4 * c4 + 1] * inv, o1[4 * c4 + 2] * inv, o1[4 * c4 + 3] * inv);
            }
        } else if (q < rows) {
            uint2 *dst = reinterpret_cast<uint2 *>(Og + (int64_t)q * HD);
#pragma unroll
            for (int i = 0; i < 8; ++i) dst[i * 2 + g] = make_uint2(0u, 0u);
        }
    }   // query blocks of this wave
}

// One wave per row of dim = 256 * C elements; lane owns elements 4 * (64 c + lane) .. + 3 of every 256-element slice.
template <int C, int DT>
__global__ __launch_bounds__(256) void add_layernorm_kernel(const uint16_t *__restrict__ x, const float *__restrict__ res,
                                                           const float *__restrict__ gamma, const float *__restrict__ beta,
                                                           float eps, float *__restrict__ out_f32,
                                                           uint16_t *__restrict__ out_bf16, int64_t rows) {
    constexpr int DIM = 256 * C;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[C][4];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int col = 4 * (64 * c + lane);
        const uint2 xb = *reinterpret_cast<const uint2 *>(x + row * DIM + col);
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (res) r = *reinterpret_cast<const float4 *>(res + row * DIM + col);
        v[c][0] = Half16<DT>::lo(xb.x) + r.x;
        v[c][1] = Half16<DT>::hi(xb.x) + r.y;
        v[c][2] = Half16<DT>::lo(xb.y) + r.z;
        v[c][3] = Half16<DT>::hi(xb.y) + r.w;
        sum += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum * (1.f / DIM);
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = v[c][j] - mean;
            sq = fmaf(d, d, sq);
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = rsqrtf(sq * (1.f / DIM) + eps);
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int col = 4 * (64 * c + lane);
        const float4 gm = *reinterpret_cast<const float4 *>(gamma + col);
        const float4 bt = *reinterpret_cast<const float4 *>(beta + col);
        float4 y;
        y.x = fmaf((v[c][0] - mean) * rstd, gm.x, bt.x);
        y.y = fmaf((v[c][1] - mean) * rstd, gm.y, bt.y);
        y.z = fmaf((v[c][2] - mean) * rstd, gm.z, bt.z);
        y.w = fmaf((v[c][3] - mean) * rstd, gm.w, bt.w);
        if (out_f32) *reinterpret_cast<float4 *>(out_f32 + row * DIM + col) = y;
        if (out_bf16) *reinterpret_cast<uint2 *>(out_bf16 + row * DIM + col) = round4<typename Half16<DT>::elem>(y.x, y.y, y.z, y.w);
    }
}

// The embedding block of the encoder (transformers BertEmbeddings.forward): (word[id] + type[t]) + position[p] in that order, fp32,
// then LayerNorm -> the fp32 residual stream and its bf16 copy.  One wave per token; ids outside a table are clamped (memory safety).
template <int C, int DT>
__global__ __launch_bounds__(256) void embed_layernorm_kernel(const float *__restrict__ word, int64_t vocab,
                                                             const float *__restrict__ pos_tab, int64_t n_pos,
                                                             const float *__restrict__ type_tab, int64_t n_types,
                                                             const int64_t *__restrict__ ids, const int64_t *__restrict__ pos,
                                                             const int64_t *__restrict__ types, const float *__restrict__ gamma,
                                                             const float *__restrict__ beta, float eps, float *__restrict__ out_f32,
                                                             uint16_t *__restrict__ out_bf16, int64_t rows) {
    constexpr int DIM = 256 * C;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    auto clamp = [](int64_t v, int64_t n) { return v < 0 ? (int64_t)0 : (v >= n ? n - 1 : v); };
    const float *w = word + clamp(ids[row], vocab) * DIM;
    const float *p = pos_tab + clamp(pos[row], n_pos) * DIM;
    const float *t = type_tab + clamp(types ? types[row] : 0, n_types) * DIM;
    float v[C][4];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int col = 4 * (64 * c + lane);
        const float4 a = *reinterpret_cast<const float4 *>(w + col);
        const float4 b = *reinterpret_cast<const float4 *>(t + col);
        const float4 d = *reinterpret_cast<const float4 *>(p + col);
        v[c][0] = (a.x + b.x) + d.x;
        v[c][1] = (a.y + b.y) + d.y;
        v[c][2] = (a.z + b.z) + d.z;
        v[c][3] = (a.w + b.w) + d.w;
        sum += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum * (1.f / DIM);
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = v[c][j] - mean;
            sq = fmaf(d, d, sq);
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = rsqrtf(sq * (1.f / DIM) + eps);
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int col = 4 * (64 * c + lane);
        const float4 gm = *reinterpret_cast<const float4 *>(gamma + col);
        const float4 bt = *reinterpret_cast<const float4 *>(beta + col);
        float4 y;
        y.x = fmaf((v[c][0] - mean) * rstd, gm.x, bt.x);
        y.y = fmaf((v[c][1] - mean) * rstd, gm.y, bt.y);
        y.z = fmaf((v[c][2] - mean) * rstd, gm.z, bt.z);
        y.w = fmaf((v[c][3] - mean) * rstd, gm.w, bt.w);
        if (out_f32) *reinterpret_cast<float4 *>(out_f32 + row * DIM + col) = y;
        if (out_bf16) *reinterpret_cast<uint2 *>(out_bf16 + row * DIM + col) = round4<typename Half16<DT>::elem>(y.x, y.y, y.z, y.w);
    }
}

// Exact (erf) GELU of a bf16 array, in place or out of place: 16 bytes per lane per access, grid-stride.  The same formula and fp32
// arithmetic as torch's GELU kernel (0.5 x (1 + erf(x / sqrt 2)), rounded to bf16 once): a pass of its own between the two FFN
// projections because the library's GEMM epilogue only offers the tanh form.
template <int DT>
__global__ __launch_bounds__(256) void gelu_kernel(const uint4 *__restrict__ x, uint4 *__restrict__ y, int64_t n16) {
    typedef Half16<DT> HT;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) {
        const uint4 v = x[i];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        float g[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = HT::lo(w[j]), b = HT::hi(w[j]);
            g[2 * j] = 0.5f * a * (1.f + erff(a * 0.70710678118654752440f));
            g[2 * j + 1] = 0.5f * b * (1.f + erff(b * 0.70710678118654752440f));
        }
        const uint2 r0 = round4<typename HT::elem>(g[0], g[1], g[2], g[3]), r1 = round4<typename HT::elem>(g[4], g[5], g[6], g[7]);
        y[i] = make_uint4(r0.x, r0.y, r1.x, r1.y);
    }
}

template <int C, int DT>
static int launch_add_layernorm(const uint16_t *x, const float *res, const float *gamma, const float *beta, float eps,
                                float *out_f32, uint16_t *out_half, int64_t rows, hipStream_t s) {
    hipLaunchKernelGGL((add_layernorm_kernel<C, DT>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, res, gamma, beta, eps,
                       out_f32, out_half, rows);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

template <int DT>
static int add_layernorm_any(const uint16_t *x, const float *res, const float *gamma, const float *beta, float eps, float *out_f32,
                             uint16_t *out_half, int64_t rows, int dim, hipStream_t s) {
    switch (dim / 256) {
        case 1: return launch_add_layernorm<1, DT>(x, res, gamma, beta, eps, out_f32, out_half, rows, s);
        case 2: return launch_add_layernorm<2, DT>(x, res, gamma, beta, eps, out_f32, out_half, rows, s);
        case 3: return launch_add_layernorm<3, DT>(x, res, gamma, beta, eps, out_f32, out_half, rows, s);
        case 4: return launch_add_layernorm<4, DT>(x, res, gamma, beta, eps, out_f32, out_half, rows, s);
        case 5: return launch_add_layernorm<5, DT>(x, res, gamma, beta, eps, out_f32, out_half, rows, s);
        case 6: return launch_add_layernorm<6, DT>(x, res, gamma, beta, eps, out_f32, out_half, rows, s);
        case 7: return launch_add_layernorm<7, DT>(x, res, gamma, beta, eps, out_f32, out_half, rows, s);
        default: return launch_add_layernorm<8, DT>(x, res, gamma, beta, eps, out_f32, out_half, rows, s);
    }
}

template <int DT>
static int embed_layernorm_any(const float *word_table, int64_t vocab, const float *position_table, int64_t n_positions,
                               const float *type_table, int64_t n_types, const int64_t *token_ids, const int64_t *positions,
                               const int64_t *token_types, const float *gamma, const float *beta, float eps, float *out_f32,
                               uint16_t *out_half, int64_t rows, int dim, hipStream_t s) {
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define CCR_EMBED_CASE(C)                                                                                                              \
    case C:                                                                                                                            \
        hipLaunchKernelGGL((embed_layernorm_kernel<C, DT>), grid, block, 0, s, word_table, vocab, position_table, n_positions,         \
                           type_table, n_types, token_ids, positions, token_types, gamma, beta, eps, out_f32, out_half, rows);         \
        break;
    switch (dim / 256) {
        CCR_EMBED_CASE(1)
        CCR_EMBED_CASE(2)
        CCR_EMBED_CASE(3)
        CCR_EMBED_CASE(4)
        CCR_EMBED_CASE(5)
        CCR_EMBED_CASE(6)
        CCR_EMBED_CASE(7)
        default:
            hipLaunchKernelGGL((embed_layernorm_kernel<8, DT>), grid, block, 0, s, word_table, vocab, position_table, n_positions,
                               type_table, n_types, token_ids, positions, token_types, gamma, beta, eps, out_f32, out_half, rows);
    }
#undef CCR_EMBED_CASE
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

template <int DT>
static int attention_any(const uint16_t *qkv, const int32_t *seq_start, const int32_t *seq_len, uint16_t *out, int n_seq, int n_heads,
                         int max_len, int pad_len, float scale, hipStream_t stream) {
    const int lk_pad = (max_len + ATT_KB - 1) / ATT_KB * ATT_KB;
    const size_t lds = attention_lds_bytes(lk_pad);
    // the opt-in is cached per (kernel, device) whatever the size: ask for the kernel's maximum once (512 keys), not for this call's
    // image -- length-sorted batches start with the shortest texts
    static const bool tr = [] {
        const char *e = getenv("CCR_ATT_TR");      // A/B knob: 0 = the transposed-image kernel of rounds 3-5
        return !(e && atoi(e) == 0);
    }();
    const int rc = ensure_dynamic_lds(tr ? reinterpret_cast<const void *>(&attention_kernel<DT, true>) : reinterpret_cast<const void *>(&attention_kernel<DT, false>),
                                      attention_lds_bytes(512));
    if (rc != CCR_OK) return rc;
    int waves = (max_len + ATT_QW - 1) / ATT_QW;   // one wave per 32 query rows, at most 8 (longer sequences: the waves loop)
    if (waves > ATT_MAX_THREADS / 64) waves = ATT_MAX_THREADS / 64;
    // 129..192 tokens: the LDS image lets three workgroups share a CU, but workgroups of 5 or 6 waves do not pack three times into
    // its four SIMDs' wave slots (measured residency ~1.5 workgroups); 4 waves, the fifth / sixth query block on a second round
    // of a wave that rotates with the head: 134 -> 120 us at 136 tokens, 125 -> 111 at 160 (longer sequences: two workgroups
    // fit either way and 7-8 waves are faster)
    if (lk_pad == 192 && waves > 4) waves = 4;
    CCR_REQUIRE(lk_pad * 8 <= ATT_KMAX * 64 * waves && lk_pad * 4 <= ATT_VMAX * 64 * waves, "ccr_attention: staging bound (internal)");
    if (tr)
        hipLaunchKernelGGL((attention_kernel<DT, true>), dim3(n_heads, n_seq), dim3(64 * waves), lds, stream, qkv, seq_start, seq_len, out,
                           n_heads, pad_len, max_len, lk_pad, scale * 1.4426950408889634f);
    else
        hipLaunchKernelGGL((attention_kernel<DT, false>), dim3(n_heads, n_seq), dim3(64 * waves), lds, stream, qkv, seq_start, seq_len, out,
                           n_heads, pad_len, max_len, lk_pad, scale * 1.4426950408889634f);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

template <int DT>
static int gelu_any(const uint16_t *x, uint16_t *y, int64_t n, hipStream_t stream) {
    const int64_t n16 = n / 8;
    int64_t blocks = (n16 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(gelu_kernel<DT>, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<const uint4 *>(x),
                       reinterpret_cast<uint4 *>(y), n16);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

}  // namespace ccr

using namespace ccr;

#define CCR_REQUIRE_HALF(dtype, who) \
    CCR_REQUIRE((dtype) == CCR_DTYPE_BF16 || (dtype) == CCR_DTYPE_F16, who ": half_dtype=%d (CCR_DTYPE_F16 or CCR_DTYPE_BF16)", (int)(dtype))

extern "C" int ccr_attention_half(const uint16_t *qkv, const int32_t *seq_start, const int32_t *seq_len, uint16_t *out, int n_seq,
                                  int n_heads, int max_len, int pad_len, float scale, int half_dtype, void *stream) {
    CCR_REQUIRE(qkv && seq_start && seq_len && out, "ccr_attention_half: null pointer");
    CCR_REQUIRE_HALF(half_dtype, "ccr_attention_half");
    CCR_REQUIRE(n_seq >= 0 && n_seq <= 65535 && n_heads > 0 && n_heads <= 1024, "ccr_attention_half: bad shape n_seq=%d n_heads=%d",
                n_seq, n_heads);
    CCR_REQUIRE(max_len > 0 && max_len <= 512 && pad_len >= 0 && pad_len <= 512,
                "ccr_attention_half: max_len=%d pad_len=%d (1..512 tokens per sequence)", max_len, pad_len);
    CCR_REQUIRE(scale > 0.f, "ccr_attention_half: scale must be positive");
    if (n_seq == 0) return CCR_OK;
    return half_dtype == CCR_DTYPE_F16
               ? attention_any<CCR_DTYPE_F16>(qkv, seq_start, seq_len, out, n_seq, n_heads, max_len, pad_len, scale, (hipStream_t)stream)
               : attention_any<CCR_DTYPE_BF16>(qkv, seq_start, seq_len, out, n_seq, n_heads, max_len, pad_len, scale, (hipStream_t)stream);
}

extern "C" int ccr_attention_bf16(const uint16_t *qkv, const int32_t *seq_start, const int32_t *seq_len, uint16_t *out,
                                  int n_seq, int n_heads, int max_len, int pad_len, float scale, void *stream) {
    return ccr_attention_half(qkv, seq_start, seq_len, out, n_seq, n_heads, max_len, pad_len, scale, CCR_DTYPE_BF16, stream);
}

extern "C" int ccr_add_layernorm_half(const uint16_t *x_half, const float *residual, const float *gamma, const float *beta, float eps,
                                      float *out_f32, uint16_t *out_half, int64_t rows, int dim, int half_dtype, void *stream) {
    CCR_REQUIRE(x_half && gamma && beta && (out_f32 || out_half), "ccr_add_layernorm_half: null pointer");
    CCR_REQUIRE_HALF(half_dtype, "ccr_add_layernorm_half");
    CCR_REQUIRE(rows >= 0 && dim > 0 && dim % 256 == 0 && dim <= 2048,
                "ccr_add_layernorm_half: rows=%lld dim=%d (dim %% 256 == 0, dim <= 2048)", (long long)rows, dim);
    if (rows == 0) return CCR_OK;
    hipStream_t s = (hipStream_t)stream;
    return half_dtype == CCR_DTYPE_F16 ? add_layernorm_any<CCR_DTYPE_F16>(x_half, residual, gamma, beta, eps, out_f32, out_half, rows, dim, s)
                                       : add_layernorm_any<CCR_DTYPE_BF16>(x_half, residual, gamma, beta, eps, out_f32, out_half, rows, dim, s);
}

extern "C" int ccr_add_layernorm(const uint16_t *x_bf16, const float *residual, const float *gamma, const float *beta, float eps,
                                 float *out_f32, uint16_t *out_bf16, int64_t rows, int dim, void *stream) {
    return ccr_add_layernorm_half(x_bf16, residual, gamma, beta, eps, out_f32, out_bf16, rows, dim, CCR_DTYPE_BF16, stream);
}

extern "C" int ccr_embed_layernorm_half(const float *word_table, int64_t vocab, const float *position_table, int64_t n_positions,
                                        const float *type_table, int64_t n_types, const int64_t *token_ids, const int64_t *positions,
                                        const int64_t *token_types, const float *gamma, const float *beta, float eps, float *out_f32,
                                        uint16_t *out_half, int64_t rows, int dim, int half_dtype, void *stream) {
    CCR_REQUIRE(word_table && position_table && type_table && token_ids && positions && gamma && beta && (out_f32 || out_half),
                "ccr_embed_layernorm_half: null pointer");
    CCR_REQUIRE_HALF(half_dtype, "ccr_embed_layernorm_half");
    CCR_REQUIRE(vocab > 0 && n_positions > 0 && n_types > 0, "ccr_embed_layernorm_half: empty table");
    CCR_REQUIRE(rows >= 0 && dim > 0 && dim % 256 == 0 && dim <= 2048,
                "ccr_embed_layernorm_half: rows=%lld dim=%d (dim %% 256 == 0, dim <= 2048)", (long long)rows, dim);
    if (rows == 0) return CCR_OK;
    hipStream_t s = (hipStream_t)stream;
    return half_dtype == CCR_DTYPE_F16
               ? embed_layernorm_any<CCR_DTYPE_F16>(word_table, vocab, position_table, n_positions, type_table, n_types, token_ids,
                                                    positions, token_types, gamma, beta, eps, out_f32, out_half, rows, dim, s)
               : embed_layernorm_any<CCR_DTYPE_BF16>(word_table, vocab, position_table, n_positions, type_table, n_types, token_ids,
                                                     positions, token_types, gamma, beta, eps, out_f32, out_half, rows, dim, s);
}

extern "C" int ccr_embed_layernorm(const float *word_table, int64_t vocab, const float *position_table, int64_t n_positions,
                                   const float *type_table, int64_t n_types, const int64_t *token_ids, const int64_t *positions,
                                   const int64_t *token_types, const float *gamma, const float *beta, float eps, float *out_f32,
                                   uint16_t *out_bf16, int64_t rows, int dim, void *stream) {
    return ccr_embed_layernorm_half(word_table, vocab, position_table, n_positions, type_table, n_types, token_ids, positions,
                                    token_types, gamma, beta, eps, out_f32, out_bf16, rows, dim, CCR_DTYPE_BF16, stream);
}

extern "C" int ccr_gelu_half(const uint16_t *x, uint16_t *y, int64_t n, int half_dtype, void *stream) {
    CCR_REQUIRE(x && y, "ccr_gelu_half: null pointer");
    CCR_REQUIRE_HALF(half_dtype, "ccr_gelu_half");
    CCR_REQUIRE(n >= 0 && n % 8 == 0, "ccr_gelu_half: n=%lld (a multiple of 8 elements)", (long long)n);
    CCR_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0, "ccr_gelu_half: 16-byte aligned arrays");
    if (n == 0) return CCR_OK;
    return half_dtype == CCR_DTYPE_F16 ? gelu_any<CCR_DTYPE_F16>(x, y, n, (hipStream_t)stream)
                                       : gelu_any<CCR_DTYPE_BF16>(x, y, n, (hipStream_t)stream);
}

extern "C" int ccr_gelu_bf16(const uint16_t *x, uint16_t *y, int64_t n, void *stream) {
    return ccr_gelu_half(x, y, n, CCR_DTYPE_BF16, stream);
}
