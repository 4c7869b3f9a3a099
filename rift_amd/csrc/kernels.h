// Non-GEMM kernels of the Pluto forward for gfx950: feature builders, small-sequence
// attention, LayerNorm, BatchNorm statistics, masked max-pool, token assembly.
// All activations are fp32 row-major in HBM; every kernel is HBM/latency bound
// (tiny arithmetic), so the rules that matter are coalesced rows and one pass.
#pragma once
#include "common.h"
#include "token.h"

namespace RIFT_NS {

#define RIFT_PI 3.14159265358979323846f

// ---------------------------------------------------------------------------
// weight packing: fp32 [N][K] (row stride K) -> [Npad][Kp] bf16 / fp32, zero padded.
// conv mode: src [N][C][3] -> dst[n][j*C + c]  (tap-major, matches AMODE_CONV3)
// tap_lo/tap_n select a subset of taps (fpn_conv only needs taps 0,1 at the last step).
// ---------------------------------------------------------------------------
template <bool BF16>
__global__ void pack_weight_kernel(const float* __restrict__ src, void* dst, int N, int K, int Npad, int Kp,
                                   int conv_C, int tap_lo, int tap_n, int src_row_off, int src_ld) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Npad * Kp) return;
  const int n = idx / Kp, k = idx - n * Kp;
  float v = 0.f;
  if (n < N && k < K) {
    if (conv_C > 0) {
      const int j = k / conv_C, c = k - j * conv_C;
      if (j < tap_n) v = src[((size_t)n * conv_C + c) * 3 + tap_lo + j];
    } else {
      v = src[(size_t)(n + src_row_off) * src_ld + k];
    }
  }
  if (BF16) reinterpret_cast<unsigned short*>(dst)[fm_index(n, k, Kp)] = f2h(v);   // fragment-major (common.h)
  else reinterpret_cast<float*>(dst)[idx] = v;
}
// the fragment-major image of an fc2 weight in hidden-layer operand words (opfmt.h: fp16 in the packed-fp16-GELU build), from the fp32 image
__global__ void pack_weight_hid_kernel(const float* __restrict__ f32, unsigned short* __restrict__ dst, int Npad, int Kp) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Npad * Kp) return;
  const int n = idx / Kp, k = idx - n * Kp;
  dst[fm_index(n, k, Kp)] = f2h_hid(f32[idx]);
}

// ---------------------------------------------------------------------------
// feature builders
// ---------------------------------------------------------------------------
// agent 9-channel difference features (agent_encoder.py:54-75) -> F[(b*A+a)*20 + t][9]
// also valid_agent[b*A+a] = any(valid[:21]) and, if asked for, hist_agent[b*A+a] = valid_agent && a != 0: the sequences the history
// encoder's output is read of (agent_encoder.py:77-87: it runs on agent_feature[valid_agent_mask], and row 0 is overwritten by the ego state token)
__device__ __forceinline__ void agent_feature_body(const float* __restrict__ pos, const float* __restrict__ head,
                                     const float* __restrict__ vel, const float* __restrict__ shp,
                                     const uint8_t* __restrict__ valid, int nA, int Tfull,
                                     float* __restrict__ F, uint8_t* __restrict__ valid_agent, const int vblk,
                                     uint8_t* __restrict__ hist_agent = nullptr, const int A = 1) {
  const int idx = vblk * (int)blockDim.x + (int)threadIdx.x;   // (agent, t) t in [0,20)
  if (idx >= nA * 20) return;
  const int a = idx / 20, t = idx - a * 20;
  const size_t b1 = (size_t)a * Tfull + t + 1, b0 = b1 - 1;
  const bool vm = valid[b0] && valid[b1];
  float* o = F + (size_t)idx * 9;
  const float dh = vm ? head[b1] - head[b0] : 0.f;
  o[0] = vm ? pos[b1 * 2] - pos[b0 * 2] : 0.f;
  o[1] = vm ? pos[b1 * 2 + 1] - pos[b0 * 2 + 1] : 0.f;
  o[2] = vm ? vel[b1 * 2] - vel[b0 * 2] : 0.f;
  o[3] = vm ? vel[b1 * 2 + 1] - vel[b0 * 2 + 1] : 0.f;
  o[4] = cosf(dh);
  o[5] = sinf(dh);
  o[6] = shp[b1 * 2];
  o[7] = shp[b1 * 2 + 1];
  o[8] = vm ? 1.f : 0.f;
  if (t == 0) {
    bool any = false;
    for (int i = 0; i < 21; ++i) any |= (valid[(size_t)a * Tfull + i] != 0);
    valid_agent[a] = any ? 1 : 0;
    if (hist_agent) hist_agent[a] = (any && (a % A) != 0) ? 1 : 0;
  }
}
__global__ void agent_feature_kernel(const float* __restrict__ pos, const float* __restrict__ head,
                                     const float* __restrict__ vel, const float* __restrict__ shp,
                                     const uint8_t* __restrict__ valid, int nA, int Tfull,
                                     float* __restrict__ F, uint8_t* __restrict__ valid_agent) {
  agent_feature_body(pos, head, vel, shp, valid, nA, Tfull, F, valid_agent, blockIdx.x);
}

// map 10-channel point features (map_encoder.py:43-59) -> F[(b*Mp+m)*20 + p][10]
__device__ __forceinline__ void map_feature_body(const float* __restrict__ pp, const float* __restrict__ pv,
                                   const float* __restrict__ po, const float* __restrict__ center,
                                   int nPoly, float* __restrict__ F, const int vblk) {
  const int idx = vblk * (int)blockDim.x + (int)threadIdx.x;   // (poly, p)
  if (idx >= nPoly * 20) return;
  const int m = idx / 20, p = idx - m * 20;
  const float* P0 = pp + ((size_t)(m * 3 + 0) * 20 + p) * 2;
  const float* P1 = pp + ((size_t)(m * 3 + 1) * 20 + p) * 2;
  const float* P2 = pp + ((size_t)(m * 3 + 2) * 20 + p) * 2;
  const float* V0 = pv + ((size_t)(m * 3 + 0) * 20 + p) * 2;
  const float o0 = po[(size_t)(m * 3 + 0) * 20 + p];
  float* o = F + (size_t)idx * 10;
  o[0] = P0[0] - center[m * 3 + 0];
  o[1] = P0[1] - center[m * 3 + 1];
  o[2] = V0[0]; o[3] = V0[1];
  o[4] = cosf(o0); o[5] = sinf(o0);
  o[6] = P1[0] - P0[0]; o[7] = P1[1] - P0[1];
  o[8] = P2[0] - P0[0]; o[9] = P2[1] - P0[1];
}
__global__ void map_feature_kernel(const float* __restrict__ pp, const float* __restrict__ pv,
                                   const float* __restrict__ po, const float* __restrict__ center,
                                   int nPoly, float* __restrict__ F) {
  map_feature_body(pp, pv, po, center, nPoly, F, blockIdx.x);
}

// reference-line 6-channel features (planning_decoder.py:145-153) -> F[(b*R+r)*120 + p][6]
__device__ __forceinline__ void ref_feature_body(const float* __restrict__ rp, const float* __restrict__ rv,
                                   const float* __restrict__ ro, int nLine, float* __restrict__ F, const int vblk) {
  const int idx = vblk * (int)blockDim.x + (int)threadIdx.x;
  if (idx >= nLine * 120) return;
  const int l = idx / 120;
  const float* p0 = rp + (size_t)l * 120 * 2;
  float* o = F + (size_t)idx * 6;
  o[0] = rp[(size_t)idx * 2] - p0[0];
  o[1] = rp[(size_t)idx * 2 + 1] - p0[1];
  o[2] = rv[(size_t)idx * 2]; o[3] = rv[(size_t)idx * 2 + 1];
  const float a = ro[idx];
  o[4] = cosf(a); o[5] = sinf(a);
}
__global__ void ref_feature_kernel(const float* __restrict__ rp, const float* __restrict__ rv,
                                   const float* __restrict__ ro, int nLine, float* __restrict__ F) {
  ref_feature_body(rp, rv, ro, nLine, F, blockIdx.x);
}

// Fourier features (fourier_embedding.py:49-50): row r, input dim d -> F[d][r][129] = [cos(2pi f x), sin(..), x]
// x = in[r*in_ld + d]; optional angle wrap on dim `wrap_dim` ((a + pi) mod 2pi - pi, pluto_model.py:133)
__global__ void fourier_feature_kernel(const float* __restrict__ in, int in_ld, int rows, int D,
                                       const float* __restrict__ freqs /*[D][64]*/, int wrap_dim,
                                       float* __restrict__ F /*[D][rows][129]*/) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // (d, r, f) f in [0,65)
  if (idx >= D * rows * 65) return;
  const int f = idx % 65, r = (idx / 65) % rows, d = idx / (65 * rows);
  float x = in[(size_t)r * in_ld + d];
  if (d == wrap_dim) { x = fmodf(x + RIFT_PI, 2.f * RIFT_PI); if (x < 0.f) x += 2.f * RIFT_PI; x -= RIFT_PI; }
  float* o = F + ((size_t)d * rows + r) * 129;
  if (f == 64) { o[128] = x; return; }
  const float arg = x * freqs[d * 64 + f] * 2.f * RIFT_PI;   // same association as the reference expression
  o[f] = cosf(arg);
  o[64 + f] = sinf(arg);
}

// ---------------------------------------------------------------------------
// LayerNorm over rows (C <= 512), one wave per row; optional ReLU; in-place safe.
// ---------------------------------------------------------------------------
__global__ void layernorm_kernel(const float* __restrict__ X, int ldx, float* __restrict__ Y, int ldy, int rows, int C,
                                 const float* __restrict__ g, const float* __restrict__ b, float eps, int relu) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float v[8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { const int k = lane + i * 64; v[i] = (k < C) ? X[(size_t)row * ldx + k] : 0.f; s += v[i]; }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { const int k = lane + i * 64; const float d = (k < C) ? v[i] - mean : 0.f; q += d * d; }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = lane + i * 64;
    if (k < C) { float y = (v[i] - mean) * rstd * g[k] + b[k]; if (relu) y = fmaxf(y, 0.f); Y[(size_t)row * ldy + k] = y; }
  }
}

// ---------------------------------------------------------------------------
// NAT 1-D neighbourhood attention (natten 0.14.6 semantics; head_dim 16).
// QKV rows [nseq*L][3C] laid out (q | k | v), each (H, 16).  One thread per (seq, pos, head).
// ---------------------------------------------------------------------------
template <int ksz>
__global__ void nat_attention_kernel(const float* __restrict__ QKV, const float* __restrict__ rpb, int nseq, int L,
                                     int H, float* __restrict__ O) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nseq * L * H) return;
  const int h = idx % H, i = (idx / H) % L, s = idx / (H * L);
  const int C = H * 16;
  const float scale = 0.25f;   // 16^-0.5
  const float* qp = QKV + ((size_t)(s * L + i)) * 3 * C + h * 16;
  float q[16];
#pragma unroll
  for (int d = 0; d < 16; ++d) q[d] = qp[d] * scale;
  int start = i - ksz / 2;
  start = start < 0 ? 0 : start;
  start = start > L - ksz ? L - ksz : start;
  float sc[ksz];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < ksz; ++j) {
    const int nb = start + j;
    const float* kp = QKV + ((size_t)(s * L + nb)) * 3 * C + C + h * 16;
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) d += q[e] * kp[e];
    d += rpb[h * (2 * ksz - 1) + (nb - i) + ksz - 1];
    sc[j] = d;
    mx = fmaxf(mx, d);
  }
  float den = 0.f;
#pragma unroll
  for (int j = 0; j < ksz; ++j) { sc[j] = expf(sc[j] - mx); den += sc[j]; }
  float o[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) o[e] = 0.f;
#pragma unroll
  for (int j = 0; j < ksz; ++j) {
    const float* vp = QKV + ((size_t)(s * L + start + j)) * 3 * C + 2 * C + h * 16;
    const float w = sc[j] / den;
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] += w * vp[e];
  }
  float* op = O + ((size_t)(s * L + i)) * C + h * 16;
#pragma unroll
  for (int e = 0; e < 16; ++e) op[e] = o[e];
}

// ---------------------------------------------------------------------------
// Generic small-sequence multi-head attention (head_dim 32), fp32, online softmax.
// batch index = (bo, bi); query row = bo*q_outer + bi*q_inner + i*q_stride, key row likewise.
// mask (True = ignore): mask[mb*Lk + j], mb = quirk ? (bo*nb_inner + bi) % mask_mod : bo.
// One thread per (batch, head, query).
// ---------------------------------------------------------------------------
struct MhaP {
  const float* Q; int ldq; const float* K; const float* V; int ldkv; float* O; int ldo;
  int nb_outer, nb_inner, H, Lq, Lk;
  int q_outer, q_inner, q_stride;      // in rows
  int kv_outer, kv_inner, kv_stride;
  int o_outer, o_inner, o_stride;
  const uint8_t* mask; int mask_quirk, mask_mod, mask_off;   // mask_off: data-parallel shard offset of the quirk row (rows of the global batch)
  float dropout_p; uint32_t seed, stream;
};

__global__ void mha_kernel(MhaP p) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = p.nb_outer * p.nb_inner * p.H * p.Lq;
  if (idx >= total) return;
  // query index fastest so that a wave shares (batch, head) -> K/V rows are wave-uniform broadcasts
  const int i = idx % p.Lq;
  const int h = (idx / p.Lq) % p.H;
  const int bi = (idx / (p.Lq * p.H)) % p.nb_inner;
  const int bo = idx / (p.Lq * p.H * p.nb_inner);
  const float scale = 0.17677669529663687f;   // 32^-0.5
  const float* qp = p.Q + ((size_t)bo * p.q_outer + (size_t)bi * p.q_inner + (size_t)i * p.q_stride) * p.ldq + h * 32;
  float q[32], acc[32];
#pragma unroll
  for (int d = 0; d < 32; ++d) { q[d] = qp[d] * scale; acc[d] = 0.f; }
  const size_t kv0 = (size_t)bo * p.kv_outer + (size_t)bi * p.kv_inner;
  const int mb = p.mask ? (p.mask_quirk ? (bo * p.nb_inner + bi + p.mask_off) % p.mask_mod : bo) : 0;
  float m = -INFINITY, l = 0.f;
  for (int j = 0; j < p.Lk; ++j) {
    if (p.mask && p.mask[(size_t)mb * p.Lk + j]) continue;
    const float* kp = p.K + (kv0 + (size_t)j * p.kv_stride) * p.ldkv + h * 32;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < 32; ++d) s += q[d] * kp[d];
    const float mn = fmaxf(m, s);
    const float corr = expf(m - mn);     // exp(-inf) = 0 on the first key
    const float pj = expf(s - mn);
    l = l * corr + pj;
    float w = pj;
    if (p.dropout_p > 0.f) {
      const float u = uniform01(p.seed, p.stream, (uint32_t)idx * (uint32_t)p.Lk + (uint32_t)j);
      w = (u < p.dropout_p) ? 0.f : pj * (1.0f / (1.0f - p.dropout_p));
    }
    const float* vp = p.V + (kv0 + (size_t)j * p.kv_stride) * p.ldkv + h * 32;
#pragma unroll
    for (int d = 0; d < 32; ++d) acc[d] = acc[d] * corr + w * vp[d];
    m = mn;
  }
  const float inv = 1.0f / l;   // all keys masked -> NaN, as torch
  float* op = p.O + ((size_t)bo * p.o_outer + (size_t)bi * p.o_inner + (size_t)i * p.o_stride) * p.ldo + h * 32;
#pragma unroll
  for (int d = 0; d < 32; ++d) op[d] = acc[d] * inv;
}

// ---------------------------------------------------------------------------
// MFMA small-sequence attention (bf16 operands, fp32 softmax / accumulate), head_dim 32.
// One workgroup per (batch, head): K [Lk][32] and V^T [32][Lk] are staged once in LDS as bf16;
// each wave owns 16-query tiles.  S^T = K Q^T is computed "swapped" (v_mfma_f32_16x16x32_bf16 with the
// key tile as A and the query tile as B), so a lane holds, for ONE query (column lane&15), the scores of
// keys 4*(lane>>4)+r of every 16-key tile: the row softmax is lane-local plus two xor-shuffles, and the
// probabilities already sit in the A-fragment layout of the P.V MFMA (k-slot (lane>>4, j) <-> key
// tile(j>>2)*16 + 4*(lane>>4) + (j&3); V^T fragments are read with the same slot->key map).
// NKT = compile-time number of 16-key tiles (keys padded to 32).
// ---------------------------------------------------------------------------
template <int NKT>
__global__ void mha_mfma_kernel(MhaP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int LKP = NKT * 16;
  constexpr int KS = 40;            // K row stride (bf16 elements): 32 + 8 pad -> conflict-free 16-byte fragment reads
  constexpr int VS = LKP + 8;       // V^T row stride
  unsigned short* Ks = reinterpret_cast<unsigned short*>(smem_raw);
  unsigned short* Vt = Ks + LKP * KS;
  unsigned char* smask = reinterpret_cast<unsigned char*>(Vt + 32 * VS);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = blockDim.x >> 6;
  const int h = blockIdx.x % p.H;
  const int bi = (blockIdx.x / p.H) % p.nb_inner;
  const int bo = blockIdx.x / (p.H * p.nb_inner);
  const size_t kv0 = (size_t)bo * p.kv_outer + (size_t)bi * p.kv_inner;
  const int mb = p.mask ? (p.mask_quirk ? (bo * p.nb_inner + bi + p.mask_off) % p.mask_mod : bo) : 0;
  // ---- stage K (row-major) and V (transposed) as bf16; keys >= Lk are zero and masked
  for (int i = tid; i < LKP * 8; i += blockDim.x) {
    const int key = i >> 3, d4 = (i & 7) * 4;
    float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
    if (key < p.Lk) {
      const size_t row = kv0 + (size_t)key * p.kv_stride;
      kv = *reinterpret_cast<const float4*>(p.K + row * p.ldkv + h * 32 + d4);
      vv = *reinterpret_cast<const float4*>(p.V + row * p.ldkv + h * 32 + d4);
    }
    uint2 u;
    u.x = (unsigned)f2h(kv.x) | ((unsigned)f2h(kv.y) << 16);
    u.y = (unsigned)f2h(kv.z) | ((unsigned)f2h(kv.w) << 16);
    *reinterpret_cast<uint2*>(Ks + key * KS + d4) = u;
    Vt[(d4 + 0) * VS + key] = f2h(vv.x); Vt[(d4 + 1) * VS + key] = f2h(vv.y);
    Vt[(d4 + 2) * VS + key] = f2h(vv.z); Vt[(d4 + 3) * VS + key] = f2h(vv.w);
  }
  for (int i = tid; i < LKP; i += blockDim.x)
    smask[i] = (i >= p.Lk) || (p.mask && p.mask[(size_t)mb * p.Lk + i]);
  __syncthreads();

  const int l15 = lane & 15, l4 = lane >> 4;
  const float scale = 0.17677669529663687f;   // 32^-0.5
  const int nqt = (p.Lq + 15) >> 4;
  for (int qt = wave; qt < nqt; qt += nwave) {
    const int q = qt * 16 + l15;
    h16x8 qf = (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
    if (q < p.Lq) {
      const float* qp = p.Q + ((size_t)bo * p.q_outer + (size_t)bi * p.q_inner + (size_t)q * p.q_stride) * p.ldq + h * 32 + l4 * 8;
      const float4 a = *reinterpret_cast<const float4*>(qp), b = *reinterpret_cast<const float4*>(qp + 4);
      qf[0] = (short)f2h(a.x * scale); qf[1] = (short)f2h(a.y * scale); qf[2] = (short)f2h(a.z * scale); qf[3] = (short)f2h(a.w * scale);
      qf[4] = (short)f2h(b.x * scale); qf[5] = (short)f2h(b.y * scale); qf[6] = (short)f2h(b.z * scale); qf[7] = (short)f2h(b.w * scale);
    }
    f32x4 s[NKT];
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const h16x8 kf = *reinterpret_cast<const h16x8*>(Ks + (kt * 16 + l15) * KS + l4 * 8);
      s[kt] = mfma_h(kf, qf, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (smask[kt * 16 + l4 * 4 + r]) s[kt][r] = -INFINITY;
        m = fmaxf(m, s[kt][r]);
      }
    }
    m = rows_max(m);
    float lsum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = expf(s[kt][r] - m);
        lsum += e;
        float w = e;
        if (p.dropout_p > 0.f) {
          const uint32_t qid = (uint32_t)((blockIdx.x * p.Lq + q) * LKP + kt * 16 + l4 * 4 + r);
          w = (uniform01(p.seed, p.stream, qid) < p.dropout_p) ? 0.f : e * (1.0f / (1.0f - p.dropout_p));
        }
        s[kt][r] = w;
      }
    lsum = rows_sum(lsum);
    f32x4 o0 = (f32x4){0.f, 0.f, 0.f, 0.f}, o1 = o0;
#pragma unroll
    for (int pt = 0; pt < NKT / 2; ++pt) {
      h16x8 pf;
#pragma unroll
      for (int j = 0; j < 4; ++j) { pf[j] = (short)f2h(s[2 * pt][j]); pf[4 + j] = (short)f2h(s[2 * pt + 1][j]); }
      const unsigned short* v0 = Vt + l15 * VS + pt * 32 + l4 * 4;
      const unsigned short* v1 = Vt + (16 + l15) * VS + pt * 32 + l4 * 4;
      h16x8 b0, b1;
      const uint2 x0 = *reinterpret_cast<const uint2*>(v0), x1 = *reinterpret_cast<const uint2*>(v0 + 16);
      const uint2 y0 = *reinterpret_cast<const uint2*>(v1), y1 = *reinterpret_cast<const uint2*>(v1 + 16);
      b0[0] = (short)(x0.x & 0xffff); b0[1] = (short)(x0.x >> 16); b0[2] = (short)(x0.y & 0xffff); b0[3] = (short)(x0.y >> 16);
      b0[4] = (short)(x1.x & 0xffff); b0[5] = (short)(x1.x >> 16); b0[6] = (short)(x1.y & 0xffff); b0[7] = (short)(x1.y >> 16);
      b1[0] = (short)(y0.x & 0xffff); b1[1] = (short)(y0.x >> 16); b1[2] = (short)(y0.y & 0xffff); b1[3] = (short)(y0.y >> 16);
      b1[4] = (short)(y1.x & 0xffff); b1[5] = (short)(y1.x >> 16); b1[6] = (short)(y1.y & 0xffff); b1[7] = (short)(y1.y >> 16);
      o0 = mfma_h(pf, b0, o0, 0, 0, 0);
      o1 = mfma_h(pf, b1, o1, 0, 0, 0);
    }
    // O[q = 4*l4 + r][d = l15 (+16)] ; the softmax denominator of query qq lives in lane qq
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qq = l4 * 4 + r;
      const float inv = 1.0f / __shfl(lsum, qq, 64);
      const int qo = qt * 16 + qq;
      if (qo < p.Lq) {
        float* op = p.O + ((size_t)bo * p.o_outer + (size_t)bi * p.o_inner + (size_t)qo * p.o_stride) * p.ldo + h * 32;
        op[l15] = o0[r] * inv;
        op[16 + l15] = o1[r] * inv;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// BatchNorm1d over valid rows (embedding.py:260,266): column sum / sum of squares in fp64,
// then fold into a per-channel affine  y = x*scale + shift  for the next GEMM's prologue.
// ---------------------------------------------------------------------------
__global__ void bn_partial_kernel(const float* __restrict__ X, int ld, int rows, int C,
                                  const uint8_t* __restrict__ valid, double* __restrict__ part /*[nblk][2][C]*/,
                                  int* __restrict__ cnt /*[nblk]*/, int rows_per_blk) {
  const int c = threadIdx.x;   // blockDim.x == C (128 or 256)
  const int r0 = blockIdx.x * rows_per_blk;
  const int r1 = min(rows, r0 + rows_per_blk);
  double s = 0.0, q = 0.0;
  int n = 0;
  for (int r = r0; r < r1; ++r) {
    if (!valid[r]) continue;
    const double x = (double)X[(size_t)r * ld + c];
    s += x; q += x * x; ++n;
  }
  part[((size_t)blockIdx.x * 2 + 0) * C + c] = s;
  part[((size_t)blockIdx.x * 2 + 1) * C + c] = q;
  if (c == 0) cnt[blockIdx.x] = n;
}

// train != 0: batch statistics (+ running-stat update, momentum 0.1, unbiased var); else running stats.
// One wave per channel: lanes stride over the per-block partial sums (deterministic order).
__global__ void bn_finalize_kernel(const double* __restrict__ part, const int* __restrict__ cnt, int nblk, int C,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* running_mean, float* running_var, long long* num_batches, int train,
                                   int update_running, float eps, float* __restrict__ scale, float* __restrict__ shift,
                                   double* sums /*[2C+1] (sum, sum of squares per channel, count)*/, int sums_mode /*0 local, 1 emit, 2 consume*/) {
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (c >= C) return;
  float mean, var;
  if (train) {
    double s = 0.0, q = 0.0;
    long long n = 0;
    if (sums_mode != 2) for (int b = lane; b < nblk; b += 64) { s += part[((size_t)b * 2) * C + c]; q += part[((size_t)b * 2 + 1) * C + c]; n += cnt[b]; }
    if (sums_mode != 2) {
    s = wave_sum_d(s); q = wave_sum_d(q);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
    }
    if (sums_mode == 1) {        // data parallel: this rank's sums go to the exchange buffer; a second launch consumes the reduced ones
      if (lane == 0) { sums[c] = s; sums[C + c] = q; if (c == 0) sums[2 * C] = (double)n; }
      return;
    }
    if (sums_mode == 2) { s = sums[c]; q = sums[C + c]; n = (long long)sums[2 * C]; }
    const double mu = s / (double)n;
    double v = q / (double)n - mu * mu;
    v = v < 0.0 ? 0.0 : v;
    mean = (float)mu; var = (float)v;
    if (update_running && lane == 0) {
      const double unb = n > 1 ? v * (double)n / (double)(n - 1) : v;
      running_mean[c] = 0.9f * running_mean[c] + 0.1f * mean;
      running_var[c] = 0.9f * running_var[c] + 0.1f * (float)unb;
      if (c == 0 && num_batches) *num_batches += 1;
    }
  } else { mean = running_mean[c]; var = running_var[c]; }
  if (lane == 0) {
    const float sc = gamma[c] * rsqrtf(var + eps);
    scale[c] = sc;
    shift[c] = beta[c] - mean * sc;
  }
}

// Data-parallel r2r quirk: `tgt_key_padding_mask.repeat(M, 1)` (planning_decoder.py:56-60) makes row b*12+m of the batch use the
// reference-line padding of scene (b*12+m) % bs -- of the GLOBAL minibatch when it is sharded over ranks.  Every rank writes its
// scenes' masks into its slots of the exchange buffer (zeros elsewhere); the SUM all-reduce that carries the BatchNorm sums gathers them.
__global__ void dp_kpm_fill_kernel(const uint8_t* __restrict__ r_kpm, int n_local, int slot0, int n_global, double* __restrict__ x) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_global) return;
  const int l = i - slot0;
  x[i] = (l >= 0 && l < n_local && r_kpm[l]) ? 1.0 : 0.0;
}
__global__ void dp_kpm_read_kernel(const double* __restrict__ x, int n_global, uint8_t* __restrict__ g_kpm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_global) g_kpm[i] = x[i] != 0.0;
}

// masked max-pool over the n points of each group (PointsEncoder: invalid points are all-zero rows)
// optional on-the-fly affine+relu is NOT applied here: X is the post-Linear feature.
__global__ void masked_maxpool_kernel(const float* __restrict__ X, int ld, int groups, int n, int C,
                                      const uint8_t* __restrict__ valid, float* __restrict__ Y) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= groups * C) return;
  const int c = idx % C, g = idx / C;
  float m = -INFINITY;
  bool any_invalid = false;
  for (int i = 0; i < n; ++i) {
    const int r = g * n + i;
    if (valid[r]) m = fmaxf(m, X[(size_t)r * ld + c]); else any_invalid = true;
  }
  if (any_invalid) m = fmaxf(m, 0.f);
  Y[(size_t)g * C + c] = m;
}

// ---------------------------------------------------------------------------
// small assembly kernels
// ---------------------------------------------------------------------------
// key padding masks: kpm[b][tok] for tok in agents | polygons | static (True = padded)
__device__ __forceinline__ void token_mask_body(const uint8_t* __restrict__ valid_agent, const uint8_t* __restrict__ map_valid,
                                  const uint8_t* __restrict__ static_valid, int bs, int A, int Mp, int S,
                                  uint8_t* __restrict__ kpm, const int vblk, const uint8_t* __restrict__ agent_valid_raw = nullptr,
                                  int Tfull = 0) {
  const int idx = vblk * (int)blockDim.x + (int)threadIdx.x;
  const int N = A + Mp + S;
  if (idx >= bs * N) return;
  const int b = idx / N, t = idx - b * N;
  uint8_t pad;
  if (t < A) {
    if (valid_agent) pad = !valid_agent[b * A + t];
    else {   // straight from the (agent, time) validity (same rule as agent_feature_body's valid_agent)
      bool any = false;
      for (int i = 0; i < 21; ++i) any |= (agent_valid_raw[(size_t)(b * A + t) * Tfull + i] != 0);
      pad = !any;
    }
  }
  else if (t < A + Mp) {
    bool any = false;
    const uint8_t* v = map_valid + ((size_t)b * Mp + (t - A)) * 20;
    for (int i = 0; i < 20; ++i) any |= (v[i] != 0);
    pad = !any;
  } else pad = !static_valid[b * S + (t - A - Mp)];
  kpm[idx] = pad;
}
__global__ void token_mask_kernel(const uint8_t* __restrict__ valid_agent, const uint8_t* __restrict__ map_valid,
                                  const uint8_t* __restrict__ static_valid, int bs, int A, int Mp, int S,
                                  uint8_t* __restrict__ kpm) {
  token_mask_body(valid_agent, map_valid, static_valid, bs, A, Mp, S, kpm, blockIdx.x);
}

// ref-line key padding: r_kpm[b*R + r] = !any(valid[b,r,:120])
// (r_tiles, optional: the 16-row tiles of the line up to its LAST valid point, 0 .. 8 -- what pe_w_kernel's packed rounds hold of it, pe_fused.h)
__device__ __forceinline__ void refline_mask_body(const uint8_t* __restrict__ rvalid, int nLine, uint8_t* __restrict__ r_kpm, const int vblk,
                                                  uint8_t* __restrict__ r_tiles = nullptr) {
  const int l = vblk * (int)blockDim.x + (int)threadIdx.x;
  if (l >= nLine) return;
  int last = -1;
  for (int i = 0; i < 120; ++i) if (rvalid[(size_t)l * 120 + i] != 0) last = i;
  r_kpm[l] = last < 0;
  if (r_tiles) r_tiles[l] = (uint8_t)((last + 16) >> 4);
}
__global__ void refline_mask_kernel(const uint8_t* __restrict__ rvalid, int nLine, uint8_t* __restrict__ r_kpm) {
  refline_mask_body(rvalid, nLine, r_kpm, blockIdx.x);
}

// token positions for pos_emb: pos[b][tok][3] = (x, y, angle) (pluto_model.py:131-146; wrap applied in fourier kernel)
__device__ __forceinline__ void token_pos_body(const float* __restrict__ agent_pos, const float* __restrict__ agent_head, int Tfull,
                                 const float* __restrict__ center, const float* __restrict__ st_pos,
                                 const float* __restrict__ st_head, int bs, int A, int Mp, int S, float* __restrict__ pos, const int vblk) {
  const int idx = vblk * (int)blockDim.x + (int)threadIdx.x;
  const int N = A + Mp + S;
  if (idx >= bs * N) return;
  const int b = idx / N, t = idx - b * N;
  float x, y, a;
  if (t < A) {
    const size_t o = ((size_t)(b * A + t)) * Tfull + 20;
    x = agent_pos[o * 2]; y = agent_pos[o * 2 + 1]; a = agent_head[o];
  } else if (t < A + Mp) {
    const float* c = center + ((size_t)b * Mp + (t - A)) * 3;
    x = c[0]; y = c[1]; a = c[2];
  } else {
    const size_t o = (size_t)b * S + (t - A - Mp);
    x = st_pos[o * 2]; y = st_pos[o * 2 + 1]; a = st_head[o];
  }
  pos[(size_t)idx * 3] = x; pos[(size_t)idx * 3 + 1] = y; pos[(size_t)idx * 3 + 2] = a;
}
__global__ void token_pos_kernel(const float* __restrict__ agent_pos, const float* __restrict__ agent_head, int Tfull,
                                 const float* __restrict__ center, const float* __restrict__ st_pos,
                                 const float* __restrict__ st_head, int bs, int A, int Mp, int S, float* __restrict__ pos) {
  token_pos_body(agent_pos, agent_head, Tfull, center, st_pos, st_head, bs, A, Mp, S, pos, blockIdx.x);
}

// FPN top-down merge restricted to the positions the last output step depends on
// (embedding.py:79-87 with F.interpolate(scale 2, linear, align_corners=False)):
//   l1'[8] = l1[8] + .25 l2[3] + .75 l2[4];  l1'[9] = l1[9] + l2[4]
//   l0'[18] = l0[18] + .25 l1'[8] + .75 l1'[9];  l0'[19] = l0[19] + l1'[9]
// lat buffers hold 2 positions per agent: lat2 rows (a*2+{0,1}) = t {3,4}; lat1 = {8,9}; lat0 = {18,19}.
// out Z[a][256] = [l0'[18] | l0'[19]]  (input of the pruned fpn_conv GEMM)
__global__ void fpn_merge_kernel(const float* __restrict__ lat0, const float* __restrict__ lat1,
                                 const float* __restrict__ lat2, int nA, float* __restrict__ Z) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nA * 128) return;
  const int c = idx & 127, a = idx >> 7;
  const float l2a = lat2[((size_t)a * 2 + 0) * 128 + c], l2b = lat2[((size_t)a * 2 + 1) * 128 + c];
  const float l1a = lat1[((size_t)a * 2 + 0) * 128 + c] + (0.25f * l2a + 0.75f * l2b);
  const float l1b = lat1[((size_t)a * 2 + 1) * 128 + c] + l2b;
  Z[(size_t)a * 256 + c] = lat0[((size_t)a * 2 + 0) * 128 + c] + (0.25f * l1a + 0.75f * l1b);
  Z[(size_t)a * 256 + 128 + c] = lat0[((size_t)a * 2 + 1) * 128 + c] + l1b;
}

// ego state tokens (agent_encoder.py:113-118): E[b][i][:] = cs[b][i] * w_i + b_i + pos_embed[i]
__global__ void ego_token_kernel(const float* __restrict__ cs, int cs_ld, const float* __restrict__ lw /*[6][128]*/,
                                 const float* __restrict__ lb /*[6][128]*/, const float* __restrict__ pos_embed, int bs,
                                 float* __restrict__ E) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= bs * 6 * 128) return;
  const int c = idx & 127, i = (idx >> 7) % 6, b = idx / (6 * 128);
  E[idx] = (cs[(size_t)b * cs_ld + i] * lw[i * 128 + c] + lb[i * 128 + c]) + pos_embed[i * 128 + c];
}

// state dropout key mask (agent_encoder.py:119-129): first 3 tokens visible, others dropped w.p. p
__global__ void ego_dropmask_kernel(int bs, float p, uint32_t seed, uint32_t stream, uint8_t* __restrict__ mask) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= bs * 6) return;
  const int i = idx % 6;
  mask[idx] = (i >= 3 && uniform01(seed, stream, (uint32_t)idx) < p) ? 1 : 0;
}

__device__ __forceinline__ void agent_token_body(const int idx, const float* __restrict__ nat, const float* __restrict__ x_ego,
                                                 const uint8_t* __restrict__ valid_agent, const int8_t* __restrict__ category,
                                                 const float* __restrict__ type_emb, int bs, int A, int N, float* __restrict__ X,
                                                 const float* __restrict__ pe /*optional (bs*N,128) positional embedding added on the way*/) {
  if (idx >= bs * A * 32) return;                                 // one thread = 4 channels of one agent token (16-byte accesses)
  const int c = (idx & 31) * 4, a = (idx >> 5) % A, b = idx / (A * 32);
  *reinterpret_cast<float4*>(X + ((size_t)b * N + a) * 128 + c) = agent_token_value(b, a, c, nat, x_ego, valid_agent, category, type_emb, A, N, pe);
}

__device__ __forceinline__ void polygon_token_body(const int idx, const float* __restrict__ pooled, const int8_t* __restrict__ ptype,
                                                   const uint8_t* __restrict__ on_route, const int8_t* __restrict__ tl,
                                                   const uint8_t* __restrict__ has_sl, const float* __restrict__ speed_emb,
                                                   const float* __restrict__ type_emb, const float* __restrict__ route_emb,
                                                   const float* __restrict__ tl_emb, const float* __restrict__ unk_emb, int bs,
                                                   int A, int Mp, int N, float* __restrict__ X, const float* __restrict__ pe) {
  if (idx >= bs * Mp * 32) return;                                // one thread = 4 channels of one polygon token
  const int c = (idx & 31) * 4, m = (idx >> 5) % Mp, b = idx / (Mp * 32);
  *reinterpret_cast<float4*>(X + ((size_t)b * N + A + m) * 128 + c) =
      polygon_token_value(b, m, c, pooled, ptype, on_route, tl, has_sl, speed_emb, type_emb, route_emb, tl_emb, unk_emb, A, Mp, N, pe);
}

// (TokenP, agent_token_value, polygon_token_value: token.h)
__global__ void token_kernel(TokenP p) {
  if ((int)blockIdx.x < p.nblk_a)
    agent_token_body(blockIdx.x * blockDim.x + threadIdx.x, p.nat, p.x_ego, p.valid_agent, p.category, p.a_type_emb, p.bs, p.A, p.N, p.X, p.pe);
  else
    polygon_token_body((blockIdx.x - p.nblk_a) * blockDim.x + threadIdx.x, p.pooled, p.ptype, p.on_route, p.tl, p.has_sl, p.speed_emb, p.p_type_emb,
                       p.route_emb, p.tl_emb, p.unk_emb, p.bs, p.A, p.Mp, p.N, p.X, p.pe);
}

// static-object tokens (static_objects_encoder.py:24-26): valid ? fourier(shape) + type_emb : 0
__global__ void static_token_kernel(const float* __restrict__ emb, const int8_t* __restrict__ cat,
                                    const uint8_t* __restrict__ valid, const float* __restrict__ type_emb, int bs,
                                    int A, int Mp, int S, int N, float* __restrict__ X) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= bs * S * 128) return;
  const int c = idx & 127, s = (idx >> 7) % S, b = idx / (S * 128);
  const int o = b * S + s;
  X[((size_t)b * N + A + Mp + s) * 128 + c] = valid[o] ? emb[(size_t)o * 128 + c] + type_emb[(int)cat[o] * 128 + c] : 0.f;
}

// Y[r][c] += Z[r][c]   (pos-embedding add)
__global__ void add_inplace_kernel(float* __restrict__ Y, const float* __restrict__ Z, size_t n) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n) Y[idx] += Z[idx];
}

// decoder query init (planning_decoder.py:162-165): q0[(b,r,m)] = Ra[(b,r)] + Mb[m]
__global__ void build_q0_kernel(const float* __restrict__ Ra, const float* __restrict__ Mb, int nLine, int M,
                                float* __restrict__ Q) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nLine * M * 128) return;
  const int c = idx & 127, m = (idx >> 7) % M, l = idx / (M * 128);
  Q[idx] = Ra[(size_t)l * 128 + c] + Mb[(size_t)m * 128 + c];
}

// r_pos[(b,r)] = (position[b,r,0,:], orientation[b,r,0])  (planning_decoder.py:159)
__device__ __forceinline__ void refline_pos_body(const float* __restrict__ rp, const float* __restrict__ ro, int nLine,
                                   float* __restrict__ pos, const int vblk) {
  const int l = vblk * (int)blockDim.x + (int)threadIdx.x;
  if (l >= nLine) return;
  pos[l * 3] = rp[(size_t)l * 240]; pos[l * 3 + 1] = rp[(size_t)l * 240 + 1]; pos[l * 3 + 2] = ro[(size_t)l * 120];
}
__global__ void refline_pos_kernel(const float* __restrict__ rp, const float* __restrict__ ro, int nLine,
                                   float* __restrict__ pos) {
  refline_pos_body(rp, ro, nLine, pos, blockIdx.x);
}

// gather rows: Y[i] = X[rowidx(i)] with rowidx = (i / per) * stride_rows + off  (token 0 of each scene etc.)
__global__ void gather_rows_kernel(const float* __restrict__ X, int ldx, float* __restrict__ Y, int ldy, int rows,
                                   int C, int per, int stride_rows, int off) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * C) return;
  const int c = idx % C, r = idx / C;
  const size_t src = (size_t)(r / per) * stride_rows + off + (r % per);
  Y[(size_t)r * ldy + c] = X[src * ldx + c];
}

// ---------------------------------------------------------------------------
// All input-only preparation of a forward in ONE launch: agent difference features + agent validity, map / reference-line point
// features, token key-padding masks and positions, reference-line masks and positions.  Seven ~6 us dispatches become one.
// ---------------------------------------------------------------------------
struct PrepP {
  const float *agent_pos, *agent_head, *agent_vel, *agent_shape; const uint8_t* agent_valid; int nA, Tfull;
  float* F9; uint8_t* valid_agent; uint8_t* hist_agent;
  const float *map_pp, *map_pv, *map_po, *map_center; int nPoly; float* F10;
  const float *ref_pos, *ref_vec, *ref_ori; const uint8_t* ref_valid; int nLine; float* F6; float* r_pos; uint8_t* r_kpm; uint8_t* r_tiles;
  const uint8_t *map_valid, *static_valid; const float *st_pos, *st_head; int bs, A, Mp, S; uint8_t* kpm; float* pos;
  int nb[7];
  // (round 6) the ranking of the history encoder's sequences inside this launch (rank_scene_body below): nrk = bs blocks at the HEAD of the
  // grid (0: the ranking is its own launch, nat_rank_kernel)
  int nrk; unsigned long long* rk_pub; unsigned int rk_epoch; int* aidx; int* cnt;
  int* fail;      // the context's sticky device flag (bit 1: the look-back gave up waiting -- rift_check_finite reports it)
  int rk_fault;   // diagnostic (RIFT_RANK_FAULT=1): scene block 0 never publishes -- what the bounded wait is there for (tests)
};

// ---------------------------------------------------------------------------
// The ranking of the compacted history-encoder launch (nat_l0w.h: aidx[3 i + c] = the i-th marked agent slot of residue class c = slot % 3
// in ascending slot order, cnt[c] = the class counts) as blocks of the preparation's own launch.  Rounds 3 - 5 ran it as nat_rank_kernel:
// ONE workgroup scanning all bs * A marks behind the preparation, 13 us on the history chain, most of it the launch and 16 K scattered
// 4-byte stores out of one CU.  Here block b (one per scene, first in the grid) derives its scene's marks from the raw validity (mark =
// any of the first 21 samples && slot % A != 0, the rule of agent_feature_body), PUBLISHES its three class counts as one 64-bit word
// (epoch << 32 | three 9-bit counts: one atomic store carries flag and value), adds up the words of the scenes before it (decoupled
// look-back: its 256 threads read up to 256 predecessors each pass, waiting for a word until it carries this launch's epoch -- workgroups
// are dispatched in block order and a block only ever waits for lower ones, so the wait ends) and writes its own ranks: a scene's ranks
// are consecutive per class, i.e. <= 64 x 12 bytes per store instruction instead of 64 cache lines.  Every scene block is done a few
// microseconds into a launch that lasts ~12; the history chain no longer waits for a ranking at all.  Bit-identical to nat_rank_kernel's
// output (the same order), any A <= 256.  rk_pub: bs words that PERSIST between launches (zero at allocation: epoch 0 is never used).
// RELAXED agent-scope atomics on purpose: the published word carries value and flag together, nothing else travels between the blocks,
// and on this chip an agent-scope release / acquire is an L2 write-back / invalidate (the eight XCDs' L2s are not coherent with each
// other: `buffer_wbl2 sc1` / `buffer_inv sc1`) -- the first version (release store, acquire loads, and before it __threadfence + a
// counter) made the preparation 34 us long instead of 12 (profiles/r06_ab_rank_in_prep.txt).  A monotonic sc1 store / load goes to the
// coherent level and costs a memory round trip.
// ---------------------------------------------------------------------------
#define RIFT_RK_M0 0x9249249249249249ull      // bits j with j % 3 == 0 (bit 63 included: 63 % 3 == 0)
#define RIFT_RK_M1 0x2492492492492492ull      // j % 3 == 1
#define RIFT_RK_M2 0x4924924924924924ull      // j % 3 == 2
__device__ __forceinline__ unsigned long long rk_class_mask(int r) { return r == 0 ? RIFT_RK_M0 : r == 1 ? RIFT_RK_M1 : RIFT_RK_M2; }
__device__ __forceinline__ void rank_scene_body(const PrepP& q, const int b) {
  __shared__ unsigned long long s_mask[4];
  __shared__ unsigned long long s_part[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, A = q.A;
  bool mark = false;
  if (tid < A && tid != 0) {                       // slot % A != 0: the ego row is replaced by the ego state token (agent_encoder.py:87)
    const uint8_t* v = q.agent_valid + ((size_t)b * A + tid) * q.Tfull;
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < 21; ++i) any |= v[i];
    mark = any != 0;
  }
  const unsigned long long m = __builtin_amdgcn_ballot_w64(mark);
  if (lane == 0) s_mask[wave] = m;
  __syncthreads();
  // class of bit j of ballot word w: (wbase[w] + j) % 3
  int wbase[4];
#pragma unroll
  for (int w = 0; w < 4; ++w) wbase[w] = (int)(((long long)b * A + 64 * w) % 3);
  int own[3] = {0, 0, 0};
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const unsigned long long mw = s_mask[w];
#pragma unroll
    for (int c = 0; c < 3; ++c) own[c] += __builtin_popcountll(mw & rk_class_mask((c - wbase[w] + 3) % 3));
  }
  if (tid == 0 && !(q.rk_fault && b == 0))
    __hip_atomic_store(q.rk_pub + b, ((unsigned long long)q.rk_epoch << 32) | ((unsigned long long)own[2] << 18) | ((unsigned long long)own[1] << 9) | (unsigned long long)own[0],
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (relaxed: the word IS the message; see below)
  // ---- look-back: the class counts of the scenes before this one (three 21-bit fields)
  unsigned long long acc = 0ull;
  for (int j = tid; j < b; j += 256) {
    unsigned long long w;
    // (bounded: ~2^21 polls of ~1 us each are a couple of seconds, five orders beyond the longest legitimate wait -- a predecessor that never
    // publishes would otherwise hang the device; the launch then finishes with wrong ranks and the sticky flag says so)
    for (int spins = 0;; ++spins) {
      w = __hip_atomic_load(q.rk_pub + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((unsigned int)(w >> 32) == q.rk_epoch) break;
      if (spins > (1 << 21)) { if (q.fail) atomicOr(q.fail, 2); w = 0ull; break; }
      __builtin_amdgcn_s_sleep(2);
    }
    acc += (w & 0x1ffull) | (((w >> 9) & 0x1ffull) << 21) | (((w >> 18) & 0x1ffull) << 42);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
  if (lane == 0) s_part[wave] = acc;
  __syncthreads();
  const unsigned long long base = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
  if (mark) {
    const int c = (wbase[wave] + lane) % 3;
    int r = (int)((base >> (21 * c)) & 0x1fffffull);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (w > wave) break;
      const unsigned long long sel = s_mask[w] & rk_class_mask((c - wbase[w] + 3) % 3);
      r += __builtin_popcountll(w < wave ? sel : (sel & ((1ull << lane) - 1ull)));
    }
    q.aidx[3 * r + c] = b * A + tid;
  }
  if (b == q.nrk - 1 && tid < 3) q.cnt[tid] = (int)((base >> (21 * tid)) & 0x1fffffull) + own[tid];
}

__device__ __forceinline__ void prep_body(const PrepP& q, int blk) {
  if (blk < q.nrk) { rank_scene_body(q, blk); return; }
  blk -= q.nrk;
  if (blk < q.nb[0]) { agent_feature_body(q.agent_pos, q.agent_head, q.agent_vel, q.agent_shape, q.agent_valid, q.nA, q.Tfull, q.F9, q.valid_agent, blk, q.hist_agent, q.A); return; }
  blk -= q.nb[0];
  if (blk < q.nb[1]) { map_feature_body(q.map_pp, q.map_pv, q.map_po, q.map_center, q.nPoly, q.F10, blk); return; }
  blk -= q.nb[1];
  if (blk < q.nb[2]) { ref_feature_body(q.ref_pos, q.ref_vec, q.ref_ori, q.nLine, q.F6, blk); return; }
  blk -= q.nb[2];
  if (blk < q.nb[3]) { refline_pos_body(q.ref_pos, q.ref_ori, q.nLine, q.r_pos, blk); return; }
  blk -= q.nb[3];
  if (blk < q.nb[4]) { refline_mask_body(q.ref_valid, q.nLine, q.r_kpm, blk, q.r_tiles); return; }
  blk -= q.nb[4];
  if (blk < q.nb[5]) { token_mask_body(nullptr, q.map_valid, q.static_valid, q.bs, q.A, q.Mp, q.S, q.kpm, blk, q.agent_valid, q.Tfull); return; }
  blk -= q.nb[5];
  token_pos_body(q.agent_pos, q.agent_head, q.Tfull, q.map_center, q.st_pos, q.st_head, q.bs, q.A, q.Mp, q.S, q.pos, blk);
}
__global__ __launch_bounds__(256) void prep_kernel(PrepP q) { prep_body(q, blockIdx.x); }

}  // namespace RIFT_NS
