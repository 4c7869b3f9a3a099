// Fused policy-head forward for gfx950: q_final = cat_x_proj(cat[q, enc_emb[:, 0]]) (planning_decoder.py:177-179; the ego-token half
// arrives as one precomputed row per scene) -> pi_head = MLPLayer(128, 128, 1) (planning_decoder.py:184, mlp_layer.py:8-16) -> logits
// with the -1e6 mask of padded reference lines (pluto_model.py:203), for a tile of 128 decoder queries.
// cat_x_proj is frozen: bf16 MFMA.  pi_head is the TRAINABLE layer: its first Linear runs as exact-fp32 MFMA (v_mfma_f32_16x16x4_f32)
// straight from the live fp32 parameters, and both its input (q_final) and its hidden pre-activation are written out in fp32 for the
// analytic backward (loss.h: pi_backward_kernel).  LayerNorm statistics and the final dot product are reduced across the 8 waves
// through small LDS tables.  Replaces two GEMM launches and the tail kernel.
#pragma once
#include "common.h"
#include "pe_fused.h"

namespace RIFT_NS {

struct PiFwdP {
  const float* Q; int rows, rows_per_scene;      // (rows, 128) decoder output
  const unsigned short* wq; const float* bq;     // cat_x_proj columns 0:128, fragment-major bf16 [128][128], and the bias
  const float* x0p;                              // (bs, 128): cat_x_proj columns 128:256 applied to the scene's ego token
  const float* w1; const float* b1;              // pi_head.mlp.0.{weight (128,128) row-major, bias}: live fp32 parameters
  const float *lng, *lnb, *w2, *b2;              // pi_head.mlp.1.{weight,bias}, mlp.3.{weight (128), bias (1)}
  const uint8_t* r_kpm; int M;                   // (rows / M) padded reference lines
  float eps;
  float* QF; float* Hpi; float* prob;            // (rows,128) (rows,128) (rows)
  int* nonfinite;                                // device flag: set when the decoder output holds a NaN / Inf (planning_decoder.py:175)
};

#define PI_ROWS 128
#define PI_QS 144
#define PI_FS 132
#define PI_LDS (PI_ROWS * PI_QS * 2 + PI_ROWS * PI_FS * 4 + 8 * PI_ROWS * 2 * 4 + 8 * PI_ROWS * 4)

__global__ __launch_bounds__(512) void pi_forward_kernel(PiFwdP p) {
  constexpr int MT = 8, NW = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* qb = reinterpret_cast<unsigned short*>(smem_raw);              // [128][PI_QS] bf16 q
  float* qf = reinterpret_cast<float*>(qb + PI_ROWS * PI_QS);                    // [128][PI_FS] fp32 q_final
  float* wpart = qf + PI_ROWS * PI_FS;                                           // [8][128][2] per-wave row sums of h, h^2
  float* zpart = wpart + 8 * PI_ROWS * 2;                                        // [8][128] per-wave partial logits
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int row0 = blockIdx.x * PI_ROWS;
  const int col = wave * 16 + l4 * 4;                                            // this lane's 4 output channels (both layers)

  PFrags<4, 1> Wq;
  p_load_w<NW, 4, 1>(Wq, p.wq, 128, 0, wave, l15, l4);
  // fp32 operand of the trainable layer, as the MFMA A operand: out channel wave*16 + l15, k = 32*l4 + s at step s (the contraction
  // order is free, so each lane takes 32 CONSECUTIVE k: eight 16-byte loads here, eight ds_read_b128 per row tile below)
  float4 w1v[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) w1v[u] = *reinterpret_cast<const float4*>(p.w1 + (size_t)(wave * 16 + l15) * 128 + l4 * 32 + u * 4);
  const float4 bq4 = *reinterpret_cast<const float4*>(p.bq + col), b14 = *reinterpret_cast<const float4*>(p.b1 + col);
  const float4 g4 = *reinterpret_cast<const float4*>(p.lng + col), e4 = *reinterpret_cast<const float4*>(p.lnb + col);
  const float4 w24 = *reinterpret_cast<const float4*>(p.w2 + col);
  const float b2 = p.b2[0];
  bool bad = false;
  {
    float4 qv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = tid + u * 512, r = i >> 5, c4 = (i & 31) * 4;
      qv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row0 + r < p.rows) qv[u] = *reinterpret_cast<const float4*>(p.Q + (size_t)(row0 + r) * 128 + c4);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int i = tid + u * 512; *reinterpret_cast<uint2*>(qb + (i >> 5) * PI_QS + (i & 31) * 4) = pack_h4(qv[u].x, qv[u].y, qv[u].z, qv[u].w); }
  }
  __syncthreads();
  {   // ---- q_final = q Wq^T + b + x0p[scene]
    f32x4 acc[MT][1];
    p_zero(acc);
    p_mma<MT, 4, 1>(acc, qb, PI_QS, 0, Wq, l15, l4);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = mt * 16 + l15, gr = row0 + row;
      float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gr < p.rows) x0 = *reinterpret_cast<const float4*>(p.x0p + (size_t)(gr / p.rows_per_scene) * 128 + col);
      const float4 v = make_float4(acc[mt][0][0] + bq4.x + x0.x, acc[mt][0][1] + bq4.y + x0.y, acc[mt][0][2] + bq4.z + x0.z, acc[mt][0][3] + bq4.w + x0.w);
      *reinterpret_cast<float4*>(qf + row * PI_FS + col) = v;
      if (gr < p.rows) {
        *reinterpret_cast<float4*>(p.QF + (size_t)gr * 128 + col) = v;
        bad |= (v.x * 0.f + v.y * 0.f) + (v.z * 0.f + v.w * 0.f) != 0.f;     // x * 0 is NaN for NaN and +-Inf, 0 otherwise
      }
    }
  }
  // the reference asserts torch.isfinite(q).all() on the decoder queries (planning_decoder.py:175); a NaN / Inf there reaches q_final
  if (__builtin_amdgcn_ballot_w64(bad) != 0ull && lane == 0 && p.nonfinite) atomicOr(p.nonfinite, 1);
  __syncthreads();
  // ---- h = q_final W1^T + b1, exact fp32: D^T[n][m] = sum_k W1[n][k] q_final[m][k]; a lane ends with 4 consecutive n of row m = l15
  f32x4 h[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    h[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 xv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) xv[u] = *reinterpret_cast<const float4*>(qf + (mt * 16 + l15) * PI_FS + l4 * 32 + u * 4);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      h[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1v[u].x, xv[u].x, h[mt], 0, 0, 0);
      h[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1v[u].y, xv[u].y, h[mt], 0, 0, 0);
      h[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1v[u].z, xv[u].z, h[mt], 0, 0, 0);
      h[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1v[u].w, xv[u].w, h[mt], 0, 0, 0);
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = mt * 16 + l15, gr = row0 + row;
    h[mt][0] += b14.x; h[mt][1] += b14.y; h[mt][2] += b14.z; h[mt][3] += b14.w;
    if (gr < p.rows) *reinterpret_cast<float4*>(p.Hpi + (size_t)gr * 128 + col) = make_float4(h[mt][0], h[mt][1], h[mt][2], h[mt][3]);
    float s = (h[mt][0] + h[mt][1]) + (h[mt][2] + h[mt][3]);
    s = rows_sum(s);
    if (l4 == 0) wpart[(wave * PI_ROWS + row) * 2] = s;
  }
  __syncthreads();
  float mean[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {   // two-pass variance (as torch's LayerNorm): mean first, then centred squares
    const int row = mt * 16 + l15;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) s += wpart[(w * PI_ROWS + row) * 2];
    mean[mt] = s * (1.0f / 128.0f);
    const float d0 = h[mt][0] - mean[mt], d1 = h[mt][1] - mean[mt], d2 = h[mt][2] - mean[mt], d3 = h[mt][3] - mean[mt];
    float q = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    q = rows_sum(q);
    if (l4 == 0) wpart[(wave * PI_ROWS + row) * 2 + 1] = q;
  }
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = mt * 16 + l15;
    float q = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) q += wpart[(w * PI_ROWS + row) * 2 + 1];
    const float rstd = rsqrtf(q * (1.0f / 128.0f) + p.eps);
    float z = fmaxf((h[mt][0] - mean[mt]) * rstd * g4.x + e4.x, 0.f) * w24.x + fmaxf((h[mt][1] - mean[mt]) * rstd * g4.y + e4.y, 0.f) * w24.y +
              fmaxf((h[mt][2] - mean[mt]) * rstd * g4.z + e4.z, 0.f) * w24.z + fmaxf((h[mt][3] - mean[mt]) * rstd * g4.w + e4.w, 0.f) * w24.w;
    z = rows_sum(z);
    if (l4 == 0) zpart[wave * PI_ROWS + row] = z;
  }
  __syncthreads();
  if (tid < PI_ROWS && row0 + tid < p.rows) {
    float z = b2;
#pragma unroll
    for (int w = 0; w < 8; ++w) z += zpart[w * PI_ROWS + tid];
    const int gr = row0 + tid;
    p.prob[gr] = p.r_kpm[gr / p.M] ? -1e6f : z;
  }
}

// Decoder query init (planning_decoder.py:160-165): q0[(b, r, m)] = q_proj(cat[r_emb[b, r], m_emb[m]]) = r_emb Wr^T + b + (m_emb Wm^T)[m];
// the mode half is weight-only (cached), the reference-line half is one K = 128 MFMA contraction for 16 lines per workgroup, expanded
// over the 12 modes on the way out.  Replaces a GEMM launch and the expansion kernel.
struct Q0P { const float* r_emb; int nL, M; const unsigned short* wr; const float* br; const float* Mb; float* Q; };

__global__ __launch_bounds__(256) void q0_fused_kernel(Q0P p) {
  __shared__ __attribute__((aligned(16))) unsigned short xb[16 * 136];
  __shared__ __attribute__((aligned(16))) float ra[16 * 132];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int l0 = blockIdx.x * 16;
  PFrags<4, 2> W;
  p_load_w<4, 4, 2>(W, p.wr, 128, 0, wave, l15, l4);
  {
    const int r = tid >> 4, c8 = (tid & 15) * 8;         // 16 rows x 16 lanes x 8 floats
    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
    if (l0 + r < p.nL) { v0 = *reinterpret_cast<const float4*>(p.r_emb + (size_t)(l0 + r) * 128 + c8); v1 = *reinterpret_cast<const float4*>(p.r_emb + (size_t)(l0 + r) * 128 + c8 + 4); }
    uint4 o; o.x = pack_h2(v0.x, v0.y); o.y = pack_h2(v0.z, v0.w); o.z = pack_h2(v1.x, v1.y); o.w = pack_h2(v1.z, v1.w);
    *reinterpret_cast<uint4*>(xb + r * 136 + c8) = o;
  }
  __syncthreads();
  f32x4 acc[1][2];
  p_zero(acc);
  p_mma<1, 4, 2>(acc, xb, 136, 0, W, l15, l4);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = (j * 4 + wave) * 16 + l4 * 4;
    const float4 b4 = *reinterpret_cast<const float4*>(p.br + col);
    *reinterpret_cast<float4*>(ra + l15 * 132 + col) = make_float4(acc[0][j][0] + b4.x, acc[0][j][1] + b4.y, acc[0][j][2] + b4.z, acc[0][j][3] + b4.w);
  }
  __syncthreads();
  for (int i = tid; i < 16 * p.M * 32; i += 256) {        // (line, mode, 4 channels): 512 B rows, coalesced
    const int c4 = (i & 31) * 4, rm = i >> 5, m = rm % p.M, r = rm / p.M;
    if (l0 + r >= p.nL) continue;
    const float4 a = *reinterpret_cast<const float4*>(ra + r * 132 + c4), mb = *reinterpret_cast<const float4*>(p.Mb + (size_t)m * 128 + c4);
    *reinterpret_cast<float4*>(p.Q + ((size_t)(l0 + r) * p.M + m) * 128 + c4) = make_float4(a.x + mb.x, a.y + mb.y, a.z + mb.z, a.w + mb.w);
  }
}

}  // namespace RIFT_NS
