// PPO critic for gfx950: CriticPPO (rift/gym_carla/utils/net.py:420-431; base :355-371) forward, and the SmoothL1 value loss
// of get_ppo_loss (ppo_trainer.py:175-176) with its analytic backward.  fp32 VALU throughout (the critic sees <= 256 rows
// per step and 4096 per buffer sweep: ~0.2 GFLOP; exactness against autograd matters more than MFMA here).
//   value = (relu(relu(sn W0^T + b0) W1^T + b1) W2^T + b2) * value_std + value_avg,   sn = (state - state_avg) / state_std
// Backward is split so that every sum has a fixed order: a row kernel (16 rows per workgroup) produces the per-row deltas,
// then each parameter gradient is a column / outer-product sum over the rows computed by exactly one thread.
#pragma once
#include "common.h"

namespace RIFT_NS {

#define RIFT_CRITIC_IN 128
#define RIFT_CRITIC_H 256
#define RIFT_CRITIC_NPARAM (RIFT_CRITIC_H * RIFT_CRITIC_IN + RIFT_CRITIC_H + RIFT_CRITIC_H * RIFT_CRITIC_H + RIFT_CRITIC_H + RIFT_CRITIC_H + 1 + 2 * RIFT_CRITIC_IN + 2)
// flat layout: w0 [256][128] | b0 [256] | w1 [256][256] | b1 [256] | w2 [256] | b2 [1] | state_avg [128] | state_std [128] | value_avg [1] | value_std [1]
// (the reference's freeze_parameters makes the four normalisation constants trainable too, ppo_trainer.py:84-96)
#define RIFT_CRITIC_OFF_B0 (RIFT_CRITIC_H * RIFT_CRITIC_IN)
#define RIFT_CRITIC_OFF_W1 (RIFT_CRITIC_OFF_B0 + RIFT_CRITIC_H)
#define RIFT_CRITIC_OFF_B1 (RIFT_CRITIC_OFF_W1 + RIFT_CRITIC_H * RIFT_CRITIC_H)
#define RIFT_CRITIC_OFF_W2 (RIFT_CRITIC_OFF_B1 + RIFT_CRITIC_H)
#define RIFT_CRITIC_OFF_B2 (RIFT_CRITIC_OFF_W2 + RIFT_CRITIC_H)
#define RIFT_CRITIC_OFF_SAVG (RIFT_CRITIC_OFF_B2 + 1)
#define RIFT_CRITIC_OFF_SSTD (RIFT_CRITIC_OFF_SAVG + RIFT_CRITIC_IN)
#define RIFT_CRITIC_OFF_VAVG (RIFT_CRITIC_OFF_SSTD + RIFT_CRITIC_IN)
#define RIFT_CRITIC_OFF_VSTD (RIFT_CRITIC_OFF_VAVG + 1)

struct CriticW { const float *w0, *b0, *w1, *b1, *w2, *b2, *savg, *sstd, *vavg, *vstd; };

struct CriticRowsP {
  CriticW w;
  const float* state;      // (n, 128)
  const float* target;     // (n) reward_sum, or null: forward only
  int n;
  float* value;            // (n) or null
  float *sn, *h1, *h2;     // (n,128) (n,256) (n,256): saved activations (backward only)
  float *dh1, *dh2, *dout; // (n,256) (n,256) (n): d SmoothL1_r / d pre-activation (sum reduction, no 1/n)
  float *gsa, *gss;        // (n,128) (n,128): d SmoothL1_r / d state_avg, d state_std
  float *gva, *gvs;        // (n) (n): d SmoothL1_r / d value_avg, d value_std
  double* sl1_part;        // [gridDim.x] per-workgroup sum of SmoothL1
};

__global__ __launch_bounds__(256) void critic_rows_kernel(CriticRowsP p) {
  constexpr int R = 16, IN = RIFT_CRITIC_IN, H = RIFT_CRITIC_H;
  __shared__ __attribute__((aligned(16))) float s[R][IN];
  __shared__ __attribute__((aligned(16))) float a1[R][H];
  __shared__ __attribute__((aligned(16))) float a2[R][H];
  __shared__ float dv[R], sl[R];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = blockIdx.x * R;
  for (int i = tid; i < R * IN; i += 256) {
    const int r = i / IN, k = i - r * IN;
    float v = 0.f;
    if (r0 + r < p.n) v = (p.state[(size_t)(r0 + r) * IN + k] - p.w.savg[k]) / p.w.sstd[k];
    s[r][k] = v;
    if (p.sn && r0 + r < p.n) p.sn[(size_t)(r0 + r) * IN + k] = v;
  }
  __syncthreads();
  {   // layer 0: thread c owns hidden unit c for the 16 rows
    const int c = tid;
    float acc[R];
    const float b = p.w.b0[c];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = b;
    const float* wr = p.w.w0 + (size_t)c * IN;
    for (int k = 0; k < IN; k += 4) {
      const float4 w4 = *reinterpret_cast<const float4*>(wr + k);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float4 x = *reinterpret_cast<const float4*>(&s[r][k]);
        acc[r] = fmaf(w4.x, x.x, acc[r]); acc[r] = fmaf(w4.y, x.y, acc[r]); acc[r] = fmaf(w4.z, x.z, acc[r]); acc[r] = fmaf(w4.w, x.w, acc[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float v = fmaxf(acc[r], 0.f);
      a1[r][c] = v;
      if (p.h1 && r0 + r < p.n) p.h1[(size_t)(r0 + r) * H + c] = v;
    }
  }
  __syncthreads();
  {   // layer 1
    const int c = tid;
    float acc[R];
    const float b = p.w.b1[c];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = b;
    const float* wr = p.w.w1 + (size_t)c * H;
    for (int k = 0; k < H; k += 4) {
      const float4 w4 = *reinterpret_cast<const float4*>(wr + k);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float4 x = *reinterpret_cast<const float4*>(&a1[r][k]);
        acc[r] = fmaf(w4.x, x.x, acc[r]); acc[r] = fmaf(w4.y, x.y, acc[r]); acc[r] = fmaf(w4.z, x.z, acc[r]); acc[r] = fmaf(w4.w, x.w, acc[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float v = fmaxf(acc[r], 0.f);
      a2[r][c] = v;
      if (p.h2 && r0 + r < p.n) p.h2[(size_t)(r0 + r) * H + c] = v;
    }
  }
  __syncthreads();
  // output layer: wave w owns rows 4w..4w+3
  const float vstd = p.w.vstd[0], vavg = p.w.vavg[0], b2 = p.w.b2[0];
  for (int rr = 0; rr < 4; ++rr) {
    const int r = wave * 4 + rr;
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) t = fmaf(p.w.w2[lane + 64 * j], a2[r][lane + 64 * j], t);
    t = wave_sum(t);
    if (lane == 0) {
      const float v = (t + b2) * vstd + vavg;
      float d = 0.f, l = 0.f;
      if (r0 + r < p.n) {
        if (p.value) p.value[r0 + r] = v;
        if (p.target) {
          const float e = v - p.target[r0 + r];
          const float ae = fabsf(e);
          l = ae < 1.f ? 0.5f * e * e : ae - 0.5f;                  // SmoothL1, beta = 1
          const float dl = ae < 1.f ? e : (e > 0.f ? 1.f : -1.f);   // d l / d value
          d = dl * vstd;                                            // d l / d (net output)
          p.dout[r0 + r] = d;
          p.gva[r0 + r] = dl;                                       // value = net * value_std + value_avg
          p.gvs[r0 + r] = dl * (t + b2);
        }
      }
      dv[r] = d; sl[r] = l;
    }
  }
  if (!p.target) return;
  __syncthreads();
  {   // dh2 = dout * w2 * (a2 > 0): thread c; kept in a2's place for the next step
    const int c = tid;
    const float w2c = p.w.w2[c];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float v = a2[r][c] > 0.f ? dv[r] * w2c : 0.f;
      a2[r][c] = v;
      if (r0 + r < p.n) p.dh2[(size_t)(r0 + r) * H + c] = v;
    }
  }
  __syncthreads();
  {   // dh1[r][k] = (a1 > 0) * sum_c dh2[r][c] * w1[c][k]: thread k, w1 column reads are coalesced across the workgroup
    const int k = tid;
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
    for (int c = 0; c < H; ++c) {
      const float w = p.w.w1[(size_t)c * H + k];
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = fmaf(a2[r][c], w, acc[r]);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float v = a1[r][k] > 0.f ? acc[r] : 0.f;
      a1[r][k] = v;                                   // kept for the input-normalisation gradients below
      if (r0 + r < p.n) p.dh1[(size_t)(r0 + r) * H + k] = v;
    }
  }
  __syncthreads();
  if (tid < IN) {   // d l / d sn[r][k] = sum_c dh1[r][c] w0[c][k];  sn = (state - avg) / std
    const int k = tid;
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
    for (int c = 0; c < H; ++c) {
      const float w = p.w.w0[(size_t)c * IN + k];
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = fmaf(a1[r][c], w, acc[r]);
    }
    const float inv = 1.0f / p.w.sstd[k];
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (r0 + r < p.n) {
        p.gsa[(size_t)(r0 + r) * IN + k] = -acc[r] * inv;
        p.gss[(size_t)(r0 + r) * IN + k] = -acc[r] * s[r][k] * inv;
      }
  }
  if (tid == 0) {
    double t = 0.0;
    for (int r = 0; r < R; ++r) t += (double)sl[r];
    p.sl1_part[blockIdx.x] = t;
  }
}

// out[c * Cx + k] = scale * sum_r D[r * Cd + c] * X[r * Cx + k]   (one thread per output, rows in order)
__global__ void critic_outer_sum_kernel(const float* __restrict__ D, int Cd, const float* __restrict__ X, int Cx, int n, float scale,
                                        float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Cd * Cx) return;
  const int c = i / Cx, k = i - c * Cx;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int r = 0;
  for (; r + 4 <= n; r += 4) {
    a0 = fmaf(D[(size_t)r * Cd + c], X[(size_t)r * Cx + k], a0);
    a1 = fmaf(D[(size_t)(r + 1) * Cd + c], X[(size_t)(r + 1) * Cx + k], a1);
    a2 = fmaf(D[(size_t)(r + 2) * Cd + c], X[(size_t)(r + 2) * Cx + k], a2);
    a3 = fmaf(D[(size_t)(r + 3) * Cd + c], X[(size_t)(r + 3) * Cx + k], a3);
  }
  for (; r < n; ++r) a0 = fmaf(D[(size_t)r * Cd + c], X[(size_t)r * Cx + k], a0);
  out[i] = scale * ((a0 + a1) + (a2 + a3));
}

// out[c] = scale * sum_r D[r * Cd + c]
__global__ void critic_col_sum_kernel(const float* __restrict__ D, int Cd, int n, float scale, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Cd) return;
  float a = 0.f;
  for (int r = 0; r < n; ++r) a += D[(size_t)r * Cd + c];
  out[c] = scale * a;
}

// stats[0] -= sum of the per-workgroup SmoothL1 sums (the objective convention of loss.h: loss = -stats[0] / stats[1])
__global__ void critic_stats_kernel(const double* __restrict__ part, int nwg, double* __restrict__ stats) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < nwg; ++i) t += part[i];
    stats[0] -= t;
  }
}

// grads = -flat / count scattered into the six caller-owned .grad tensors
__global__ void critic_finalize_kernel(const float* __restrict__ flat, const double* __restrict__ stats, float* g_w0, float* g_b0,
                                       float* g_w1, float* g_b1, float* g_w2, float* g_b2, float* g_savg, float* g_sstd, float* g_vavg,
                                       float* g_vstd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= RIFT_CRITIC_NPARAM) return;
  const double cnt = stats[1];
  const float sc = cnt > 0.0 ? (float)(-1.0 / cnt) : 0.f;
  float* dst; int o;
  if (i < RIFT_CRITIC_OFF_B0) { dst = g_w0; o = i; }
  else if (i < RIFT_CRITIC_OFF_W1) { dst = g_b0; o = i - RIFT_CRITIC_OFF_B0; }
  else if (i < RIFT_CRITIC_OFF_B1) { dst = g_w1; o = i - RIFT_CRITIC_OFF_W1; }
  else if (i < RIFT_CRITIC_OFF_W2) { dst = g_b1; o = i - RIFT_CRITIC_OFF_B1; }
  else if (i < RIFT_CRITIC_OFF_B2) { dst = g_w2; o = i - RIFT_CRITIC_OFF_W2; }
  else if (i < RIFT_CRITIC_OFF_SAVG) { dst = g_b2; o = 0; }
  else if (i < RIFT_CRITIC_OFF_SSTD) { dst = g_savg; o = i - RIFT_CRITIC_OFF_SAVG; }
  else if (i < RIFT_CRITIC_OFF_VAVG) { dst = g_sstd; o = i - RIFT_CRITIC_OFF_SSTD; }
  else if (i < RIFT_CRITIC_OFF_VSTD) { dst = g_vavg; o = 0; }
  else { dst = g_vstd; o = 0; }
  if (dst) dst[o] = flat[i] * sc;
}

}  // namespace RIFT_NS
