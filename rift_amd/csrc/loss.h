// pi-head tail, RLFT objectives (RIFT / GRPO / PPO-actor / REINFORCE) and the analytic
// backward into the only trainable tensors (planning_decoder.pi_head.*) for gfx950.
// Row softmax / LayerNorm reductions are wavefront shuffles; objective sums are fp64
// (the reference promotes fp64 advantage x fp32 ratio, rift_trainer.py:160-178).
#pragma once
#include "common.h"

namespace RIFT_NS {

enum { LOSS_RIFT = 0, LOSS_GRPO = 1, LOSS_PPO = 2, LOSS_REINFORCE = 3, LOSS_SFT = 4 };

// pi_head tail (mlp_layer.py:8-13 after the first Linear): per row  LN(128) -> ReLU -> dot(w2) + b2.
// One wave per row.  Writes raw logits and the -1e6-masked `probability` (pluto_model.py:203).
__global__ void pi_tail_kernel(const float* __restrict__ Hpi /*[rows][128]*/, int rows, int M,
                               const float* __restrict__ g, const float* __restrict__ be,
                               const float* __restrict__ w2, const float* __restrict__ b2,
                               const uint8_t* __restrict__ r_kpm /*[rows/M]*/, float eps,
                               float* __restrict__ prob /*[rows]*/, int* __restrict__ nonfinite) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float h0 = Hpi[(size_t)row * 128 + lane], h1 = Hpi[(size_t)row * 128 + 64 + lane];
  const float mean = wave_sum(h0 + h1) * (1.f / 128.f);
  const float d0 = h0 - mean, d1 = h1 - mean;
  const float rstd = rsqrtf(wave_sum(d0 * d0 + d1 * d1) * (1.f / 128.f) + eps);
  const float a0 = fmaxf(d0 * rstd * g[lane] + be[lane], 0.f);
  const float a1 = fmaxf(d1 * rstd * g[lane + 64] + be[lane + 64], 0.f);
  const float z = wave_sum(a0 * w2[lane] + a1 * w2[lane + 64]) + b2[0];
  // a NaN / Inf in the decoder queries (the reference's assert, planning_decoder.py:175) reaches this hidden row; test the row itself
  // (x * 0 is NaN for NaN and +-Inf) -- the ReLU's fmaxf would swallow a NaN before it reached the logit
  const float bad = wave_sum(h0 * 0.f + h1 * 0.f);
  if (lane == 0) {
    prob[row] = r_kpm[row / M] ? -1e6f : z;
    if (nonfinite && bad != 0.f) atomicOr(nonfinite, 1);
  }
}

struct LossP {
  int kind, bs, G, M;                 // G = R*M candidates per scene
  const float* prob;                  // [bs][G]  model output (already -1e6 on padded ref lines)
  const uint8_t* r_kpm;               // [bs][R]
  const float* old_logits;            // [bs][G]  (RIFT, GRPO)
  const float* ref_logits;            // [bs][G]  (GRPO)
  const double* adv64;                // [bs][G]  (RIFT, GRPO) fp64 group advantage
  const uint8_t* valid;               // [bs][G]  (RIFT, GRPO)
  const long long* action_mode;       // [bs][2]  (PPO) int64 (r_idx, m_idx)
  const float* scal_a;                // [bs]     PPO advantage / REINFORCE return
  const float* old_log_prob;          // [bs]     (PPO)
  float clip_eps, lambda_entropy;
  double* S;                          // [bs] per-scene objective sum
  double* cnt;                        // [bs] per-scene normaliser count
  float* dlogit;                      // [bs][G]  dS/dlogit
  long long* argmax_rm;               // [bs][2]  (REINFORCE) chosen (r, m), bit-exact integer output
};

// One 64-lane wave per scene (G <= 64*MAXPL candidates held in registers).
template <int MAXPL>
__global__ void loss_kernel(LossP p) {
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (b >= p.bs) return;
  const int G = p.G;
  float z[MAXPL], lp[MAXPL];
  bool pad[MAXPL];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < MAXPL; ++i) {
    const int j = lane + i * 64;
    pad[i] = true; z[i] = -INFINITY;
    if (j < G) {
      pad[i] = p.r_kpm[(size_t)b * (G / p.M) + j / p.M] != 0;
      z[i] = pad[i] ? -1e8f : p.prob[(size_t)b * G + j];     // masked_fill_(-1e8), rift_trainer.py:153
      mx = fmaxf(mx, z[i]);
    }
  }
  mx = wave_max(mx);
  float se = 0.f;
#pragma unroll
  for (int i = 0; i < MAXPL; ++i) if (lane + i * 64 < G) se += expf(z[i] - mx);
  const float lse = mx + logf(wave_sum(se));
#pragma unroll
  for (int i = 0; i < MAXPL; ++i) lp[i] = z[i] - lse;

  double S = 0.0, cnt = 0.0;
  float gi[MAXPL];   // dS/dlp_i
#pragma unroll
  for (int i = 0; i < MAXPL; ++i) gi[i] = 0.f;

  if (p.kind == LOSS_RIFT || p.kind == LOSS_GRPO) {
    // old-policy log-softmax
    float zo[MAXPL], zr[MAXPL];
    float mo = -INFINITY, mr = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAXPL; ++i) {
      const int j = lane + i * 64;
      zo[i] = -INFINITY; zr[i] = -INFINITY;
      if (j < G) {
        zo[i] = pad[i] ? -1e8f : p.old_logits[(size_t)b * G + j];
        mo = fmaxf(mo, zo[i]);
        if (p.kind == LOSS_GRPO) { zr[i] = pad[i] ? -1e8f : p.ref_logits[(size_t)b * G + j]; mr = fmaxf(mr, zr[i]); }
      }
    }
    mo = wave_max(mo);
    float so = 0.f, sr = 0.f;
#pragma unroll
    for (int i = 0; i < MAXPL; ++i) if (lane + i * 64 < G) so += expf(zo[i] - mo);
    const float lseo = mo + logf(wave_sum(so));
    float lser = 0.f;
    if (p.kind == LOSS_GRPO) {
      mr = wave_max(mr);
#pragma unroll
      for (int i = 0; i < MAXPL; ++i) if (lane + i * 64 < G) sr += expf(zr[i] - mr);
      lser = mr + logf(wave_sum(sr));
    }
#pragma unroll
    for (int i = 0; i < MAXPL; ++i) {
      const int j = lane + i * 64;
      if (j >= G || !p.valid[(size_t)b * G + j]) continue;
      const double A = p.adv64[(size_t)b * G + j];
      const float ratio = expf(lp[i] - (zo[i] - lseo));
      const double u = A * (double)ratio;
      const double c = A * (double)fminf(fmaxf(ratio, 0.8f), 1.2f);
      double obj = u < c ? u : c;
      // d min(u,c)/d lp: u is active (or tied with c inside the clamp range) -> A*ratio
      bool pass = (ratio >= 0.8f && ratio <= 1.2f) || (u < c);
      if (p.kind == LOSS_RIFT) {
        if (A < 0.0) {                                   // dual clip, rift_trainer.py:168-170
          const double lo = A * 3.0;
          if (obj < lo) { obj = lo; pass = false; }
        }
      } else {
        const float rp = expf(zr[i] - lser);             // softmax(ref logits)
        // F.kl_div(log p, ref_p) = ref_p * (log ref_p - log p); xlogy(0, 0) = 0
        const double kl = rp > 0.f ? (double)(rp * (logf(rp) - lp[i])) : 0.0;
        obj -= 0.2 * kl;
        gi[i] += 0.2f * rp;
      }
      if (pass) gi[i] += (float)(A * (double)ratio);
      S += obj; cnt += 1.0;
    }
  } else if (p.kind == LOSS_PPO) {
    const int ridx = (int)p.action_mode[(size_t)b * 2], midx = (int)p.action_mode[(size_t)b * 2 + 1];
    const int chosen = ridx * p.M + midx;
    float ent_part = 0.f, cur = 0.f;
#pragma unroll
    for (int i = 0; i < MAXPL; ++i) {
      const int j = lane + i * 64;
      if (j < G) { ent_part += expf(lp[i]) * lp[i]; if (j == chosen) cur = lp[i]; }
    }
    const float ent = -wave_sum(ent_part);
    cur = wave_sum(cur);
    const float A = p.scal_a[b];
    const float ratio = expf(cur - p.old_log_prob[b]);
    const float l1 = A * ratio, l2 = A * fminf(fmaxf(ratio, 1.f - p.clip_eps), 1.f + p.clip_eps);
    const bool pass = (ratio >= 1.f - p.clip_eps && ratio <= 1.f + p.clip_eps) || (l1 < l2);
    const float gsel = pass ? A * ratio : 0.f;
    S = (double)(l1 < l2 ? l1 : l2) + (double)p.lambda_entropy * (double)ent;
    cnt = 1.0;
#pragma unroll
    for (int i = 0; i < MAXPL; ++i) {
      const int j = lane + i * 64;
      if (j >= G) continue;
      // dH/dlp_j routed through dlogit below: dH/dz_j = -p_j (lp_j + H); expressed as gi on lp: g_j = -(p_j (lp_j + 1))
      gi[i] = -p.lambda_entropy * expf(lp[i]) * (lp[i] + 1.f) + (j == chosen ? gsel : 0.f);
    }
  } else {   // REINFORCE, reinforce_trainer.py:125-170; SFT shares the argmax
    // argmax over the masked logits, first index on ties (torch.argmax)
    float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < MAXPL; ++i) {
      const int j = lane + i * 64;
      if (j < G && (z[i] > bv)) { bv = z[i]; bi = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    // SFT (sft_trainer.py:123-184): the target keeps the policy's own best reference line and takes the MODE of the teacher
    // (generate_target_label :193-199); cross entropy against that one-hot label = -log p[target], mean over the batch
    float ret = 1.0f;
    if (p.kind == LOSS_SFT) bi = (bi / p.M) * p.M + (int)p.action_mode[(size_t)b * 2 + 1];
    else ret = p.scal_a[b];
    float cur = 0.f;
#pragma unroll
    for (int i = 0; i < MAXPL; ++i) if (lane + i * 64 == bi) cur = lp[i];
    cur = wave_sum(cur);
    S = (double)(cur * ret); cnt = 1.0;
#pragma unroll
    for (int i = 0; i < MAXPL; ++i) if (lane + i * 64 == bi) gi[i] = ret;
    if (lane == 0 && p.argmax_rm) { p.argmax_rm[(size_t)b * 2] = bi / p.M; p.argmax_rm[(size_t)b * 2 + 1] = bi % p.M; }
  }

  // dS/dz_j = g_j - p_j * sum_i g_i   (lp_i = z_i - lse)
  float gs = 0.f;
#pragma unroll
  for (int i = 0; i < MAXPL; ++i) gs += gi[i];
  gs = wave_sum(gs);
#pragma unroll
  for (int i = 0; i < MAXPL; ++i) {
    const int j = lane + i * 64;
    if (j < G) p.dlogit[(size_t)b * G + j] = pad[i] ? 0.f : gi[i] - expf(lp[i]) * gs;
  }
  if (p.kind == LOSS_RIFT || p.kind == LOSS_GRPO) { S = wave_sum_d(S); cnt = wave_sum_d(cnt); }
  if (lane == 0) { p.S[b] = S; p.cnt[b] = cnt; }
}

// SFT teacher label, the mode index (sft_trainer.py:186-199 with sft/utils.py:10-32 and pid_controller.py:108-125): every candidate's
// trajectory is sub-sampled every `fr` frames, moved to the teacher's local frame (fp32, as the reference's einsum does), its target
// speed is the mean distance between consecutive sub-sampled points, and the label takes the mode index of the candidate -- padded
// reference lines included, as in the reference -- whose target speed is closest to the teacher's (first minimum).  One wave per scene.
__global__ __launch_bounds__(64) void sft_teacher_mode_kernel(const float* __restrict__ traj /*(bs,G,T,6)*/, const float* __restrict__ teacher /*(bs,5)*/,
                                                              int bs, int G, int M, int T, int fr, long long* __restrict__ mode_out /*(bs,2)*/) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= bs) return;
  const float tsp = teacher[b * 5], ox = teacher[b * 5 + 1], oy = teacher[b * 5 + 2], hd = teacher[b * 5 + 3];
  const float c = cosf(hd), s = sinf(hd);
  const int np_ = T < fr ? 1 : T / fr;
  float best = INFINITY; int bi = 0x7fffffff;
  for (int g = lane; g < G; g += 64) {
    const float* q = traj + ((size_t)b * G + g) * T * 6;
    float px = 0.f, py = 0.f, acc = 0.f, speed;
    for (int k = 0; k < np_; ++k) {
      const int t = T < fr ? T - 1 : fr - 1 + k * fr;
      const float dx = q[t * 6] - ox, dy = q[t * 6 + 1] - oy;
      const float lx = dx * c + dy * s, ly = dx * (-s) + dy * c;          // (p - origin) . [[cos, -sin], [sin, cos]]
      if (k > 0) acc += sqrtf((lx - px) * (lx - px) + (ly - py) * (ly - py));
      px = lx; py = ly;
    }
    speed = np_ == 1 ? sqrtf(px * px + py * py) : acc / (float)(np_ - 1);
    const float d = fabsf(speed - tsp);
    if (d < best) { best = d; bi = g; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
    if (ov < best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) { mode_out[(size_t)b * 2] = bi / M; mode_out[(size_t)b * 2 + 1] = bi % M; }
}

// Backward through pi_head for a chunk of rows.  partial layout per workgroup (floats):
//   [0,16384) dW1[c][k] | +128 db1 | +128 dgamma | +128 dbeta | +128 dw2 | +1 db2      (= 16897)
#define RIFT_PI_NPARAM 16897
#define RIFT_PI_BWD_ROWS 128
// Rows are handled 16 lanes per row (8 channels per lane): the LayerNorm-backward reductions are DPP row sums, four
// rows per wave in flight; dW1 += dh^T q accumulates an 8 x 8 register block per thread from LDS-staged 32-row slabs.
__global__ __launch_bounds__(256) void pi_backward_kernel(
    const float* __restrict__ Q /*[rows][128] pi_head input*/, const float* __restrict__ Hpi /*[rows][128]*/,
    const float* __restrict__ dz /*[rows]*/, int rows, const float* __restrict__ g, const float* __restrict__ be,
    const float* __restrict__ w2, float eps, float* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) float s_dh[32][132];
  __shared__ __attribute__((aligned(16))) float s_q[32][132];
  __shared__ float s_red[16][4][128];
  __shared__ float s_db2[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, rsub = lane >> 4;
  const int r0 = blockIdx.x * RIFT_PI_BWD_ROWS;
  const int r1 = min(rows, r0 + RIFT_PI_BWD_ROWS);
  float accW[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) accW[i][j] = 0.f;
  float a_db1[8], a_dg[8], a_dbe[8], a_dw2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a_db1[i] = 0.f; a_dg[i] = 0.f; a_dbe[i] = 0.f; a_dw2[i] = 0.f; }
  float a_db2 = 0.f;
  const int c0 = (tid >> 4) * 8, k0 = (tid & 15) * 8;
  float gv[8], bv[8], wv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { gv[i] = g[l15 * 8 + i]; bv[i] = be[l15 * 8 + i]; wv[i] = w2[l15 * 8 + i]; }
  for (int base = r0; base < r1; base += 32) {
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const int rr = st * 16 + wave * 4 + rsub;
      const int row = base + rr;
      float dh[8], q[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { dh[i] = 0.f; q[i] = 0.f; }
      float hsum = 0.f, h[8], d = 0.f;
      const bool live = row < r1;
      if (live) {
        const float4 h0 = *reinterpret_cast<const float4*>(Hpi + (size_t)row * 128 + l15 * 8), h1 = *reinterpret_cast<const float4*>(Hpi + (size_t)row * 128 + l15 * 8 + 4);
        const float4 q0 = *reinterpret_cast<const float4*>(Q + (size_t)row * 128 + l15 * 8), q1 = *reinterpret_cast<const float4*>(Q + (size_t)row * 128 + l15 * 8 + 4);
        h[0] = h0.x; h[1] = h0.y; h[2] = h0.z; h[3] = h0.w; h[4] = h1.x; h[5] = h1.y; h[6] = h1.z; h[7] = h1.w;
        q[0] = q0.x; q[1] = q0.y; q[2] = q0.z; q[3] = q0.w; q[4] = q1.x; q[5] = q1.y; q[6] = q1.z; q[7] = q1.w;
        d = dz[row];
#pragma unroll
        for (int i = 0; i < 8; ++i) hsum += h[i];
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) h[i] = 0.f;
      }
      // the reductions run for every lane (DPP needs full rows of 16 lanes); dead rows contribute zeros
      const float mean = sum16(hsum) * (1.f / 128.f);
      float e[8], vs = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) { e[i] = h[i] - mean; vs += e[i] * e[i]; }
      const float rstd = rsqrtf(sum16(vs) * (1.f / 128.f) + eps);
      float n[8], dn[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        n[i] = e[i] * rstd;
        const float y = n[i] * gv[i] + bv[i];
        const float dy = (live && y > 0.f) ? d * wv[i] : 0.f;
        if (live) { a_dw2[i] += d * fmaxf(y, 0.f); a_dg[i] += dy * n[i]; a_dbe[i] += dy; }
        dn[i] = dy * gv[i];
        s1 += dn[i]; s2 += dn[i] * n[i];
      }
      if (live && l15 == 0) a_db2 += d;
      const float m1 = sum16(s1) * (1.f / 128.f), m2 = sum16(s2) * (1.f / 128.f);
      if (live) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { dh[i] = rstd * (dn[i] - m1 - n[i] * m2); a_db1[i] += dh[i]; }
      }
      *reinterpret_cast<float4*>(&s_dh[rr][l15 * 8]) = make_float4(dh[0], dh[1], dh[2], dh[3]);
      *reinterpret_cast<float4*>(&s_dh[rr][l15 * 8 + 4]) = make_float4(dh[4], dh[5], dh[6], dh[7]);
      *reinterpret_cast<float4*>(&s_q[rr][l15 * 8]) = make_float4(q[0], q[1], q[2], q[3]);
      *reinterpret_cast<float4*>(&s_q[rr][l15 * 8 + 4]) = make_float4(q[4], q[5], q[6], q[7]);
    }
    __syncthreads();
    for (int rr = 0; rr < 32; ++rr) {
      const float4 d0 = *reinterpret_cast<const float4*>(&s_dh[rr][c0]), d1 = *reinterpret_cast<const float4*>(&s_dh[rr][c0 + 4]);
      const float4 q0 = *reinterpret_cast<const float4*>(&s_q[rr][k0]), q1 = *reinterpret_cast<const float4*>(&s_q[rr][k0 + 4]);
      const float dv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w}, qv[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) accW[i][j] += dv[i] * qv[j];
    }
    __syncthreads();
  }
  float* out = partial + (size_t)blockIdx.x * RIFT_PI_NPARAM;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) out[(c0 + i) * 128 + k0 + j] = accW[i][j];   // (the 16897-float stride is not 16-byte aligned)
  // reduction of the per-channel vectors over the 16 (wave, row-slot) groups
  const int grp = wave * 4 + rsub;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    s_red[grp][0][l15 * 8 + i] = a_db1[i]; s_red[grp][1][l15 * 8 + i] = a_dg[i];
    s_red[grp][2][l15 * 8 + i] = a_dbe[i]; s_red[grp][3][l15 * 8 + i] = a_dw2[i];
  }
  if (l15 == 0) s_db2[grp] = a_db2;
  __syncthreads();
  for (int i = tid; i < 512; i += 256) {
    const int v = i >> 7, c = i & 127;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += s_red[k][v][c];
    out[16384 + v * 128 + c] = t;
  }
  if (tid == 0) {
    float t = 0.f;
    for (int k = 0; k < 16; ++k) t += s_db2[k];
    out[16384 + 512] = t;
  }
}

// flat[i] = sum_wg partial[wg][i]; stats[0] = sum S_b, stats[1] = sum cnt_b  (deterministic order)
__global__ void loss_reduce_kernel(const float* __restrict__ partial, int nwg, float* __restrict__ flat,
                                   const double* __restrict__ S, const double* __restrict__ cnt, int bs,
                                   double* __restrict__ stats, double* __restrict__ xchg /*optional [NPARAM + 2]*/) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < RIFT_PI_NPARAM) {
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // eight loads in flight per thread; fixed summation order
    int w = 0;
    for (; w + 8 <= nwg; w += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) s[u] += partial[(size_t)(w + u) * RIFT_PI_NPARAM + i];
    }
    for (; w < nwg; ++w) s[0] += partial[(size_t)w * RIFT_PI_NPARAM + i];
    const float tot = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    flat[i] = tot;
    if (xchg) xchg[i] = (double)tot;
  }
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    double a = 0.0, c = 0.0;
    for (int b = threadIdx.x; b < bs; b += 64) { a += S[b]; c += cnt[b]; }
    a = wave_sum_d(a); c = wave_sum_d(c);
    if (threadIdx.x == 0) { stats[0] = a; stats[1] = c; if (xchg) { xchg[RIFT_PI_NPARAM] = a; xchg[RIFT_PI_NPARAM + 1] = c; } }
  }
}

// loss = -S/cnt ; grads = -flat/cnt scattered into the six caller-owned .grad tensors (accumulate or overwrite)
__global__ void loss_finalize_kernel(const float* __restrict__ flat, const double* __restrict__ stats_in,
                                     float* gW1, float* gb1, float* gg, float* gbe, float* gw2, float* gb2,
                                     double* __restrict__ loss_out, int accumulate, const double* __restrict__ xchg,
                                     double* __restrict__ stats_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const double* stats = xchg ? xchg + RIFT_PI_NPARAM : stats_in;     // the (all-reduced) exchange buffer wins when present
  if (xchg && i == 0 && stats_out) { stats_out[0] = stats[0]; stats_out[1] = stats[1]; }
  const double cnt = stats[1];
  const float sc = cnt > 0.0 ? (float)(-1.0 / cnt) : 0.f;
  if (i == 0 && loss_out) loss_out[0] = cnt > 0.0 ? -stats[0] / cnt : 0.0;
  if (i >= RIFT_PI_NPARAM) return;
  float* dst; int o;
  if (i < 16384) { dst = gW1; o = i; }
  else if (i < 16384 + 128) { dst = gb1; o = i - 16384; }
  else if (i < 16384 + 256) { dst = gg; o = i - 16384 - 128; }
  else if (i < 16384 + 384) { dst = gbe; o = i - 16384 - 256; }
  else if (i < 16384 + 512) { dst = gw2; o = i - 16384 - 384; }
  else { dst = gb2; o = 0; }
  if (!dst) return;
  const float v = (xchg ? (float)xchg[i] : flat[i]) * sc;
  dst[o] = accumulate ? dst[o] + v : v;
}

// loss_finalize + clip_grad_norm_ over exactly the six pi_head gradients in ONE single-workgroup launch (the update's tail is a
// chain of latency-bound launches; these were three).  Same arithmetic: grads = -sum/count (+ existing .grad when accumulating),
// total = sqrt(sum of squares in f64), grads *= min(1, max_norm / (total + 1e-6)).
__global__ __launch_bounds__(1024) void loss_finalize_clip_kernel(const float* __restrict__ flat, const double* __restrict__ stats_in,
                                                                  float* gW1, float* gb1, float* gg, float* gbe, float* gw2, float* gb2,
                                                                  double* __restrict__ loss_out, int accumulate, const double* __restrict__ xchg,
                                                                  double* __restrict__ stats_out, float max_norm, float* __restrict__ total_norm) {
  constexpr int NU = (RIFT_PI_NPARAM + 1023) / 1024;
  const int tid = threadIdx.x;
  const double* stats = xchg ? xchg + RIFT_PI_NPARAM : stats_in;
  const double s0 = stats[0], cnt = stats[1];
  const float sc = cnt > 0.0 ? (float)(-1.0 / cnt) : 0.f;
  float v[NU]; float* dp[NU];
  double ss = 0.0;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int i = tid + u * 1024;
    float* dst = nullptr; int o = 0;
    if (i < 16384) { dst = gW1; o = i; }
    else if (i < 16384 + 128) { dst = gb1; o = i - 16384; }
    else if (i < 16384 + 256) { dst = gg; o = i - 16384 - 128; }
    else if (i < 16384 + 384) { dst = gbe; o = i - 16384 - 256; }
    else if (i < 16384 + 512) { dst = gw2; o = i - 16384 - 384; }
    else if (i < RIFT_PI_NPARAM) { dst = gb2; o = 0; }
    dp[u] = dst ? dst + o : nullptr;
    v[u] = 0.f;
    if (dp[u]) {
      v[u] = (xchg ? (float)xchg[i] : flat[i]) * sc;
      if (accumulate) v[u] += *dp[u];
      ss += (double)v[u] * (double)v[u];
    }
  }
  ss = wave_sum_d(ss);
  __shared__ double ws[16];
  if ((tid & 63) == 0) ws[tid >> 6] = ss;
  __syncthreads();
  double tot = 0.0;
#pragma unroll
  for (int w = 0; w < 16; ++w) tot += ws[w];                 // every thread: same order, same value
  const float total = (float)sqrt(tot);
  const float coef = fminf(max_norm / (total + 1e-6f), 1.0f);
  if (tid == 0) {
    if (total_norm) *total_norm = total;
    if (loss_out) loss_out[0] = cnt > 0.0 ? -s0 / cnt : 0.0;
    if (xchg && stats_out) { stats_out[0] = s0; stats_out[1] = cnt; }
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) if (dp[u]) *dp[u] = v[u] * coef;
}

// ---- gradient-norm clipping over a list of tensors (torch.nn.utils.clip_grad_norm_, norm_type 2, as Lightning's
// gradient_clip_val applies it, custom_lightning.yaml:40-41): total = sqrt(sum_t ||g_t||^2), g *= min(1, max_norm / (total + 1e-6))
struct ClipList { float* g[16]; long long n[16]; int count; };

__global__ __launch_bounds__(256) void clip_norm_partial_kernel(ClipList L, double* __restrict__ part /*[gridDim.x]*/) {
  double s = 0.0;
  for (int t = 0; t < L.count; ++t)
    for (long long i = blockIdx.x * 256 + threadIdx.x; i < L.n[t]; i += (long long)gridDim.x * 256) { const double v = L.g[t][i]; s += v * v; }
  s = wave_sum_d(s);
  __shared__ double ws[4];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

__global__ __launch_bounds__(256) void clip_scale_kernel(ClipList L, const double* __restrict__ part, int nparts, float max_norm,
                                                         float* __restrict__ total_norm) {
  double s = 0.0;
  for (int i = 0; i < nparts; ++i) s += part[i];          // every thread: same order, same value
  const float total = (float)sqrt(s);
  if (blockIdx.x == 0 && threadIdx.x == 0 && total_norm) *total_norm = total;
  float coef = max_norm / (total + 1e-6f);
  if (coef >= 1.0f) return;
  for (int t = 0; t < L.count; ++t)
    for (long long i = blockIdx.x * 256 + threadIdx.x; i < L.n[t]; i += (long long)gridDim.x * 256) L.g[t][i] *= coef;
}

// ---- AdamW over a short list of tensors (torch.optim.AdamW, amsgrad = False, maximize = False; the update of the reference's
// configure_optimizers, rift_trainer.py:279-362) on torch's own state tensors, ONE launch for every parameter group: the torch fast
// path costs four launches (two groups x (_foreach_add_ on the step counters + _fused_adamw_)) for 16,897 parameters.
//   p *= 1 - lr wd;  m = lerp(m, g, 1 - b1);  v = b2 v + (1 - b2) g^2;  p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
struct AdamList {
  float* p[16]; const float* g[16]; float* m[16]; float* v[16]; float* step[16];
  long long n[16]; float step_size[16], decay[16], step_new[16];     // lr / (1 - b1^t) and 1 - lr wd, formed in double on the host
  int count; float beta1, beta2, omb1, omb2, bc2_sqrt, eps;           // omb = 1 - beta (double subtraction, then rounded, as torch does)
};

// one element's update -- ONE definition for both kernels that apply it (adamw_kernel, update_tail_kernel): which products the compiler
// contracts into FMAs is decided per expression tree, and the two have to agree to the last bit
__device__ __forceinline__ void adamw_elem(const float g, float& p, float& m, float& v, const float step_size, const float decay,
                                           const float omb1, const float beta2, const float omb2, const float bc2_sqrt, const float eps) {
  float pv = p * decay;
  const float mn = m + (g - m) * omb1;
  const float vn = v * beta2 + omb2 * g * g;
  const float denom = sqrtf(vn) / bc2_sqrt + eps;
  pv -= step_size * (mn / denom);
  p = pv; m = mn; v = vn;
}

__global__ __launch_bounds__(256) void adamw_kernel(AdamList L) {
  for (int t = 0; t < L.count; ++t) {
    const float step_size = L.step_size[t], decay = L.decay[t];
    if (blockIdx.x == 0 && threadIdx.x == 0) L.step[t][0] = L.step_new[t];
    for (long long i = blockIdx.x * 256 + threadIdx.x; i < L.n[t]; i += (long long)gridDim.x * 256) {
      float pv = L.p[t][i], m = L.m[t][i], v = L.v[t][i];
      adamw_elem(L.g[t][i], pv, m, v, step_size, decay, L.omb1, L.beta2, L.omb2, L.bc2_sqrt, L.eps);
      L.p[t][i] = pv; L.m[t][i] = m; L.v[t][i] = v;
    }
  }
}

// ---- finalize + clip + AdamW of the six pi_head tensors in ONE single-workgroup launch (rift_update_tail).  The first half is
// loss_finalize_clip_kernel line for line (same thread <-> element mapping, same f64 reduction order), the second half adamw_kernel's
// arithmetic on the element the thread holds: bit-identical to the two launches.
struct TailAdam {            // per gradient segment, in the flat order W1 | b1 | ln_w | ln_b | w2 | b2
  float* p[6]; float* m[6]; float* v[6]; float* step[6];
  float step_size[6], decay[6], step_new;
  float beta2, omb1, omb2, bc2_sqrt, eps;
};
__global__ __launch_bounds__(1024) void update_tail_kernel(const float* __restrict__ flat, const double* __restrict__ stats_in,
                                                           float* gW1, float* gb1, float* gg, float* gbe, float* gw2, float* gb2,
                                                           double* __restrict__ loss_out, int accumulate, const double* __restrict__ xchg,
                                                           double* __restrict__ stats_out, float max_norm, float* __restrict__ total_norm, TailAdam A) {
  constexpr int NU = (RIFT_PI_NPARAM + 1023) / 1024;
  const int tid = threadIdx.x;
  const double* stats = xchg ? xchg + RIFT_PI_NPARAM : stats_in;
  const double s0 = stats[0], cnt = stats[1];
  const float sc = cnt > 0.0 ? (float)(-1.0 / cnt) : 0.f;
  float v[NU]; float* dp[NU]; int seg[NU], off[NU];
  double ss = 0.0;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int i = tid + u * 1024;
    float* dst = nullptr; int o = 0, sg = 0;
    if (i < 16384) { dst = gW1; o = i; sg = 0; }
    else if (i < 16384 + 128) { dst = gb1; o = i - 16384; sg = 1; }
    else if (i < 16384 + 256) { dst = gg; o = i - 16384 - 128; sg = 2; }
    else if (i < 16384 + 384) { dst = gbe; o = i - 16384 - 256; sg = 3; }
    else if (i < 16384 + 512) { dst = gw2; o = i - 16384 - 384; sg = 4; }
    else if (i < RIFT_PI_NPARAM) { dst = gb2; o = 0; sg = 5; }
    dp[u] = dst ? dst + o : nullptr; seg[u] = sg; off[u] = o;
    v[u] = 0.f;
    if (dp[u]) {
      v[u] = (xchg ? (float)xchg[i] : flat[i]) * sc;
      if (accumulate) v[u] += *dp[u];
      ss += (double)v[u] * (double)v[u];
    }
  }
  ss = wave_sum_d(ss);
  __shared__ double ws[16];
  if ((tid & 63) == 0) ws[tid >> 6] = ss;
  __syncthreads();
  double tot = 0.0;
#pragma unroll
  for (int w = 0; w < 16; ++w) tot += ws[w];
  const float total = (float)sqrt(tot);
  const float coef = fminf(max_norm / (total + 1e-6f), 1.0f);
  if (tid == 0) {
    if (total_norm) *total_norm = total;
    if (loss_out) loss_out[0] = cnt > 0.0 ? -s0 / cnt : 0.0;
    if (xchg && stats_out) { stats_out[0] = s0; stats_out[1] = cnt; }
  }
  if (tid < 6) A.step[tid][0] = A.step_new;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    if (!dp[u]) continue;
    float g = v[u] * coef;
    asm volatile("" : "+v"(g));          // the ROUNDED clipped gradient (what the separate AdamW launch reads back from .grad): no FMA across this point
    *dp[u] = g;
    const int sg = seg[u], o = off[u];
    float pv = A.p[sg][o], m = A.m[sg][o], vv = A.v[sg][o];
    adamw_elem(g, pv, m, vv, A.step_size[sg], A.decay[sg], A.omb1, A.beta2, A.omb2, A.bc2_sqrt, A.eps);
    A.p[sg][o] = pv; A.m[sg][o] = m; A.v[sg][o] = vv;
  }
}

}  // namespace RIFT_NS
