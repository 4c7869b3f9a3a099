// Row-resident MFMA GEMM for gfx950:  Y[M,N] = epilogue( prologue(X)[M,K] . W[N,K]^T )
//
// Every dense layer of the Pluto policy (nn.Linear, Conv1d k=3 as im2col, the
// halves of nn.MultiheadAttention in/out projections) has K <= 512, so a
// workgroup keeps a 64-row A tile with the WHOLE K extent in LDS.  That makes
// the row prologues (LayerNorm over K, BatchNorm affine + ReLU) exact and free,
// and lets the 4 waves stream the weight fragments straight from L2 into
// registers (each weight element is read once per workgroup).
//
//   bf16 mode : A tile bf16 in LDS, weights bf16 [Npad][Kp], v_mfma_f32_16x16x32_bf16, fp32 accumulate
//   fp32 mode : A tile fp32 in LDS, weights fp32 [Npad][Kp], v_mfma_f32_16x16x4_f32 (exact fp32 fma chain)
#pragma once
#include "common.h"

namespace rift {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2 };
enum { PRO_NONE = 0, PRO_LN = 1, PRO_AFFINE = 2 };
enum { AMODE_LINEAR = 0, AMODE_CONV3 = 1 };

struct GemmP {
  const float* X; int ldx;
  int M, N, K, Kp;
  const void* W;            // packed weights [Npad][Kp]
  const float* bias;        // [N] or null
  float* Y; int ldy;
  // A loader
  int amode;
  int cv_C, cv_Lin, cv_nout, cv_t0, cv_stride;   // conv3: row r -> (seq = r / nout, t = t0 + r % nout), K = 3*C
  // prologue over the K extent of each row
  int pro; const float* pg; const float* pb; int pro_relu; float ln_eps;
  // epilogue
  const float* gbias; int gb_div, gb_mod;        // + gbias[((row / gb_div) % gb_mod) * N + col]   (gb_mod = 0: no modulo)
  int act;
  float dropout_p; uint32_t seed; uint32_t stream;   // elementwise dropout (train mode)
  float droppath_p; int dp_div;                  // per-sample stochastic depth: sample = row / dp_div
  const float* residual; int ldr;                // Y = residual + val
  const uint8_t* rowzero; int rz_div;            // rowzero[row / rz_div] != 0 -> Y = 0
};

template <bool BF16> struct Prec;
template <> struct Prec<true> {
  typedef unsigned short lds_t;
  static constexpr int PAD = 8;
  __device__ static __forceinline__ lds_t cvt(float f) { return f2bf(f); }
};
template <> struct Prec<false> {
  typedef float lds_t;
  static constexpr int PAD = 4;
  __device__ static __forceinline__ lds_t cvt(float f) { return f; }
};

template <bool BF16, int MT, int NT, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_rows_kernel(GemmP p) {
  static_assert(WM * WN == 4 && MT * WM == 4, "64-row tile, 4 waves");
  typedef typename Prec<BF16>::lds_t lds_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  lds_t* As = reinterpret_cast<lds_t*>(smem_raw);
  const int lda = p.Kp + Prec<BF16>::PAD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * 64;

  // ---------------- stage the A tile: one wave per row, lanes stride over k ----------------
  for (int rr = wave; rr < 64; rr += 4) {
    const int row = row0 + rr;
    float v[8];
    const bool rvalid = row < p.M;
    const float* src = nullptr;
    int seq = 0, t = 0;
    if (rvalid) {
      if (p.amode == AMODE_LINEAR) src = p.X + (size_t)row * p.ldx;
      else { seq = row / p.cv_nout; t = (p.cv_t0 + row % p.cv_nout) * p.cv_stride - 1; }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = lane + i * 64;
      float x = 0.f;
      if (rvalid && k < p.K) {
        if (p.amode == AMODE_LINEAR) x = src[k];
        else {
          const int j = k / p.cv_C, c = k - j * p.cv_C, tin = t + j;
          if (tin >= 0 && tin < p.cv_Lin) x = p.X[((size_t)seq * p.cv_Lin + tin) * p.ldx + c];
        }
      }
      v[i] = x;
    }
    if (p.pro == PRO_LN) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) s += v[i];
      const float mean = wave_sum(s) / (float)p.K;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) { const int k = lane + i * 64; const float d = (k < p.K) ? v[i] - mean : 0.f; q += d * d; }
      const float rstd = rsqrtf(wave_sum(q) / (float)p.K + p.ln_eps);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = lane + i * 64;
        if (k < p.K) v[i] = (v[i] - mean) * rstd * p.pg[k] + p.pb[k];
      }
    } else if (p.pro == PRO_AFFINE) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { const int k = lane + i * 64; if (k < p.K) v[i] = v[i] * p.pg[k] + p.pb[k]; }
    }
    if (p.pro_relu) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = lane + i * 64;
      if (k < p.Kp) As[rr * lda + k] = Prec<BF16>::cvt((rvalid && k < p.K) ? v[i] : 0.f);
    }
  }
  __syncthreads();

  // ---------------- MFMA main loop ----------------
  const int wm = wave / WN, wn = wave % WN;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int KS = p.Kp >> 5;
  const int Npad = (p.N + 15) & ~15;
  constexpr int PASSN = 16 * NT * WN;
  for (int n0 = 0; n0 < p.N; n0 += PASSN) {
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int ncol0 = n0 + wn * (16 * NT);
    if constexpr (BF16) {
      const unsigned short* W = reinterpret_cast<const unsigned short*>(p.W);
      for (int ks = 0; ks < KS; ++ks) {
        bf16x8 b[NT], a[MT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int nb = ncol0 + nt * 16;
          if (nb < Npad) b[nt] = *reinterpret_cast<const bf16x8*>(W + (size_t)(nb + l15) * p.Kp + ks * 32 + l4 * 8);
          else b[nt] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          a[mt] = *reinterpret_cast<const bf16x8*>(As + ((wm * MT + mt) * 16 + l15) * lda + ks * 32 + l4 * 8);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
      }
    } else {
      const float* W = reinterpret_cast<const float*>(p.W);
      for (int k0 = 0; k0 < p.Kp; k0 += 4) {
        float b[NT], a[MT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int nb = ncol0 + nt * 16;
          b[nt] = (nb < Npad) ? W[(size_t)(nb + l15) * p.Kp + k0 + l4] : 0.f;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = As[((wm * MT + mt) * 16 + l15) * lda + k0 + l4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
      }
    }

    // ---------------- epilogue: C/D layout col = lane&15, row = 4*(lane>>4) + reg ----------------
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int col = ncol0 + nt * 16 + l15;
        if (col >= p.N) continue;
        const float bcol = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = row0 + (wm * MT + mt) * 16 + l4 * 4 + r;
          if (row >= p.M) continue;
          float val = acc[mt][nt][r] + bcol;
          if (p.gbias) {
            int g = row / p.gb_div;
            if (p.gb_mod) g %= p.gb_mod;
            val += p.gbias[(size_t)g * p.N + col];
          }
          if (p.act == ACT_RELU) val = fmaxf(val, 0.f);
          else if (p.act == ACT_GELU) val = gelu_erf(val);
          if (p.dropout_p > 0.f) {
            const float u = uniform01(p.seed, p.stream, (uint32_t)row * (uint32_t)p.N + (uint32_t)col);
            val = (u < p.dropout_p) ? 0.f : val * (1.0f / (1.0f - p.dropout_p));
          }
          if (p.droppath_p > 0.f) {
            const float u = uniform01(p.seed, p.stream ^ 0x5bd1e995u, (uint32_t)(row / p.dp_div));
            val = (u < p.droppath_p) ? 0.f : val * (1.0f / (1.0f - p.droppath_p));
          }
          if (p.residual) val += p.residual[(size_t)row * p.ldr + col];
          if (p.rowzero && p.rowzero[row / p.rz_div]) val = 0.f;
          p.Y[(size_t)row * p.ldy + col] = val;
        }
      }
    }
  }
}

}  // namespace rift
