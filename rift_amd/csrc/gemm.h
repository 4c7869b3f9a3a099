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

namespace RIFT_NS {

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
  int cs_off;                                    // byte offset of the C staging area in dynamic LDS (0 = aliases the A tile)
  int evec;                                      // epilogue may use 16-byte accesses (N, ldy, ldr % 4 == 0, aligned pointers)
  int dbg;                                       // timing experiments only; 0 in production
  int stage;                                     // 0 scalar staging; 1..5 vectorised (LPR,CH) = (8,1)(16,1)(32,1)(64,1)(64,2)
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
  __device__ static __forceinline__ lds_t cvt(float f) { return f2h(f); }
};
template <> struct Prec<false> {
  typedef float lds_t;
  static constexpr int PAD = 4;
  __device__ static __forceinline__ lds_t cvt(float f) { return f; }
};

// Row descriptor of the A operand: element k of output row `row` is src[k] for k in [klo, khi), else 0.
// Linear rows: src = X + row*ldx.  Conv1d(k=3, pad=1) rows are a CONTIGUOUS 3*C window of the dense
// (L, C) sequence starting one step before the output position (taps outside [0, Lin) are the zero padding).
struct RowDesc { const float* src; int klo, khi; bool valid; };

__device__ __forceinline__ RowDesc row_desc(const GemmP& p, int row) {
  RowDesc d;
  d.valid = row < p.M;
  d.klo = 0; d.khi = d.valid ? p.K : 0; d.src = p.X;
  if (!d.valid) return d;
  if (p.amode == AMODE_LINEAR) {
    d.src = p.X + (size_t)row * p.ldx;
  } else {
    const int seq = row / p.cv_nout, t = p.cv_t0 + row - seq * p.cv_nout;
    const int tin0 = t * p.cv_stride - 1;
    d.src = p.X + ((long long)seq * p.cv_Lin + tin0) * p.cv_C;
    d.klo = tin0 < 0 ? p.cv_C : 0;
    const int ntap = p.cv_Lin - tin0;
    d.khi = (ntap < 3 ? ntap : 3) * p.cv_C;
  }
  return d;
}

__device__ __forceinline__ void lds_store4(unsigned short* dst, float4 v) {
  uint2 u;
  u.x = (unsigned)f2h(v.x) | ((unsigned)f2h(v.y) << 16);
  u.y = (unsigned)f2h(v.z) | ((unsigned)f2h(v.w) << 16);
  *reinterpret_cast<uint2*>(dst) = u;
}
__device__ __forceinline__ void lds_store4(float* dst, float4 v) { *reinterpret_cast<float4*>(dst) = v; }

// Vectorised staging of this wave's 16 rows: LPR lanes share a row (16-byte loads), up to 8 loads per
// lane are issued before the first use, LayerNorm statistics are xor-shuffles over the LPR lanes.
template <bool BF16, int LPR, int CH>
__device__ __forceinline__ void stage_rows_vec(const GemmP& p, typename Prec<BF16>::lds_t* As, int lda, int row0,
                                               int wave, int lane) {
  constexpr int RPS = 64 / LPR;            // rows per step
  constexpr int NS = 16 / RPS;             // steps for 16 rows
  constexpr int B = (NS * CH > 8) ? (8 / CH) : NS;
  const int lr = lane % LPR, rsub = lane / LPR;
  float4 gam[CH], bet[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int k4 = (lr + c * LPR) * 4;
    gam[c] = make_float4(1.f, 1.f, 1.f, 1.f); bet[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.pro != PRO_NONE && k4 < p.K) {
      gam[c] = *reinterpret_cast<const float4*>(p.pg + k4);
      bet[c] = *reinterpret_cast<const float4*>(p.pb + k4);
    }
  }
  const float invK = 1.0f / (float)p.K;
  for (int s0 = 0; s0 < NS; s0 += B) {
    float4 v[B][CH];
#pragma unroll
    for (int s = 0; s < B; ++s) {
      const int rr = wave * 16 + (s0 + s) * RPS + rsub;
      const RowDesc d = row_desc(p, row0 + rr);
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int k4 = (lr + c * LPR) * 4;
        v[s][c] = (k4 >= d.klo && k4 < d.khi) ? *reinterpret_cast<const float4*>(d.src + k4)
                                              : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    if (p.pro == PRO_LN) {
#pragma unroll
      for (int s = 0; s < B; ++s) {
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) sum += (v[s][c].x + v[s][c].y) + (v[s][c].z + v[s][c].w);
        sum = group_sum<LPR>(sum);
        const float mean = sum * invK;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          if ((lr + c * LPR) * 4 < p.K) {
            const float a = v[s][c].x - mean, b = v[s][c].y - mean, cc = v[s][c].z - mean, dd = v[s][c].w - mean;
            q += (a * a + b * b) + (cc * cc + dd * dd);
          }
        }
        q = group_sum<LPR>(q);
        const float rstd = rsqrtf(q * invK + p.ln_eps);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          v[s][c].x = (v[s][c].x - mean) * rstd * gam[c].x + bet[c].x;
          v[s][c].y = (v[s][c].y - mean) * rstd * gam[c].y + bet[c].y;
          v[s][c].z = (v[s][c].z - mean) * rstd * gam[c].z + bet[c].z;
          v[s][c].w = (v[s][c].w - mean) * rstd * gam[c].w + bet[c].w;
        }
      }
    } else if (p.pro == PRO_AFFINE) {
#pragma unroll
      for (int s = 0; s < B; ++s)
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          v[s][c].x = v[s][c].x * gam[c].x + bet[c].x; v[s][c].y = v[s][c].y * gam[c].y + bet[c].y;
          v[s][c].z = v[s][c].z * gam[c].z + bet[c].z; v[s][c].w = v[s][c].w * gam[c].w + bet[c].w;
        }
    }
#pragma unroll
    for (int s = 0; s < B; ++s) {
      const int rr = wave * 16 + (s0 + s) * RPS + rsub;
      const bool rvalid = (row0 + rr) < p.M;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int k4 = (lr + c * LPR) * 4;
        if (k4 >= p.Kp) continue;
        float4 o = v[s][c];
        if (p.pro_relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        if (!rvalid || k4 >= p.K) o = make_float4(0.f, 0.f, 0.f, 0.f);
        lds_store4(As + rr * lda + k4, o);
      }
    }
  }
}

// Scalar staging (odd K / unaligned rows): one wave per row, lanes stride over k.
template <bool BF16>
__device__ __forceinline__ void stage_rows_scalar(const GemmP& p, typename Prec<BF16>::lds_t* As, int lda, int row0,
                                                  int wave, int lane) {
  for (int rr = wave * 16; rr < wave * 16 + 16; ++rr) {
    const RowDesc d = row_desc(p, row0 + rr);
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = lane + i * 64;
      v[i] = (k >= d.klo && k < d.khi) ? d.src[k] : 0.f;
    }
    if (p.pro == PRO_LN) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) s += v[i];
      const float mean = wave_sum(s) / (float)p.K;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) { const int k = lane + i * 64; const float e = (k < p.K) ? v[i] - mean : 0.f; q += e * e; }
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
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = lane + i * 64;
      float o = p.pro_relu ? fmaxf(v[i], 0.f) : v[i];
      if (k < p.Kp) As[rr * lda + k] = Prec<BF16>::cvt((d.valid && k < p.K) ? o : 0.f);
    }
  }
}

// register-array element by a runtime (wave-uniform) index, as a v_cndmask chain (no scratch)
template <int N>
__device__ __forceinline__ float4 sel4(const float4 (&a)[N], int i) {
  float4 r = a[0];
#pragma unroll
  for (int j = 1; j < N; ++j) { const bool c = (i == j); r.x = c ? a[j].x : r.x; r.y = c ? a[j].y : r.y; r.z = c ? a[j].z : r.z; r.w = c ? a[j].w : r.w; }
  return r;
}

template <bool BF16, int MT, int NT, int WM, int WN>
__global__ __launch_bounds__(256, 2) void gemm_rows_kernel(GemmP p) {
  static_assert(WM * WN == 4 && MT * WM == 4, "64-row tile, 4 waves");
  typedef typename Prec<BF16>::lds_t lds_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  lds_t* As = reinterpret_cast<lds_t*>(smem_raw);
  const int lda = p.Kp + Prec<BF16>::PAD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * 64;
  // C staging (4 waves x 16 rows x (16*NT+4) fp32): aliases the A tile when a single N pass suffices
  float* Cs = reinterpret_cast<float*>(smem_raw + p.cs_off);

  // ---------------- stage the A tile (64 rows x Kp) into LDS ----------------
  switch (p.stage) {
    case 1: stage_rows_vec<BF16, 8, 1>(p, As, lda, row0, wave, lane); break;
    case 2: stage_rows_vec<BF16, 16, 1>(p, As, lda, row0, wave, lane); break;
    case 3: stage_rows_vec<BF16, 32, 1>(p, As, lda, row0, wave, lane); break;
    case 4: stage_rows_vec<BF16, 64, 1>(p, As, lda, row0, wave, lane); break;
    case 5: stage_rows_vec<BF16, 64, 2>(p, As, lda, row0, wave, lane); break;
    default: stage_rows_scalar<BF16>(p, As, lda, row0, wave, lane); break;
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
    constexpr int CW = 16 * NT, CWP = CW + 4, C4 = CW / 4, NI = CW / 16;
    if constexpr (BF16) {
      const unsigned short* W = reinterpret_cast<const unsigned short*>(p.W);
      for (int ks = 0; ks < KS; ++ks) {
        h16x8 b[NT], a[MT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int nb = ncol0 + nt * 16;
          if (nb < Npad) b[nt] = fm_load(W, p.Kp, nb, ks * 32, l4 * 16 + l15);   // fragment-major weight image (common.h)
          else b[nt] = (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          a[mt] = *reinterpret_cast<const h16x8*>(As + ((wm * MT + mt) * 16 + l15) * lda + ks * 32 + l4 * 8);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = mfma_h(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
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

    // ---------------- epilogue through LDS: C/D layout (col = lane&15, row = 4*(lane>>4) + reg) is
    // transposed to row-major in a WAVE-PRIVATE staging area, so that residual loads and Y stores are
    // 16-byte row-contiguous accesses; only wave-level ordering is needed ----------------
    float* Cw = Cs + wave * (16 * CWP);
    if (p.cs_off == 0) __syncthreads();   // C staging aliases the A tile: every wave must be done reading A
#pragma unroll 1
    for (int mt = 0; mt < MT; ++mt) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        f32x4 t = acc[0][nt];
#pragma unroll
        for (int m2 = 1; m2 < MT; ++m2) {
          const bool c = (mt == m2);
          t[0] = c ? acc[m2][nt][0] : t[0]; t[1] = c ? acc[m2][nt][1] : t[1];
          t[2] = c ? acc[m2][nt][2] : t[2]; t[3] = c ? acc[m2][nt][3] : t[3];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) Cw[(l4 * 4 + r) * CWP + nt * 16 + l15] = t[r];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll 1
      for (int i = 0; i < NI; ++i) {
        const int idx = lane + 64 * i;
        const int rr = idx / C4, c4 = (idx - rr * C4) * 4;
        const int row = row0 + (wm * MT + mt) * 16 + rr;
        const int col = ncol0 + c4;
        if (row >= p.M || col >= p.N) continue;
        const float4 cv = *reinterpret_cast<const float4*>(Cw + rr * CWP + c4);
        float val[4] = {cv.x, cv.y, cv.z, cv.w};
        const int nv = (p.N - col) < 4 ? (p.N - col) : 4;
        float* yrow = p.Y + (size_t)row * p.ldy + col;
        const bool vec = p.evec && nv == 4;
        float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (vec) {
          // issue every global operand load of this slot before the first use
          float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), g4 = b4;
          if (p.residual) r4 = *reinterpret_cast<const float4*>(p.residual + (size_t)row * p.ldr + col);
          if (p.bias) b4 = *reinterpret_cast<const float4*>(p.bias + col);
          if (p.gbias) {
            int g = row / p.gb_div;
            if (p.gb_mod) g %= p.gb_mod;
            g4 = *reinterpret_cast<const float4*>(p.gbias + (size_t)g * p.N + col);
          }
          val[0] += b4.x + g4.x; val[1] += b4.y + g4.y; val[2] += b4.z + g4.z; val[3] += b4.w + g4.w;
        } else {
          for (int e = 0; e < nv; ++e) {
            if (p.bias) val[e] += p.bias[col + e];
            if (p.gbias) {
              int g = row / p.gb_div;
              if (p.gb_mod) g %= p.gb_mod;
              val[e] += p.gbias[(size_t)g * p.N + col + e];
            }
          }
        }
        if (p.act == ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) val[e] = fmaxf(val[e], 0.f);
        } else if (p.act == ACT_GELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) val[e] = BF16 ? gelu_fast(val[e]) : gelu_erf(val[e]);
        }
        if (p.dropout_p > 0.f) {
          const float sc = 1.0f / (1.0f - p.dropout_p);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float u = uniform01(p.seed, p.stream, (uint32_t)row * (uint32_t)p.N + (uint32_t)(col + e));
            val[e] = (u < p.dropout_p) ? 0.f : val[e] * sc;
          }
        }
        if (p.droppath_p > 0.f) {
          const float u = uniform01(p.seed, p.stream ^ 0x5bd1e995u, (uint32_t)(row / p.dp_div));
          const float sc = (u < p.droppath_p) ? 0.f : 1.0f / (1.0f - p.droppath_p);
#pragma unroll
          for (int e = 0; e < 4; ++e) val[e] *= sc;
        }
        if (p.residual) {
          if (vec) {
            val[0] += r4.x; val[1] += r4.y; val[2] += r4.z; val[3] += r4.w;
          } else {
            const float* rrow = p.residual + (size_t)row * p.ldr + col;
            for (int e = 0; e < nv; ++e) val[e] += rrow[e];
          }
        }
        if (p.rowzero && p.rowzero[row / p.rz_div]) { val[0] = val[1] = val[2] = val[3] = 0.f; }
        if (vec) *reinterpret_cast<float4*>(yrow) = make_float4(val[0], val[1], val[2], val[3]);
        else for (int e = 0; e < nv; ++e) yrow[e] = val[e];
      }
    }
    if (p.cs_off == 0 && n0 + PASSN < p.N) __syncthreads();
  }
}

}  // namespace RIFT_NS
