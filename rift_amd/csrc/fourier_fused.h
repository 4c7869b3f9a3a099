// Fused FourierEmbedding for gfx950 (fourier_embedding.py:45-55): for each input dim d,
//   feat_d = [cos(2 pi f_d x_d), sin(2 pi f_d x_d), x_d] (129) -> Linear(129,128) -> LayerNorm -> ReLU -> Linear(128,128);
//   sum over d -> LayerNorm -> ReLU -> Linear(128,128).
// One workgroup owns 64 rows; the Fourier features, both hidden activations and the sum stay in LDS / registers, so a
// call is one launch instead of 1 + 2 D + 1 (the 129-wide feature matrix never exists in HBM).  The 129th input
// (x_d itself) is applied as an fp32 rank-1 update in the epilogue, which keeps the MFMA contraction at K = 128.
// Weight fragment = MFMA A operand (see enc_fused.h): a lane holds four consecutive output channels of one row.
#pragma once
#include "common.h"
#include "pe_fused.h"

namespace RIFT_NS {

struct FourierP {
  const float* in; int in_ld, rows, D, wrap_dim;
  const float* freqs;                    // [D][64]
  const unsigned short* w0[3];           // bf16 [128][128]: input columns 0..127 of mlps.d.0
  const float* wl[3]; int wl_ld;         // input column 128 of mlps.d.0 (fp32, element n at wl[d][n * wl_ld])
  const float* b0[3]; const float* lng[3]; const float* lnb[3];
  const unsigned short* w3[3]; const float* b3[3];
  const float* og; const float* ob;      // to_out.0
  const unsigned short* wo; const float* bo;
  float* Y; int accumulate;              // Y (rows, 128) = or += result
};

#define FO_ROWS 64
#define FO_FS 144
#define FO_HS 132
#define FO_NPAR (3 * 5 * 128 + 3 * 128)
#define FO_LDS (FO_ROWS * FO_FS * 2 * 2 + FO_ROWS * FO_HS * 4 + FO_ROWS * 4 * 4 + FO_NPAR * 4)

__device__ __forceinline__ void fourier_fused_body(const FourierP& p, const int blk) {
  constexpr int MT = 4, NW = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* feat = reinterpret_cast<unsigned short*>(smem_raw);
  unsigned short* hn = feat + FO_ROWS * FO_FS;
  float* hf = reinterpret_cast<float*>(hn + FO_ROWS * FO_FS);
  float* xs = hf + FO_ROWS * FO_HS;            // [64][4]
  float* par = xs + FO_ROWS * 4;               // per dim: b0 | lng | lnb | wl | b3 (640), then og | ob | bo
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int row0 = blk * FO_ROWS;
  const int D = p.D;

  PFrags<4, 2> W0, W3;
  p_load_w<NW, 4, 2>(W0, p.w0[0], 128, 0, wave, l15, l4);
  p_load_w<NW, 4, 2>(W3, p.w3[0], 128, 0, wave, l15, l4);
  for (int e = tid; e < D * 640; e += 256) {
    const int d = e / 640, k = e - d * 640, n = k & 127;
    const int which = k >> 7;
    par[e] = which == 0 ? p.b0[d][n] : which == 1 ? p.lng[d][n] : which == 2 ? p.lnb[d][n] : which == 3 ? p.wl[d][(size_t)n * p.wl_ld] : p.b3[d][n];
  }
  for (int e = tid; e < 384; e += 256) par[1920 + e] = e < 128 ? p.og[e] : e < 256 ? p.ob[e - 128] : p.bo[e - 256];
  for (int i = tid; i < FO_ROWS * 4; i += 256) {
    const int r = i >> 2, d = i & 3;
    float x = 0.f;
    if (d < D && row0 + r < p.rows) {
      x = p.in[(size_t)(row0 + r) * p.in_ld + d];
      if (d == p.wrap_dim) { x = fmodf(x + RIFT_PI, 2.f * RIFT_PI); if (x < 0.f) x += 2.f * RIFT_PI; x -= RIFT_PI; }
    }
    xs[i] = x;
  }
  __syncthreads();

  auto layer_norm_relu = [&](const float* g, const float* be) {   // hf -> hn (bf16); 16 lanes per row, 8 columns per lane
    const float4 g0 = *reinterpret_cast<const float4*>(g + l15 * 8), g1 = *reinterpret_cast<const float4*>(g + l15 * 8 + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(be + l15 * 8), b1 = *reinterpret_cast<const float4*>(be + l15 * 8 + 4);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int r = it * 16 + wave * 4 + l4;
      const float4 v0 = *reinterpret_cast<const float4*>(hf + r * FO_HS + l15 * 8), v1 = *reinterpret_cast<const float4*>(hf + r * FO_HS + l15 * 8 + 4);
      float s = ((v0.x + v0.y) + (v0.z + v0.w)) + ((v1.x + v1.y) + (v1.z + v1.w));
      s = sum16(s);
      const float mean = s * (1.0f / 128.0f);
      const float d0 = v0.x - mean, d1 = v0.y - mean, d2 = v0.z - mean, d3 = v0.w - mean;
      const float d4 = v1.x - mean, d5 = v1.y - mean, d6 = v1.z - mean, d7 = v1.w - mean;
      float q = ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) + ((d4 * d4 + d5 * d5) + (d6 * d6 + d7 * d7));
      q = sum16(q);
      const float rstd = rsqrtf(q * (1.0f / 128.0f) + 1e-5f);
      uint4 o;
      o.x = pack_h2(fmaxf(d0 * rstd * g0.x + b0.x, 0.f), fmaxf(d1 * rstd * g0.y + b0.y, 0.f));
      o.y = pack_h2(fmaxf(d2 * rstd * g0.z + b0.z, 0.f), fmaxf(d3 * rstd * g0.w + b0.w, 0.f));
      o.z = pack_h2(fmaxf(d4 * rstd * g1.x + b1.x, 0.f), fmaxf(d5 * rstd * g1.y + b1.y, 0.f));
      o.w = pack_h2(fmaxf(d6 * rstd * g1.z + b1.z, 0.f), fmaxf(d7 * rstd * g1.w + b1.w, 0.f));
      *reinterpret_cast<uint4*>(hn + r * FO_FS + l15 * 8) = o;
    }
  };

  f32x4 sum[MT][2];
  p_zero(sum);
  for (int d = 0; d < D; ++d) {
    const float* pd = par + d * 640;
    for (int i = tid; i < FO_ROWS * 64; i += 256) {
      const int r = i >> 6, fq = i & 63;
      // cos / sin(2 pi f x): the hardware v_sin / v_cos take their argument in revolutions, so only fract(f x) is needed
      // (|error| ~ 1e-6, far below the bf16 rounding the features get as MFMA operands)
      const float rev = __builtin_amdgcn_fractf(xs[r * 4 + d] * p.freqs[d * 64 + fq]);
      const float sn = __builtin_amdgcn_sinf(rev), cs = __builtin_amdgcn_cosf(rev);
      feat[r * FO_FS + fq] = f2h(cs);
      feat[r * FO_FS + 64 + fq] = f2h(sn);
    }
    __syncthreads();
    {
      f32x4 acc[MT][2];
      p_zero(acc);
      p_mma<MT, 4, 2>(acc, feat, FO_FS, 0, W0, l15, l4);
      if (d + 1 < D) p_load_w<NW, 4, 2>(W0, p.w0[d + 1], 128, 0, wave, l15, l4);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = (j * NW + wave) * 16 + l4 * 4;
        const float4 b = *reinterpret_cast<const float4*>(pd + col), wl = *reinterpret_cast<const float4*>(pd + 384 + col);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int row = mt * 16 + l15;
          const float x = xs[row * 4 + d];
          *reinterpret_cast<float4*>(hf + row * FO_HS + col) =
              make_float4(acc[mt][j][0] + b.x + wl.x * x, acc[mt][j][1] + b.y + wl.y * x, acc[mt][j][2] + b.z + wl.z * x, acc[mt][j][3] + b.w + wl.w * x);
        }
      }
    }
    __syncthreads();
    layer_norm_relu(pd + 128, pd + 256);
    __syncthreads();
    p_mma<MT, 4, 2>(sum, hn, FO_FS, 0, W3, l15, l4);
    if (d + 1 < D) p_load_w<NW, 4, 2>(W3, p.w3[d + 1], 128, 0, wave, l15, l4);
    else p_load_w<NW, 4, 2>(W3, p.wo, 128, 0, wave, l15, l4);
  }
  // ---- to_out: LayerNorm -> ReLU -> Linear
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = (j * NW + wave) * 16 + l4 * 4;
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int d = 0; d < D; ++d) {
      const float4 t = *reinterpret_cast<const float4*>(par + d * 640 + 512 + col);
      b.x += t.x; b.y += t.y; b.z += t.z; b.w += t.w;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      *reinterpret_cast<float4*>(hf + (mt * 16 + l15) * FO_HS + col) =
          make_float4(sum[mt][j][0] + b.x, sum[mt][j][1] + b.y, sum[mt][j][2] + b.z, sum[mt][j][3] + b.w);
  }
  __syncthreads();
  layer_norm_relu(par + 1920, par + 2048);
  __syncthreads();
  {
    f32x4 acc[MT][2];
    p_zero(acc);
    p_mma<MT, 4, 2>(acc, hn, FO_FS, 0, W3, l15, l4);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = (j * NW + wave) * 16 + l4 * 4;
      const float4 b = *reinterpret_cast<const float4*>(par + 2176 + col);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int row = row0 + mt * 16 + l15;
        if (row >= p.rows) continue;
        float4* dst = reinterpret_cast<float4*>(p.Y + (size_t)row * 128 + col);
        float4 v = make_float4(acc[mt][j][0] + b.x, acc[mt][j][1] + b.y, acc[mt][j][2] + b.z, acc[mt][j][3] + b.w);
        if (p.accumulate) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        *dst = v;
      }
    }
  }
}

// up to three embeddings (token positions, speed limits, reference-line positions) in one launch: blocks are dealt in order
struct FourierP3 { FourierP e[3]; int nblk[3]; int count; };

__global__ __launch_bounds__(256) void fourier_fused_kernel(FourierP3 q) {
  int blk = blockIdx.x;
  if (blk < q.nblk[0]) { fourier_fused_body(q.e[0], blk); return; }
  blk -= q.nblk[0];
  if (q.count > 1 && blk < q.nblk[1]) { fourier_fused_body(q.e[1], blk); return; }
  blk -= q.nblk[1];
  if (q.count > 2) fourier_fused_body(q.e[2], blk);
}

}  // namespace RIFT_NS
