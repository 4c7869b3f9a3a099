// Fused trajectory heads for gfx950: the three MLPLayer(128, 256, 160) heads (Linear -> LayerNorm -> ReLU -> Linear; mlp_layer.py:8-16)
// of the planning decoder (loc / yaw / vel, planning_decoder.py:178-183) or of the agent predictor (agent_predictor.py:17-29) on a tile
// of 128 rows, writing the interleaved (rows, 80, 6) = [x, y, cos, sin, vx, vy] layout directly.  The 256-wide hidden layer never
// leaves the CU: its LayerNorm statistics are reduced across the 8 waves through a 4 KB LDS table, the normalised values go to LDS
// as bf16 MFMA operands.  Replaces 6 GEMM launches, an interleave kernel and ~0.4 GB of HBM traffic per call.
#pragma once
#include "common.h"
#include "pe_fused.h"

namespace RIFT_NS {

struct Heads3P {
  const float* X; int ldx;            // (rows, 128) fp32 input rows at stride ldx; row r of scene-structured inputs: see gather
  int rows;
  int gather_per, gather_stride, gather_off;   // gather_per > 0: input row r lives at X[((r / per) * stride + off + r % per) * ldx]
  const unsigned short* w1[3]; const float* b1[3]; const float* lng[3]; const float* lnb[3];
  const unsigned short* w2[3]; const float* b2[3];   // fragment-major bf16 [256][128] and [160][256]
  float* out;                         // (rows, 80, 6)
};

#define HD_ROWS 128
#define HD_XS 144
#define HD_HS 272
#define HD_LDS (HD_ROWS * HD_XS * 2 + HD_ROWS * HD_HS * 2 + 8 * HD_ROWS * 2 * 4 + 3 * 768 * 4 + 3 * 160 * 4)

__global__ __launch_bounds__(512) void heads3_fused_kernel(Heads3P p) {
  constexpr int MT = 8, NW = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* xb = reinterpret_cast<unsigned short*>(smem_raw);          // [128][HD_XS] bf16 input tile
  unsigned short* hn = xb + HD_ROWS * HD_XS;                                 // [128][HD_HS] bf16 relu(LN(hidden))
  float* wpart = reinterpret_cast<float*>(hn + HD_ROWS * HD_HS);             // [8 waves][128 rows][2]: per-wave row sums (x, x^2)
  float* par = wpart + 8 * HD_ROWS * 2;                                      // per head: b1 256 | ln g 256 | ln b 256, then b2 160 x 3
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  // one (row tile, head) pair per workgroup: 144 + 126 row tiles alone would leave half of the 256 CUs idle.  The three heads of a tile write
  // 8-byte pieces of the same 24-byte (step) records: their workgroups are numbered 8 apart, so that they run on ONE XCD (workgroups go to
  // the XCDs round-robin) and its L2 merges the pieces into whole lines -- as neighbours (blockIdx % 3) they sat on three XCDs, every line
  // left three L2s partially written and the launch moved 97 MB for a 31 MB output
  const int grp = blockIdx.x / 24, rem = blockIdx.x % 24;
  const int row0 = (grp * 8 + (rem & 7)) * HD_ROWS, h0 = rem >> 3;
  if (row0 >= p.rows) return;
  PFrags<4, 2> W1;
  p_load_w<NW, 4, 2>(W1, p.w1[h0], 128, 0, wave, l15, l4);
  {
    float4 xv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = tid + u * 512, r = i >> 5, c4 = (i & 31) * 4;
      xv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      const int gr = row0 + r;
      if (gr < p.rows) {
        const size_t src = p.gather_per > 0 ? (size_t)(gr / p.gather_per) * p.gather_stride + p.gather_off + gr % p.gather_per : (size_t)gr;
        xv[u] = *reinterpret_cast<const float4*>(p.X + src * p.ldx + c4);
      }
    }
    float pv[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int e = tid + u * 512;
      pv[u] = 0.f;
      if (e < 3 * 768) { const int h = e / 768, k = e - h * 768; pv[u] = k < 256 ? p.b1[h][k] : k < 512 ? p.lng[h][k - 256] : p.lnb[h][k - 512]; }
      else if (e < 3 * 768 + 480) { const int k = e - 3 * 768; pv[u] = p.b2[k / 160][k % 160]; }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int i = tid + u * 512; *reinterpret_cast<uint2*>(xb + (i >> 5) * HD_XS + (i & 31) * 4) = pack_h4(xv[u].x, xv[u].y, xv[u].z, xv[u].w); }
#pragma unroll
    for (int u = 0; u < 6; ++u) { const int e = tid + u * 512; if (e < 3 * 768 + 480) par[e] = pv[u]; }
  }
  __syncthreads();
  for (int h = h0; h < h0 + 1; ++h) {
    const float* ph = par + h * 768;
    // ---- hidden = x W1^T + b1 (256 columns: n-tiles wave, wave + 8), LayerNorm statistics across the waves, ReLU -> hn
    f32x4 acc[MT][2];
    p_zero(acc);
    p_mma<MT, 4, 2>(acc, xb, HD_XS, 0, W1, l15, l4);
    // second layer: 160 outputs = n-tiles 0..9, K = 256 in two halves.  Every wave owns n-tile `wave`; waves 0 and 1 own n-tiles 8 and 9 as well
    // (round 6: the other six waves used to multiply zero fragments for a second n-tile that does not exist -- half of their MFMAs)
    const bool two = wave < 2;                          // (wave-uniform)
    PFrags<4, 1> W2a, W2b;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      W2a.f[ks][0] = fm_load(p.w2[h], 256, wave * 16, ks * 32, l4 * 16 + l15);
      W2b.f[ks][0] = fm_load(p.w2[h], 256, wave * 16, 128 + ks * 32, l4 * 16 + l15);
    }
    float bsum[MT], bsq[MT];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = (j * NW + wave) * 16 + l4 * 4;
      const float4 b4 = *reinterpret_cast<const float4*>(ph + col);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        acc[mt][j][0] += b4.x; acc[mt][j][1] += b4.y; acc[mt][j][2] += b4.z; acc[mt][j][3] += b4.w;
        const float s = (acc[mt][j][0] + acc[mt][j][1]) + (acc[mt][j][2] + acc[mt][j][3]);
        const float q = (acc[mt][j][0] * acc[mt][j][0] + acc[mt][j][1] * acc[mt][j][1]) + (acc[mt][j][2] * acc[mt][j][2] + acc[mt][j][3] * acc[mt][j][3]);
        if (j == 0) { bsum[mt] = s; bsq[mt] = q; } else { bsum[mt] += s; bsq[mt] += q; }
      }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {                   // the four l4 groups hold different columns of the same row
      float s = bsum[mt], q = bsq[mt];
      s = rows_sum(s);
      q = rows_sum(q);
      if (l4 == 0) { wpart[(wave * HD_ROWS + mt * 16 + l15) * 2] = s; wpart[(wave * HD_ROWS + mt * 16 + l15) * 2 + 1] = q; }
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = mt * 16 + l15;
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) { s += wpart[(w * HD_ROWS + row) * 2]; q += wpart[(w * HD_ROWS + row) * 2 + 1]; }
      const float mean = s * (1.0f / 256.0f);
      const float rstd = rsqrtf(fmaxf(q * (1.0f / 256.0f) - mean * mean, 0.f) + 1e-5f);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = (j * NW + wave) * 16 + l4 * 4;
        const float4 g4 = *reinterpret_cast<const float4*>(ph + 256 + col), e4 = *reinterpret_cast<const float4*>(ph + 512 + col);
        *reinterpret_cast<uint2*>(hn + row * HD_HS + col) =
            pack_h4(fmaxf((acc[mt][j][0] - mean) * rstd * g4.x + e4.x, 0.f), fmaxf((acc[mt][j][1] - mean) * rstd * g4.y + e4.y, 0.f),
                        fmaxf((acc[mt][j][2] - mean) * rstd * g4.z + e4.z, 0.f), fmaxf((acc[mt][j][3] - mean) * rstd * g4.w + e4.w, 0.f));
      }
    }
    __syncthreads();
    // ---- out = hn W2^T + b2: 10 n-tiles (waves 0, 1 own two, the others one), scattered into (row, step, component)
    {
      auto store = [&](const f32x4 (&o)[MT][1], int nt) {
        const int col = nt * 16 + l4 * 4;              // head output column: step col / 2, component col % 2
        const float4 b4 = *reinterpret_cast<const float4*>(par + 3 * 768 + h * 160 + col);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int gr = row0 + mt * 16 + l15;
          if (gr >= p.rows) continue;
          float* dst = p.out + (size_t)gr * 480 + (col >> 1) * 6 + h * 2;
          *reinterpret_cast<float2*>(dst) = make_float2(o[mt][0][0] + b4.x, o[mt][0][1] + b4.y);
          *reinterpret_cast<float2*>(dst + 6) = make_float2(o[mt][0][2] + b4.z, o[mt][0][3] + b4.w);
        }
      };
      {
        f32x4 o[MT][1];
        p_zero(o);
        p_mma<MT, 4, 1>(o, hn, HD_HS, 0, W2a, l15, l4);
        p_mma<MT, 4, 1>(o, hn, HD_HS, 128, W2b, l15, l4);
        store(o, wave);
      }
      if (two) {                                        // (its weights are fetched here: held across the first product they spilled)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          W2a.f[ks][0] = fm_load(p.w2[h], 256, (NW + wave) * 16, ks * 32, l4 * 16 + l15);
          W2b.f[ks][0] = fm_load(p.w2[h], 256, (NW + wave) * 16, 128 + ks * 32, l4 * 16 + l15);
        }
        f32x4 o[MT][1];
        p_zero(o);
        p_mma<MT, 4, 1>(o, hn, HD_HS, 0, W2a, l15, l4);
        p_mma<MT, 4, 1>(o, hn, HD_HS, 128, W2b, l15, l4);
        store(o, NW + wave);
      }
    }
    __syncthreads();     // hn / wpart are rewritten by the next head
  }
}

// ---- the two heads on the ego token (pluto_model.py:173-180), one launch ------------------------------------------------------------------------
// hidden = hidden_proj(x) = Linear(128, 128) -> ReLU -> Linear(128, 128) and ref_free_trajectory = MLPLayer(128, 256, 320)(x) on the bs ego rows
// of the encoder output.  As four gemm_rows launches these 75 MFLOP took 26 + 13 + 26 + 13 us of a queue of their own and 26 us of the
// every-output step (0.637 -> 0.611 ms without them): latency, not work.  Here: 16 rows per workgroup of four waves, the same rounding points
// as the layer-wise path (16-bit operands, fp32 accumulation, fp32 LayerNorm on the fp32 pre-activation, ReLU, 16-bit hidden operands).
struct EgoHeadsP {
  const float* X; int ldx; int rows;              // ego rows: X[r * ldx .. + 128]
  const unsigned short *wh0, *wh2, *wr0, *wr3;    // fragment-major images: [128][128], [128][128], [256][128], [320][256]
  const float *bh0, *bh2, *br0, *br3, *lng, *lnb;
  float* hidden;                                  // (rows, 128) or null
  float* ref;                                     // (rows, 320) or null
};
#define EH_XS 144
#define EH_TS 272

__global__ __launch_bounds__(256) void ego_heads_kernel(EgoHeadsP p) {
  __shared__ __attribute__((aligned(16))) unsigned short xb[16 * EH_XS];       // x rows as operands
  __shared__ __attribute__((aligned(16))) unsigned short h1[16 * EH_XS];       // relu(hidden_proj.0)
  __shared__ __attribute__((aligned(16))) unsigned short t1[16 * EH_TS];       // relu(LN(ref mlp.0))
  __shared__ float wpart[4][16][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int row0 = blockIdx.x * 16;
  {
    const int r = tid >> 4, c8 = (tid & 15) * 8, gr = min(row0 + r, p.rows - 1);
    const float4 a = *reinterpret_cast<const float4*>(p.X + (size_t)gr * p.ldx + c8), b = *reinterpret_cast<const float4*>(p.X + (size_t)gr * p.ldx + c8 + 4);
    *reinterpret_cast<uint2*>(xb + r * EH_XS + c8) = pack_h4(a.x, a.y, a.z, a.w);
    *reinterpret_cast<uint2*>(xb + r * EH_XS + c8 + 4) = pack_h4(b.x, b.y, b.z, b.w);
  }
  __syncthreads();
  const bool rok = row0 + l15 < p.rows;
  const int li = l4 * 16 + l15;
  // ---- first layers: hidden_proj.0 (n-tiles 2 wave, 2 wave + 1) and ref mlp.0 (n-tiles 4 wave .. + 3), K = 128
  f32x4 ah[2], ar[4];
#pragma unroll
  for (int j = 0; j < 2; ++j) { const float4 b = *reinterpret_cast<const float4*>(p.bh0 + (2 * wave + j) * 16 + l4 * 4); ah[j] = (f32x4){b.x, b.y, b.z, b.w}; }
#pragma unroll
  for (int j = 0; j < 4; ++j) { const float4 b = *reinterpret_cast<const float4*>(p.br0 + (4 * wave + j) * 16 + l4 * 4); ar[j] = (f32x4){b.x, b.y, b.z, b.w}; }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const h16x8 a = *reinterpret_cast<const h16x8*>(xb + l15 * EH_XS + ks * 32 + l4 * 8);
#pragma unroll
    for (int j = 0; j < 2; ++j) ah[j] = mfma_h(fm_load(p.wh0, 128, (2 * wave + j) * 16, ks * 32, li), a, ah[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) ar[j] = mfma_h(fm_load(p.wr0, 128, (4 * wave + j) * 16, ks * 32, li), a, ar[j], 0, 0, 0);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j)
    *reinterpret_cast<uint2*>(h1 + l15 * EH_XS + (2 * wave + j) * 16 + l4 * 4) = pack_h4(fmaxf(ah[j][0], 0.f), fmaxf(ah[j][1], 0.f), fmaxf(ah[j][2], 0.f), fmaxf(ah[j][3], 0.f));
  {   // LayerNorm statistics of the 256-wide pre-activation: this wave's 64 columns, then across the four waves
    float sm = 0.f, sq = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) { sm += ar[j][r]; sq += ar[j][r] * ar[j][r]; }
    sm = rows_sum(sm); sq = rows_sum(sq);
    if (l4 == 0) { wpart[wave][l15][0] = sm; wpart[wave][l15][1] = sq; }
  }
  __syncthreads();
  {
    float sm = 0.f, sq = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { sm += wpart[w][l15][0]; sq += wpart[w][l15][1]; }
    const float mean = sm * (1.0f / 256.0f), rstd = rsqrtf(fmaxf(sq * (1.0f / 256.0f) - mean * mean, 0.f) + 1e-5f);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = (4 * wave + j) * 16 + l4 * 4;
      const float4 g = *reinterpret_cast<const float4*>(p.lng + col), e = *reinterpret_cast<const float4*>(p.lnb + col);
      *reinterpret_cast<uint2*>(t1 + l15 * EH_TS + col) =
          pack_h4(fmaxf((ar[j][0] - mean) * rstd * g.x + e.x, 0.f), fmaxf((ar[j][1] - mean) * rstd * g.y + e.y, 0.f),
                  fmaxf((ar[j][2] - mean) * rstd * g.z + e.z, 0.f), fmaxf((ar[j][3] - mean) * rstd * g.w + e.w, 0.f));
    }
  }
  __syncthreads();
  // ---- second layers: hidden_proj.2 (K = 128 from h1, n-tiles 2 wave ..) and ref mlp.3 (K = 256 from t1, n-tiles 5 wave .. + 4 of 20)
  if (p.hidden) {
    f32x4 o[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { const float4 b = *reinterpret_cast<const float4*>(p.bh2 + (2 * wave + j) * 16 + l4 * 4); o[j] = (f32x4){b.x, b.y, b.z, b.w}; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const h16x8 a = *reinterpret_cast<const h16x8*>(h1 + l15 * EH_XS + ks * 32 + l4 * 8);
#pragma unroll
      for (int j = 0; j < 2; ++j) o[j] = mfma_h(fm_load(p.wh2, 128, (2 * wave + j) * 16, ks * 32, li), a, o[j], 0, 0, 0);
    }
    if (rok)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        *reinterpret_cast<float4*>(p.hidden + (size_t)(row0 + l15) * 128 + (2 * wave + j) * 16 + l4 * 4) = make_float4(o[j][0], o[j][1], o[j][2], o[j][3]);
  }
  if (p.ref) {
    f32x4 o[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) { const float4 b = *reinterpret_cast<const float4*>(p.br3 + (5 * wave + j) * 16 + l4 * 4); o[j] = (f32x4){b.x, b.y, b.z, b.w}; }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const h16x8 a = *reinterpret_cast<const h16x8*>(t1 + l15 * EH_TS + ks * 32 + l4 * 8);
#pragma unroll
      for (int j = 0; j < 5; ++j) o[j] = mfma_h(fm_load(p.wr3, 256, (5 * wave + j) * 16, ks * 32, li), a, o[j], 0, 0, 0);
    }
    if (rok)
#pragma unroll
      for (int j = 0; j < 5; ++j)
        *reinterpret_cast<float4*>(p.ref + (size_t)(row0 + l15) * 320 + (5 * wave + j) * 16 + l4 * 4) = make_float4(o[j][0], o[j][1], o[j][2], o[j][3]);
  }
}

}  // namespace RIFT_NS
