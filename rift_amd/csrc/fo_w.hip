// Wave-private, weight-streaming FourierEmbedding (see fo_w.h).  Its own translation unit: built with `-fno-honor-nans -mno-amdgpu-ieee`.
#include "common.h"
#include "fo_w.h"
#include "wp_stream.h"

namespace RIFT_NS {

#ifndef RIFT_PI
#define RIFT_PI 3.14159265358979323846f
#endif

// image: group g = 2 d (+ 1) -> mlps.d.0[:, :128] in plain K order (the features are the operand as they come: k < 64 cos, k >= 64 sin) /
// mlps.d.3 K-permuted (an n-tile pair of the hidden activation is a k-step); group 2 D -> to_out.2 K-permuted.  Fragment f = ks * 8 + nt.
__global__ void pack_fow_kernel(FoWSrc s, unsigned short* __restrict__ img, float* __restrict__ par) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < (2 * s.D + 1) * 32 * 512) {
    const int j = e & 7, lane = (e >> 3) & 63, fr = e >> 9, g = fr >> 5, f = fr & 31, ks = f >> 3, nt = f & 7, l15 = lane & 15, l4 = lane >> 4;
    const int o = nt * 16 + l15, ch = l0w_chan(l4, j, 2 * ks);
    float v;
    if (g == 2 * s.D) v = s.wo[o * 128 + ch];
    else if (g & 1) v = s.w3[g >> 1][o * 128 + ch];
    else v = s.w0[g >> 1][o * 129 + 32 * ks + 8 * l4 + j];
    img[e] = f2h(v);
  }
  if (e < FOW_NPAR) {
    float v = 0.f;
    if (e < FOW_PAR_OUT) {
      const int d = e / FOW_PAR_DIM, k = e % FOW_PAR_DIM, n = k & 127, which = k >> 7;
      if (d < s.D) v = which == 0 ? s.b0[d][n] : which == 1 ? s.lng[d][n] : which == 2 ? s.lnb[d][n] : which == 3 ? s.w0[d][n * 129 + 128] : s.b3[d][n];
    } else if (e < FOW_PAR_FREQ) {
      const int k = e - FOW_PAR_OUT;
      v = k < 128 ? s.og[k] : k < 256 ? s.ob[k - 128] : s.bo[k - 256];
    } else if (e < FOW_PAR_B3SUM) {
      const int k = e - FOW_PAR_FREQ;
      if (k < s.D * 64) v = s.freqs[k];
    } else if (e < FOW_PAR_B3SUM + 128) {
      for (int d = 0; d < s.D; ++d) v += s.b3[d][e - FOW_PAR_B3SUM];
    }
    par[e] = v;
  }
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void fo_w_kernel(FoWP q) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char* ring = smem_raw;
  float* par = reinterpret_cast<float*>(smem_raw + 2 * 32768);
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem_raw);
  const uint32_t voff = (uint32_t)lane * 16u;
  int wg = blockIdx.x, ei = 0;
  while (ei + 1 < q.count && wg >= q.e[ei].nwg) { wg -= q.e[ei].nwg; ++ei; }
  const FoWSide& p = q.e[ei];
  const int D = p.D, NG = 2 * D + 1;
  const unsigned char* img = reinterpret_cast<const unsigned char*>(p.img);
  const f32x4 Z = {0.f, 0.f, 0.f, 0.f};

  auto dma = [&](int g, uint32_t slot) { decw_dma_share(img + (size_t)g * 32768, voff, lds0 + slot * 32768u, 32, wv, 8); };
  auto sync = [&]() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  int gc = 0;                                  // groups consumed: group gc of the endless sequence (g = gc mod NG) sits in slot gc & 1
  const int total = NG * p.rep;
  auto boundary = [&]() {                      // the current group has landed for everybody; the next one goes to the other slot
    sync();
    if (gc + 1 < total) dma((gc + 1) % NG, (uint32_t)((gc + 1) & 1));
  };
  auto gemm = [&](const h16x8 (&x)[4], f32x4 (&c)[8]) {
    decw_gemm<false>((uint32_t)(uintptr_t)ring + (uint32_t)(gc & 1) * 32768u + voff, x, c);
  };
  // LayerNorm over the 128 channels of a row (32 per lane, four lanes per row) + ReLU -> bf16 operands of the four k-steps
  auto ln_relu = [&](const f32x4 (&v)[8], h16x8 (&xb)[4], const float* g, const float* b) {
    f32x4 s4 = (v[0] + v[1]) + (v[2] + v[3]);
    s4 += (v[4] + v[5]) + (v[6] + v[7]);
    const float mean = rows_sum((s4[0] + s4[1]) + (s4[2] + s4[3])) * (1.0f / 128.0f);
    f32x4 d[8];
    f32x4 q4 = Z;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) { d[nt] = v[nt] - mean; q4 += d[nt] * d[nt]; }
    const float r = rsqrtf(rows_sum((q4[0] + q4[1]) + (q4[2] + q4[3])) * (1.0f / 128.0f) + 1e-5f);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      f32x4 y[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int nt = 2 * ks + u;
        const float4 gg = *reinterpret_cast<const float4*>(g + nt * 16 + l4 * 4), bb = *reinterpret_cast<const float4*>(b + nt * 16 + l4 * 4);
        y[u] = d[nt] * ((f32x4){gg.x, gg.y, gg.z, gg.w} * r) + (f32x4){bb.x, bb.y, bb.z, bb.w};
        y[u] = (f32x4){fmaxf(y[u][0], 0.f), fmaxf(y[u][1], 0.f), fmaxf(y[u][2], 0.f), fmaxf(y[u][3], 0.f)};
      }
      xb[ks] = l0w_pack8(y[0], y[1]);
    }
  };

  dma(0, 0);
  {                                            // the parameter block: every load issued before the first LDS store (published by the first group barrier)
    float4 t[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) t[u] = tid + u * 512 < FOW_NPAR / 4 ? reinterpret_cast<const float4*>(p.par)[tid + u * 512] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 2; ++u) if (tid + u * 512 < FOW_NPAR / 4) reinterpret_cast<float4*>(par)[tid + u * 512] = t[u];
  }

#pragma unroll 1
  for (int r = 0; r < p.rep; ++r) {
    const int tile = (wg * p.rep + r) * 8 + wv;
    const int row = tile * 16 + l15;
    const bool ex = row < p.rows;
    float x[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if (d < D && ex) {
        float t = p.in[(size_t)row * p.in_ld + d];
        if (d == p.wrap_dim) { t = fmodf(t + RIFT_PI, 2.f * RIFT_PI); if (t < 0.f) t += 2.f * RIFT_PI; t -= RIFT_PI; }
        x[d] = t;
      }
    }
    f32x4 sum[8];
#pragma unroll 1
    for (int d = 0; d < D; ++d) {
      const float* pd = par + d * FOW_PAR_DIM;
      const float xd = d == 0 ? x[0] : d == 1 ? x[1] : x[2];
      boundary();                                                  // ---- mlps.d.0 on [cos | sin] (x_d itself: an fp32 rank-1 term in the accumulator)
      if (d == 0) {
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) { const float4 b = *reinterpret_cast<const float4*>(par + FOW_PAR_B3SUM + nt * 16 + l4 * 4); sum[nt] = (f32x4){b.x, b.y, b.z, b.w}; }
      }
      h16x8 fk[4];
      {
        // cos / sin(2 pi f x): v_sin / v_cos take revolutions, so only fract(f x) is needed; lane quarter l4 holds frequencies 8 l4 + j and 32 + 8 l4 + j
        float cs[16], sn[16];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float4 f0 = *reinterpret_cast<const float4*>(par + FOW_PAR_FREQ + d * 64 + 32 * h + 8 * l4), f1 = *reinterpret_cast<const float4*>(par + FOW_PAR_FREQ + d * 64 + 32 * h + 8 * l4 + 4);
          const float fr[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float rev = __builtin_amdgcn_fractf(xd * fr[j]);
            cs[8 * h + j] = __builtin_amdgcn_cosf(rev); sn[8 * h + j] = __builtin_amdgcn_sinf(rev);
          }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          fk[h] = l0w_from_u2(make_uint2(pack_h2(cs[8 * h], cs[8 * h + 1]), pack_h2(cs[8 * h + 2], cs[8 * h + 3])),
                              make_uint2(pack_h2(cs[8 * h + 4], cs[8 * h + 5]), pack_h2(cs[8 * h + 6], cs[8 * h + 7])));
          fk[2 + h] = l0w_from_u2(make_uint2(pack_h2(sn[8 * h], sn[8 * h + 1]), pack_h2(sn[8 * h + 2], sn[8 * h + 3])),
                                  make_uint2(pack_h2(sn[8 * h + 4], sn[8 * h + 5]), pack_h2(sn[8 * h + 6], sn[8 * h + 7])));
        }
      }
      f32x4 acc[8];
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const float4 b = *reinterpret_cast<const float4*>(pd + nt * 16 + l4 * 4), w = *reinterpret_cast<const float4*>(pd + 384 + nt * 16 + l4 * 4);
        acc[nt] = (f32x4){b.x + w.x * xd, b.y + w.y * xd, b.z + w.z * xd, b.w + w.w * xd};
      }
      gemm(fk, acc);
      h16x8 hb[4];
      ln_relu(acc, hb, pd + 128, pd + 256);
      ++gc;
      boundary();                                                  // ---- mlps.d.3, summed over the dims
      gemm(hb, sum);
      ++gc;
    }
    boundary();                                                    // ---- to_out: LayerNorm, ReLU, Linear
    h16x8 ob[4];
    ln_relu(sum, ob, par + FOW_PAR_OUT, par + FOW_PAR_OUT + 128);
    f32x4 acc[8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) { const float4 b = *reinterpret_cast<const float4*>(par + FOW_PAR_OUT + 256 + nt * 16 + l4 * 4); acc[nt] = (f32x4){b.x, b.y, b.z, b.w}; }
    gemm(ob, acc);
    ++gc;
    if (ex) {
      float* dst = p.Y + (size_t)row * 128 + l4 * 4;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        float4 v = make_float4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]);
        if (p.accumulate) { const float4 o = *reinterpret_cast<const float4*>(dst + nt * 16); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        *reinterpret_cast<float4*>(dst + nt * 16) = v;
      }
    }
  }
}

int fow_set_attributes() {
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(fo_w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, FOW_LDS_BYTES);
}

void fow_pack(const FoWSrc& src, unsigned short* img, float* par, hipStream_t stream) {
  const int n = (2 * src.D + 1) * 32 * 512;
  hipLaunchKernelGGL(pack_fow_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, src, img, par);
}

void fow_launch(const FoWP& p, hipStream_t stream) {
  int grid = 0;
  for (int i = 0; i < p.count; ++i) grid += p.e[i].nwg;
  if (grid > 0) hipLaunchKernelGGL(fo_w_kernel, dim3(grid), dim3(512), FOW_LDS_BYTES, stream, p);
}

}  // namespace RIFT_NS
