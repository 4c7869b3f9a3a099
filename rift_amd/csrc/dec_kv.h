// Cross-attention K | V^T operand fragments of the planning decoder for batches whose scene encoder ran layer-wise (more than 96 tokens: the
// dense-traffic shapes): the job the fused encoder kernel's tail does for the standard shapes (enc_fused.h), here as its own launch from
// the encoder output in global memory.  One workgroup per scene; the scene's tokens (N <= 192) are staged as bf16 operand rows, the four
// layers' K | V projections (planning_decoder.py:74-79, in_proj rows 128:384) run as MFMA tiles and are stored in the per-head fragment
// order dec_w_kernel<., true> streams: per (scene, layer) 4 heads x (12 K fragments (key tile kt) | 12 V^T fragments (dim tile d, key
// pair pt)), a fragment = [lane][8] with k slot j of lane (l15, l4) <-> dim / key 16 (j / 4) + 4 l4 + j % 4.
#pragma once
#include "enc_fused.h"

namespace RIFT_NS {

struct DecKvP {
  const float* ENC;             // (bs*N, 128) encoder output
  int bs, N;                    // N <= 192
  const unsigned short* wkv;    // fragment-major bf16 [4 * 256][128]: per layer (k 128 rows | v 128 rows)
  const float* bkv;             // [4 * 256]
  unsigned short* KV;           // (bs, 4, 96, 512)
};

#define DEC_KV_XN 144
#define DEC_KV_LDS (192 * DEC_KV_XN * 2)

__global__ __launch_bounds__(512) void dec_kv_frag_kernel(DecKvP p) {
  constexpr int MT = 12, XN = DEC_KV_XN, NW = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* xn = reinterpret_cast<unsigned short*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int b = blockIdx.x, N = p.N;
  for (int i = tid; i < 192 * 32; i += 512) {
    const int r = i >> 5, c4 = (i & 31) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);          // keys beyond N: zero rows (masked by the decoder; finite operands)
    if (r < N) v = *reinterpret_cast<const float4*>(p.ENC + ((size_t)b * N + r) * 128 + c4);
    *reinterpret_cast<uint2*>(xn + r * XN + c4) = pack_h4(v.x, v.y, v.z, v.w);
  }
  __syncthreads();
  for (int l = 0; l < 4; ++l) {
    EFrags<4, 2> Wk;                                      // n-tile `wave` of K (swapped order) and n-tile 8 + wave = V (plain order)
    e_load_b(Wk, p.wkv, 128, l * 256, 0, wave, l15, l4, EWaves<NW>());
    f32x4 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    e_mma<MT, 4, 2, 1>(acc, xn, XN, Wk, l15, l4);
    const int h = wave >> 1;
    unsigned short* base = p.KV + (((size_t)b * 4 + l) * 96 + h * 24) * 512 + lane * 8;
    {   // K: this lane holds channels wave * 16 + 4 l4 .. + 3 of key mt * 16 + l15 = k slots (wave & 1) * 4 .. + 3 of fragment kt = mt
      const float4 b4 = *reinterpret_cast<const float4*>(p.bkv + l * 256 + wave * 16 + l4 * 4);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        *reinterpret_cast<uint2*>(base + mt * 512 + (wave & 1) * 4) =
            pack_h4(acc[mt][0][0] + b4.x, acc[mt][0][1] + b4.y, acc[mt][0][2] + b4.z, acc[mt][0][3] + b4.w);
    }
    {   // V^T: channel wave * 16 + l15 (dim tile d = wave & 1), keys mt * 16 + 4 l4 .. + 3 = k slots (mt & 1) * 4 .. of fragment (d, pt = mt >> 1)
      const float bias = p.bkv[l * 256 + 128 + wave * 16 + l15];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        *reinterpret_cast<uint2*>(base + (12 + (wave & 1) * 6 + (mt >> 1)) * 512 + (mt & 1) * 4) =
            pack_h4(acc[mt][1][0] + bias, acc[mt][1][1] + bias, acc[mt][1][2] + bias, acc[mt][1][3] + bias);
    }
  }
}

}  // namespace RIFT_NS
