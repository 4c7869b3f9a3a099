// Fused ego state token for gfx950 (StateAttentionEncoder, agent_encoder.py:99-140): six scalar-state tokens -> K | V projection
// (bf16 MFMA) -> 4-head attention of the learned query over the six tokens with the state-dropout key mask (fp32) -> out_proj.
// One workgroup per scene; replaces five launches (token build, K|V GEMM, drop mask, attention, out_proj GEMM).
#pragma once
#include "enc_fused.h"

namespace RIFT_NS {

struct EgoP {
  const float* cs; int cs_ld;        // (bs, cs_ld) current_state, first 6 entries used
  const float* lw; const float* lb;  // (6, 128) per-token Linear(1, 128) weight / bias
  const float* pos;                  // pos_embed (6, 128)
  const unsigned short* wkv; const float* bkv;   // in_proj rows 128:384, fragment-major bf16 [256][128]
  const float* q;                    // (128) projected query (weight-only, cached)
  const unsigned short* wo; const float* bo;     // out_proj [128][128]
  float* out;                        // (bs, 128)
  int bs;
  float drop_p;                      // state dropout probability (0 = off): tokens 3..5 are masked with this probability
  uint32_t seed, stream;
  DropStats ds;                      // diagnostic build only (dropstats.h)
};

#define EGO_ES 136
#define EGO_KS 260
#define EGO_LDS_BYTES (2 * 16 * EGO_ES * 2 + 6 * EGO_KS * 4 + 16)
// (smem: EGO_LDS_BYTES of 16-byte aligned LDS)
__device__ __forceinline__ void ego_body(const EgoP& p, const int b, unsigned char* smem) {
  constexpr int ES = EGO_ES, KS_ = EGO_KS;
  unsigned short* e = reinterpret_cast<unsigned short*>(smem);
  unsigned short* ao = e + 16 * ES;
  float* kvs = reinterpret_cast<float*>(ao + 16 * ES);
  unsigned char* msk = reinterpret_cast<unsigned char*>(kvs + 6 * KS_);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  // (K | V weights in two halves of 128 output columns: 32 fragment registers at a time instead of 64, so that the kernel stays within 64 VGPRs)
  EFrags<4, 2> Wkv;
  e_load_b(Wkv, p.wkv, 128, 0, 0, wave, l15, l4);
  float tv[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int i = tid + k * 256, tok = i >> 7, c = i & 127;
    tv[k] = (p.cs[(size_t)b * p.cs_ld + tok] * p.lw[tok * 128 + c] + p.lb[tok * 128 + c]) + p.pos[tok * 128 + c];
  }
  const float qv = p.q[(wave * 32 + (lane & 31))];
#pragma unroll
  for (int k = 0; k < 3; ++k) { const int i = tid + k * 256; e[(i >> 7) * ES + (i & 127)] = f2h(tv[k]); }
  for (int i = tid; i < 10 * 128; i += 256) e[(6 + (i >> 7)) * ES + (i & 127)] = 0;
  for (int i = tid; i < 16 * 128; i += 256) ao[(i >> 7) * ES + (i & 127)] = 0;
  if (tid < 6) msk[tid] = (p.drop_p > 0.f && tid >= 3 && uniform01(p.seed, p.stream, (uint32_t)(b * 6 + tid)) < p.drop_p) ? 1 : 0;
  if (tid < 6 && p.drop_p > 0.f) ds_sample(p.ds, RIFT_DS_EGO, b * 6 + tid, msk[tid] ? 0.f : 1.f);      // (a masked key, no rescaling: agent_encoder.py:119-129)
  __syncthreads();
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    if (hf) e_load_b(Wkv, p.wkv, 128, 128, 0, wave, l15, l4);
    f32x4 acc[1][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[0][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    e_mma<1, 4, 2>(acc, e, ES, Wkv, l15, l4);
    if (l15 < 6) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = 128 * hf + (j * 4 + wave) * 16 + l4 * 4;
        const float4 b4 = *reinterpret_cast<const float4*>(p.bkv + col);
        *reinterpret_cast<float4*>(kvs + l15 * KS_ + col) = make_float4(acc[0][j][0] + b4.x, acc[0][j][1] + b4.y, acc[0][j][2] + b4.z, acc[0][j][3] + b4.w);
      }
    }
  }
  EFrags<4, 2> Wo;
  e_load_b(Wo, p.wo, 128, 0, 0, wave, l15, l4);
  __syncthreads();
  {   // head = wave; lanes 0..31 (and their mirror 32..63) own one of the 32 head dims
    const int d = wave * 32 + (lane & 31);
    float s[6], mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float t = group_sum<32>(qv * kvs[j * KS_ + d]) * 0.17677669529663687f;
      if (msk[j]) t = -INFINITY;
      s[j] = t; mx = fmaxf(mx, t);
    }
    float den = 0.f, o = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) { const float w = __expf(s[j] - mx); den += w; o += w * kvs[j * KS_ + 128 + d]; }
    if (lane < 32) ao[d] = f2h(o / den);
  }
  __syncthreads();
  {
    f32x4 acc[1][2];
    acc[0][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[0][1] = acc[0][0];
    e_mma<1, 4, 2>(acc, ao, ES, Wo, l15, l4);
    if (l15 == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = (j * 4 + wave) * 16 + l4 * 4;
        const float4 b4 = *reinterpret_cast<const float4*>(p.bo + col);
        *reinterpret_cast<float4*>(p.out + (size_t)b * 128 + col) = make_float4(acc[0][j][0] + b4.x, acc[0][j][1] + b4.y, acc[0][j][2] + b4.z, acc[0][j][3] + b4.w);
      }
    }
  }
}
// (64 VGPRs: a workgroup that fits on a CU BESIDE the decoder's -- 224 VGPRs x 2 waves per SIMD leave 64 -- so that the ego tokens of step k + 1 are
// done inside the decoder of step k instead of opening the map chain behind it, like the preparation and the ranking)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void ego_fused_kernel(EgoP p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[EGO_LDS_BYTES];
  ego_body(p, blockIdx.x, smem);
}

}  // namespace RIFT_NS
