// Token assembly of the scene encoder (agent_encoder.py:87-96, map_encoder.py:82-91): the parameter block and the two value functions, shared by
// token_kernel (kernels.h), the fused encoder's optional in-kernel assembly (enc_fused.h) and its 112-row build (enc112.hip).  No kernels here:
// every translation unit may include it.
#pragma once
#include "common.h"

namespace RIFT_NS {

// agent tokens: x[b][a] = (a == 0 ? x_ego[b] : valid ? nat[b*A+a] : 0) + type_emb[cat] (+ positional embedding of token row b*N + a); four channels
__device__ __forceinline__ float4 agent_token_value(const int b, const int a, const int c, const float* __restrict__ nat, const float* __restrict__ x_ego,
                                                    const uint8_t* __restrict__ valid_agent, const int8_t* __restrict__ category,
                                                    const float* __restrict__ type_emb, int A, int N, const float* __restrict__ pe) {
  const int ag = b * A + a;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a == 0) v = *reinterpret_cast<const float4*>(x_ego + (size_t)b * 128 + c);
  else if (valid_agent[ag]) v = *reinterpret_cast<const float4*>(nat + (size_t)ag * 128 + c);
  const float4 t = *reinterpret_cast<const float4*>(type_emb + (int)category[ag] * 128 + c);
  v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
  if (pe) { const float4 q = *reinterpret_cast<const float4*>(pe + ((size_t)b * N + a) * 128 + c); v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
  return v;
}

// polygon tokens (map_encoder.py:82-91): x = pooled + type + on_route + tl + (has ? speed_emb : unknown) (+ positional embedding of row b*N + A + m)
__device__ __forceinline__ float4 polygon_token_value(const int b, const int m, const int c, const float* __restrict__ pooled, const int8_t* __restrict__ ptype,
                                                      const uint8_t* __restrict__ on_route, const int8_t* __restrict__ tl,
                                                      const uint8_t* __restrict__ has_sl, const float* __restrict__ speed_emb,
                                                      const float* __restrict__ type_emb, const float* __restrict__ route_emb,
                                                      const float* __restrict__ tl_emb, const float* __restrict__ unk_emb,
                                                      int A, int Mp, int N, const float* __restrict__ pe) {
  const int pg = b * Mp + m;
  auto ld = [&](const float* p) { return *reinterpret_cast<const float4*>(p + c); };
  const float4 a0 = ld(pooled + (size_t)pg * 128), a1 = ld(type_emb + (int)ptype[pg] * 128), a2 = ld(route_emb + (on_route[pg] ? 1 : 0) * 128),
               a3 = ld(tl_emb + (int)tl[pg] * 128), a4 = has_sl[pg] ? ld(speed_emb + (size_t)pg * 128) : ld(unk_emb);
  // same association as the scalar form: ((pooled + type) + route) + tl, then + speed
  float4 v = make_float4(((a0.x + a1.x) + a2.x) + a3.x, ((a0.y + a1.y) + a2.y) + a3.y, ((a0.z + a1.z) + a2.z) + a3.z, ((a0.w + a1.w) + a2.w) + a3.w);
  v.x += a4.x; v.y += a4.y; v.z += a4.z; v.w += a4.w;
  if (pe) { const float4 q = *reinterpret_cast<const float4*>(pe + ((size_t)b * N + A + m) * 128 + c); v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
  return v;
}

// agent and polygon tokens of the scene encoder in one launch (blocks [0, nblk_a) build agent tokens): two ~5 us launches in a row cost their
// fixed part twice
struct TokenP {
  const float *nat, *x_ego; const uint8_t* valid_agent; const int8_t* category; const float* a_type_emb;
  const float* pooled; const int8_t* ptype; const uint8_t* on_route; const int8_t* tl; const uint8_t* has_sl;
  const float *speed_emb, *p_type_emb, *route_emb, *tl_emb, *unk_emb;
  int bs, A, Mp, N, nblk_a; float* X; const float* pe;
};

}  // namespace RIFT_NS
