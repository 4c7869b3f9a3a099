// Diagnostic instrumentation of the train-mode stochastic operations (dropout, DropPath, state dropout).  Compiled in only with
// -DRIFT_DROP_STATS=1, which build.py uses for the separate librift_hip_stats.so that tests/test_gpu_dropstats.py loads; the product
// library carries none of it (the struct below is an unused, zeroed member of the kernel parameter blocks there).
//
// What the kernels record, next to applying their decisions as usual:
//   per-SAMPLE decisions (DropPath of the NAT blocks -- sample = agent sequence --, DropPath of the encoder layers -- sample = scene --,
//   state dropout of the ego token's six state tokens -- sample = (scene, token)): for every (site, sample) the number of lanes that
//   applied the decision, the OR and the AND of "kept" over those lanes (OR == AND: all rows of the sample saw ONE decision), and the keep
//   multiplier used;
//   per-ELEMENT decisions (the eight dropout sites of a decoder layer): kept / drawn counts per site, and the keep multiplier.
#pragma once
#include <hip/hip_runtime.h>

#include "opfmt.h"

#ifndef RIFT_DROP_STATS
#define RIFT_DROP_STATS 0
#endif

namespace RIFT_NS {

#define RIFT_DS_NAT(lv, bi, br) ((lv) * 4 + (bi) * 2 + (br))      // NAT level lv, block bi, branch br (0 attention, 1 MLP): 12 sites
#define RIFT_DS_ENC(li, br) (12 + (li) * 2 + (br))                // encoder layer li, branch br: 8 sites
#define RIFT_DS_EGO 20                                            // state dropout of the ego token (sample = scene * 6 + state token)
#define RIFT_DS_SITES 21
#define RIFT_DS_DEC_SITES 8   // r2r weights, r2r branch, m2m weights, m2m branch, cross weights, cross branch, FFN hidden, FFN branch

struct DropStats {
  unsigned int* cnt;            // [RIFT_DS_SITES][nmax]
  unsigned int* any;            // [RIFT_DS_SITES][nmax]  OR of kept
  unsigned int* all;            // [RIFT_DS_SITES][nmax]  AND of kept (initialised to all ones)
  float* scale;                 // [RIFT_DS_SITES + RIFT_DS_DEC_SITES] keep multiplier 1 / (1 - p) as the kernel used it
  unsigned long long* elem;     // [RIFT_DS_DEC_SITES][2] kept, drawn
  int nmax;
};

#if RIFT_DROP_STATS
__device__ __forceinline__ void ds_sample(const DropStats& d, int site, int sample, float mult) {
  if (!d.cnt || sample < 0 || sample >= d.nmax) return;
  const size_t i = (size_t)site * d.nmax + sample;
  atomicAdd(&d.cnt[i], 1u);
  atomicOr(&d.any[i], mult != 0.f ? 1u : 0u);
  atomicAnd(&d.all[i], mult != 0.f ? 0xffffffffu : 0u);
  if (mult != 0.f) d.scale[site] = mult;
}
#else
__device__ __forceinline__ void ds_sample(const DropStats&, int, int, float) {}
#endif

}  // namespace RIFT_NS
