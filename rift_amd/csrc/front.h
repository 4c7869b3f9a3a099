// (RIFT_FRONT_FUSED=1; measured, NOT the default -- engine.hip says why.)  The head of a forward in ONE launch (round 4): input preparation (kernels.h: prep_body), the ranking of the history encoder's sequences
// (nat_l0w.h: nat_rank_body, from the raw validity instead of the preparation's marks) and the ego state token (ego_fused.h: ego_body, which
// reads nothing but current_state).  Round 3 ran them as three launches: the ranking (12.6 us of ONE workgroup) behind the preparation on the
// history chain, the ego token (17 us, one workgroup per scene) at the head of the map chain -- and this round's packing experiment (DESIGN.md
// section 4) showed that at a chip-filling batch the step waits for those chains' LATENCY: a single-workgroup launch on a chain costs its
// whole duration, a block of a launch that is there anyway costs nothing.  Block 0 is the ranking (the longest block, dispatched first), blocks
// [1, 1 + bs) the ego tokens, the rest the preparation.
#pragma once
#include "ego_fused.h"
#include "kernels.h"
#include "nat_l0w.h"

namespace RIFT_NS {

struct FrontP {
  PrepP prep;
  EgoP ego; int n_ego;                   // 0: no ego blocks (the ego token runs as its own launch)
  int* aidx; int* cnt; int rank_on;      // 0: no ranking block
};

// EGO = false: preparation + ranking only -- a kernel of <= 64 VGPRs, which still fits on a CU beside the decoder's workgroups (224 VGPRs x 2
// waves per SIMD leave 64), as prep_kernel and nat_rank_kernel did; with the ego blocks (MFMA weight fragments: 116 VGPRs) it does not.
template <bool EGO>
__global__ __launch_bounds__(256) void front_kernel(FrontP q) {
  static_assert(NAT_RANK_THREADS == 256, "one block size for all roles");
  __shared__ __attribute__((aligned(16))) unsigned char smem[EGO ? EGO_LDS_BYTES : 64];
  int blk = blockIdx.x;
  if (q.rank_on) {
    if (blk == 0) {
      nat_rank_body(nullptr, q.prep.nA, q.aidx, q.cnt, reinterpret_cast<unsigned long long*>(smem), q.prep.agent_valid, q.prep.Tfull, q.prep.A);
      return;
    }
    --blk;
  }
  if (EGO) {
    if (blk < q.n_ego) { ego_body(q.ego, blk, smem); return; }
    blk -= q.n_ego;
  }
  prep_body(q.prep, blk);
}

}  // namespace RIFT_NS
