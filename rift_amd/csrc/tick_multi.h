// The rollout tick's per-CBV kernels with the CBV as a second grid dimension (rift_group_advantage_tick, engine.hip): everything of the
// evaluation chain except the candidate rollout itself is independent between the CBVs of a tick, so one launch per STAGE serves all of
// them -- reference-line deviations and neighbour forecasts ahead of the rollouts, kinematics / collision / off-road flags, returns and
// z-scores behind them.  The rollouts stay one launch per CBV: CBV k + 1 starts from the PID state CBV k leaves (the reference shares one
// never-reset controller).  The bodies are the single-CBV kernels' (adv.h, rollout.h), unchanged: same arithmetic, same results.
#pragma once
#include "adv.h"
#include "rollout.h"

namespace RIFT_NS {

#define RIFT_TICK_CHUNK 8                        // CBVs per launch (the descriptors travel as kernel arguments)

struct TickK {                                   // one CBV: inputs and its own slices of the library's scratch
  const float* traj; int G, Tfull, Pmax, N, H, W;
  const float *ref_pos, *ref_ang; const int* ref_len;
  float *dd, *da; int* ci;                       // (G, 40)
  const double* actors; double* ov;              // [actions | speed | location | yaw | extent]; (N, 40, 4, 2)
  RolloutP ro;                                   // rollout outputs (and the closed-loop kernel's own arguments)
  const uint8_t* mask; double ox, oy, ch, sh;    // raster, origin, cos / sin of its heading
  uint8_t *col, *offr;                           // (G, 40), (G, 80)
  double *ret, *adv;                             // (G), (G)
};
struct TickArr { TickK d[RIFT_TICK_CHUNK]; int K; double gamma; };

// STAGE 0: reference-line deviations (block 128)   1: neighbour forecast (block 64)   2: kinematics + corners (block 256)
//       3: collision flags (256)   4: off-road flags (256)   5: returns (256)   6: z-score (256)
template <int STAGE>
__global__ void tick_multi_kernel(const TickArr a) {
  if ((int)blockIdx.y >= a.K) return;
  const TickK& d = a.d[blockIdx.y];
  constexpr int Ts = 40, M = 12, TR = RIFT_RO_LEN;
  if (STAGE == 0) ref_line_info_body(d.traj, d.G, d.Tfull, Ts, M, d.ref_pos, d.ref_ang, d.ref_len, d.Pmax, d.dd, d.da, d.ci);
  if (STAGE == 1) { if (d.N > 0) { const double* p = d.actors; const size_t N = (size_t)d.N; other_vehicle_rollout_body(p, p + 3 * N, p + 4 * N, p + 7 * N, p + 8 * N, d.N, Ts, 1, 1.1, d.ov); } }
  if (STAGE == 2) rollout_kinematics_body(d.ro);
  if (STAGE == 3) collision_matrix_body(d.ro.vertices, d.G, TR, d.ov, d.N, Ts, d.col);          // (no neighbours: the loop over them is empty, every flag 0)
  if (STAGE == 4) {
    if (d.mask) off_road_body(d.ro.center, d.G * TR, d.mask, d.H, d.W, d.ox, d.oy, d.ch, d.sh, 0.5, -0.5, 200.0, 200.0, d.offr);
    else { const int i = blockIdx.x * 256 + threadIdx.x; if (i < d.G * TR) d.offr[i] = 0; }
  }
  if (STAGE == 5) rollout_return_body(d.dd, d.da, d.ro.speed, d.ro.acc, d.ro.ang_vel, d.ro.ang_acc, d.col, Ts, d.offr, TR, d.G, Ts, a.gamma, d.ret, TR);
  if (STAGE == 6) group_zscore_body(d.ret, 1, d.G, d.adv);
}

}  // namespace RIFT_NS
