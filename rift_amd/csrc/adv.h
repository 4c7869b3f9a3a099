// Advantage / return kernels of the RLFT update (gfx950): GAE and discounted-return reverse
// scans as wavefront scans of affine maps, buffer-wide and per-group z-scores with shuffle
// reductions, the dense-reward rollout return, and the replay-arena gather (collation).
// All of them are latency / HBM bound scalar work -- no MFMA.
#pragma once
#include "common.h"

namespace RIFT_NS {

// ---------------------------------------------------------------------------
// Reverse affine scan  x_t = b_t + a_t * x_{t+1},  x_n = 0, in fp64, by ONE workgroup of 16 waves:
// thread i owns the contiguous chunk [i*ch, (i+1)*ch); pass 1 composes the chunk's map x_hi -> x_lo, a DPP/shuffle scan from the
// right composes the maps inside each wave, the 16 wave composites are chained through LDS, pass 2 replays the chunk from its
// incoming value.  A 4096-step PPO buffer is 4 steps per thread (the reference walks it as a 4096-iteration Python loop).
// ---------------------------------------------------------------------------
struct GaeCoef {    // get_advantages_GAE, ppo_datamodule.py:22-37 (dtype promotion mirrored: fp64 rewards, fp32 rest)
  const double* rewards; const float* undones; const float* values; const float* next_values; const float* unterminated;
  float gamma, lambda_;
  __device__ __forceinline__ void get(int t, double& a, double& b) const {
    const float t2 = (unterminated[t] * gamma) * next_values[t];
    b = rewards[t] + (double)t2 - (double)values[t];
    a = (double)((undones[t] * gamma) * lambda_);
  }
};
struct ReturnCoef { // compute_return, reinforce_datamodule.py:19-38
  const double* rewards; const float* dones; double gamma;
  __device__ __forceinline__ void get(int t, double& a, double& b) const {
    b = rewards[t];
    a = (dones[t] == 1.0f) ? 0.0 : gamma;
  }
};

#define RIFT_SCAN_THREADS 1024
template <class Coef, class OutT>
__global__ __launch_bounds__(RIFT_SCAN_THREADS) void affine_scan_reverse_kernel(Coef c, int n, OutT* __restrict__ out) {
  constexpr int NWV = RIFT_SCAN_THREADS / 64;
  __shared__ double sA[NWV], sB[NWV];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ch = (n + RIFT_SCAN_THREADS - 1) / RIFT_SCAN_THREADS;
  const int lo = min(n, tid * ch), hi = min(n, lo + ch);
  // chunk composite: x_lo = B + A * x_hi   (identity for an empty chunk)
  double A = 1.0, B = 0.0;
  for (int t = hi - 1; t >= lo; --t) { double a, b; c.get(t, a, b); B = b + a * B; A = a * A; }
  // inclusive scan from the right inside the wave: (A,B) then maps x at the wave's right end to x at this lane's chunk start
  for (int o = 1; o < 64; o <<= 1) {
    const double Ar = __shfl_down(A, o, 64), Br = __shfl_down(B, o, 64);
    if (lane + o < 64) { B = B + A * Br; A = A * Ar; }
  }
  if (lane == 0) { sA[wave] = A; sB[wave] = B; }
  __syncthreads();
  double xw = 0.0;                                   // x at this wave's right end: the later waves' maps applied to x_n = 0
  for (int w = NWV - 1; w > wave; --w) xw = sB[w] + sA[w] * xw;
  // incoming value of this chunk = x at the start of the next lane's chunk
  const double An = __shfl_down(A, 1, 64), Bn = __shfl_down(B, 1, 64);
  double x = lane == 63 ? xw : Bn + An * xw;
  for (int t = hi - 1; t >= lo; --t) { double a, b; c.get(t, a, b); x = b + a * x; out[t] = (OutT)x; }
}

// (x - mean) / (std_unbiased + 1e-5) in place over n fp32 values (ppo_datamodule.py:166); one workgroup.
__global__ __launch_bounds__(256) void normalize_unbiased_kernel(float* __restrict__ x, int n) {
  __shared__ double s_a[4], s_b[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double s = 0.0;
  for (int i = tid; i < n; i += 256) s += (double)x[i];
  s = wave_sum_d(s);
  if (lane == 0) s_a[wave] = s;
  __syncthreads();
  const double mean = (s_a[0] + s_a[1] + s_a[2] + s_a[3]) / (double)n;
  double q = 0.0;
  for (int i = tid; i < n; i += 256) { const double d = (double)x[i] - mean; q += d * d; }
  q = wave_sum_d(q);
  if (lane == 0) s_b[wave] = q;
  __syncthreads();
  const double var = (s_b[0] + s_b[1] + s_b[2] + s_b[3]) / (double)(n > 1 ? n - 1 : 1);
  const float meanf = (float)mean, den = (float)sqrt(var) + 1e-5f;
  for (int i = tid; i < n; i += 256) x[i] = (x[i] - meanf) / den;
}

// GRPO group z-score (traj_evaluator.py:467-470): one wave per group of G fp64 returns, ddof 0, +1e-5.
__device__ __forceinline__ void group_zscore_body(const double* __restrict__ ret, int n_groups, int G, double* __restrict__ adv) {
  const int g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (g >= n_groups) return;
  double s = 0.0;
  for (int i = lane; i < G; i += 64) s += ret[(size_t)g * G + i];
  const double mean = wave_sum_d(s) / (double)G;
  double q = 0.0;
  for (int i = lane; i < G; i += 64) { const double d = ret[(size_t)g * G + i] - mean; q += d * d; }
  const double sd = sqrt(wave_sum_d(q) / (double)G) + 1e-5;
  for (int i = lane; i < G; i += 64) adv[(size_t)g * G + i] = (ret[(size_t)g * G + i] - mean) / sd;
}
__global__ void group_zscore_kernel(const double* __restrict__ ret, int n_groups, int G, double* __restrict__ adv) { group_zscore_body(ret, n_groups, G, adv); }

// Dense reward (gym_carla/reward/reward_model.py:34-50) with the dtype promotion of the
// reference environment (numpy 1.24: np.float32 scalar (op) python float -> float64; f32 (op) f32 -> f32).
__device__ __forceinline__ double dense_reward(float dd_abs, float da_abs, float speed, float acc, float ang_acc,
                                               int collision, int offroad) {
  const float aspeed = fabsf(speed);
  const double r_collision = -(20.0 + (double)aspeed) * (double)collision;
  const double r_offroad = -5.0 * (double)offroad;
  const double r_comfort = -0.8 * (double)((fabsf(acc) > 4.f ? 1 : 0) + (fabsf(ang_acc) > 4.f ? 1 : 0));
  const float c = cosf(da_abs);
  const float cs = c * speed;                                  // f32 * f32
  const double r_l_align = 0.5 * ((double)fminf(c, 0.f) + 0.05 * (double)fminf(cs, 0.f) +
                                  0.25 * (1.0 - (double)da_abs / (3.141592653589793 / 2.0)));
  const double dd = (double)dd_abs;                            // abs(delta_dis - 0.0) in f64
  const double r_l_center = -0.6 * (double)(c > 0.5f ? 1 : 0) * (dd - 0.05 / exp(dd - 0.5));
  const double r_velocity = 0.1 * (double)fmaxf(c, 0.f) * (double)((aspeed > 3.f && aspeed < 20.f) ? 1 : 0) * (double)aspeed;
  const double r_timestep = -0.1 * (double)((aspeed > 0.f || fabsf(acc) > 0.f) ? 1 : 0);
  return r_collision + r_offroad + r_comfort + r_l_align + r_l_center + r_velocity + r_timestep;
}

// get_rollout_return (traj_evaluator.py:333-370): one WAVE per candidate, lane j = time step j (Ts <= 64): the 7-term dense reward
// and gamma^j of every step in parallel, the first colliding step from a ballot (steps after it are dropped, the colliding step
// itself counts), one wave sum in fp64.  (One lane per candidate walking 40 steps with a double pow each took 39 us per group.)
__device__ __forceinline__ void rollout_return_body(const float* __restrict__ delta_dis, const float* __restrict__ delta_angle,
                                      const float* __restrict__ speed, const float* __restrict__ acc,
                                      const float* __restrict__ ang_vel, const float* __restrict__ ang_acc,
                                      const uint8_t* __restrict__ collision, int col_ld,
                                      const uint8_t* __restrict__ off_road, int off_ld, int G, int Ts, double gamma,
                                      double* __restrict__ ret, int ro_ld) {          // ro_ld: row stride of speed / acc / ang_acc (Ts, or the rollout's 80)
  const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= G) return;
  (void)ang_vel;
  double r = 0.0;
  for (int j0 = 0; j0 < Ts; j0 += 64) {                     // one round for Ts <= 64
    const int j = j0 + lane;
    const bool live = j < Ts;
    const size_t o = (size_t)i * Ts + (live ? j : 0), o2 = (size_t)i * ro_ld + (live ? j : 0);
    const int col = (live && collision[(size_t)i * col_ld + j]) ? 1 : 0;
    const int off = (live && off_road[(size_t)i * off_ld + j]) ? 1 : 0;
    const unsigned long long hit = __ballot(col);
    const int first = hit ? __ffsll((long long)hit) - 1 : 64;          // lane of the first collision of this round
    double term = 0.0;
    if (live && lane <= first)
      term = dense_reward(fabsf(delta_dis[o]), fabsf(delta_angle[o]), speed[o2], acc[o2], ang_acc[o2], col, off) * pow(gamma, (double)j);
    r += wave_sum_d(term);
    if (hit) break;
  }
  if (lane == 0) ret[i] = r;
}
__global__ __launch_bounds__(256) void rollout_return_kernel(const float* __restrict__ delta_dis, const float* __restrict__ delta_angle,
                                      const float* __restrict__ speed, const float* __restrict__ acc,
                                      const float* __restrict__ ang_vel, const float* __restrict__ ang_acc,
                                      const uint8_t* __restrict__ collision, int col_ld,
                                      const uint8_t* __restrict__ off_road, int off_ld, int G, int Ts, double gamma,
                                      double* __restrict__ ret, int ro_ld) {
  rollout_return_body(delta_dis, delta_angle, speed, acc, ang_vel, ang_acc, collision, col_ld, off_road, off_ld, G, Ts, gamma, ret, ro_ld);
}

// get_other_vehicle_rollout (traj_evaluator.py:160-239): constant-control kinematic-bicycle forecast of the nearby actors
// (KinematicBicycleModel.forecast_other_vehicles, rift/ego/pdm_lite/kinematic_bicycle_model.py:33-62, constants from
// rift/ego/pdm_lite/config.py:186-199,336-347), speed-dependent footprint inflation, corners FL RL RR FR in the right-handed frame
// (compute_agents_vertices, traj_evaluator.py:33-79).  One thread per actor, T sequential steps, all in fp64 as numpy computes it.
__device__ __forceinline__ void other_vehicle_rollout_body(const double* __restrict__ actions /*(N,3) steer throttle brake*/,
                                                                   const double* __restrict__ speed, const double* __restrict__ location /*(N,3)*/,
                                                                   const double* __restrict__ yaw_deg, const double* __restrict__ extent /*(N,2)*/,
                                                                   int N, int T, int near_lane_change, double inflation,
                                                                   double* __restrict__ vertices /*(N,T,4,2)*/) {
  const int a = blockIdx.x * 64 + threadIdx.x;
  if (a >= N) return;
  const double dt = 0.1, Lf = -0.090769015, Lr = 1.4178275, gain = 0.36848336, brake_acc = -4.952399, thr_acc = 0.5633837;
  const double slow_f = 1.0, v_thr = 1.0, min_y = 1.0, fac_y = 1.3, min_x = 1.2, min_x_lc = 2.0;
  const double steer = actions[3 * a], throttle = actions[3 * a + 1];
  const bool brake = ((unsigned char)actions[3 * a + 2]) != 0;                 // .astype(np.uint8): truncation
  const double slip = atan(Lr / (Lf + Lr) * tan(gain * steer));
  double x = location[3 * a], y = location[3 * a + 1], h = yaw_deg[a] * (3.14159265358979323846 / 180.0), v = speed[a];
  const double s = near_lane_change ? min_x_lc : min_x;
  const double acc = brake ? brake_acc : __dmul_rn(throttle, thr_acc);
  for (int i = 0; i < T; ++i) {
    const double nx = __dadd_rn(x, __dmul_rn(__dmul_rn(v, cos(h + slip)), dt));
    const double ny = __dadd_rn(y, __dmul_rn(__dmul_rn(v, sin(h + slip)), dt));
    const double nh = __dadd_rn(h, __dmul_rn(__dmul_rn(v / Lr, sin(slip)), dt));
    const double nv = fmax(0.0, __dadd_rn(v, __dmul_rn(dt, acc)));
    x = nx; y = ny; h = nh; v = nv;
    const bool slow = v < v_thr;
    const double fr = (double)i / (double)T;
    double ex = extent[2 * a] * (slow ? slow_f : fmax(s, min_x * fr));
    double ey = extent[2 * a + 1] * (slow ? slow_f : fmax(min_y, fac_y * fr));
    ex *= inflation; ey *= inflation;
    const double hw = (ey * 2) / 2, hl = (ex * 2) / 2;
    const double cx = x, cy = -y, c = cos(-h), sn = sin(-h);
    const double ol[4] = {hl, -hl, -hl, hl}, ow[4] = {hw, hw, -hw, -hw};
    double* o = vertices + ((size_t)a * T + i) * 8;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      o[2 * k] = __dadd_rn(__dadd_rn(__dmul_rn(ol[k], c), __dmul_rn(ow[k], -sn)), cx);
      o[2 * k + 1] = __dadd_rn(__dadd_rn(__dmul_rn(ol[k], sn), __dmul_rn(ow[k], c)), cy);
    }
  }
}
__global__ __launch_bounds__(64) void other_vehicle_rollout_kernel(const double* __restrict__ actions, const double* __restrict__ speed, const double* __restrict__ location,
                                                                   const double* __restrict__ yaw_deg, const double* __restrict__ extent, int N, int T, int near_lane_change,
                                                                   double inflation, double* __restrict__ vertices) {
  other_vehicle_rollout_body(actions, speed, location, yaw_deg, extent, N, T, near_lane_change, inflation, vertices);
}

// get_collision_matrix (traj_evaluator.py:241-275): the reference builds an STRtree of the other vehicles' footprints per step and
// calls tree.query(ego_polygon) WITHOUT a predicate, which returns the geometries whose ENVELOPES intersect the candidate's
// envelope (closed intervals: touching counts) -- so collision[g][j] = any_n AABB(center[g][j]) overlaps AABB(other[n][j]).
// One thread per (candidate, step); candidate vertices fp32 (exact in fp64), other vertices fp64 as numpy holds them.
__device__ __forceinline__ void collision_matrix_body(const float* __restrict__ cv /*(G,Tc,4,2)*/, int G, int Tc,
                                                               const double* __restrict__ ov /*(N,Ts,4,2)*/, int N, int Ts,
                                                               uint8_t* __restrict__ out /*(G,Ts)*/) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= G * Ts) return;
  const int g = i / Ts, j = i - g * Ts;
  const float* e = cv + ((size_t)g * Tc + j) * 8;
  double ex0 = e[0], ex1 = e[0], ey0 = e[1], ey1 = e[1];
#pragma unroll
  for (int k = 1; k < 4; ++k) { ex0 = fmin(ex0, (double)e[2 * k]); ex1 = fmax(ex1, (double)e[2 * k]); ey0 = fmin(ey0, (double)e[2 * k + 1]); ey1 = fmax(ey1, (double)e[2 * k + 1]); }
  uint8_t hit = 0;
  for (int n = 0; n < N; ++n) {
    const double* o = ov + ((size_t)n * Ts + j) * 8;
    double ox0 = o[0], ox1 = o[0], oy0 = o[1], oy1 = o[1];
#pragma unroll
    for (int k = 1; k < 4; ++k) { ox0 = fmin(ox0, o[2 * k]); ox1 = fmax(ox1, o[2 * k]); oy0 = fmin(oy0, o[2 * k + 1]); oy1 = fmax(oy1, o[2 * k + 1]); }
    if (!(ox0 > ex1 || ox1 < ex0 || oy0 > ey1 || oy1 < ey0)) { hit = 1; break; }
  }
  out[i] = hit;
}
__global__ __launch_bounds__(256) void collision_matrix_kernel(const float* __restrict__ cv, int G, int Tc, const double* __restrict__ ov, int N, int Ts, uint8_t* __restrict__ out) {
  collision_matrix_body(cv, G, Tc, ov, N, Ts, out);
}

// get_off_road_matrix, lookup part (traj_evaluator.py:299-318): pixel = round(((p - origin) . rot) / resolution_hw + offset) in fp64
// (np.round = half to even = rint), inside the raster and mask == 1 -> off road.  rot = [[c, -s], [s, c]] of the centre heading.
__device__ __forceinline__ void off_road_body(const float* __restrict__ pts /*(n,2)*/, int n, const uint8_t* __restrict__ mask, int H,
                                                       int W, double ox, double oy, double c, double s, double res_x, double res_y,
                                                       double off_x, double off_y, uint8_t* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double dx = (double)pts[2 * i] - ox, dy = (double)pts[2 * i + 1] - oy;
  const double lx = __dadd_rn(__dmul_rn(dx, c), __dmul_rn(dy, s));        // (d . rot)[0] = dx*c + dy*s   (no fused multiply-add, as numpy)
  const double ly = __dadd_rn(__dmul_rn(dx, -s), __dmul_rn(dy, c));       // (d . rot)[1] = -dx*s + dy*c
  const double px = rint(lx / res_x + off_x), py = rint(ly / res_y + off_y);
  uint8_t off = 0;
  if (px >= 0.0 && px < (double)W && py >= 0.0 && py < (double)H) off = mask[(size_t)(int)py * W + (int)px] == 1;
  out[i] = off;
}
__global__ __launch_bounds__(256) void off_road_kernel(const float* __restrict__ pts, int n, const uint8_t* __restrict__ mask, int H, int W, double ox, double oy, double c, double s,
                                                       double res_x, double res_y, double off_x, double off_y, uint8_t* __restrict__ out) {
  off_road_body(pts, n, mask, H, W, ox, oy, c, s, res_x, res_y, off_x, off_y, out);
}

// ---------------------------------------------------------------------------
// Replay-arena gather (collation): for tensor k, scene b:
//   dst_k[b*dst_bytes .. +dst_bytes) = src_k[idx[b]*src_bytes .. +dst_bytes)
// (the ragged dimension R leads each per-scene block, so cropping to the batch
// maximum is a prefix copy; arena rows beyond a scene's own count are stored as zeros).
// ---------------------------------------------------------------------------
#define RIFT_COLLATE_MAXT 40
struct CollateP {
  int nt;
  const unsigned char* src[RIFT_COLLATE_MAXT];
  unsigned char* dst[RIFT_COLLATE_MAXT];
  int src_bytes[RIFT_COLLATE_MAXT];
  int dst_bytes[RIFT_COLLATE_MAXT];
};

__global__ void collate_kernel(CollateP p, const int32_t* __restrict__ scene_idx) {
  const int k = blockIdx.y, b = blockIdx.x;
  const unsigned char* s = p.src[k] + (size_t)scene_idx[b] * p.src_bytes[k];
  unsigned char* d = p.dst[k] + (size_t)b * p.dst_bytes[k];
  const int n = p.dst_bytes[k];
  if ((((uintptr_t)s | (uintptr_t)d | (uintptr_t)n) & 15) == 0) {
    const uint4* s4 = reinterpret_cast<const uint4*>(s);
    uint4* d4 = reinterpret_cast<uint4*>(d);
    for (int i = threadIdx.x; i < (n >> 4); i += blockDim.x) d4[i] = s4[i];
  } else if ((((uintptr_t)s | (uintptr_t)d | (uintptr_t)n) & 3) == 0) {
    const uint32_t* s4 = reinterpret_cast<const uint32_t*>(s);
    uint32_t* d4 = reinterpret_cast<uint32_t*>(d);
    for (int i = threadIdx.x; i < (n >> 2); i += blockDim.x) d4[i] = s4[i];
  } else {
    for (int i = threadIdx.x; i < n; i += blockDim.x) d[i] = s[i];
  }
}

}  // namespace RIFT_NS
