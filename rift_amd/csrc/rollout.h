// GRPO advantage pipeline, candidate side (traj_eval/traj_evaluator.py:115-158,372-420 and
// traj_eval/track_propogate.py:16-780) for gfx950: reference-line deviation of every candidate trajectory and
// the 79-step closed-loop PID + kinematic-bicycle rollout with Savitzky-Golay kinematics and box corners.
// One lane = one candidate; the 79 steps are sequential per lane (registers + a per-wave LDS slab holding the
// candidate's 40-point reference path and its PID ring buffers), so a group of G = R*12 candidates is one or two
// wavefronts; the smoothed kinematics and box corners of the 80 frames are a second kernel, one thread per
// (candidate, frame).  Replaces 79 x ~40 tiny torch launches per CBV per tick.
#pragma once
#include "common.h"

namespace RIFT_NS {

// ---- reference-line deviation: one thread per (candidate, frame), ragged reference lines ----
__device__ __forceinline__ void ref_line_info_body(const float* __restrict__ traj /*(G,Tfull,6)*/, int G, int Tfull, int Ts, int M,
                                     const float* __restrict__ ref_pos /*(R,Pmax,2)*/, const float* __restrict__ ref_ang /*(R,Pmax)*/,
                                     const int* __restrict__ ref_len /*(R)*/, int Pmax, float* __restrict__ delta_dis,
                                     float* __restrict__ delta_angle, int* __restrict__ closest_idx) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= G * Ts) return;
  const int g = idx / Ts, t = idx - g * Ts, r = g / M;
  const float* p = traj + ((size_t)g * Tfull + t) * 6;
  const float x = p[0], y = p[1];
  const float cand_angle = atan2f(p[3], p[2]);
  const float* rp = ref_pos + (size_t)r * Pmax * 2;
  const int n = ref_len[r];
  float best = INFINITY; int bi = 0;
  for (int j = 0; j < n; ++j) {
    const float dx = x - rp[j * 2], dy = y - rp[j * 2 + 1];
    const float d = sqrtf(dx * dx + dy * dy);          // torch.norm then argmin: first minimum wins
    if (d < best) { best = d; bi = j; }
  }
  const float ca = ref_ang[(size_t)r * Pmax + bi];
  const float ad = cand_angle - ca;
  delta_angle[idx] = atan2f(sinf(ad), cosf(ad));
  const float rx = x - rp[bi * 2], ry = y - rp[bi * 2 + 1];
  delta_dis[idx] = -(rx * sinf(ca) - ry * cosf(ca));
  closest_idx[idx] = bi;
}
__global__ void ref_line_info_kernel(const float* __restrict__ traj, int G, int Tfull, int Ts, int M, const float* __restrict__ ref_pos, const float* __restrict__ ref_ang,
                                     const int* __restrict__ ref_len, int Pmax, float* __restrict__ delta_dis, float* __restrict__ delta_angle, int* __restrict__ closest_idx) {
  ref_line_info_body(traj, G, Tfull, Ts, M, ref_pos, ref_ang, ref_len, Pmax, delta_dis, delta_angle, closest_idx);
}

struct RolloutP {
  const float* traj;        // (G, Tfull, 6) raw candidate trajectories (x, y, cos, sin, vx, vy); first 40 frames used
  int G, Tfull, Gper;       // Gper candidates share one centre-vehicle state
  const float* state;       // (G/Gper, 6): x, y, heading, speed, width, length of the centre vehicle
  float* turn_buf; int* turn_ptr; int* turn_len;       // persistent PID state of BatchPIDTorch(1.25, .75, .3): (Gcap,20),(Gcap),(Gcap)
  float* speed_buf; int* speed_ptr; int* speed_len;    // BatchPIDTorch(5, .5, 1)
  float* center; float* angle; float* speed; float* acc; float* ang_vel; float* ang_acc;   // (G,80[,2])
  float* vertices;          // (G,80,4,2)
  int* closest_index;       // (G,79) closest reference index after every step (track_propogate.py:778)
  int* aim_idx;             // (G,79) PID aim waypoint index (track_propogate.py:468)
  float* raw_speed;         // (G,80) scratch: the unsmoothed speed history, from the closed-loop kernel to the kinematics kernel
};

#define RIFT_RO_T 40
#define RIFT_RO_LEN 80
#define RIFT_RO_LDS_BYTES ((2 * RIFT_RO_T + 40) * 64 * 4)

__device__ __forceinline__ float pid_step(float* buf /*[20][64] lane-interleaved*/, int& ptr, int& len, float err, int lane,
                                          float kp, float ki, float kd) {
  const float prev = buf[ptr * 64 + lane];
  buf[ptr * 64 + lane] = err;
  ptr = (ptr + 1) % 20;
  len = len + 1 > 20 ? 20 : len + 1;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 20; ++i) s += buf[i * 64 + lane];
  const float integral = s / (float)(len < 1 ? 1 : len);
  return kp * err + ki * integral + kd * (err - prev);
}

__global__ __launch_bounds__(64) void rollout_kernel(RolloutP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];   // RIFT_RO_LDS_BYTES
  float* s_rx = reinterpret_cast<float*>(smem_raw);          // [40][64] reference path x (lane-interleaved: conflict-free)
  float* s_ry = s_rx + RIFT_RO_T * 64;
  float* s_tb = s_ry + RIFT_RO_T * 64;                       // [20][64] turn-PID ring buffer
  float* s_sb = s_tb + 20 * 64;                              // [20][64] speed-PID ring buffer
  const int lane = threadIdx.x;
  const int g = blockIdx.x * 64 + lane;
  const bool live = g < p.G;
  const int gs = live ? g : p.G - 1;
  const float* st = p.state + (size_t)(gs / p.Gper) * 6;
  const float cx = st[0], cy = st[1], ch = st[2], v0 = st[3], width = st[4], length = st[5];
  // ---- global reference path of the candidate (get_center_rollout, traj_evaluator.py:115-153), incl. the
  // in-place aliasing quirk: only point 0 is moved to the origin
  const float cosh_ = cosf(ch), sinh_ = sinf(ch);
  float head0 = 0.f;
  for (int t = 0; t < RIFT_RO_T; ++t) {
    const float* q = p.traj + ((size_t)gs * p.Tfull + t) * 6;
    float x = q[0], y = q[1];
    if (t == 0) { x = 0.f; y = 0.f; head0 = atan2f(q[3], q[2]) + ch; }
    s_rx[t * 64 + lane] = (x * cosh_ + y * (-sinh_)) + cx;
    s_ry[t * 64 + lane] = (x * sinh_ + y * cosh_) + cy;
  }
  int tptr = p.turn_ptr[gs], tlen = p.turn_len[gs], sptr = p.speed_ptr[gs], slen = p.speed_len[gs];
  for (int i = 0; i < 20; ++i) { s_tb[i * 64 + lane] = p.turn_buf[(size_t)gs * 20 + i]; s_sb[i * 64 + lane] = p.speed_buf[(size_t)gs * 20 + i]; }

  float px = s_rx[lane], py = s_ry[lane], hd = head0, spd = v0;
  int closest = 0;
  if (live) { p.center[((size_t)g * RIFT_RO_LEN) * 2] = px; p.center[((size_t)g * RIFT_RO_LEN) * 2 + 1] = py; }
  // (the speed / heading histories go to memory: the Savitzky-Golay kinematics and the box corners of all 80 frames are a kernel of their
  // own, one thread per (candidate, frame) -- as the tail of this kernel they were 80 serial iterations of the one wave it runs as)
  if (live) { p.raw_speed[(size_t)g * RIFT_RO_LEN] = spd; p.angle[(size_t)g * RIFT_RO_LEN] = hd; }
  const float Lf = -0.090769015f, Lr = 1.4178275f, gain = 0.36848336f, dt = 0.1f;
  for (int step = 0; step < RIFT_RO_LEN - 1; ++step) {
    // ---- local waypoints 9, 19, 29 ahead of the closest reference point (get_local_traj_pos + [9::10])
    const float c = cosf(hd), s = sinf(hd);
    float wx[3], wy[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      int id = closest + 9 + 10 * k;
      id = id > RIFT_RO_T - 1 ? RIFT_RO_T - 1 : id;
      const float dx = s_rx[id * 64 + lane] - px, dy = s_ry[id * 64 + lane] - py;
      wx[k] = dx * c + dy * s;
      wy[k] = dx * (-s) + dy * c;
    }
    // ---- BatchPIDController.control_pid (track_propogate.py:436-491)
    const float n01 = sqrtf((wx[1] - wx[0]) * (wx[1] - wx[0]) + (wy[1] - wy[0]) * (wy[1] - wy[0]));
    const float n12 = sqrtf((wx[2] - wx[1]) * (wx[2] - wx[1]) + (wy[2] - wy[1]) * (wy[2] - wy[1]));
    const float desired_v = (n01 + n12) / 2.0f;
    const float aim_dist = fminf(fmaxf(0.5f * spd + 2.5f, 5.0f), 8.0f);
    const float nr0 = sqrtf(wx[0] * wx[0] + wy[0] * wy[0]), nr1 = sqrtf(wx[1] * wx[1] + wy[1] * wy[1]);
    const int aidx = (fabsf(nr1 - aim_dist) < fabsf(nr0 - aim_dist)) ? 1 : 0;
    const float ax = aidx ? wx[1] : wx[0], ay = aidx ? wy[1] : wy[0];
    const bool brake = (desired_v < 0.4f) || ((spd / fmaxf(desired_v, 1e-4f)) > 1.1f);
    const float delta = fminf(fmaxf(desired_v - spd, 0.0f), 0.25f);
    float throttle = pid_step(s_sb, sptr, slen, delta, lane, 5.0f, 0.5f, 1.0f);
    throttle = fminf(fmaxf(throttle, 0.0f), 0.75f);
    if (brake) throttle = 0.f;
    float ang = (atan2f(ay, ax) * (180.0f / 3.14159265358979323846f)) / 90.0f;
    if (spd < 0.01f || brake) ang = 0.f;
    float steer = pid_step(s_tb, tptr, tlen, ang, lane, 1.25f, 0.75f, 0.3f);
    steer = fminf(fmaxf(steer, -1.0f), 1.0f);
    // ---- BatchKinematicBicycleModel.forward (track_propogate.py:214-279)
    const float wheel = gain * steer;
    const float slip = atanf((Lr / (Lf + Lr)) * tanf(wheel));
    const float nx = px + spd * cosf(hd + slip) * dt;
    const float ny = py + spd * sinf(hd + slip) * dt;
    const float nh = hd + (spd / Lr) * sinf(slip) * dt;
    const float kph = spd * 3.6f;
    float nk = kph;
    if (brake) {
      const float bv[7] = {9.31711370e-03f, 8.20967431e-02f, -2.83832427e-03f, 5.06587474e-05f, -4.90357228e-07f,
                           2.44419284e-09f, -4.91381935e-12f};
      float pw = kph, acc_ = 0.f;
#pragma unroll
      for (int i = 0; i < 7; ++i) { acc_ += pw * bv[i]; pw *= kph; }
      nk = acc_;
    } else if (throttle >= 0.3f) {
      const float tv[8] = {9.63873001e-01f, 4.37535692e-04f, -3.80192912e-01f, 1.74950069e+00f, 9.16787414e-02f,
                           -7.05461530e-02f, -1.05996152e-03f, 6.71079346e-04f};
      const float v = kph, v2 = v * v, t = throttle, t2 = t * t;
      nk = v * tv[0] + v2 * tv[1] + t * tv[2] + t2 * tv[3] + (v * t) * tv[4] + (v * t2) * tv[5] + (v2 * t) * tv[6] + (v2 * t2) * tv[7];
    }
    const float ns = fmaxf(nk / 3.6f, 0.0f);
    px = nx; py = ny; hd = nh; spd = ns;
    // ---- find_closest_ref_pos: argmin of the squared distance over the 40 reference points (first minimum)
    float best = INFINITY; int bi = 0;
    for (int j = 0; j < RIFT_RO_T; ++j) {
      const float dx = s_rx[j * 64 + lane] - px, dy = s_ry[j * 64 + lane] - py;
      const float d = dx * dx + dy * dy;
      if (d < best) { best = d; bi = j; }
    }
    closest = bi;
    if (live) {
      p.raw_speed[(size_t)g * RIFT_RO_LEN + step + 1] = spd; p.angle[(size_t)g * RIFT_RO_LEN + step + 1] = hd;
      p.center[((size_t)g * RIFT_RO_LEN + step + 1) * 2] = px; p.center[((size_t)g * RIFT_RO_LEN + step + 1) * 2 + 1] = py;
      p.closest_index[(size_t)g * (RIFT_RO_LEN - 1) + step] = bi;
      p.aim_idx[(size_t)g * (RIFT_RO_LEN - 1) + step] = aidx;
    }
  }
  if (live) {
    p.turn_ptr[g] = tptr; p.turn_len[g] = tlen; p.speed_ptr[g] = sptr; p.speed_len[g] = slen;
    for (int i = 0; i < 20; ++i) { p.turn_buf[(size_t)g * 20 + i] = s_tb[i * 64 + lane]; p.speed_buf[(size_t)g * 20 + i] = s_sb[i * 64 + lane]; }
  }
}

// The same closed loop with EIGHT lanes per candidate: every lane of a group runs the whole step on identical state (in lockstep that costs
// nothing), except the 40-point closest-point search -- a quarter of a step's instructions -- of which each lane takes five points; the group's
// (distance, index) pairs are reduced with the sequential loop's tie rule (first minimum = smallest index).  Bit-identical to rollout_kernel
// (tests/test_gpu_properties.py); 72 candidates are nine waves on nine CUs instead of one full wave and one of eight lanes.
__global__ __launch_bounds__(64) void rollout8_kernel(RolloutP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];   // RIFT_RO_LDS_BYTES
  float* s_rx = reinterpret_cast<float*>(smem_raw);          // [40][64] reference path x (lane-interleaved: conflict-free)
  float* s_ry = s_rx + RIFT_RO_T * 64;
  float* s_tb = s_ry + RIFT_RO_T * 64;                       // [20][64] turn-PID ring buffer
  float* s_sb = s_tb + 20 * 64;                              // [20][64] speed-PID ring buffer
  const int lane = threadIdx.x, sub = lane & 7;
  const int g = blockIdx.x * 8 + (lane >> 3);
  const bool live = g < p.G, writer = live && sub == 0;
  const int gs = live ? g : p.G - 1;
  const float* st = p.state + (size_t)(gs / p.Gper) * 6;
  const float cx = st[0], cy = st[1], ch = st[2], v0 = st[3], width = st[4], length = st[5];
  // ---- global reference path of the candidate (get_center_rollout, traj_evaluator.py:115-153), incl. the
  // in-place aliasing quirk: only point 0 is moved to the origin
  const float cosh_ = cosf(ch), sinh_ = sinf(ch);
  float head0 = 0.f;
  for (int t = 0; t < RIFT_RO_T; ++t) {
    const float* q = p.traj + ((size_t)gs * p.Tfull + t) * 6;
    float x = q[0], y = q[1];
    if (t == 0) { x = 0.f; y = 0.f; head0 = atan2f(q[3], q[2]) + ch; }
    s_rx[t * 64 + lane] = (x * cosh_ + y * (-sinh_)) + cx;
    s_ry[t * 64 + lane] = (x * sinh_ + y * cosh_) + cy;
  }
  int tptr = p.turn_ptr[gs], tlen = p.turn_len[gs], sptr = p.speed_ptr[gs], slen = p.speed_len[gs];
  for (int i = 0; i < 20; ++i) { s_tb[i * 64 + lane] = p.turn_buf[(size_t)gs * 20 + i]; s_sb[i * 64 + lane] = p.speed_buf[(size_t)gs * 20 + i]; }

  float px = s_rx[lane], py = s_ry[lane], hd = head0, spd = v0;
  int closest = 0;
  if (writer) { p.center[((size_t)g * RIFT_RO_LEN) * 2] = px; p.center[((size_t)g * RIFT_RO_LEN) * 2 + 1] = py; }
  // (the speed / heading histories go to memory: the Savitzky-Golay kinematics and the box corners of all 80 frames are a kernel of their
  // own, one thread per (candidate, frame) -- as the tail of this kernel they were 80 serial iterations of the one wave it runs as)
  if (writer) { p.raw_speed[(size_t)g * RIFT_RO_LEN] = spd; p.angle[(size_t)g * RIFT_RO_LEN] = hd; }
  const float Lf = -0.090769015f, Lr = 1.4178275f, gain = 0.36848336f, dt = 0.1f;
  for (int step = 0; step < RIFT_RO_LEN - 1; ++step) {
    // ---- local waypoints 9, 19, 29 ahead of the closest reference point (get_local_traj_pos + [9::10])
    const float c = cosf(hd), s = sinf(hd);
    float wx[3], wy[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      int id = closest + 9 + 10 * k;
      id = id > RIFT_RO_T - 1 ? RIFT_RO_T - 1 : id;
      const float dx = s_rx[id * 64 + lane] - px, dy = s_ry[id * 64 + lane] - py;
      wx[k] = dx * c + dy * s;
      wy[k] = dx * (-s) + dy * c;
    }
    // ---- BatchPIDController.control_pid (track_propogate.py:436-491)
    const float n01 = sqrtf((wx[1] - wx[0]) * (wx[1] - wx[0]) + (wy[1] - wy[0]) * (wy[1] - wy[0]));
    const float n12 = sqrtf((wx[2] - wx[1]) * (wx[2] - wx[1]) + (wy[2] - wy[1]) * (wy[2] - wy[1]));
    const float desired_v = (n01 + n12) / 2.0f;
    const float aim_dist = fminf(fmaxf(0.5f * spd + 2.5f, 5.0f), 8.0f);
    const float nr0 = sqrtf(wx[0] * wx[0] + wy[0] * wy[0]), nr1 = sqrtf(wx[1] * wx[1] + wy[1] * wy[1]);
    const int aidx = (fabsf(nr1 - aim_dist) < fabsf(nr0 - aim_dist)) ? 1 : 0;
    const float ax = aidx ? wx[1] : wx[0], ay = aidx ? wy[1] : wy[0];
    const bool brake = (desired_v < 0.4f) || ((spd / fmaxf(desired_v, 1e-4f)) > 1.1f);
    const float delta = fminf(fmaxf(desired_v - spd, 0.0f), 0.25f);
    float throttle = pid_step(s_sb, sptr, slen, delta, lane, 5.0f, 0.5f, 1.0f);
    throttle = fminf(fmaxf(throttle, 0.0f), 0.75f);
    if (brake) throttle = 0.f;
    float ang = (atan2f(ay, ax) * (180.0f / 3.14159265358979323846f)) / 90.0f;
    if (spd < 0.01f || brake) ang = 0.f;
    float steer = pid_step(s_tb, tptr, tlen, ang, lane, 1.25f, 0.75f, 0.3f);
    steer = fminf(fmaxf(steer, -1.0f), 1.0f);
    // ---- BatchKinematicBicycleModel.forward (track_propogate.py:214-279)
    const float wheel = gain * steer;
    const float slip = atanf((Lr / (Lf + Lr)) * tanf(wheel));
    const float nx = px + spd * cosf(hd + slip) * dt;
    const float ny = py + spd * sinf(hd + slip) * dt;
    const float nh = hd + (spd / Lr) * sinf(slip) * dt;
    const float kph = spd * 3.6f;
    float nk = kph;
    if (brake) {
      const float bv[7] = {9.31711370e-03f, 8.20967431e-02f, -2.83832427e-03f, 5.06587474e-05f, -4.90357228e-07f,
                           2.44419284e-09f, -4.91381935e-12f};
      float pw = kph, acc_ = 0.f;
#pragma unroll
      for (int i = 0; i < 7; ++i) { acc_ += pw * bv[i]; pw *= kph; }
      nk = acc_;
    } else if (throttle >= 0.3f) {
      const float tv[8] = {9.63873001e-01f, 4.37535692e-04f, -3.80192912e-01f, 1.74950069e+00f, 9.16787414e-02f,
                           -7.05461530e-02f, -1.05996152e-03f, 6.71079346e-04f};
      const float v = kph, v2 = v * v, t = throttle, t2 = t * t;
      nk = v * tv[0] + v2 * tv[1] + t * tv[2] + t2 * tv[3] + (v * t) * tv[4] + (v * t2) * tv[5] + (v2 * t) * tv[6] + (v2 * t2) * tv[7];
    }
    const float ns = fmaxf(nk / 3.6f, 0.0f);
    px = nx; py = ny; hd = nh; spd = ns;
    // ---- find_closest_ref_pos: argmin of the squared distance over the 40 reference points (first minimum)
    float best = INFINITY; int bi = 0;
#pragma unroll
    for (int jj = 0; jj < RIFT_RO_T / 8; ++jj) {
      const int j = sub * (RIFT_RO_T / 8) + jj;
      const float dx = s_rx[j * 64 + lane] - px, dy = s_ry[j * 64 + lane] - py;
      const float d = dx * dx + dy * dy;
      if (d < best) { best = d; bi = j; }
    }
    // (the group's eight pairs through DPP: lane ^ 1, lane ^ 2 inside the quad, then the mirrored lane of the other quad -- every lane ends with the group's pair)
#define RIFT_RO8_STEP(CTRL) do { const float ob = dpp_f<CTRL>(best); const int oj = __builtin_amdgcn_update_dpp(0, bi, CTRL, 0xF, 0xF, true); \
                                 if (ob < best || (ob == best && oj < bi)) { best = ob; bi = oj; } } while (0)
    RIFT_RO8_STEP(0xB1); RIFT_RO8_STEP(0x4E); RIFT_RO8_STEP(0x141);
#undef RIFT_RO8_STEP
    closest = bi;
    if (writer) {
      p.raw_speed[(size_t)g * RIFT_RO_LEN + step + 1] = spd; p.angle[(size_t)g * RIFT_RO_LEN + step + 1] = hd;
      p.center[((size_t)g * RIFT_RO_LEN + step + 1) * 2] = px; p.center[((size_t)g * RIFT_RO_LEN + step + 1) * 2 + 1] = py;
      p.closest_index[(size_t)g * (RIFT_RO_LEN - 1) + step] = bi;
      p.aim_idx[(size_t)g * (RIFT_RO_LEN - 1) + step] = aidx;
    }
  }
  if (writer) {
    p.turn_ptr[g] = tptr; p.turn_len[g] = tlen; p.speed_ptr[g] = sptr; p.speed_len[g] = slen;
    for (int i = 0; i < 20; ++i) { p.turn_buf[(size_t)g * 20 + i] = s_tb[i * 64 + lane]; p.speed_buf[(size_t)g * 20 + i] = s_sb[i * 64 + lane]; }
  }
}

// ---- derive_kinematics (track_propogate.py:500-596): SG(5,2) smoothing with reflect padding, central differences; box corners FL, RL, RR, FR
// (track_propogate.py:16-74).  One thread per (candidate, frame): every output is a pure function of the two histories.
__device__ __forceinline__ void rollout_kinematics_body(const RolloutP& p) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= p.G * RIFT_RO_LEN) return;
  const int g = e / RIFT_RO_LEN, t = e - g * RIFT_RO_LEN;
  const float* spd = p.raw_speed + (size_t)g * RIFT_RO_LEN;
  const float* ang = p.angle + (size_t)g * RIFT_RO_LEN;          // (the heading output IS the raw history)
  const float* st = p.state + (size_t)(g / p.Gper) * 6;
  const float width = st[4], length = st[5], dt = 0.1f;
  auto refl = [](int i) { return i < 0 ? -i : (i >= RIFT_RO_LEN ? 2 * (RIFT_RO_LEN - 1) - i : i); };
  const float k0 = -3.0f / 35.0f, k1 = 12.0f / 35.0f, k2 = 17.0f / 35.0f;
  auto sg = [&](const float* a, int i) {
    return k0 * a[refl(i - 2)] + k1 * a[refl(i - 1)] + k2 * a[i] + k1 * a[refl(i + 1)] + k0 * a[refl(i + 2)];
  };
  auto wrapf = [](float d) { return atan2f(sinf(d), cosf(d)); };
  auto yaw_rate = [&](int i) {
    if (i == 0) return wrapf(sg(ang, 1) - sg(ang, 0)) / dt;
    if (i == RIFT_RO_LEN - 1) return wrapf(sg(ang, RIFT_RO_LEN - 1) - sg(ang, RIFT_RO_LEN - 2)) / dt;
    return wrapf(sg(ang, i + 1) - sg(ang, i - 1)) / (2.0f * dt);
  };
  const float sp = sg(spd, t);
  float ac;
  if (t == 0) ac = (sg(spd, 1) - sp) / dt;
  else if (t == RIFT_RO_LEN - 1) ac = (sp - sg(spd, t - 1)) / dt;
  else ac = (sg(spd, t + 1) - sg(spd, t - 1)) / (2.0f * dt);
  const float yr_cur = yaw_rate(t);
  float ya;
  if (t == 0) ya = (yaw_rate(1) - yr_cur) / dt;
  else if (t == RIFT_RO_LEN - 1) ya = (yr_cur - yaw_rate(t - 1)) / dt;
  else ya = (yaw_rate(t + 1) - yaw_rate(t - 1)) / (2.0f * dt);
  const size_t o = (size_t)g * RIFT_RO_LEN + t;
  const float h = ang[t];
  p.speed[o] = sp; p.acc[o] = ac; p.ang_vel[o] = yr_cur; p.ang_acc[o] = ya;
  const float hw = 0.5f * width, hl = 0.5f * length;
  const float cc = cosf(h), ss = sinf(h);
  const float ctx = p.center[o * 2], cty = p.center[o * 2 + 1];
  const float dxs[4] = {hl, -hl, -hl, hl}, dys[4] = {hw, hw, -hw, -hw};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    p.vertices[(o * 4 + k) * 2] = (dxs[k] * cc - dys[k] * ss) + ctx;
    p.vertices[(o * 4 + k) * 2 + 1] = (dxs[k] * ss + dys[k] * cc) + cty;
  }
}
__global__ __launch_bounds__(256) void rollout_kinematics_kernel(RolloutP p) { rollout_kinematics_body(p); }

}  // namespace RIFT_NS
