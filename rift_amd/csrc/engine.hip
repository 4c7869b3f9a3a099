// librift_hip.so : host orchestration + C-ABI of the MI355X-native RIFT policy-update path.
// See include/rift_hip.h for the interface and DESIGN.md for the kernel inventory.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <set>
#include <cmath>
#include <string>
#include <unordered_map>
#include <dlfcn.h>
#include <vector>

#include "../../include/rift_hip.h"
#include "adv.h"
#include "gemm.h"
#include "kernels.h"
#include "loss.h"
#include "nat_l0w.h"
#include "nat_l1w.h"
#include "enc_fused.h"
#include "enc_fused.h"
#include "dec_w.h"
#include "dec_kv.h"
#include "nat_l2w.h"
#include "enc_w.h"
#include "pe_w.h"
#include "fo_w.h"
#include "pe_fused.h"
#include "fourier_fused.h"
#include "critic.h"
#include "fpn_fused.h"
#include "ego_fused.h"
#include "front.h"
#include "heads_fused.h"
#include "pi_fused.h"
#include "rollout.h"
#include "tick_multi.h"

#define ENC_NW 8

using namespace RIFT_NS;

namespace {

struct Param { void* data; int64_t numel; int ndim; int64_t shape[4]; };

struct PW {   // packed weight image of one linear map Y = X W^T (+ b)
  void* bf = nullptr; float* f32 = nullptr; const float* bias = nullptr;
  int N = 0, K = 0, Kp = 0, Npad = 0;
  void* hid = nullptr;          // (fc2 weights of the fused encoder only) the same image in hidden-layer operand words (opfmt.h); pack_hid
};

struct Tap { float* p; int64_t numel; };

}  // namespace

struct RiftCtx {
  int opfmt = 0;                         // FIRST member (set by rift_ctx_create of the build): abi.cpp reads it to route a call to the build (bf16 / fp16 operands) that owns the context
  int device = 0;
  std::string err;
  std::unordered_map<std::string, Param> params;
  std::unordered_map<std::string, PW> pw;
  std::vector<void*> owned;          // packed weight allocations
  struct WConst { float* p = nullptr; bool ready = false; };
  std::unordered_map<std::string, WConst> wconst;   // activations that depend on (frozen) weights only, computed once per model load
  // activation arena (bump allocator, reset every forward).  A forward with RIFT_F_DEFER_HEAD alternates between two arenas, so that the
  // policy head / loss / backward of step k (rift_forward_head, rift_loss_backward on another stream) read step k's activations while the
  // frozen trunk of step k + 1 already writes its own.
  char* arena = nullptr; size_t arena_cap = 0, arena_off = 0;
  long long dry_key[11] = {}; size_t dry_need = 0;      // the arena size of the last sized forward and what it depended on
  char* arenas[RIFT_DEFER_SLOTS] = {}; size_t arena_caps[RIFT_DEFER_SLOTS] = {}; int parity = 0;     // (parity: the arena of the current forward)
  struct Head {   // what the policy head of a forward needs (pi_forward .. trajectory heads); kept per arena for the deferred form
    bool valid = false, fp32 = false, need_traj = false;
    float *Q = nullptr, *x0p = nullptr, *QF = nullptr, *Hpi = nullptr, *prob = nullptr, *traj = nullptr, *o3[3] = {nullptr, nullptr, nullptr}, *oT[3] = {nullptr, nullptr, nullptr};
    uint8_t* r_kpm = nullptr; int bs = 0, R = 0, nQ = 0;
    bool dec_pending = false; DecWP dec;      // the planning decoder deferred along with the head (small batches, see dec_defer_max)
  } head[RIFT_DEFER_SLOTS];
  // (round 4) Below a chip-filling batch a step is the LATENCY of token assembly -> encoder -> decoder on the caller's queue (one workgroup per
  // scene whatever the batch) while most CUs idle.  Nothing behind the encoder belongs to the frozen trunk's critical path of the NEXT
  // step, so with the head deferred the decoder goes with it: rift_forward_head_back launches decoder -> head on the caller's update
  // stream, and the caller's queue holds token assembly -> encoder of step k + 1 beside them.  All its operands live in the forward's
  // arena (four slots).  Which batch sizes: forward_impl's measured table, or bs <= RIFT_DEC_DEFER (0 = never).
  int dec_defer_max = -1; bool dec_split = true;   // (-1: the measured table in forward_impl; RIFT_DEC_DEFER=<n>: bs <= n)            // (RIFT_DEC_SPLIT=0: the deferred decoder as one launch)
  bool ro8 = true;                                         // candidate rollout with eight lanes per candidate (rollout.h; RIFT_RO8=0: one lane, the round-1 form)
  float* ro_raw = nullptr; size_t ro_cap = 0;              // rift_rollout: the unsmoothed speed history handed from the closed-loop kernel to the kinematics kernel
  char* tick_scratch = nullptr; size_t tick_cap = 0;       // rift_group_advantage_tick: per-CBV intermediates (reused CBV by CBV in stream order)
  // (Measured and not kept: the deferred decoder on a third stream of the engine's own, so that decoder k would also run beside tail k - 1 --
  // 0.21 -> 0.40 ms at 32 scenes, with 4 or 8 hardware queues; every stream beyond the four the step uses has made cross-queue waits slower.)
  bool dry = false;
  hipStream_t stream = nullptr;
  std::unordered_map<std::string, Tap> taps;
  // state of the last forward, consumed by rift_loss_backward
  float* last_qfinal = nullptr; float* last_hpi = nullptr; float* last_prob = nullptr;
  uint8_t* last_rkpm = nullptr; int last_bs = 0, last_R = 0;
  // loss scratch
  double* l_S = nullptr; double* l_cnt = nullptr; float* l_dz = nullptr; float* l_partial = nullptr;
  size_t l_cap_bs = 0, l_cap_rows = 0, l_cap_wg = 0;
  float* ego_w = nullptr; float* ego_b = nullptr;   // packed (6,128) linears of StateAttentionEncoder
  bool nat_fused = true; int gemm_dbg = 0;
  unsigned short* l0w_img = nullptr; float* l0w_par = nullptr;   // wave-private level-0 NAT kernel (nat_l0w.h)
  unsigned short* encw_img = nullptr; float* encw_par = nullptr;   // weight stream / parameters of the dense-traffic scene encoder (enc_w.h)
  unsigned short* l2w_img = nullptr; float* l2w_par = nullptr;   // wave-private, weight-streaming level-2 NAT kernel (nat_l2w.h)
  unsigned short* l1w_img = nullptr; float* l1w_par = nullptr;   // wave-private level-1 NAT kernel (nat_l1w.h)
  unsigned short* enc_wqkv[4] = {nullptr, nullptr, nullptr, nullptr};   // chunked (q|k|q|k|v|v) bf16 in_proj images
  float* enc_bqkv[4] = {nullptr, nullptr, nullptr, nullptr};
  int* enc_idx = nullptr; bool enc_fused = true;
  int poison_lds = -1;                   // RIFT_POISON_LDS diagnostic (see lds_poison_kernel)
  // the per-call diagnostic switches of the environment, read ONCE when the context is made (nine getenv calls per forward were ~10 us of a
  // call whose host time bounds a small-batch step): RIFT_{PE,PEW,NAT,ENC,DEC}_TS, RIFT_{PEW,DEC}_DBG, RIFT_POISON_ARENA
  struct { std::string delay_label; long long delay_ticks = 0; int pe_ts = -1, pew_dbg = 0, pew_ts = 0, nat_ts = 0, enc_ts = 0, dec_ts = 0, dec_dbg = 0, poison_arena = -1; } dg;
  hipEvent_t param_event = nullptr;      // rift_set_param_event: the trainable parameters are valid once this event has passed
  bool dec_fused = true;
  bool two_streams = true; bool nat_on_main = true; bool nat_compact = true; bool pe_live = true; bool pe_pack = true; bool tok_fused = false; bool keep_tokens = false; bool front_fused = false; bool front_ego = false; bool ego_nofit = false;
  hipStream_t prep_stream = nullptr; bool prep_set = false; hipEvent_t ev_prep = nullptr; int side_gate = 0;      // rift_set_prepare_stream
  hipEvent_t ev_join2 = nullptr; bool nat_aside = true; int join_once = -1;      // (the history chain behind the preparation on the prepare stream: its join event)
  hipStream_t side = nullptr; bool side_owned = false; hipEvent_t ev_fork = nullptr, ev_join = nullptr;   // (RIFT_TWO_STREAMS=0 switches it off) the agent-history chain (NAT levels + FPN tail) on a second stream beside the map / reference-line chain
  bool fo_w = true; unsigned short* fow_img[3] = {nullptr, nullptr, nullptr}; float* fow_par[3] = {nullptr, nullptr, nullptr};   // wave-private Fourier embeddings (fo_w.h): tokens, speed limits, reference-line positions
  bool pe_w = true; unsigned short* pew_img[2] = {nullptr, nullptr};   // wave-private PointsEncoder pass B (pe_w.h): weight streams of the map / reference-line encoders
  unsigned short* decw_img = nullptr; float* decw_par = nullptr;   // weight stream / parameter blocks of the decoder kernel (dec_w.h)
  double* clip_part = nullptr;
  struct Dp { bool on = false; int off = 0, gbs = 0; double* xchg = nullptr; long long len = 0; RiftExchangeFn fn = nullptr; void* user = nullptr; } dp;   // rift_set_dp
  int* nonfinite = nullptr;              // device flag set by the policy-head kernels when the decoder output is not finite
  // the in-launch ranking's published per-scene counts (kernels.h: rank_scene_body), one array per arena; they persist between launches
  // (a word is valid when it carries the launch's epoch; zero at allocation, epochs start at 1)
  unsigned long long* rk_pub[RIFT_DEFER_SLOTS] = {}; int rk_cap[RIFT_DEFER_SLOTS] = {}; unsigned int rk_epoch[RIFT_DEFER_SLOTS] = {};
  bool rank_fault = false;               // RIFT_RANK_FAULT=1 (diagnostic): the first scene block of the in-launch ranking never publishes its counts
  bool enc112 = true;                    // RIFT_ENC112=0: scenes of 97 .. 112 token slots on enc_w_kernel (rounds 3 - 5) instead of the fused kernel's 112-row layout
  bool rank_in_prep = true;              // RIFT_RANK_IN_PREP=0: the ranking as its own launch behind the preparation (nat_rank_kernel, rounds 3 - 5)
  void* comm = nullptr; int comm_rank = 0, comm_world = 1;      // library-owned RCCL communicator (rift_comm_init), or null
  float* cr_buf = nullptr; size_t cr_cap = 0; double* cr_part = nullptr;   // PPO critic scratch (rows x 1153 floats)
  bool pe_fused = true; bool fo_fused = true; int nat_grid = 256; bool fpn_fused = true; bool ego_fused = true; bool heads_fused = true; bool pi_fused = true;
  bool loaded = false;
  // optional per-launch HIP-event profiling (bench roofline leg; off on the timed path)
  bool prof_on = false; double prof_flops = 0.0; bool prof_shapes = false;
  std::set<std::string> prof_names;
  std::vector<hipEvent_t> prof_pool; size_t prof_used = 0;
  struct Rec { const char* label; hipEvent_t e0, e1; double flops; };
  std::vector<Rec> prof_recs;
};

static void prof_events(RiftCtx* c, hipEvent_t* e0, hipEvent_t* e1) {
  while (c->prof_pool.size() < c->prof_used + 2) { hipEvent_t e; (void)hipEventCreate(&e); c->prof_pool.push_back(e); }
  *e0 = c->prof_pool[c->prof_used++]; *e1 = c->prof_pool[c->prof_used++];
}
static void prof_push(RiftCtx* c, const char* label, hipEvent_t e0, hipEvent_t e1, double flops) {
  c->prof_recs.push_back(RiftCtx::Rec{label, e0, e1, flops});
}

#define HIPCHK(ctx, expr)                                                                 \
  do {                                                                                    \
    hipError_t e__ = (expr);                                                              \
    if (e__ != hipSuccess) {                                                              \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e__);                    \
      return RIFT_ERR_HIP;                                                                \
    }                                                                                     \
  } while (0)

namespace {

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

template <class T>
T* A_alloc(RiftCtx* c, size_t n) {
  size_t bytes = (n * sizeof(T) + 255) & ~(size_t)255;
  char* p = c->dry ? nullptr : c->arena + c->arena_off;
  c->arena_off += bytes;
  return reinterpret_cast<T*>(p);
}

void tap(RiftCtx* c, const char* name, float* p, int64_t numel) {
  if (!c->dry) c->taps[name] = Tap{p, numel};
}

// diagnostic (RIFT_POISON_LDS=<byte>): LDS keeps its contents between kernels, so a kernel that reads LDS it never wrote sees whatever
// the previous kernel on that CU left there.  With the switch set every launch is preceded by a kernel that fills all 160 KB of every
// CU's LDS with the byte (0xFF = NaN pattern): such a read then shows up in the parity tests instead of depending on history.
__global__ __launch_bounds__(256) void lds_poison_kernel(unsigned int pattern) {
  extern __shared__ unsigned int lds_all[];
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 256) lds_all[i] = pattern;
  __syncthreads();
  if (lds_all[threadIdx.x] != pattern) asm volatile("s_nop 0");      // keep the stores
}

// diagnostic (RIFT_DELAY=<launch label>:<microseconds>): one lane spins for that long on the stream right behind every launch of that
// label -- the successors of the kernel start later, no CU is taken from anybody.  d(step time) / d(delay) is the kernel's share of the
// step's critical path: ~1 on it, ~0 where the step pipeline has slack (tools/critical_path.py).
__global__ void delay_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
inline void delay_behind(RiftCtx* c, const char* label) {
  if (c->dg.delay_ticks > 0 && c->dg.delay_label == label) hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(1), 0, c->stream, c->dg.delay_ticks);
}

template <class... KArgs, class... Args>
void launch(RiftCtx* c, const char* label, void (*kern)(KArgs...), dim3 grid, dim3 block, size_t shmem, Args... args) {
  if (c->dry || grid.x == 0 || grid.y == 0) return;
  if (c->poison_lds >= 0) {
    const unsigned int b = (unsigned int)c->poison_lds & 0xffu;
    hipLaunchKernelGGL(lds_poison_kernel, dim3(2048), dim3(256), 160 * 1024, c->stream, b | (b << 8) | (b << 16) | (b << 24));
  }
  if (c->prof_on) {
    hipEvent_t e0, e1;
    prof_events(c, &e0, &e1);
    (void)hipEventRecord(e0, c->stream);
    hipLaunchKernelGGL(kern, grid, block, shmem, c->stream, static_cast<KArgs>(args)...);
    (void)hipEventRecord(e1, c->stream);
    prof_push(c, label, e0, e1, c->prof_flops);
    c->prof_flops = 0.0;
    return;
  }
  hipLaunchKernelGGL(kern, grid, block, shmem, c->stream, static_cast<KArgs>(args)...);
  delay_behind(c, label);
}

// same bookkeeping for a kernel that lives in another translation unit (its launch is the callable)
template <class F>
void launch_call(RiftCtx* c, const char* label, F&& f) {
  if (c->dry) return;
  if (c->poison_lds >= 0) {
    const unsigned int b = (unsigned int)c->poison_lds & 0xffu;
    hipLaunchKernelGGL(lds_poison_kernel, dim3(2048), dim3(256), 160 * 1024, c->stream, b | (b << 8) | (b << 16) | (b << 24));
  }
  if (c->prof_on) {
    hipEvent_t e0, e1;
    prof_events(c, &e0, &e1);
    (void)hipEventRecord(e0, c->stream);
    f();
    (void)hipEventRecord(e1, c->stream);
    prof_push(c, label, e0, e1, c->prof_flops);
    c->prof_flops = 0.0;
    return;
  }
  f();
  delay_behind(c, label);
}

template <class... KArgs>
void launch_gemm(RiftCtx* c, const char* label, void (*kern)(KArgs...), dim3 grid, dim3 block, size_t shmem, const GemmP& g) {
  c->prof_flops = 2.0 * (double)g.M * (double)g.N * (double)g.K;
  if (c->prof_on && c->prof_shapes) {   // RIFT_PROF_SHAPES=1: one label per GEMM shape (interned)
    char tmp[160];
    snprintf(tmp, sizeof(tmp), "%s:%dx%dx%d%s%s%s", label, g.M, g.N, g.K, g.pro == PRO_LN ? ":ln" : (g.pro == PRO_AFFINE ? ":bn" : ""),
             g.amode == AMODE_CONV3 ? ":conv" : "", g.stage == 0 ? ":scalar" : "");
    label = c->prof_names.insert(tmp).first->c_str();
  }
  launch(c, label, kern, grid, block, shmem, g);
}

// Weight-only products (mode embeddings through q_proj / m2m in-projection, the ego query projection): input
// independent, so they are computed by the first forward after rift_model_load and kept (per precision mode).
// Returns the buffer and whether the caller must fill it now.
float* wconst_get(RiftCtx* c, const std::string& key, size_t n, bool fp32, bool* fill) {
  RiftCtx::WConst& w = c->wconst[key + (fp32 ? "#32" : "#16")];
  if (!w.p) { if (hipMalloc((void**)&w.p, n * sizeof(float)) != hipSuccess) { w.p = nullptr; *fill = false; return nullptr; } c->owned.push_back(w.p); }
  *fill = !w.ready && !c->dry;
  if (!c->dry) w.ready = true;
  return w.p;
}

const Param* find(RiftCtx* c, const std::string& name) {
  auto it = c->params.find(name);
  return it == c->params.end() ? nullptr : &it->second;
}

const float* fptr(RiftCtx* c, const std::string& name) {
  const Param* p = find(c, name);
  if (!p) { if (c->err.empty()) c->err = "missing parameter: " + name; return nullptr; }
  return reinterpret_cast<const float*>(p->data);
}

// ---- weight packing -------------------------------------------------------------------
int pack(RiftCtx* c, const std::string& key, const float* src, int n_src, int N, int K, int src_row_off,
         int src_col_off, int src_ld, int conv_C, int tap_lo, int tap_n, const float* bias) {
  PW w;
  w.N = N; w.K = K; w.Kp = (K + 31) & ~31; w.Npad = (N + 15) & ~15; w.bias = bias;
  const size_t cnt = (size_t)w.Npad * w.Kp;
  HIPCHK(c, hipMalloc(&w.bf, cnt * 2));
  HIPCHK(c, hipMalloc((void**)&w.f32, cnt * 4));
  c->owned.push_back(w.bf); c->owned.push_back(w.f32);
  const int blocks = cdiv((long long)cnt, 256);
  hipLaunchKernelGGL(pack_weight_kernel<true>, dim3(blocks), dim3(256), 0, c->stream, src + src_col_off, w.bf, n_src, K,
                     w.Npad, w.Kp, conv_C, tap_lo, tap_n, src_row_off, src_ld);
  hipLaunchKernelGGL(pack_weight_kernel<false>, dim3(blocks), dim3(256), 0, c->stream, src + src_col_off, (void*)w.f32,
                     n_src, K, w.Npad, w.Kp, conv_C, tap_lo, tap_n, src_row_off, src_ld);
  c->pw[key] = w;
  return RIFT_OK;
}

// + the hidden-layer-operand image of an already packed weight (the fused scene encoder's fc2: its A rows are GELU outputs, common.h: gelu4_hid)
int pack_hid(RiftCtx* c, const std::string& key) {
  PW& w = c->pw[key];
  const size_t cnt = (size_t)w.Npad * w.Kp;
  HIPCHK(c, hipMalloc(&w.hid, cnt * 2));
  c->owned.push_back(w.hid);
  hipLaunchKernelGGL(pack_weight_hid_kernel, dim3(cdiv((long long)cnt, 256)), dim3(256), 0, c->stream, (const float*)w.f32, (unsigned short*)w.hid, w.Npad, w.Kp);
  return RIFT_OK;
}

int pack_linear(RiftCtx* c, const std::string& prefix) {   // nn.Linear
  const Param* w = find(c, prefix + ".weight");
  if (!w || w->ndim != 2) { c->err = "missing linear " + prefix; return RIFT_ERR_ARG; }
  const Param* b = find(c, prefix + ".bias");
  return pack(c, prefix, (const float*)w->data, (int)w->shape[0], (int)w->shape[0], (int)w->shape[1], 0, 0,
              (int)w->shape[1], 0, 0, 0, b ? (const float*)b->data : nullptr);
}

int pack_conv(RiftCtx* c, const std::string& prefix) {     // nn.Conv1d k=3 -> tap-major [N][3*C]
  const Param* w = find(c, prefix + ".weight");
  if (!w || w->ndim != 3 || w->shape[2] != 3) { c->err = "missing conv " + prefix; return RIFT_ERR_ARG; }
  const Param* b = find(c, prefix + ".bias");
  const int C = (int)w->shape[1];
  return pack(c, prefix, (const float*)w->data, (int)w->shape[0], (int)w->shape[0], 3 * C, 0, 0, 0, C, 0, 3,
              b ? (const float*)b->data : nullptr);
}

// rows [r0, r0+n_src) of an (R, K) matrix as a GEMM with N output columns (rows beyond n_src are zero)
int pack_rows(RiftCtx* c, const std::string& key, const std::string& wname, const std::string& bname, int r0,
              int n_src, int N) {
  const Param* w = find(c, wname);
  if (!w || w->ndim != 2) { c->err = "missing matrix " + wname; return RIFT_ERR_ARG; }
  const Param* b = bname.empty() ? nullptr : find(c, bname);
  return pack(c, key, (const float*)w->data, n_src, N, (int)w->shape[1], r0, 0, (int)w->shape[1], 0, 0, 0,
              b ? (const float*)b->data + r0 : nullptr);
}

// columns [c0, c0+K) of a Linear weight (N, Ktot)
int pack_cols(RiftCtx* c, const std::string& key, const std::string& prefix, int c0, int K, bool with_bias) {
  const Param* w = find(c, prefix + ".weight");
  if (!w || w->ndim != 2) { c->err = "missing linear " + prefix; return RIFT_ERR_ARG; }
  const Param* b = with_bias ? find(c, prefix + ".bias") : nullptr;
  return pack(c, key, (const float*)w->data, (int)w->shape[0], (int)w->shape[0], K, 0, c0, (int)w->shape[1], 0, 0, 0,
              b ? (const float*)b->data : nullptr);
}

#define TRY(x) do { int rc__ = (x); if (rc__ != RIFT_OK) return rc__; } while (0)

// rows [r0, r0+n) of `nsrc` (N_src, K) matrices stacked into one (nsrc*n, K) GEMM weight (+ stacked bias, engine-owned)
int pack_stacked_rows(RiftCtx* c, const std::string& key, const std::vector<std::string>& wnames, const std::vector<std::string>& bnames,
                      int r0, int n) {
  const Param* w0 = find(c, wnames[0]);
  if (!w0 || w0->ndim != 2) { c->err = "missing matrix " + wnames[0]; return RIFT_ERR_ARG; }
  const int K = (int)w0->shape[1], ns = (int)wnames.size();
  PW w;
  w.N = ns * n; w.K = K; w.Kp = (K + 31) & ~31; w.Npad = (w.N + 15) & ~15;
  if (n % 16) { c->err = "pack_stacked_rows: n % 16"; return RIFT_ERR_ARG; }
  const size_t cnt = (size_t)w.Npad * w.Kp;
  float* bias = nullptr;
  HIPCHK(c, hipMalloc(&w.bf, cnt * 2));
  HIPCHK(c, hipMalloc((void**)&w.f32, cnt * 4));
  HIPCHK(c, hipMalloc((void**)&bias, (size_t)w.N * 4));
  c->owned.push_back(w.bf); c->owned.push_back(w.f32); c->owned.push_back(bias);
  for (int i = 0; i < ns; ++i) {
    const Param* wi = find(c, wnames[i]);
    const Param* bi = find(c, bnames[i]);
    if (!wi || !bi || wi->shape[1] != K) { c->err = "missing matrix " + wnames[i]; return RIFT_ERR_ARG; }
    const size_t off = (size_t)i * n * w.Kp;     // a 16-row block of a fragment-major image starts at the row-major offset of its first row
    const int blocks = cdiv((long long)n * w.Kp, 256);
    hipLaunchKernelGGL(pack_weight_kernel<true>, dim3(blocks), dim3(256), 0, c->stream, (const float*)wi->data, (void*)((unsigned short*)w.bf + off),
                       n, K, n, w.Kp, 0, 0, 0, r0, K);
    hipLaunchKernelGGL(pack_weight_kernel<false>, dim3(blocks), dim3(256), 0, c->stream, (const float*)wi->data, (void*)(w.f32 + off),
                       n, K, n, w.Kp, 0, 0, 0, r0, K);
    HIPCHK(c, hipMemcpyAsync(bias + (size_t)i * n, (const float*)bi->data + r0, (size_t)n * 4, hipMemcpyDeviceToDevice, c->stream));
  }
  w.bias = bias;
  c->pw[key] = w;
  return RIFT_OK;
}

int pack_mlp_layer(RiftCtx* c, const std::string& p) { TRY(pack_linear(c, p + ".mlp.0")); return pack_linear(c, p + ".mlp.3"); }

int pack_fourier(RiftCtx* c, const std::string& p, int D) {
  for (int d = 0; d < D; ++d) {
    TRY(pack_linear(c, p + ".mlps." + std::to_string(d) + ".0"));
    TRY(pack_cols(c, p + ".mlps." + std::to_string(d) + ".0.lo", p + ".mlps." + std::to_string(d) + ".0", 0, 128, true));     // fused kernel:
    TRY(pack_cols(c, p + ".mlps." + std::to_string(d) + ".0.last", p + ".mlps." + std::to_string(d) + ".0", 128, 1, false));  // K = 128 + rank-1
    TRY(pack_linear(c, p + ".mlps." + std::to_string(d) + ".3"));
  }
  return pack_linear(c, p + ".to_out.2");
}

int pack_points_encoder(RiftCtx* c, const std::string& p) {
  TRY(pack_linear(c, p + ".first_mlp.0"));
  TRY(pack_linear(c, p + ".first_mlp.3"));
  TRY(pack_cols(c, p + ".second_mlp.0.feat", p + ".second_mlp.0", 0, 256, true));
  TRY(pack_cols(c, p + ".second_mlp.0.pool", p + ".second_mlp.0", 256, 256, false));
  return pack_linear(c, p + ".second_mlp.3");
}

int pack_self_mha(RiftCtx* c, const std::string& p) {
  TRY(pack_rows(c, p + ".qkv", p + ".in_proj_weight", p + ".in_proj_bias", 0, 384, 384));
  return pack_linear(c, p + ".out_proj");
}

// ---- GEMM dispatch -----------------------------------------------------------------------
GemmP mk(const float* X, int ldx, int M, const PW& w, float* Y, int ldy) {
  GemmP g;
  memset(&g, 0, sizeof(g));
  g.X = X; g.ldx = ldx; g.M = M; g.N = w.N; g.K = w.K; g.Kp = w.Kp; g.bias = w.bias; g.Y = Y; g.ldy = ldy;
  g.ln_eps = 1e-5f; g.gb_div = 1; g.dp_div = 1; g.rz_div = 1;
  return g;
}

// Tile shapes (64 rows per workgroup, 4 waves):
//   m1n8 : wave = 16 rows x 128 cols   -- small K (B fragments are few, A fragments read once)
//   m4n2 : wave = 64 rows x  32 cols   -- N <= 128 with K >= 128 (each weight fragment feeds 4 MFMAs)
//   m2n6 : wave = 32 rows x  96 cols   -- 128 < N <= 192
//   m4n4 : wave = 64 rows x  64 cols   -- N > 192, 256 columns per pass
template <bool BF16>
void gemm_launch(RiftCtx* c, GemmP g) {
  const size_t a_bytes = (((size_t)64 * (g.Kp + Prec<BF16>::PAD) * sizeof(typename Prec<BF16>::lds_t)) + 15) & ~(size_t)15;
  int variant;   // 0 m1n8, 1 m4n2, 2 m2n6, 3 m4n4
  if (g.N <= 128) variant = (g.Kp >= 128 && g.N > 32) ? 1 : 0;
  else if (g.N <= 192) variant = 2;
  else variant = 3;
  static const int NTs[4] = {8, 2, 6, 4}, WNs[4] = {1, 4, 2, 4};
  const int NT = NTs[variant], WN = WNs[variant];
  const size_t c_bytes = (size_t)4 * 16 * (16 * NT + 4) * 4;
  const bool single = g.N <= 16 * NT * WN;
  g.cs_off = single ? 0 : (int)a_bytes;
  const size_t lds = single ? (a_bytes > c_bytes ? a_bytes : c_bytes) : a_bytes + c_bytes;
  const dim3 grid(cdiv(g.M, 64)), block(256);
  switch (variant) {
    case 0: launch_gemm(c, BF16 ? "gemm_bf16_m1n8" : "gemm_fp32_m1n8", gemm_rows_kernel<BF16, 1, 8, 4, 1>, grid, block, lds, g); break;
    case 1: launch_gemm(c, BF16 ? "gemm_bf16_m4n2" : "gemm_fp32_m4n2", gemm_rows_kernel<BF16, 4, 2, 1, 4>, grid, block, lds, g); break;
    case 2: launch_gemm(c, BF16 ? "gemm_bf16_m2n6" : "gemm_fp32_m2n6", gemm_rows_kernel<BF16, 2, 6, 2, 2>, grid, block, lds, g); break;
    default: launch_gemm(c, BF16 ? "gemm_bf16_m4n4" : "gemm_fp32_m4n4", gemm_rows_kernel<BF16, 4, 4, 1, 4>, grid, block, lds, g); break;
  }
}

void gemm(RiftCtx* c, GemmP g, const PW& w, bool fp32) {
  if (g.M <= 0) return;
  // vectorised A staging needs 16-byte aligned rows: dense conv windows (C % 4 == 0) or ldx % 4 == 0, K % 4 == 0
  const bool conv = g.amode == AMODE_CONV3;
  const bool al = (((uintptr_t)g.X) & 15) == 0 && (g.K % 4 == 0) && (conv ? (g.cv_C % 4 == 0 && g.ldx == g.cv_C) : (g.ldx % 4 == 0)) &&
                  (g.pro == PRO_NONE || ((((uintptr_t)g.pg) | ((uintptr_t)g.pb)) & 15) == 0);
  const int KV = g.Kp / 4;
  g.dbg = c->gemm_dbg;
  g.stage = !al ? 0 : (KV <= 8 ? 1 : KV <= 16 ? 2 : KV <= 32 ? 3 : KV <= 64 ? 4 : 5);
  g.evec = (g.N % 4 == 0) && (g.ldy % 4 == 0) && ((((uintptr_t)g.Y) & 15) == 0) &&
           (!g.bias || (((uintptr_t)g.bias) & 15) == 0) && (!g.gbias || (((uintptr_t)g.gbias) & 15) == 0) &&
           (!g.residual || ((g.ldr % 4 == 0) && (((uintptr_t)g.residual) & 15) == 0));
  if (fp32) { g.W = w.f32; gemm_launch<false>(c, g); }
  else { g.W = w.bf; gemm_launch<true>(c, g); }
}

int set_lds_attrs(RiftCtx* c) {
  const int big = 160 * 1024;
#define SETATTR(K) HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, big))
  SETATTR(rollout_kernel);
  SETATTR(enc_fused_kernel<ENC_NW>);
  HIPCHK(c, (hipError_t)enc112_set_attributes());
  HIPCHK(c, (hipError_t)decw_set_attributes());
  HIPCHK(c, (hipError_t)l2w_set_attributes());
  HIPCHK(c, (hipError_t)encw_set_attributes());
  HIPCHK(c, (hipError_t)pew_set_attributes());
  HIPCHK(c, (hipError_t)fow_set_attributes());
  SETATTR((gemm_rows_kernel<true, 1, 8, 4, 1>));
  SETATTR((gemm_rows_kernel<true, 4, 2, 1, 4>));
  SETATTR((gemm_rows_kernel<false, 4, 2, 1, 4>));
  SETATTR((gemm_rows_kernel<true, 2, 6, 2, 2>));
  SETATTR((gemm_rows_kernel<true, 4, 4, 1, 4>));
  SETATTR((gemm_rows_kernel<false, 1, 8, 4, 1>));
  SETATTR((gemm_rows_kernel<false, 2, 6, 2, 2>));
  SETATTR((gemm_rows_kernel<false, 4, 4, 1, 4>));
#undef SETATTR
#define SETATTR_N(K, N) HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(N)))
  SETATTR_N(lds_poison_kernel, 160 * 1024);
  SETATTR_N(dec_kv_frag_kernel, DEC_KV_LDS);
  SETATTR_N(pe_mid_kernel, PE_MID_LDS);   // these kernels also hold a few hundred bytes of static LDS
  SETATTR_N(pe_out_kernel, PE_OUT_LDS);
  SETATTR_N(fourier_fused_kernel, FO_LDS);
  SETATTR_N(fpn_tail_kernel, FPN_LDS);
  SETATTR_N(heads3_fused_kernel, HD_LDS);
  SETATTR_N(pi_forward_kernel, PI_LDS);
  HIPCHK(c, (hipError_t)l0w_set_attributes());
  HIPCHK(c, (hipError_t)l1w_set_attributes());
#undef SETATTR_N
  return RIFT_OK;
}

struct Fwd {   // per-forward context
  RiftCtx* c; bool train, drop, fp32, need_traj, bn_update; uint32_t seed; uint32_t stream_id = 1;
  uint32_t next_stream() { return stream_id++; }
  // data parallel (rift_set_dp): xchg[0, kb) = quirk-mask slots of the global minibatch, xchg[kb, ...) = BatchNorm sums of the current point
  bool dp = false, kpm_pending = false; int kb = 0; uint8_t* g_rkpm = nullptr;
  uint8_t* r_tiles = nullptr;                        // tiles of every reference line up to its last valid point (prep_kernel), for pe_w_kernel's packed rounds
};

// RCCL through dlopen (rift_comm_*): the library has no link-time dependency on a communication library; an RCCL that is already in the
// process (torch's) is reused so that one process does not run two of them.
struct RcclApi {
  struct UniqueId { char internal[128]; };
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};
static RcclApi& rccl_api(std::string* why) {
  static RcclApi api;
  static bool tried = false;
  static std::string err;
  if (!tried) {
    tried = true;
    void* h = nullptr;
    const char* names[] = {getenv("RIFT_RCCL_LIB"), "librccl.so", "librccl.so.1"};
    for (const char* n : names) if (n && !h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);      // one that is loaded already
    for (const char* n : names) if (n && !h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) err = std::string("librccl.so not found (set RIFT_RCCL_LIB): ") + (dlerror() ? dlerror() : "");
    else {
      api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
      api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
      api.AllReduce = (decltype(api.AllReduce))dlsym(h, "ncclAllReduce");
      api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
      api.GetErrorString = (decltype(api.GetErrorString))dlsym(h, "ncclGetErrorString");
      api.ok = api.GetUniqueId && api.CommInitRank && api.AllReduce && api.CommDestroy;
      if (!api.ok) err = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy";
    }
  }
  if (why) *why = err;
  return api;
}
static int comm_sum_f64(RiftCtx* c, double* buf, long long count, hipStream_t stream) {
  RcclApi& r = rccl_api(nullptr);
  if (!c->comm || !r.ok) return -1;
  return r.AllReduce(buf, buf, (size_t)count, /*ncclDouble*/ 8, /*ncclSum*/ 0, c->comm, stream);
}

// All-reduce xchg[kb, kb + n) over the ranks (BatchNorm sums); the first exchange of a forward also carries the mask slots [0, kb)
// and is followed by their conversion into the gathered padding mask.  Dry (arena sizing) passes exchange nothing.
void dp_exchange(Fwd& f, long long n) {
  RiftCtx* c = f.c;
  if (!f.dp) return;
  const bool with_kpm = f.kpm_pending;
  f.kpm_pending = false;
  if (c->dry) return;
  const long long off = with_kpm ? 0 : f.kb, cnt = with_kpm ? f.kb + n : n;
  if (cnt > 0) {
    const int rc = c->dp.fn ? c->dp.fn(c->dp.user, off, cnt, (void*)c->stream) : comm_sum_f64(c, c->dp.xchg + off, cnt, c->stream);      // (no callback: the library's own communicator)
    if (rc != 0 && c->err.empty()) c->err = c->dp.fn ? "data-parallel exchange callback failed" : "data-parallel exchange over the library communicator failed";
  }
  if (with_kpm) launch(c, "dp_kpm_read_kernel", dp_kpm_read_kernel, dim3(cdiv(f.kb, 256)), dim3(256), 0, (const double*)c->dp.xchg, f.kb, f.g_rkpm);
}

void layernorm(Fwd& f, const float* X, int ldx, float* Y, int ldy, int rows, int C, const std::string& name, int relu = 0) {
  RiftCtx* c = f.c;
  launch(c, "layernorm_kernel", layernorm_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, X, ldx, Y, ldy, rows, C, fptr(c, name + ".weight"),
         fptr(c, name + ".bias"), 1e-5f, relu);
}

// FourierEmbedding (fourier_embedding.py:45-55): in (rows, D) -> out (rows, 128)
FourierP fourier_desc(Fwd& f, const float* in, int in_ld, int rows, int D, const std::string& p, int wrap_dim, float* add_to) {
  RiftCtx* c = f.c;
  FourierP q; memset(&q, 0, sizeof(q));
  q.in = in; q.in_ld = in_ld; q.rows = rows; q.D = D; q.wrap_dim = wrap_dim; q.freqs = fptr(c, p + ".freqs.weight");
  for (int d = 0; d < D; ++d) {
    const std::string m = p + ".mlps." + std::to_string(d);
    const PW &lo = c->pw[m + ".0.lo"], &last = c->pw[m + ".0.last"], &w3 = c->pw[m + ".3"];
    q.w0[d] = (const unsigned short*)lo.bf; q.b0[d] = lo.bias; q.wl[d] = last.f32; q.wl_ld = last.Kp;
    q.lng[d] = fptr(c, m + ".1.weight"); q.lnb[d] = fptr(c, m + ".1.bias");
    q.w3[d] = (const unsigned short*)w3.bf; q.b3[d] = w3.bias;
  }
  q.og = fptr(c, p + ".to_out.0.weight"); q.ob = fptr(c, p + ".to_out.0.bias");
  q.wo = (const unsigned short*)c->pw[p + ".to_out.2"].bf; q.bo = c->pw[p + ".to_out.2"].bias;
  q.Y = add_to ? add_to : A_alloc<float>(c, (size_t)rows * 128);
  q.accumulate = add_to ? 1 : 0;
  return q;
}

// add_to != nullptr: the embedding is added into that (rows, 128) buffer, which is returned
float* fourier(Fwd& f, const float* in, int in_ld, int rows, int D, const std::string& p, int wrap_dim, float* add_to = nullptr) {
  RiftCtx* c = f.c;
  if (!f.fp32 && c->fo_fused && D <= 3) {
    FourierP3 q3; memset(&q3, 0, sizeof(q3));
    q3.e[0] = fourier_desc(f, in, in_ld, rows, D, p, wrap_dim, add_to);
    q3.nblk[0] = cdiv(rows, FO_ROWS); q3.count = 1;
    c->prof_flops = 2.0 * rows * 128.0 * (D * (129.0 + 128.0) + 128.0);
    launch(c, "fourier_fused_kernel", fourier_fused_kernel, dim3(q3.nblk[0]), dim3(256), (size_t)FO_LDS, q3);
    return q3.e[0].Y;
  }
  float* FF = A_alloc<float>(c, (size_t)D * rows * 129);
  launch(c, "fourier_feature_kernel", fourier_feature_kernel, dim3(cdiv((long long)D * rows * 65, 256)), dim3(256), 0, in, in_ld, rows, D,
         fptr(c, p + ".freqs.weight"), wrap_dim, FF);
  float* T1 = A_alloc<float>(c, (size_t)rows * 128);
  float* acc = A_alloc<float>(c, (size_t)rows * 128);
  for (int d = 0; d < D; ++d) {
    const std::string m = p + ".mlps." + std::to_string(d);
    GemmP g = mk(FF + (size_t)d * rows * 129, 129, rows, c->pw[m + ".0"], T1, 128);
    gemm(c, g, c->pw[m + ".0"], f.fp32);
    GemmP g2 = mk(T1, 128, rows, c->pw[m + ".3"], acc, 128);
    g2.pro = PRO_LN; g2.pg = fptr(c, m + ".1.weight"); g2.pb = fptr(c, m + ".1.bias"); g2.pro_relu = 1;
    if (d > 0) { g2.residual = acc; g2.ldr = 128; }
    gemm(c, g2, c->pw[m + ".3"], f.fp32);
  }
  float* out = A_alloc<float>(c, (size_t)rows * 128);
  GemmP g3 = mk(acc, 128, rows, c->pw[p + ".to_out.2"], out, 128);
  g3.pro = PRO_LN; g3.pg = fptr(c, p + ".to_out.0.weight"); g3.pb = fptr(c, p + ".to_out.0.bias"); g3.pro_relu = 1;
  gemm(c, g3, c->pw[p + ".to_out.2"], f.fp32);
  if (add_to) {
    launch(c, "add_inplace_kernel", add_inplace_kernel, dim3(cdiv((long long)rows * 128, 256)), dim3(256), 0, add_to, (const float*)out, (size_t)rows * 128);
    return add_to;
  }
  return out;
}

// BatchNorm1d folded to an affine (scale, shift) for the next GEMM prologue.
void batchnorm_affine(Fwd& f, const float* X, int rows, int C, const uint8_t* valid, const std::string& name,
                      float** scale, float** shift) {
  RiftCtx* c = f.c;
  *scale = A_alloc<float>(c, C); *shift = A_alloc<float>(c, C);
  const int rows_per_blk = 128;
  const int nblk = cdiv(rows, rows_per_blk);
  double* part = A_alloc<double>(c, (size_t)nblk * 2 * C);
  int* cnt = A_alloc<int>(c, nblk);
  if (f.train) launch(c, "bn_partial_kernel", bn_partial_kernel, dim3(nblk), dim3(C), 0, X, C, rows, C, valid, part, cnt, rows_per_blk);
  const Param* nb = find(c, name + ".num_batches_tracked");
  const bool dpx = f.dp && f.train;      // data parallel: batch statistics over the GLOBAL minibatch (sums all-reduced between two launches)
  double* sums = dpx ? c->dp.xchg + f.kb : nullptr;
  for (int mode = dpx ? 1 : 0; mode <= (dpx ? 2 : 0); ++mode) {
    launch(c, "bn_finalize_kernel", bn_finalize_kernel, dim3(cdiv(C, 4)), dim3(256), 0, (const double*)part, (const int*)cnt, nblk, C,
           fptr(c, name + ".weight"), fptr(c, name + ".bias"), (float*)fptr(c, name + ".running_mean"),
           (float*)fptr(c, name + ".running_var"), nb ? (long long*)nb->data : (long long*)nullptr, f.train ? 1 : 0,
           f.bn_update ? 1 : 0, 1e-5f, *scale, *shift, sums, mode);
    if (mode == 1) dp_exchange(f, 2 * C + 1);
  }
}

// The two PointsEncoders (map polygons 10 -> 128 with 20 points, reference lines 6 -> 128 with 120 points) in the fused
// three-pass form (pe_fused.h), side by side: five launches for both (statistics, BN finalize, mid, BN finalize, out).
static void pe_fill(Fwd& f, PeP& q, const float* F, int Cin, int groups, int n, const uint8_t* valid, const std::string& p) {
  RiftCtx* c = f.c;
  memset(&q, 0, sizeof(q));
  const int rows = groups * n;
  q.F = F; q.Cin = Cin; q.valid = valid; q.rows = rows; q.ntiles = cdiv(rows, 120);
  const PW &w1 = c->pw[p + ".first_mlp.0"], &w2 = c->pw[p + ".first_mlp.3"], &w3a = c->pw[p + ".second_mlp.0.feat"],
           &w3b = c->pw[p + ".second_mlp.0.pool"], &w4 = c->pw[p + ".second_mlp.3"];
  q.w1 = (const unsigned short*)w1.bf; q.w2 = (const unsigned short*)w2.bf; q.w3a = (const unsigned short*)w3a.bf;
  q.w3b = (const unsigned short*)w3b.bf; q.w4 = (const unsigned short*)w4.bf;
  q.b1 = w1.bias; q.b2 = w2.bias; q.b3 = w3a.bias; q.b4 = w4.bias;
  q.s1 = A_alloc<float>(c, 128); q.t1 = A_alloc<float>(c, 128); q.s2 = A_alloc<float>(c, 256); q.t2 = A_alloc<float>(c, 256);
  q.part1 = A_alloc<float>(c, (size_t)2 * 128 * q.ntiles);
  q.part2 = A_alloc<float>(c, (size_t)2 * 256 * q.ntiles);
  q.cnt = A_alloc<int>(c, q.ntiles);
  q.Fmid = A_alloc<unsigned short>(c, (size_t)rows * 256);
  q.gp = A_alloc<float>(c, (size_t)groups * 256);
  q.out = A_alloc<float>(c, (size_t)groups * 128);
  q.do_stats = f.train ? 1 : 0;
  { if (c->dg.pe_ts == n) { q.ts = A_alloc<long long>(c, 64); q.ts_tile = 0; tap(c, "pe_ts", (float*)q.ts, 128); } }
}

static BnFinP bn_fin(RiftCtx* c, const PeP& q, const std::string& name, int C, const float* part, const float* sc, const float* sh,
                     double* sums = nullptr, int sums_mode = 0) {
  BnFinP b; memset(&b, 0, sizeof(b));
  b.sums = sums; b.sums_mode = sums_mode;
  const Param* nb = find(c, name + ".num_batches_tracked");
  b.part = part; b.cnt = q.cnt; b.nblk = q.ntiles; b.C = C; b.gamma = fptr(c, name + ".weight"); b.beta = fptr(c, name + ".bias");
  b.running_mean = (float*)fptr(c, name + ".running_mean"); b.running_var = (float*)fptr(c, name + ".running_var");
  b.num_batches = nb ? (long long*)nb->data : nullptr; b.scale = (float*)sc; b.shift = (float*)sh;
  return b;
}

void points_encoder_pair(Fwd& f, const float* Fm, int gm, const uint8_t* vm, const std::string& pm, const float* Fr, int gr,
                         const uint8_t* vr, const std::string& pr, float** out_m, float** out_r) {
  RiftCtx* c = f.c;
  PeP2 q;
  pe_fill(f, q.a, Fm, 10, gm, 20, vm, pm);
  pe_fill(f, q.b, Fr, 6, gr, 120, vr, pr);
  const int nt = q.a.ntiles + q.b.ntiles;
  const double rows = (double)q.a.rows + q.b.rows, groups = (double)gm + gr;
  const bool stats_p = f.train && c->pe_w;          // persistent pass A: one partial per workgroup (pe_fused.h)
  // reference lines packed at tile granularity (pe_fused.h: pe_pack_body): table + round count in device memory, built by one extra block
  // of pass A's launch (eval mode: a launch of its own)
  const bool packb = c->pe_w && c->pe_pack && gr > 0 && f.r_tiles != nullptr;
  int* pk_tab = nullptr; int* pk_hdr = nullptr;
  PePackP pk; memset(&pk, 0, sizeof(pk));
  if (packb) {
    const int maxr = cdiv(q.b.rows, PEW_ROUND_ROWS);
    pk_tab = A_alloc<int>(c, (size_t)std::max(maxr, 1) * PEW_TAB_INTS); pk_hdr = A_alloc<int>(c, 4);
    pk.tiles = f.r_tiles; pk.nlines = gr; pk.tab = pk_tab; pk.hdr = pk_hdr; pk.max_rounds = maxr;
    tap(c, "pe_pack_hdr", (float*)pk_hdr, 4); tap(c, "pe_pack_tab", (float*)pk_tab, (int64_t)std::max(maxr, 1) * PEW_TAB_INTS);
    if (!stats_p) { c->prof_flops = 0.0; launch(c, "pe_pack_lines_kernel", pe_pack_lines_kernel, dim3(1), dim3(256), (size_t)gr * 2 * sizeof(int), pk); }
  }
  if (stats_p) {
    int g = std::min(nt, 4 * c->nat_grid);          // four 256-thread workgroups per CU
    int ga = (int)(((long long)g * q.a.ntiles + nt / 2) / nt);
    if (q.a.ntiles > 0 && ga < 1) ga = 1;
    if (q.b.ntiles > 0 && ga > g - 1) ga = g - 1;
    if (q.b.ntiles == 0) ga = g;
    ga = std::min(ga, q.a.ntiles);
    q.a.nwg1 = ga; q.b.nwg1 = std::min(g - ga, q.b.ntiles);
    PeP* sd[2] = {&q.a, &q.b};
    for (int i = 0; i < 2; ++i) { sd[i]->part1w = A_alloc<float>(c, (size_t)2 * 128 * std::max(sd[i]->nwg1, 1)); sd[i]->cnt1w = A_alloc<int>(c, std::max(sd[i]->nwg1, 1)); }
    c->prof_flops = 2.0 * 128.0 * (q.a.rows * 10.0 + q.b.rows * 6.0);
    launch(c, "pe_stats1_kernel", pe_stats1p_kernel, dim3(q.a.nwg1 + q.b.nwg1 + (packb ? 1 : 0)), dim3(256), packb ? (size_t)gr * 2 * sizeof(int) : 0, q, pk);
  } else if (f.train) {
    c->prof_flops = 2.0 * 128.0 * (q.a.rows * 10.0 + q.b.rows * 6.0);
    launch(c, "pe_stats1_kernel", pe_stats1_kernel, dim3(nt), dim3(256), 0, q);
  }
  const bool dpx = f.dp && f.train;      // data parallel: the four BatchNorms see the statistics of the GLOBAL minibatch
  double* xs = dpx ? c->dp.xchg + f.kb : nullptr;
  // pass B's live rounds and workgroup split, from pass A's counts (pe_fused.h: PeLiveP): one extra block of the BatchNorm-1 finalize
  PeLiveP lv; memset(&lv, 0, sizeof(lv));
  const bool live = stats_p && c->pe_live;
  int pew_grid = c->nat_grid, pew_ga = 0;
  if (c->pe_w) pew_split(cdiv(q.a.rows, PEW_ROUND_ROWS), cdiv(q.b.rows, PEW_ROUND_ROWS), &pew_grid, &pew_ga);
  if (live || packb) {
    const PeP* sd[2] = {&q.a, &q.b};
    for (int i = 0; i < 2; ++i) {
      lv.cnt[i] = live ? sd[i]->cnt : nullptr; lv.nt[i] = sd[i]->ntiles; lv.nr[i] = cdiv(sd[i]->rows, PEW_ROUND_ROWS);
      lv.live[i] = A_alloc<int>(c, std::max(lv.nr[i], 1));
    }
    lv.grid = pew_grid; lv.hdr = A_alloc<int>(c, 4);
    lv.packed_b = pk_hdr;
    tap(c, "pe_live_hdr", (float*)lv.hdr, 4);
  }
  const bool lvblock = live || packb;       // the extra block of the BatchNorm-1 finalize launch
  for (int mode = dpx ? 1 : 0; mode <= (dpx ? 2 : 0); ++mode) {
    BnFinP f1a = bn_fin(c, q.a, pm + ".first_mlp.1", 128, q.a.part1, q.a.s1, q.a.t1, xs, mode);
    BnFinP f1b = bn_fin(c, q.b, pr + ".first_mlp.1", 128, q.b.part1, q.b.s1, q.b.t1, xs ? xs + 257 : nullptr, mode);
    if (stats_p) { f1a.part = q.a.part1w; f1a.cnt = q.a.cnt1w; f1a.nblk = q.a.nwg1; f1b.part = q.b.part1w; f1b.cnt = q.b.cnt1w; f1b.nblk = q.b.nwg1; }
    const bool with_lv = lvblock && mode == (dpx ? 2 : 0);
    PeLiveP none; memset(&none, 0, sizeof(none));
    launch(c, "bn_finalize_t_kernel", bn_finalize_t_kernel, dim3(256 + (with_lv ? 1 : 0)), dim3(256), 0, f1a, f1b, f.train ? 1 : 0, f.bn_update ? 1 : 0, 1e-5f,
           with_lv ? lv : none);
    if (mode == 1) dp_exchange(f, 2 * 257);
  }
  c->prof_flops = 2.0 * rows * (128.0 * 8 + 128.0 * 256 + (f.train ? 256.0 * 256 : 0.0)) + 2.0 * groups * 256.0 * 256;
  BnFinP fa, fb;
  if (c->pe_w) {     // wave-private pass B: rounds of 240 rows, one statistics partial per workgroup
    PeWP w; memset(&w, 0, sizeof(w));
    const PeP* src[2] = {&q.a, &q.b};
    PeWSide* dst[2] = {&w.a, &w.b};
    const int npts[2] = {20, 120};
    const int grid = pew_grid, ga = pew_ga;
    const int nwg[2] = {lvblock ? grid : ga, lvblock ? grid : grid - ga};       // (live / packed: the split is made on the device; room for either extreme)
    for (int i = 0; i < 2; ++i) {
      const PeP& o = *src[i]; PeWSide& d = *dst[i];
      d.F = o.F; d.Cin = o.Cin; d.valid = o.valid; d.rows = o.rows; d.npts = npts[i]; d.nrounds = cdiv(o.rows, PEW_ROUND_ROWS);
      d.img = c->pew_img[i]; d.w3b = o.w3b; d.b1 = o.b1; d.b2 = o.b2; d.b3 = o.b3; d.s1 = o.s1; d.t1 = o.t1;
      d.live = (live && !(i == 1 && packb)) ? lv.live[i] : nullptr;
      if (i == 1) { d.ptab = pk_tab; d.phdr = pk_hdr; }
      d.part2 = A_alloc<float>(c, (size_t)2 * 256 * std::max(nwg[i], 1)); d.cnt2 = A_alloc<int>(c, std::max(nwg[i], 1));
      d.Fmid = o.Fmid;
    }
    w.ga = ga; w.hdr = lvblock ? lv.hdr : nullptr;
    w.do_stats = f.train ? 1 : 0;
    w.dbg = c->dg.pew_dbg;
    { if (c->dg.pew_ts) { w.ts = A_alloc<long long>(c, 128); tap(c, "pew_ts", (float*)w.ts, 256); } }
    launch_call(c, "pe_w_kernel", [&] { pew_launch(w, grid, c->stream); });
    fa = bn_fin(c, q.a, pm + ".second_mlp.1", 256, w.a.part2, q.a.s2, q.a.t2, xs, 0); fa.cnt = w.a.cnt2; fa.nblk = nwg[0]; fa.nblk_dev = lvblock ? lv.hdr + 2 : nullptr;
    fb = bn_fin(c, q.b, pr + ".second_mlp.1", 256, w.b.part2, q.b.s2, q.b.t2, xs ? xs + 513 : nullptr, 0); fb.cnt = w.b.cnt2; fb.nblk = nwg[1]; fb.nblk_dev = lvblock ? lv.hdr + 3 : nullptr;
  } else {
    launch(c, "pe_mid_kernel", pe_mid_kernel, dim3(nt), dim3(512), (size_t)PE_MID_LDS, q);
    fa = bn_fin(c, q.a, pm + ".second_mlp.1", 256, q.a.part2, q.a.s2, q.a.t2, xs, 0);
    fb = bn_fin(c, q.b, pr + ".second_mlp.1", 256, q.b.part2, q.b.s2, q.b.t2, xs ? xs + 513 : nullptr, 0);
  }
  for (int mode = dpx ? 1 : 0; mode <= (dpx ? 2 : 0); ++mode) {
    fa.sums_mode = mode; fb.sums_mode = mode;
    PeLiveP none; memset(&none, 0, sizeof(none));
    launch(c, "bn_finalize_t_kernel", bn_finalize_t_kernel, dim3(512), dim3(256), 0, fa, fb, f.train ? 1 : 0, f.bn_update ? 1 : 0, 1e-5f, none);
    if (mode == 1) dp_exchange(f, 2 * 513);
  }
  c->prof_flops = 2.0 * rows * (256.0 * 256 + 256.0 * 128);
  launch(c, "pe_out_kernel", pe_out_kernel, dim3(nt), dim3(512), (size_t)PE_OUT_LDS, q);
  *out_m = q.a.out; *out_r = q.b.out;
}

// PointsEncoder (embedding.py:271-296): F (groups*n, Cin) -> (groups, 128)
float* points_encoder(Fwd& f, const float* F, int Cin, int groups, int n, const uint8_t* valid, const std::string& p) {
  RiftCtx* c = f.c;
  const int rows = groups * n;
  float* H1 = A_alloc<float>(c, (size_t)rows * 128);
  gemm(c, mk(F, Cin, rows, c->pw[p + ".first_mlp.0"], H1, 128), c->pw[p + ".first_mlp.0"], f.fp32);
  float *s1, *t1;
  batchnorm_affine(f, H1, rows, 128, valid, p + ".first_mlp.1", &s1, &t1);
  float* F256 = A_alloc<float>(c, (size_t)rows * 256);
  GemmP g = mk(H1, 128, rows, c->pw[p + ".first_mlp.3"], F256, 256);
  g.pro = PRO_AFFINE; g.pg = s1; g.pb = t1; g.pro_relu = 1;
  gemm(c, g, c->pw[p + ".first_mlp.3"], f.fp32);
  float* pooled = A_alloc<float>(c, (size_t)groups * 256);
  launch(c, "masked_maxpool_kernel", masked_maxpool_kernel, dim3(cdiv((long long)groups * 256, 256)), dim3(256), 0, (const float*)F256, 256, groups, n, 256,
         valid, pooled);
  float* G1 = A_alloc<float>(c, (size_t)groups * 256);
  gemm(c, mk(pooled, 256, groups, c->pw[p + ".second_mlp.0.pool"], G1, 256), c->pw[p + ".second_mlp.0.pool"], f.fp32);
  float* H2 = A_alloc<float>(c, (size_t)rows * 256);
  GemmP g2 = mk(F256, 256, rows, c->pw[p + ".second_mlp.0.feat"], H2, 256);
  g2.gbias = G1; g2.gb_div = n; g2.gb_mod = 0;
  gemm(c, g2, c->pw[p + ".second_mlp.0.feat"], f.fp32);
  float *s2, *t2;
  batchnorm_affine(f, H2, rows, 256, valid, p + ".second_mlp.1", &s2, &t2);
  float* O = A_alloc<float>(c, (size_t)rows * 128);
  GemmP g3 = mk(H2, 256, rows, c->pw[p + ".second_mlp.3"], O, 128);
  g3.pro = PRO_AFFINE; g3.pg = s2; g3.pb = t2; g3.pro_relu = 1;
  gemm(c, g3, c->pw[p + ".second_mlp.3"], f.fp32);
  float* out = A_alloc<float>(c, (size_t)groups * 128);
  launch(c, "masked_maxpool_kernel", masked_maxpool_kernel, dim3(cdiv((long long)groups * 128, 256)), dim3(256), 0, (const float*)O, 128, groups, n, 128,
         valid, out);
  return out;
}

template <int NKT>
void run_mha_mfma(RiftCtx* c, const MhaP& p) {
  const int nqt = (p.Lq + 15) / 16;
  const int waves = nqt < 4 ? nqt : 4;
  const size_t lds = (size_t)NKT * 16 * 40 * 2 + (size_t)32 * (NKT * 16 + 8) * 2 + NKT * 16;
  launch(c, "mha_mfma_kernel", mha_mfma_kernel<NKT>, dim3(p.nb_outer * p.nb_inner * p.H), dim3(64 * waves), lds, p);
}

// bf16 mode: MFMA attention; fp32 mode (or unaligned / very long inputs): exact-fp32 VALU kernel
void run_mha(RiftCtx* c, const MhaP& p, bool fp32) {
  const bool al = ((((uintptr_t)p.Q) | ((uintptr_t)p.K) | ((uintptr_t)p.V)) & 15) == 0 && p.ldq % 4 == 0 && p.ldkv % 4 == 0;
  if (!fp32 && al && p.Lk <= 192) {
    if (p.Lk <= 32) run_mha_mfma<2>(c, p);
    else if (p.Lk <= 96) run_mha_mfma<6>(c, p);
    else run_mha_mfma<12>(c, p);
    return;
  }
  const long long total = (long long)p.nb_outer * p.nb_inner * p.H * p.Lq;
  launch(c, "mha_kernel", mha_kernel, dim3(cdiv(total, 64)), dim3(64), 0, p);
}

// NAT block (embedding.py:196-202) on X (rows, C) in place
void nat_layer(Fwd& f, float* X, int rows, int C, int H, int ksz, int L, const std::string& p, float droppath) {
  RiftCtx* c = f.c;
  float* QKV = A_alloc<float>(c, (size_t)rows * 3 * C);
  GemmP g = mk(X, C, rows, c->pw[p + ".attn.qkv"], QKV, 3 * C);
  g.pro = PRO_LN; g.pg = fptr(c, p + ".norm1.weight"); g.pb = fptr(c, p + ".norm1.bias");
  gemm(c, g, c->pw[p + ".attn.qkv"], f.fp32);
  float* AO = A_alloc<float>(c, (size_t)rows * C);
  const int nthreads = rows * H;
  if (ksz == 3) launch(c, "nat_attention_kernel", nat_attention_kernel<3>, dim3(cdiv(nthreads, 256)), dim3(256), 0, (const float*)QKV, fptr(c, p + ".attn.rpb"), rows / L, L, H, AO);
  else launch(c, "nat_attention_kernel", nat_attention_kernel<5>, dim3(cdiv(nthreads, 256)), dim3(256), 0, (const float*)QKV, fptr(c, p + ".attn.rpb"), rows / L, L, H, AO);
  GemmP g2 = mk(AO, C, rows, c->pw[p + ".attn.proj"], X, C);
  g2.residual = X; g2.ldr = C;
  if (f.drop && droppath > 0.f) { g2.droppath_p = droppath; g2.dp_div = L; g2.seed = f.seed; g2.stream = f.next_stream(); }
  gemm(c, g2, c->pw[p + ".attn.proj"], f.fp32);
  float* Hh = A_alloc<float>(c, (size_t)rows * 3 * C);
  GemmP g3 = mk(X, C, rows, c->pw[p + ".mlp.fc1"], Hh, 3 * C);
  g3.pro = PRO_LN; g3.pg = fptr(c, p + ".norm2.weight"); g3.pb = fptr(c, p + ".norm2.bias"); g3.act = ACT_GELU;
  gemm(c, g3, c->pw[p + ".mlp.fc1"], f.fp32);
  GemmP g4 = mk(Hh, 3 * C, rows, c->pw[p + ".mlp.fc2"], X, C);
  g4.residual = X; g4.ldr = C;
  if (f.drop && droppath > 0.f) { g4.droppath_p = droppath; g4.dp_div = L; g4.seed = f.seed; g4.stream = f.next_stream(); }
  gemm(c, g4, c->pw[p + ".mlp.fc2"], f.fp32);
}

// MLPLayer (mlp_layer.py:8-16): X (rows,128) -> out (rows, Nout) with hidden width Hd
// three MLPLayer(128, 256, 160) heads -> interleaved (rows, 80, 6), one launch (heads_fused.h); bf16 mode
void heads3_fused(Fwd& f, const float* X, int ldx, int rows, int g_per, int g_stride, int g_off, const std::string names[3], float* out) {
  RiftCtx* c = f.c;
  Heads3P q; memset(&q, 0, sizeof(q));
  q.X = X; q.ldx = ldx; q.rows = rows; q.gather_per = g_per; q.gather_stride = g_stride; q.gather_off = g_off; q.out = out;
  for (int i = 0; i < 3; ++i) {
    const PW &w1 = c->pw[names[i] + ".mlp.0"], &w2 = c->pw[names[i] + ".mlp.3"];
    q.w1[i] = (const unsigned short*)w1.bf; q.b1[i] = w1.bias; q.w2[i] = (const unsigned short*)w2.bf; q.b2[i] = w2.bias;
    q.lng[i] = fptr(c, names[i] + ".mlp.1.weight"); q.lnb[i] = fptr(c, names[i] + ".mlp.1.bias");
  }
  c->prof_flops = 3.0 * 2.0 * rows * (128.0 * 256 + 256.0 * 160);
  launch(c, "heads3_fused_kernel", heads3_fused_kernel, dim3(24 * cdiv(cdiv(rows, HD_ROWS), 8)), dim3(512), (size_t)HD_LDS, q);      // (groups of 8 tiles x 3 heads: heads_fused.h)
}

void mlp_layer(Fwd& f, const float* X, int ldx, int rows, const std::string& p, float* out, int ldo, bool fp32) {
  RiftCtx* c = f.c;
  const PW& w0 = c->pw[p + ".mlp.0"];
  float* T = A_alloc<float>(c, (size_t)rows * w0.N);
  gemm(c, mk(X, ldx, rows, w0, T, w0.N), w0, fp32);
  const PW& w3 = c->pw[p + ".mlp.3"];
  GemmP g = mk(T, w0.N, rows, w3, out, ldo);
  g.pro = PRO_LN; g.pg = fptr(c, p + ".mlp.1.weight"); g.pb = fptr(c, p + ".mlp.1.bias"); g.pro_relu = 1;
  gemm(c, g, w3, fp32);
}

__global__ void interleave_traj_kernel(const float* __restrict__ loc, const float* __restrict__ yaw,
                                       const float* __restrict__ vel, int rows, float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // (row, t, c6)
  if (idx >= rows * 480) return;
  const int c6 = idx % 6, t = (idx / 6) % 80, r = idx / 480;
  const float* src = c6 < 2 ? loc : (c6 < 4 ? yaw : vel);
  out[idx] = src[(size_t)r * 160 + t * 2 + (c6 & 1)];
}

// The policy head of a forward: cat_x_proj -> pi_head (read LIVE: the only trainable parameters) -> masked logits, then the trajectory
// heads on its q_final.  Launched on c->stream; leaves what rift_loss_backward consumes.
int head_impl(RiftCtx* c, const RiftCtx::Head& h) {
  const std::string PD = "planning_decoder";
  const int nQ = h.nQ, R = h.R, M = 12;
  Fwd f; f.c = c; f.fp32 = h.fp32; f.need_traj = h.need_traj; f.train = f.drop = f.bn_update = false; f.seed = 0;
  // Everything before reads frozen, load-time-packed weights only; pi_head.* is read LIVE from here on.  A host that updates pi_head on
  // another stream (all-reduce + clip + AdamW of the previous step) hands over the event that marks the update's end.
  if (c->param_event && !c->dry) HIPCHK(c, hipStreamWaitEvent(c->stream, c->param_event, 0));
  if (!h.fp32 && c->pi_fused) {
    // cat_x_proj (bf16) -> pi_head first Linear (exact fp32, live parameters) -> LayerNorm -> ReLU -> Linear -> masked logits: one launch
    PiFwdP q; memset(&q, 0, sizeof(q));
    q.Q = h.Q; q.rows = nQ; q.rows_per_scene = R * M;
    q.wq = (const unsigned short*)c->pw[PD + ".cat_x_proj.q"].bf; q.bq = c->pw[PD + ".cat_x_proj.q"].bias; q.x0p = h.x0p;
    q.w1 = fptr(c, PD + ".pi_head.mlp.0.weight"); q.b1 = fptr(c, PD + ".pi_head.mlp.0.bias");
    q.lng = fptr(c, PD + ".pi_head.mlp.1.weight"); q.lnb = fptr(c, PD + ".pi_head.mlp.1.bias");
    q.w2 = fptr(c, PD + ".pi_head.mlp.3.weight"); q.b2 = fptr(c, PD + ".pi_head.mlp.3.bias");
    q.r_kpm = h.r_kpm; q.M = M; q.eps = 1e-5f; q.QF = h.QF; q.Hpi = h.Hpi; q.prob = h.prob; q.nonfinite = c->nonfinite;
    c->prof_flops = 2.0 * nQ * (2.0 * 128 * 128 + 128);
    launch(c, "pi_forward_kernel", pi_forward_kernel, dim3(cdiv(nQ, PI_ROWS)), dim3(512), (size_t)PI_LDS, q);
  } else {
    GemmP g = mk(h.Q, 128, nQ, c->pw[PD + ".cat_x_proj.q"], h.QF, 128);
    g.gbias = h.x0p; g.gb_div = R * M; g.gb_mod = 0;
    gemm(c, g, c->pw[PD + ".cat_x_proj.q"], h.fp32);
    // pi head: first Linear straight from the live (trainable) fp32 parameters, exact-fp32 MFMA
    PW w; w.N = 128; w.K = 128; w.Kp = 128; w.Npad = 128;
    w.f32 = (float*)fptr(c, PD + ".pi_head.mlp.0.weight"); w.bias = fptr(c, PD + ".pi_head.mlp.0.bias");
    gemm(c, mk(h.QF, 128, nQ, w, h.Hpi, 128), w, true);
    launch(c, "pi_tail_kernel", pi_tail_kernel, dim3(cdiv(nQ, 4)), dim3(256), 0, (const float*)h.Hpi, nQ, M, fptr(c, PD + ".pi_head.mlp.1.weight"),
           fptr(c, PD + ".pi_head.mlp.1.bias"), fptr(c, PD + ".pi_head.mlp.3.weight"), fptr(c, PD + ".pi_head.mlp.3.bias"),
           (const uint8_t*)h.r_kpm, 1e-5f, h.prob, c->nonfinite);
  }
  tap(c, "q_final", h.QF, (int64_t)nQ * 128);
  if (h.need_traj && !h.fp32 && c->heads_fused) {
    const std::string nm3[3] = {PD + ".loc_head", PD + ".yaw_head", PD + ".vel_head"};
    heads3_fused(f, h.QF, 128, nQ, 0, 0, 0, nm3, h.traj);
  } else if (h.need_traj) {
    const char* nm[3] = {"loc_head", "yaw_head", "vel_head"};
    for (int i = 0; i < 3; ++i) {      // MLPLayer(128, 256, 160) as in mlp_layer(), on the buffers the trunk reserved
      const PW& w0 = c->pw[PD + "." + nm[i] + ".mlp.0"];
      const PW& w3 = c->pw[PD + "." + nm[i] + ".mlp.3"];
      float* T = h.oT[i];
      gemm(c, mk(h.QF, 128, nQ, w0, T, w0.N), w0, h.fp32);
      GemmP g = mk(T, w0.N, nQ, w3, h.o3[i], 160);
      g.pro = PRO_LN; g.pg = fptr(c, PD + "." + nm[i] + ".mlp.1.weight"); g.pb = fptr(c, PD + "." + nm[i] + ".mlp.1.bias"); g.pro_relu = 1;
      gemm(c, g, w3, h.fp32);
    }
    launch(c, "interleave_traj_kernel", interleave_traj_kernel, dim3(cdiv((long long)nQ * 480, 256)), dim3(256), 0, (const float*)h.o3[0], (const float*)h.o3[1],
           (const float*)h.o3[2], nQ, h.traj);
  }
  if (!c->dry) {
    c->last_qfinal = h.QF; c->last_hpi = h.Hpi; c->last_prob = h.prob; c->last_rkpm = h.r_kpm; c->last_bs = h.bs; c->last_R = h.R;
  }
  return RIFT_OK;
}

int forward_impl(RiftCtx* c, const RiftFeatureBatch* B, const RiftOutputs* out, int flags, uint32_t seed) {
  Fwd f;
  f.c = c; f.train = (flags & RIFT_F_TRAIN) != 0; f.drop = f.train && !(flags & RIFT_F_NO_DROP);
  f.fp32 = (flags & RIFT_F_FP32) != 0; f.need_traj = (flags & RIFT_F_NEED_TRAJ) != 0;
  f.bn_update = f.train && !(flags & RIFT_F_NO_BN_UPDATE); f.seed = seed;
  const int bs = B->bs, A = B->A, Mp = B->Mp, R = B->R, S = B->S, T = B->T;
  const int N = A + Mp + S, M = 12;
  const int nA = bs * A, nP = bs * Mp, nL = bs * R, nQ = nL * M, nT = bs * N;
  const std::string HE = "agent_encoder.history_encoder";

  // ================= agent encoder (agent_encoder.py:54-96, embedding.py:62-87) =================
  // ---- input-only preparation: one launch (prep_kernel) for the seven feature / mask / position builders
  (void)A_alloc<float>(c, 64);          // front padding: the level-0 NAT kernel's window loads start up to 9 floats before a sequence's first row
  float* F9 = A_alloc<float>(c, (size_t)nA * 20 * 9 + 64);
  uint8_t* valid_agent = A_alloc<uint8_t>(c, nA);
  // the fused history encoder runs on the sequences its output is read of -- valid agents other than the ego (agent_encoder.py:77-87) -- in
  // compacted order (nat_l0w.h ranks them); the layer-wise / fp32 path keeps all nA sequences
  const bool nat_compact = c->nat_compact && c->nat_fused && !f.fp32 && c->fpn_fused;
  uint8_t* hist_agent = nat_compact ? A_alloc<uint8_t>(c, nA) : nullptr;
  int* nat_aidx = nat_compact ? A_alloc<int>(c, nA) : nullptr;
  int* nat_cnt = nat_compact ? A_alloc<int>(c, 4) : nullptr;
  float* F10 = A_alloc<float>(c, (size_t)nP * 20 * 10);
  float* F6 = A_alloc<float>(c, (size_t)nL * 120 * 6);
  uint8_t* kpm = A_alloc<uint8_t>(c, nT);
  float* pos = A_alloc<float>(c, (size_t)nT * 3);
  float* r_pos = A_alloc<float>(c, (size_t)nL * 3);
  uint8_t* r_kpm = A_alloc<uint8_t>(c, nL);
  bool prefetched = false;
  bool front_fused = false; float* x_ego_front = nullptr; EgoP ego_front; memset(&ego_front, 0, sizeof(ego_front));
  f.r_tiles = A_alloc<uint8_t>(c, (size_t)std::max(nL, 1));
  {
    PrepP q; memset(&q, 0, sizeof(q));
    q.agent_pos = B->agent_position; q.agent_head = B->agent_heading; q.agent_vel = B->agent_velocity; q.agent_shape = B->agent_shape;
    q.agent_valid = B->agent_valid_mask; q.nA = nA; q.Tfull = T; q.F9 = F9; q.valid_agent = valid_agent; q.hist_agent = hist_agent;
    q.map_pp = B->map_point_position; q.map_pv = B->map_point_vector; q.map_po = B->map_point_orientation; q.map_center = B->map_polygon_center;
    q.nPoly = nP; q.F10 = F10;
    q.ref_pos = B->ref_position; q.ref_vec = B->ref_vector; q.ref_ori = B->ref_orientation; q.ref_valid = B->ref_valid_mask; q.nLine = nL;
    q.F6 = F6; q.r_pos = r_pos; q.r_kpm = r_kpm; q.r_tiles = f.r_tiles;
    q.map_valid = B->map_valid_mask; q.static_valid = B->static_valid_mask; q.st_pos = B->static_position; q.st_head = B->static_heading;
    q.bs = bs; q.A = A; q.Mp = Mp; q.S = S; q.kpm = kpm; q.pos = pos;
    q.nb[0] = cdiv((long long)nA * 20, 256); q.nb[1] = cdiv((long long)nP * 20, 256); q.nb[2] = cdiv((long long)nL * 120, 256);
    q.nb[3] = cdiv(nL, 256); q.nb[4] = cdiv(nL, 256); q.nb[5] = cdiv(nT, 256); q.nb[6] = cdiv(nT, 256);
    int tot = 0;
    for (int i = 0; i < 7; ++i) tot += q.nb[i];
    // (round 6) the ranking as the first bs blocks of this launch (kernels.h: rank_scene_body): no launch of its own on the history chain
    const bool rank_in_prep = nat_compact && c->rank_in_prep && A <= 256 && !(c->front_fused && !f.fp32 && c->ego_fused);
    if (rank_in_prep) {
      const int sl = c->parity;
      if (c->rk_cap[sl] < bs) {                          // (a batch beyond the arrays of rift_ctx_create: a fresh zeroed array, epochs start over)
        HIPCHK(c, hipDeviceSynchronize());               // nothing in flight reads the old array; and the zeros below are in place before ANY stream's launch
        if (c->rk_pub[sl]) { HIPCHK(c, hipFree(c->rk_pub[sl])); c->rk_pub[sl] = nullptr; c->rk_cap[sl] = 0; }
        const int cap = std::max(2 * bs, 4096);
        HIPCHK(c, hipMalloc((void**)&c->rk_pub[sl], (size_t)cap * sizeof(unsigned long long)));
        HIPCHK(c, hipMemset(c->rk_pub[sl], 0, (size_t)cap * sizeof(unsigned long long)));
        HIPCHK(c, hipDeviceSynchronize());               // (hipMemset of device memory may return before the fill has run, and the preparation is launched on a non-blocking stream)
        c->rk_cap[sl] = cap; c->rk_epoch[sl] = 0;
      }
      q.nrk = bs; q.rk_pub = c->rk_pub[sl]; q.aidx = nat_aidx; q.cnt = nat_cnt; q.fail = c->nonfinite; q.rk_fault = c->rank_fault ? 1 : 0;
      if (!c->dry) { if (++c->rk_epoch[sl] == 0u) c->rk_epoch[sl] = 1u; }      // (epoch 0 = "never written")
      q.rk_epoch = c->rk_epoch[sl];
      q.hist_agent = nullptr;                            // (the scene blocks derive the marks themselves)
      tot += bs;
    }
    // (round 4) the ranking and the ego token as blocks of the preparation's own launch (front.h): neither reads anything the preparation writes
    front_fused = c->front_fused && !f.fp32 && c->ego_fused && !RIFT_DROP_STATS;      // (the diagnostic twin sets its counters up behind the preparation: it keeps the three launches)
    FrontP fq; memset(&fq, 0, sizeof(fq));
    if (front_fused) {
      const std::string EG = "agent_encoder.ego_state_emb";
      bool fill_eq;
      float* eq = wconst_get(c, "ego_q", 128, f.fp32, &fill_eq);
      if (fill_eq) gemm(c, mk(fptr(c, EG + ".query"), 128, 1, c->pw[EG + ".attn.q"], eq, 128), c->pw[EG + ".attn.q"], f.fp32);
      x_ego_front = A_alloc<float>(c, (size_t)bs * 128);
      EgoP& e = fq.ego;
      e.cs = B->current_state; e.cs_ld = B->cs_ld; e.lw = c->ego_w; e.lb = c->ego_b; e.pos = fptr(c, EG + ".pos_embed");
      e.wkv = (const unsigned short*)c->pw[EG + ".attn.kv"].bf; e.bkv = c->pw[EG + ".attn.kv"].bias; e.q = eq;
      e.wo = (const unsigned short*)c->pw[EG + ".attn.out_proj"].bf; e.bo = c->pw[EG + ".attn.out_proj"].bias;
      e.out = x_ego_front; e.bs = bs; e.drop_p = f.drop ? 0.75f : 0.f; e.seed = f.seed; e.stream = 0x45474Fu;      // (a stream id of its own: every other kernel keeps the id it had)
      ego_front = e;
      fq.n_ego = c->front_ego ? bs : 0;
      fq.rank_on = nat_compact ? 1 : 0; fq.aidx = nat_aidx; fq.cnt = nat_cnt;
      if (nat_compact) q.hist_agent = nullptr;          // (the ranking block derives the marks itself)
      fq.prep = q;
    }
    auto launch_front = [&]() {
      if (front_fused) {
        c->prof_flops = 2.0 * bs * (6.0 * 128 * 256 + 128.0 * 128);
        if (fq.n_ego) launch(c, "front_kernel", front_kernel<true>, dim3(tot + fq.n_ego + fq.rank_on), dim3(256), 0, fq);
        else launch(c, "front_kernel", front_kernel<false>, dim3(tot + fq.rank_on), dim3(256), 0, fq);
      } else {
        launch(c, "prep_kernel", prep_kernel, dim3(tot), dim3(256), 0, q);
        if (nat_compact && !rank_in_prep) launch(c, "nat_rank_kernel", nat_rank_kernel, dim3(1), dim3(NAT_RANK_THREADS), 0, (const uint8_t*)hist_agent, nA, nat_aidx, nat_cnt);
      }
    };
    // on the caller's prepare stream if there is one (rift_set_prepare_stream): behind the gather of the batch, beside the previous step
    prefetched = c->prep_set && !c->prof_on && !c->dry;
    if (prefetched) {
      if (!c->ev_prep) HIPCHK(c, hipEventCreateWithFlags(&c->ev_prep, hipEventDisableTiming));
      hipStream_t own = c->stream;
      c->stream = c->prep_stream;
      launch_front();
      c->stream = own;
      HIPCHK(c, hipEventRecord(c->ev_prep, c->prep_stream));
      // (round 6) when the history chain stays on the prepare stream and the map chain forks onto the side stream (the update loop's case),
      // the caller's queue gets nothing before the join, and both joined chains are behind the preparation already: its own wait for the
      // preparation would be a third one on the same fact (one event operation costs the host what a launch does)
      const bool join_covers = c->two_streams && c->nat_fused && !f.fp32 && c->nat_aside && c->side_gate <= 0 && !c->dp.on;
      if (!join_covers) HIPCHK(c, hipStreamWaitEvent(own, c->ev_prep, 0));
    } else {
      launch_front();
    }
  }
  // data parallel: the r2r mask quirk indexes padding rows of the GLOBAL minibatch -> gather them (slots in the exchange buffer; the
  // first BatchNorm exchange carries them, an eval forward exchanges them on their own before the decoder)
  const uint8_t* q_kpm = r_kpm; int q_bs = bs, q_off = 0;
  if (c->dp.on) {
    f.dp = true; f.kb = c->dp.gbs * R; f.kpm_pending = true;
    f.g_rkpm = A_alloc<uint8_t>(c, (size_t)f.kb);
    if ((long long)f.kb + 2 * 513 > c->dp.len || c->dp.off < 0 || c->dp.off + bs > c->dp.gbs) { c->err = "rift_set_dp: exchange buffer too small or shard outside the global minibatch"; return RIFT_ERR_ARG; }
    // (with the preparation prefetched the fill waits for the head of the map chain: the exchange buffer is the one the previous forward's
    // map chain exchanged through, and that chain's stream is what orders the two)
    if (!prefetched) launch(c, "dp_kpm_fill_kernel", dp_kpm_fill_kernel, dim3(cdiv(f.kb, 256)), dim3(256), 0, (const uint8_t*)r_kpm, nL, c->dp.off * R, f.kb, c->dp.xchg);
    q_kpm = f.g_rkpm; q_bs = c->dp.gbs; q_off = c->dp.off;
  }
  const bool dp_fill_late = c->dp.on && prefetched;
#if RIFT_DROP_STATS      // diagnostic build (dropstats.h): the counters of this forward's stochastic decisions, readable as taps afterwards
  DropStats ds; memset(&ds, 0, sizeof(ds));
  if (f.drop) {
    ds.nmax = std::max(nA, bs * 6);
    const size_t n = (size_t)RIFT_DS_SITES * ds.nmax;
    ds.cnt = A_alloc<unsigned int>(c, n); ds.any = A_alloc<unsigned int>(c, n); ds.all = A_alloc<unsigned int>(c, n);
    ds.scale = A_alloc<float>(c, RIFT_DS_SITES + RIFT_DS_DEC_SITES); ds.elem = A_alloc<unsigned long long>(c, 2 * RIFT_DS_DEC_SITES);
    if (!c->dry) {
      HIPCHK(c, hipMemsetAsync(ds.cnt, 0, n * 4, c->stream)); HIPCHK(c, hipMemsetAsync(ds.any, 0, n * 4, c->stream));
      HIPCHK(c, hipMemsetAsync(ds.all, 0xff, n * 4, c->stream));
      HIPCHK(c, hipMemsetAsync(ds.scale, 0, (RIFT_DS_SITES + RIFT_DS_DEC_SITES) * 4, c->stream));
      HIPCHK(c, hipMemsetAsync(ds.elem, 0, 2 * RIFT_DS_DEC_SITES * 8, c->stream));
    }
    tap(c, "drop_cnt", (float*)ds.cnt, (int64_t)n); tap(c, "drop_any", (float*)ds.any, (int64_t)n); tap(c, "drop_all", (float*)ds.all, (int64_t)n);
    tap(c, "drop_scale", ds.scale, RIFT_DS_SITES + RIFT_DS_DEC_SITES); tap(c, "drop_elem", (float*)ds.elem, 4 * RIFT_DS_DEC_SITES);
  }
#define RIFT_SET_DS(x) (x).ds = ds
#else
#define RIFT_SET_DS(x)
#endif
  if (nat_compact) { tap(c, "nat_aidx", (float*)nat_aidx, nA); tap(c, "nat_cnt", (float*)nat_cnt, 3); }      // (int32 words read back through the float tap)
  static const float dpr[6] = {0.f, 0.04f, 0.08f, 0.12f, 0.16f, 0.2f};   // linspace(0, 0.2, 6), embedding.py:30
  const bool fused = c->nat_fused && !f.fp32;
  // fork: the agent-history chain depends on prep_kernel only and joins at the token assembly; on its own stream it fills the CUs the
  // map / reference-line chain leaves idle (partial last rounds, 238-workgroup launches) and vice versa.  Off while profiling per kernel.
  hipStream_t main_stream = c->stream;
  const bool forked = c->two_streams && fused && !c->prof_on && !c->dry;
  bool nat_aside = false;
  if (forked) {
    if (!c->side) { HIPCHK(c, hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking)); c->side_owned = true; }
    if (!c->ev_fork) { HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming)); HIPCHK(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming)); }
    // With the preparation prefetched, neither front chain of step k + 1 needs anything of step k: the map chain (side stream) waits for
    // the preparation only, the agent-history chain follows it on the prepare stream, and the caller's queue holds token assembly ->
    // encoder -> decoder of step k, then of step k + 1 -- the fronts run beside the previous step's one-workgroup-per-scene encoder /
    // decoder (which leave most of the chip idle below 256 scenes; at 256 they hold every CU whole, and what is gained is that the fronts
    // start the moment CUs come free, with gather and preparation long done -- profiles/r03_timeline_256.txt).  ms per step, fronts behind the caller's queue
    // / beside it: 32 scenes 0.372 / 0.237, 64 0.394 / 0.245, 128 0.462 / 0.389, 192 0.597 / 0.524, 256 0.701 / 0.678.  What it took:
    // RIFT_DEFER_SLOTS = 4 arenas (with two, tail k - 1 -> front k + 1 -> encoder / decoder k + 1 -> tail k + 1 is a cycle two steps long)
    // and no further hardware queue for the history chain (on a stream of its own every cross-queue wait of the step got slower: 0.372).
    // RIFT_SIDE_GATE=1 keeps the fronts behind the caller's queue (the event record costs that queue ~5 us), RIFT_NAT_ASIDE=0 the history
    // chain on it.
    const bool gate = c->side_gate > 0;
    if (prefetched) HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_prep, 0));
    if (!prefetched || gate) {
      HIPCHK(c, hipEventRecord(c->ev_fork, main_stream));
      HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
    }
    nat_aside = prefetched && !gate && c->nat_aside;
    if (nat_aside) {
      if (!c->ev_join2) HIPCHK(c, hipEventCreateWithFlags(&c->ev_join2, hipEventDisableTiming));
      c->stream = c->prep_stream;
    } else if (!c->nat_on_main) c->stream = c->side;
  }
  static const int Ll[3] = {20, 10, 5}, Cl[3] = {32, 64, 128}, Hl[3] = {2, 4, 8}, Kl[3] = {3, 3, 5};
  float* Oc[3];   // LayerNorm(norm_i) of the last 3 steps of level i: all that out[:, :, -1] of the FPN depends on
  for (int i = 0; i < 3; ++i) Oc[i] = A_alloc<float>(c, (size_t)nA * 3 * Cl[i]);
  // the wave-private level kernels hand these rows to fpn_tail_kernel as bf16 (it rounds them to bf16 first thing anyway): same values, half the bytes
  const bool oc_bf16 = fused && c->fpn_fused;
  unsigned short* Ocb[3] = {nullptr, nullptr, nullptr};
  if (oc_bf16) for (int i = 0; i < 3; ++i) Ocb[i] = A_alloc<unsigned short>(c, (size_t)nA * 3 * Cl[i]);
  if (fused) {
    // one launch per level: [ConvTokenizer ->] 2 NAT blocks -> {FPN LayerNorm of the last 3 steps, downsample conv + LN}
    float* Xin[3] = {nullptr, A_alloc<float>(c, (size_t)nA * 10 * 64), A_alloc<float>(c, (size_t)nA * 5 * 128)};
    for (int lv = 0; lv < 3; ++lv) {
      const int C = Cl[lv], H = Hl[lv], ksz = Kl[lv], L = Ll[lv], rows = nA * L;
      if (lv == 0) {   // level 0 as wave-private, register-resident tiles (no workgroup barriers): nat_l0w.h
        NatL0WP q; memset(&q, 0, sizeof(q));
        q.aidx = nat_aidx; q.cnt = nat_cnt;
        q.F9 = F9; q.nseq = nA; q.img = c->l0w_img; q.par = c->l0w_par; q.Oc = Oc[0]; q.Ocb = Ocb[0]; q.Xnext = Xin[1];
        { if (c->dg.nat_ts == 1) { q.ts = A_alloc<long long>(c, 64); tap(c, "nat_ts", (float*)q.ts, 128); } }
        q.droppath[0] = f.drop ? dpr[0] : 0.f; q.droppath[1] = f.drop ? dpr[1] : 0.f; q.seed = f.seed; q.stream = f.next_stream(); f.stream_id += 4;
        RIFT_SET_DS(q);
        c->prof_flops = 2.0 * rows * (20.0 * C * C + 4.0 * ksz * C) + 2.0 * rows * 27 * 32 + (rows / 2) * 2.0 * 3 * C * 2 * C;
        { const int g0 = std::min(cdiv(cdiv(nA, 4), L0W_NWV), c->nat_grid); launch_call(c, "nat_l0w_kernel", [&] { l0w_launch(q, g0, c->stream); }); }
        continue;
      }
      if (lv == 1) {   // level 1 likewise (weights swapped through LDS between the two layers): nat_l1w.h
        NatL1WP q; memset(&q, 0, sizeof(q));
        q.cnt = nat_cnt;
        q.X = Xin[1]; q.nseq = nA; q.img = c->l1w_img; q.par = c->l1w_par; q.Oc = Oc[1]; q.Ocb = Ocb[1]; q.Xnext = Xin[2];
        q.droppath[0] = f.drop ? dpr[2] : 0.f; q.droppath[1] = f.drop ? dpr[3] : 0.f; q.seed = f.seed; q.stream = f.next_stream(); f.stream_id += 4;
        RIFT_SET_DS(q);
        c->prof_flops = 2.0 * rows * (20.0 * C * C + 4.0 * ksz * C) + (rows / 2) * 2.0 * 3 * C * 2 * C;
        { const int g1 = std::min(cdiv(cdiv(nA, 4), L1W_NWV), c->nat_grid); launch_call(c, "nat_l1w_kernel", [&] { l1w_launch(q, g1, c->stream); }); }
        continue;
      }
      {   // level 2: wave-private tiles of 3 agents, the two layers' weights streamed through LDS (nat_l2w.h)
        NatL2WP q; memset(&q, 0, sizeof(q));
        q.cnt = nat_cnt;
        q.X = Xin[2]; q.nseq = nA; q.img = c->l2w_img; q.par = c->l2w_par; q.Oc = Oc[2]; q.Ocb = Ocb[2];
        q.droppath[0] = f.drop ? dpr[4] : 0.f; q.droppath[1] = f.drop ? dpr[5] : 0.f; q.seed = f.seed; q.stream = f.next_stream(); f.stream_id += 4;
        RIFT_SET_DS(q);
        { if (c->dg.nat_ts == 3) { q.ts = A_alloc<long long>(c, 64); tap(c, "nat_ts", (float*)q.ts, 128); } }
        c->prof_flops = 2.0 * rows * (20.0 * C * C + 4.0 * ksz * C);
        // (round 6: up to one tile per workgroup -- a launch that does not fill the chip deals its tiles wave-major and waves without a tile skip
        // the arithmetic, so a small batch runs two or three waves on each of many CUs instead of eight on a few; nat_l2w.hip)
        const int l2grid = std::min(cdiv(nA, 3), c->nat_grid);
        launch_call(c, "nat_l2w_kernel", [&] { l2w_launch(q, l2grid, c->stream); });
        continue;
      }
    }
  } else {
    float* X0 = A_alloc<float>(c, (size_t)nA * 20 * 32);
    {
      GemmP g = mk(F9, 9, nA * 20, c->pw[HE + ".embed.proj"], X0, 32);
      g.amode = AMODE_CONV3; g.cv_C = 9; g.cv_Lin = 20; g.cv_nout = 20; g.cv_t0 = 0; g.cv_stride = 1;
      gemm(c, g, c->pw[HE + ".embed.proj"], f.fp32);
    }
    auto nat_level = [&](float* Xl, int lv, int rows, int C, int H, int ksz, int L) {
      nat_layer(f, Xl, rows, C, H, ksz, L, HE + ".levels." + std::to_string(lv) + ".blocks.0", dpr[2 * lv]);
      nat_layer(f, Xl, rows, C, H, ksz, L, HE + ".levels." + std::to_string(lv) + ".blocks.1", dpr[2 * lv + 1]);
    };
    nat_level(X0, 0, nA * 20, 32, 2, 3, 20);
    float* X1 = A_alloc<float>(c, (size_t)nA * 10 * 64);
    {
      GemmP g = mk(X0, 32, nA * 10, c->pw[HE + ".levels.0.downsample.reduction"], X1, 64);
      g.amode = AMODE_CONV3; g.cv_C = 32; g.cv_Lin = 20; g.cv_nout = 10; g.cv_t0 = 0; g.cv_stride = 2;
      gemm(c, g, c->pw[HE + ".levels.0.downsample.reduction"], f.fp32);
      layernorm(f, X1, 64, X1, 64, nA * 10, 64, HE + ".levels.0.downsample.norm");
    }
    nat_level(X1, 1, nA * 10, 64, 4, 3, 10);
    // level outputs are the PRE-downsample activations (NATBlock returns (downsample(x), x)); X0/X1 are
    // still needed below, so the downsample writes new buffers.
    float* X2 = A_alloc<float>(c, (size_t)nA * 5 * 128);
    {
      GemmP g = mk(X1, 64, nA * 5, c->pw[HE + ".levels.1.downsample.reduction"], X2, 128);
      g.amode = AMODE_CONV3; g.cv_C = 64; g.cv_Lin = 10; g.cv_nout = 5; g.cv_t0 = 0; g.cv_stride = 2;
      gemm(c, g, c->pw[HE + ".levels.1.downsample.reduction"], f.fp32);
      layernorm(f, X2, 128, X2, 128, nA * 5, 128, HE + ".levels.1.downsample.norm");
    }
    nat_level(X2, 2, nA * 5, 128, 8, 5, 5);
    tap(c, "nat_level2", X2, (int64_t)nA * 5 * 128);
    float* Xl[3] = {X0, X1, X2};
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)   // rows (a, j) <- level rows (a*L + L-3 + j): 3 strided LN calls
        launch(c, "layernorm_kernel", layernorm_kernel, dim3(cdiv(nA, 4)), dim3(256), 0, (const float*)(Xl[i] + (size_t)(Ll[i] - 3 + j) * Cl[i]),
               Ll[i] * Cl[i], Oc[i] + (size_t)j * Cl[i], 3 * Cl[i], nA, Cl[i],
               fptr(c, HE + ".norm" + std::to_string(i) + ".weight"), fptr(c, HE + ".norm" + std::to_string(i) + ".bias"),
               1e-5f, 0);
  }
  // FPN restricted to what out[:, :, -1] depends on
  float* nat_out = A_alloc<float>(c, (size_t)nA * 128);
  if (fused && c->fpn_fused) {
    FpnP q; memset(&q, 0, sizeof(q));
    for (int i = 0; i < 3; ++i) {
      const PW& w = c->pw[HE + ".lateral_convs." + std::to_string(i)];
      q.oc[i] = Oc[i]; q.ocb[i] = Ocb[i]; q.wl[i] = (const unsigned short*)w.bf; q.bl[i] = w.bias;
    }
    q.wf = (const unsigned short*)c->pw[HE + ".fpn_conv.last"].bf; q.bf_ = c->pw[HE + ".fpn_conv.last"].bias;
    q.out = nat_out; q.nA = nA; q.cnt = fused ? nat_cnt : nullptr; q.aidx = fused ? nat_aidx : nullptr;
    c->prof_flops = 2.0 * nA * (2.0 * 128 * (96 + 192 + 384) + 256.0 * 128);
    launch(c, "fpn_tail_kernel", fpn_tail_kernel, dim3(cdiv(nA, FPN_AG)), dim3(512), (size_t)FPN_LDS, q);
  } else {
    float* lat[3];
    for (int i = 0; i < 3; ++i) {
      lat[i] = A_alloc<float>(c, (size_t)nA * 2 * 128);
      const std::string lc = HE + ".lateral_convs." + std::to_string(i);
      GemmP g = mk(Oc[i], Cl[i], nA * 2, c->pw[lc], lat[i], 128);
      g.amode = AMODE_CONV3; g.cv_C = Cl[i]; g.cv_Lin = 3; g.cv_nout = 2; g.cv_t0 = 1; g.cv_stride = 1;
      gemm(c, g, c->pw[lc], f.fp32);
    }
    float* Z = A_alloc<float>(c, (size_t)nA * 256);
    launch(c, "fpn_merge_kernel", fpn_merge_kernel, dim3(cdiv((long long)nA * 128, 256)), dim3(256), 0, (const float*)lat[0], (const float*)lat[1],
           (const float*)lat[2], nA, Z);
    gemm(c, mk(Z, 256, nA, c->pw[HE + ".fpn_conv.last"], nat_out, 128), c->pw[HE + ".fpn_conv.last"], f.fp32);
  }
  // the longer chain (agent history: ~310 of the front's ~510 us at 256 scenes) stays on the caller's queue, so neither its start nor the
  // join pays a cross-queue hop (12-15 us each by the kernel trace); the map / reference-line chain is the one that forks
  if (forked) {
    if (nat_aside) { HIPCHK(c, hipEventRecord(c->ev_join2, c->prep_stream)); c->stream = c->side; }
    else if (c->nat_on_main) c->stream = c->side;
    else { HIPCHK(c, hipEventRecord(c->ev_join, c->side)); c->stream = main_stream; }
  }
  if (dp_fill_late)      // (on the map chain's stream if there is one, else on the caller's: behind the previous forward's exchanges either way)
    launch(c, "dp_kpm_fill_kernel", dp_kpm_fill_kernel, dim3(cdiv(f.kb, 256)), dim3(256), 0, (const uint8_t*)r_kpm, nL, c->dp.off * R, f.kb, c->dp.xchg);
  tap(c, "nat_out", nat_out, (int64_t)nA * 128);

  // ego state token (StateAttentionEncoder, agent_encoder.py:99-140)
  const std::string EG = "agent_encoder.ego_state_emb";
  bool fill_eq;
  float* eq = wconst_get(c, "ego_q", 128, f.fp32, &fill_eq);
  if (fill_eq) gemm(c, mk(fptr(c, EG + ".query"), 128, 1, c->pw[EG + ".attn.q"], eq, 128), c->pw[EG + ".attn.q"], f.fp32);
  float* x_ego = front_fused ? x_ego_front : A_alloc<float>(c, (size_t)bs * 128);
  if (front_fused && c->front_ego) {
    if (f.drop) (void)f.next_stream();          // (the id the ego token's own launch used to take: the kernels behind it keep theirs)
  } else if (!f.fp32 && c->ego_fused) {
    EgoP q; memset(&q, 0, sizeof(q));
    if (front_fused) { q = ego_front; q.stream = f.drop ? f.next_stream() : 0; } else {
    q.cs = B->current_state; q.cs_ld = B->cs_ld; q.lw = c->ego_w; q.lb = c->ego_b; q.pos = fptr(c, EG + ".pos_embed");
    q.wkv = (const unsigned short*)c->pw[EG + ".attn.kv"].bf; q.bkv = c->pw[EG + ".attn.kv"].bias; q.q = eq;
    q.wo = (const unsigned short*)c->pw[EG + ".attn.out_proj"].bf; q.bo = c->pw[EG + ".attn.out_proj"].bias;
    q.out = x_ego; q.bs = bs; q.drop_p = f.drop ? 0.75f : 0.f; q.seed = f.seed; q.stream = f.drop ? f.next_stream() : 0;
    }
    RIFT_SET_DS(q);
    c->prof_flops = 2.0 * bs * (6.0 * 128 * 256 + 128.0 * 128);
    // (diagnostic RIFT_EGO_NOFIT=1: 40 KB of unused dynamic LDS keep the workgroup from fitting beside the decoder's -- the round-3 behaviour, for A/B)
    launch(c, "ego_fused_kernel", ego_fused_kernel, dim3(bs), dim3(256), (size_t)(c->ego_nofit ? 40960 : 0), q);
  } else {
  float* E = A_alloc<float>(c, (size_t)bs * 6 * 128);
  launch(c, "ego_token_kernel", ego_token_kernel, dim3(cdiv((long long)bs * 6 * 128, 256)), dim3(256), 0, B->current_state, B->cs_ld,
         (const float*)c->ego_w, (const float*)c->ego_b, fptr(c, EG + ".pos_embed"), bs, E);
  float* EKV = A_alloc<float>(c, (size_t)bs * 6 * 256);
  gemm(c, mk(E, 128, bs * 6, c->pw[EG + ".attn.kv"], EKV, 256), c->pw[EG + ".attn.kv"], f.fp32);
  uint8_t* edrop = nullptr;
  if (f.drop) {
    edrop = A_alloc<uint8_t>(c, (size_t)bs * 6);
    launch(c, "ego_dropmask_kernel", ego_dropmask_kernel, dim3(cdiv(bs * 6, 256)), dim3(256), 0, bs, 0.75f, f.seed, f.next_stream(), edrop);
  }
  float* EAO = A_alloc<float>(c, (size_t)bs * 128);
  {
    MhaP p; memset(&p, 0, sizeof(p));
    p.Q = eq; p.ldq = 128; p.K = EKV; p.V = EKV + 128; p.ldkv = 256; p.O = EAO; p.ldo = 128;
    p.nb_outer = bs; p.nb_inner = 1; p.H = 4; p.Lq = 1; p.Lk = 6;
    p.q_outer = 0; p.q_inner = 0; p.q_stride = 0; p.kv_outer = 6; p.kv_inner = 0; p.kv_stride = 1;
    p.o_outer = 1; p.o_inner = 0; p.o_stride = 0; p.mask = edrop; p.mask_quirk = 0; p.mask_mod = 1;
    run_mha(c, p, f.fp32);
  }
  gemm(c, mk(EAO, 128, bs, c->pw[EG + ".attn.out_proj"], x_ego, 128), c->pw[EG + ".attn.out_proj"], f.fp32);
  }
  tap(c, "x_ego", x_ego, (int64_t)bs * 128);

  // ================= tokens =================
  float* X = A_alloc<float>(c, (size_t)nT * 128);
  // map encoder (map_encoder.py:31-93)
  // reference-line features are input-only too: both PointsEncoders run side by side (fused path)
  const std::string PD = "planning_decoder";
  float *poly = nullptr, *r_emb = nullptr;
  const bool pe_pair = !f.fp32 && c->pe_fused;
  if (pe_pair) points_encoder_pair(f, F10, nP, B->map_valid_mask, "map_encoder.polygon_encoder", F6, nL, B->ref_valid_mask, PD + ".r_encoder", &poly, &r_emb);
  else poly = points_encoder(f, F10, 10, nP, 20, B->map_valid_mask, "map_encoder.polygon_encoder");
  tap(c, "poly_pe", poly, (int64_t)nP * 128);
  // the three Fourier embeddings (token positions, speed limits, reference-line positions) depend on inputs only: with both
  // PointsEncoders done they run as ONE launch, and the token kernels add the positional embedding on the way
  const bool fo3 = pe_pair && !f.fp32 && c->fo_fused && S == 0;
  float *speed_emb = nullptr, *PEtok = nullptr;
  bool rpe_done = false;
  if (fo3) {
    tap(c, "r_pe", r_emb, (int64_t)nL * 128);
    FourierP3 q3; memset(&q3, 0, sizeof(q3));
    q3.e[0] = fourier_desc(f, pos, 3, nT, 3, "pos_emb", 2, nullptr);
    q3.e[1] = fourier_desc(f, B->map_polygon_speed_limit, 1, nP, 1, "map_encoder.speed_limit_emb", -1, nullptr);
    q3.e[2] = fourier_desc(f, r_pos, 3, nL, 3, PD + ".r_pos_emb", -1, r_emb);
    q3.nblk[0] = cdiv(nT, FO_ROWS); q3.nblk[1] = cdiv(nP, FO_ROWS); q3.nblk[2] = cdiv(nL, FO_ROWS); q3.count = 3;
    c->prof_flops = 2.0 * 128.0 * ((nT + nL) * (3 * 257.0 + 128.0) + nP * (257.0 + 128.0));
    if (c->fo_w) {          // wave-private form: one 16-row tile per wave, 8 tiles per pass, the one-dimensional embedding two passes per workgroup
      FoWP w; memset(&w, 0, sizeof(w));
      w.count = 3;
      for (int i = 0; i < 3; ++i) {
        const FourierP& o = q3.e[i]; FoWSide& d = w.e[i];
        d.in = o.in; d.in_ld = o.in_ld; d.rows = o.rows; d.D = o.D; d.wrap_dim = o.wrap_dim; d.img = c->fow_img[i]; d.par = c->fow_par[i];
        d.Y = o.Y; d.accumulate = o.accumulate; d.rep = o.D == 1 ? 2 : 1; d.nwg = cdiv(cdiv(o.rows, 16), 8 * d.rep);
      }
      launch_call(c, "fo_w_kernel", [&] { fow_launch(w, c->stream); });
    } else
    launch(c, "fourier_fused_kernel", fourier_fused_kernel, dim3(q3.nblk[0] + q3.nblk[1] + q3.nblk[2]), dim3(256), (size_t)FO_LDS, q3);
    PEtok = q3.e[0].Y; speed_emb = q3.e[1].Y; rpe_done = true;
  } else {
    speed_emb = fourier(f, B->map_polygon_speed_limit, 1, nP, 1, "map_encoder.speed_limit_emb", -1);
  }
  // decoder queries q0 = q_proj(cat[r_emb, m_emb]) (planning_decoder.py:149-154): they depend on the reference-line embedding only, so with the fused
  // embeddings done they are launched here, ahead of the join, and run beside the tail of the agent-history chain instead of between encoder and decoder
  float* Q = nullptr;
  auto build_q0 = [&]() -> int {
  bool fill;
  float* Mb = wconst_get(c, "Mb", (size_t)M * 128, f.fp32, &fill);
  if (fill) gemm(c, mk(fptr(c, PD + ".m_emb"), 128, M, c->pw[PD + ".q_proj.m"], Mb, 128), c->pw[PD + ".q_proj.m"], f.fp32);
  Q = A_alloc<float>(c, (size_t)nQ * 128);
  if (!f.fp32 && c->pi_fused) {
    Q0P q; memset(&q, 0, sizeof(q));
    q.r_emb = r_emb; q.nL = nL; q.M = M; q.wr = (const unsigned short*)c->pw[PD + ".q_proj.r"].bf; q.br = c->pw[PD + ".q_proj.r"].bias;
    q.Mb = Mb; q.Q = Q;
    c->prof_flops = 2.0 * nL * 128.0 * 128;
    launch(c, "q0_fused_kernel", q0_fused_kernel, dim3(cdiv(nL, 16)), dim3(256), 0, q);
  } else {
    float* Ra = A_alloc<float>(c, (size_t)nL * 128);
    gemm(c, mk(r_emb, 128, nL, c->pw[PD + ".q_proj.r"], Ra, 128), c->pw[PD + ".q_proj.r"], f.fp32);
    launch(c, "build_q0_kernel", build_q0_kernel, dim3(cdiv((long long)nQ * 128, 256)), dim3(256), 0, (const float*)Ra, (const float*)Mb, nL, M, Q);
  }
    return RIFT_OK;
  };
  if (forked && rpe_done) { const int rc0 = build_q0(); if (rc0 != RIFT_OK) return rc0; }
  // join: the agent tokens need both chains.  Small batches: the map chain waits for the history chain first, so that the caller's queue --
  // token assembly, encoder, decoder: the step, below a chip-filling batch -- takes ONE cross-queue wait per forward (32 scenes: 0.233 ->
  // 0.226 ms); at 128 scenes it makes no difference and at 256 it costs 1-3 % (the map chain of the step after next stalls behind the wait)
  const bool join_once = nat_aside && (c->join_once >= 0 ? c->join_once != 0 : bs <= 64);
  if (join_once) HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_join2, 0));
  if (forked && (c->nat_on_main || nat_aside)) { HIPCHK(c, hipEventRecord(c->ev_join, c->side)); c->stream = main_stream; }
  if (forked) HIPCHK(c, hipStreamWaitEvent(main_stream, c->ev_join, 0));
  if (nat_aside && !join_once) HIPCHK(c, hipStreamWaitEvent(main_stream, c->ev_join2, 0));
  // token assembly: fused into the scene encoder's prologue where that kernel runs (N <= 96, no static objects, position embedding from
  // the one-launch Fourier pass), a launch of its own otherwise
  const bool tok_fused = c->tok_fused && c->enc_fused && !f.fp32 && N <= 96 && S == 0 && fo3 && PEtok != nullptr;
  TokenP tokq; memset(&tokq, 0, sizeof(tokq));
  {
    TokenP& q = tokq;
    q.nat = nat_out; q.x_ego = x_ego; q.valid_agent = (const uint8_t*)valid_agent; q.category = B->agent_category; q.a_type_emb = fptr(c, "agent_encoder.type_emb.weight");
    q.pooled = poly; q.ptype = B->map_polygon_type; q.on_route = B->map_polygon_on_route; q.tl = B->map_polygon_tl_status; q.has_sl = B->map_polygon_has_speed_limit;
    q.speed_emb = speed_emb; q.p_type_emb = fptr(c, "map_encoder.type_emb.weight"); q.route_emb = fptr(c, "map_encoder.on_route_emb.weight");
    q.tl_emb = fptr(c, "map_encoder.traffic_light_emb.weight"); q.unk_emb = fptr(c, "map_encoder.unknown_speed_emb.weight");
    q.bs = bs; q.A = A; q.Mp = Mp; q.N = N; q.X = X; q.pe = PEtok;
    q.nblk_a = cdiv((long long)nA * 32, 256);
    if (!tok_fused) launch(c, "token_kernel", token_kernel, dim3(q.nblk_a + cdiv((long long)nP * 32, 256)), dim3(256), 0, q);
  }
  if (S > 0) {
    float* semb = fourier(f, B->static_shape, 2, bs * S, 2, "static_objects_encoder.obj_encoder", -1);
    launch(c, "static_token_kernel", static_token_kernel, dim3(cdiv((long long)bs * S * 128, 256)), dim3(256), 0, (const float*)semb, B->static_category,
           B->static_valid_mask, fptr(c, "static_objects_encoder.type_emb.weight"), bs, A, Mp, S, N, X);
  }
  if (!fo3) fourier(f, pos, 3, nT, 3, "pos_emb", 2, X);
  tap(c, "x_tokens", X, (int64_t)nT * 128);

  // ================= encoder blocks (transformer.py:73-94) =================
  static const float edpr[4] = {0.f, 0.2f / 3.f, 0.4f / 3.f, 0.2f};   // linspace(0, 0.2, 4), pluto_model.py:80-83
  float* ENC = A_alloc<float>(c, (size_t)nT * 128);
  unsigned short* enc_KT = nullptr;   // the decoder's cross-attention K | V^T operand fragments, written by the encoder kernel's tail
  uint8_t* kpm_c = nullptr;           // (RIFT_ENC_COMPACT) the key padding of the compacted encoder rows (bs, 96): what the decoder masks those fragments with
  float* enc_x0p = nullptr;   // cat_x_proj's ego-token half, written by the encoder kernel's tail
  // 97 .. 112 token slots (what train_cbv collates: 49 + 60 = 109): the fused kernel on its 112-row layout (enc_fused.h: EncLay<112>) when the
  // decoder's eight-key-tile variant consumes its K | V^T image; RIFT_ENC112=0 keeps those shapes on enc_w_kernel
  const bool enc_wide = N > 96 && N <= 112 && c->enc112 && c->dec_fused && R <= 8 && ENC_NW == 8 && !tok_fused;
  if (c->enc_fused && !f.fp32 && (N <= 96 || enc_wide)) {
    EncFusedP ep; memset(&ep, 0, sizeof(ep));
    ep.X = X; ep.Y = ENC; ep.kpm = kpm; ep.bs = bs; ep.N = N; ep.seed = f.seed; ep.stream = f.next_stream(); f.stream_id += 8;
    if (tok_fused) { ep.tok = tokq; ep.tok_on = 1; ep.Xout = c->keep_tokens ? X : nullptr; }
    for (int i = 0; i < 4; ++i) {
      const std::string p = "encoder_blocks." + std::to_string(i);
      EncBlockW& w = ep.blk[i];
      w.ln1_g = fptr(c, p + ".norm1.weight"); w.ln1_b = fptr(c, p + ".norm1.bias");
      w.ln2_g = fptr(c, p + ".norm2.weight"); w.ln2_b = fptr(c, p + ".norm2.bias");
      w.wqkv = c->enc_wqkv[i]; w.bqkv = c->enc_bqkv[i];
      w.wo = (const unsigned short*)c->pw[p + ".attn.out_proj"].bf; w.bo = c->pw[p + ".attn.out_proj"].bias;
      w.w1 = (const unsigned short*)c->pw[p + ".mlp.fc1"].bf; w.b1 = c->pw[p + ".mlp.fc1"].bias;
      w.w2 = (const unsigned short*)c->pw[p + ".mlp.fc2"].hid; w.b2 = c->pw[p + ".mlp.fc2"].bias;      // (hidden-layer operand words: pack_hid)
      w.droppath = f.drop ? edpr[i] : 0.f;
    }
    ep.norm_g = fptr(c, "norm.weight"); ep.norm_b = fptr(c, "norm.bias"); ep.nonfinite = c->nonfinite;
    RIFT_SET_DS(ep);
    if (c->dg.enc_ts) { ep.ts = A_alloc<long long>(c, 256); tap(c, "enc_ts", (float*)ep.ts, 512); }
    c->prof_flops = 4.0 * bs * N * (2.0 * 128 * 384 + 4.0 * N * 128 + 2.0 * 128 * 128 + 4.0 * 128 * 512);
    if (c->dec_fused && R <= 8 && ENC_NW == 8) {   // the decoder kernel will run: emit its cross-attention K | V operand fragments here
      enc_KT = A_alloc<unsigned short>(c, (size_t)bs * 4 * (enc_wide ? 96 : DECW_KV_FRAGS) * 512);      // (wide: the dense per-head image, dec_kv.h)
      ep.wkv = (const unsigned short*)c->pw["planning_decoder.kv_all"].bf; ep.bkv = c->pw["planning_decoder.kv_all"].bias;
      ep.KT = enc_KT;
#if RIFT_ENC_COMPACT
      kpm_c = A_alloc<uint8_t>(c, (size_t)bs * (enc_wide ? 112 : 96)); ep.kpm_c = kpm_c;
#else
      if (enc_wide) { kpm_c = A_alloc<uint8_t>(c, (size_t)bs * 112); ep.kpm_c = kpm_c; }
#endif
      enc_x0p = A_alloc<float>(c, (size_t)bs * 128);
      ep.wx0 = (const unsigned short*)c->pw["planning_decoder.cat_x_proj.x"].bf; ep.x0p = enc_x0p;
      c->prof_flops += 2.0 * bs * N * 128.0 * 1024;
    }
    if (enc_wide) launch_call(c, "enc_fused112_kernel", [&] { enc112_launch(ep, c->stream); });
    else launch(c, "enc_fused_kernel", enc_fused_kernel<ENC_NW>, dim3(bs), dim3(64 * ENC_NW), (size_t)RIFT_ENC_LDS_BYTES, ep);
  } else if (c->enc_fused && !f.fp32 && N <= 192) {   // dense-traffic shapes: the wave-private, weight-streaming encoder (two passes per layer)
    EncWP eq; memset(&eq, 0, sizeof(eq));
    eq.X = X; eq.Y = ENC; eq.kpm = kpm; eq.bs = bs; eq.N = N; eq.seed = f.seed; eq.stream = f.next_stream(); f.stream_id += 8;
    eq.KVs = A_alloc<unsigned short>(c, (size_t)bs * 96 * 512);
    eq.img = c->encw_img; eq.par = c->encw_par; eq.nonfinite = c->nonfinite;
    if (c->dec_fused && R <= 16) { enc_KT = A_alloc<unsigned short>(c, (size_t)bs * 4 * 96 * 512); eq.DKV = enc_KT; }   // the decoder's (dense-variant) operands
    for (int i = 0; i < 4; ++i) eq.droppath[i] = f.drop ? edpr[i] : 0.f;
    RIFT_SET_DS(eq);
    c->prof_flops = 4.0 * bs * N * (2.0 * 128 * 384 + 4.0 * N * 128 + 2.0 * 128 * 128 + 4.0 * 128 * 512);
    launch_call(c, "enc_w_kernel", [&] { encw_launch(eq, c->stream); });
  } else {
  float* QKV = A_alloc<float>(c, (size_t)nT * 384);
  float* AO = A_alloc<float>(c, (size_t)nT * 128);
  float* H512 = A_alloc<float>(c, (size_t)nT * 512);
  for (int i = 0; i < 4; ++i) {
    const std::string p = "encoder_blocks." + std::to_string(i);
    GemmP g = mk(X, 128, nT, c->pw[p + ".attn.qkv"], QKV, 384);
    g.pro = PRO_LN; g.pg = fptr(c, p + ".norm1.weight"); g.pb = fptr(c, p + ".norm1.bias");
    gemm(c, g, c->pw[p + ".attn.qkv"], f.fp32);
    MhaP m; memset(&m, 0, sizeof(m));
    m.Q = QKV; m.ldq = 384; m.K = QKV + 128; m.V = QKV + 256; m.ldkv = 384; m.O = AO; m.ldo = 128;
    m.nb_outer = bs; m.nb_inner = 1; m.H = 4; m.Lq = N; m.Lk = N;
    m.q_outer = N; m.q_stride = 1; m.kv_outer = N; m.kv_stride = 1; m.o_outer = N; m.o_stride = 1;
    m.mask = kpm; m.mask_mod = 1;
    run_mha(c, m, f.fp32);
    GemmP g2 = mk(AO, 128, nT, c->pw[p + ".attn.out_proj"], X, 128);
    g2.residual = X; g2.ldr = 128;
    if (f.drop && edpr[i] > 0.f) { g2.droppath_p = edpr[i]; g2.dp_div = N; g2.seed = f.seed; g2.stream = f.next_stream(); }
    gemm(c, g2, c->pw[p + ".attn.out_proj"], f.fp32);
    GemmP g3 = mk(X, 128, nT, c->pw[p + ".mlp.fc1"], H512, 512);
    g3.pro = PRO_LN; g3.pg = fptr(c, p + ".norm2.weight"); g3.pb = fptr(c, p + ".norm2.bias"); g3.act = ACT_GELU;
    gemm(c, g3, c->pw[p + ".mlp.fc1"], f.fp32);
    GemmP g4 = mk(H512, 512, nT, c->pw[p + ".mlp.fc2"], X, 128);
    g4.residual = X; g4.ldr = 128;
    if (f.drop && edpr[i] > 0.f) { g4.droppath_p = edpr[i]; g4.dp_div = N; g4.seed = f.seed; g4.stream = f.next_stream(); }
    gemm(c, g4, c->pw[p + ".mlp.fc2"], f.fp32);
  }
  layernorm(f, X, 128, ENC, 128, nT, 128, "norm");
  }
  tap(c, "enc_out", ENC, (int64_t)nT * 128);

  // ================= agent predictor (agent_predictor.py:17-29) =================
  if (f.need_traj && out->prediction && A > 1) {
    const int rows = bs * (A - 1);
    if (!f.fp32 && c->heads_fused) {
      const std::string nm3[3] = {"agent_predictor.loc_predictor", "agent_predictor.yaw_predictor", "agent_predictor.vel_predictor"};
      heads3_fused(f, ENC, 128, rows, A - 1, N, 1, nm3, out->prediction);
    } else {
    float* Xa = A_alloc<float>(c, (size_t)rows * 128);
    launch(c, "gather_rows_kernel", gather_rows_kernel, dim3(cdiv((long long)rows * 128, 256)), dim3(256), 0, (const float*)ENC, 128, Xa, 128, rows, 128,
           A - 1, N, 1);
    float* o3[3];
    const char* nm[3] = {"loc_predictor", "yaw_predictor", "vel_predictor"};
    for (int i = 0; i < 3; ++i) {
      o3[i] = A_alloc<float>(c, (size_t)rows * 160);
      mlp_layer(f, Xa, 128, rows, std::string("agent_predictor.") + nm[i], o3[i], 160, f.fp32);
    }
    launch(c, "interleave_traj_kernel", interleave_traj_kernel, dim3(cdiv((long long)rows * 480, 256)), dim3(256), 0, (const float*)o3[0], (const float*)o3[1],
           (const float*)o3[2], rows, out->prediction);
    }
  }

  // ================= planning decoder (planning_decoder.py:135-188) =================
  if (!pe_pair) r_emb = points_encoder(f, F6, 6, nL, 120, B->ref_valid_mask, PD + ".r_encoder");
  if (!rpe_done) {
    tap(c, "r_pe", r_emb, (int64_t)nL * 128);
    fourier(f, r_pos, 3, nL, 3, PD + ".r_pos_emb", -1, r_emb);
  }
  tap(c, "r_emb", r_emb, (int64_t)nL * 128);
  if (!Q) { const int rc0 = build_q0(); if (rc0 != RIFT_OK) return rc0; }
  tap(c, "q0", Q, (int64_t)nQ * 128);

  bool dec_deferred = false; DecWP dec_later; memset(&dec_later, 0, sizeof(dec_later));
  dp_exchange(f, 0);                      // (eval forward under data parallelism: the mask slots have not travelled yet)
  const float dp = f.drop ? 0.1f : 0.f;   // pluto_model.py:35,93
  const bool dec_dense = (R > 8 || N > 96) && R <= 16 && N <= 192;      // dense-traffic shapes: the kernel's round-of-eight-tiles variant
  if (c->dec_fused && !f.fp32 && dec_dense && !enc_KT) {   // its K | V^T operand fragments from the (layer-wise) encoder's output
    enc_KT = A_alloc<unsigned short>(c, (size_t)bs * 4 * 96 * 512);
    DecKvP kq; memset(&kq, 0, sizeof(kq));
    kq.ENC = ENC; kq.bs = bs; kq.N = N; kq.wkv = (const unsigned short*)c->pw[PD + ".kv_all"].bf; kq.bkv = c->pw[PD + ".kv_all"].bias; kq.KV = enc_KT;
    c->prof_flops = 2.0 * bs * N * 128.0 * 1024;
    launch(c, "dec_kv_frag_kernel", dec_kv_frag_kernel, dim3(bs), dim3(512), (size_t)DEC_KV_LDS, kq);
  }
  if (c->dec_fused && !f.fp32 && ((R <= 8 && N <= 96) || dec_dense) && enc_KT) {
    DecWP dq; memset(&dq, 0, sizeof(dq));
    dq.Q = Q; dq.kpm = kpm; dq.r_kpm = r_kpm; dq.q_kpm = q_kpm; dq.q_bs = q_bs; dq.q_off = q_off; dq.bs = bs; dq.N = N; dq.R = R; dq.dropout = dp; dq.seed = f.seed;
    if (kpm_c) { dq.kpm = kpm_c; dq.N = enc_wide ? 112 : 96; dq.compact = 1; }      // (the encoder compacted its rows: its own key padding, valid keys a prefix)
    dq.stream = f.next_stream(); f.stream_id += 64;
    dq.KV = enc_KT; dq.img = c->decw_img; dq.par = c->decw_par; dq.nonfinite = c->nonfinite;
    RIFT_SET_DS(dq);
    if (c->dg.dec_ts) { dq.ts = A_alloc<long long>(c, 1024); tap(c, "dec_ts", (float*)dq.ts, 2048); }      // [0, 128): boundaries of wave 0; [128 + 112 w, ...): arrivals of wave w
    dq.dbg = c->dg.dec_dbg;
    c->prof_flops = 4.0 * bs * (R * M) * (2.0 * 128 * (384 + 128) * 2 + 2.0 * 128 * 128 * 2 + 4.0 * 128 * 512 + 4.0 * 128 * (N + R + M));
    // (with the trajectory heads on, the tail behind the decoder is longer and the caller's queue carries the prediction head too: measured
    // worth it up to twice the batch -- 128 scenes 0.419 -> 0.391 ms, 256 scenes 0.707 -> 0.713)
    dq.l0 = 0; dq.l1 = 4;
    // Which batches: measured per size (profiles/r04_batch_sweep.txt; ms per step with / without): 32 0.18 / 0.23, 64 0.21 / 0.24, 80 0.253 / 0.264, 88 0.259 / 0.254,
    // 96 0.315 / 0.300, 104 0.325 / 0.321, 112 0.338 / 0.359, 128 0.368 / 0.381, 160 0.433 / 0.440, 192 0.503 / 0.520, 256 0.656 / 0.65 -- a win except between 84 and 108
    // scenes and at a chip-filling batch.  RIFT_DEC_DEFER=<n> replaces the table by bs <= n (0: never).
    const bool dec_size_ok = c->dec_defer_max >= 0 ? bs <= c->dec_defer_max : (f.need_traj ? bs <= 128 : (bs <= 84 || (bs >= 108 && bs <= 192)));
    const bool dec_may_defer = (flags & RIFT_F_DEFER_HEAD) && dec_size_ok && !c->prof_on;      // (the same in the sizing pass)
    // ... and split: layers 0 - 1 here, behind the encoder, layers 2 - 3 in front of the deferred head -- token assembly + encoder + half a decoder
    // on the caller's queue against half a decoder + tail on the update stream (9 + 88 + 55 against 55 + 80 us at 32 scenes) instead of 97 against
    // 190.  The dropout streams of the second launch carry on from the states the first one leaves (rng_io): the draws are a single launch's.
    // Measured (ms per step, split / one launch): 64 scenes 0.217 / 0.229, 32 scenes 0.20 - 0.21 / 0.20 - 0.21 (the host issues a 32-scene step in 0.15 - 0.2 ms: it
    // bounds that one now); with the trajectory heads on the caller's queue carries the prediction head as well and the split LOSES (0.30 / 0.26 at 64): off there.
    uint32_t* rng_io = (dec_may_defer && c->dec_split && !f.need_traj) ? A_alloc<uint32_t>(c, (size_t)bs * 512 * 4) : nullptr;
    dec_deferred = dec_may_defer && !c->dry && !dq.ts;
    if (dec_deferred) {
      dec_later = dq;
      if (rng_io) {
        dq.l1 = 2; dq.rng_io = rng_io;
        launch_call(c, "dec_w_kernel", [&] { decw_launch(dq, c->stream); });
        dec_later.l0 = 2; dec_later.rng_io = rng_io;
      }
    } else launch_call(c, "dec_w_kernel", [&] { decw_launch(dq, c->stream); });
  } else {
  float* DQKV = A_alloc<float>(c, (size_t)nQ * 384);
  float* DAO = A_alloc<float>(c, (size_t)nQ * 128);
  float* DQc = A_alloc<float>(c, (size_t)nQ * 128);
  float* DH = A_alloc<float>(c, (size_t)nQ * 512);
  float* KVm = A_alloc<float>(c, (size_t)nT * 256);
  float* MP = A_alloc<float>(c, (size_t)M * 384);
  for (int i = 0; i < 4; ++i) {
    const std::string p = PD + ".decoder_blocks." + std::to_string(i);
    // ---- r2r self attention over the R reference lines of each (scene, mode), with the mask quirk
    GemmP g = mk(Q, 128, nQ, c->pw[p + ".r2r_attn.qkv"], DQKV, 384);
    g.pro = PRO_LN; g.pg = fptr(c, p + ".norm1.weight"); g.pb = fptr(c, p + ".norm1.bias");
    gemm(c, g, c->pw[p + ".r2r_attn.qkv"], f.fp32);
    {
      MhaP m; memset(&m, 0, sizeof(m));
      m.Q = DQKV; m.ldq = 384; m.K = DQKV + 128; m.V = DQKV + 256; m.ldkv = 384; m.O = DAO; m.ldo = 128;
      m.nb_outer = bs; m.nb_inner = M; m.H = 4; m.Lq = R; m.Lk = R;
      m.q_outer = R * M; m.q_inner = 1; m.q_stride = M; m.kv_outer = R * M; m.kv_inner = 1; m.kv_stride = M;
      m.o_outer = R * M; m.o_inner = 1; m.o_stride = M;
      m.mask = q_kpm; m.mask_quirk = 1; m.mask_mod = q_bs; m.mask_off = q_off * M;   // tgt_key_padding_mask.repeat(M, 1), planning_decoder.py:56-60
      if (dp > 0.f) { m.dropout_p = dp; m.seed = f.seed; m.stream = f.next_stream(); }
      run_mha(c, m, f.fp32);
    }
    GemmP g2 = mk(DAO, 128, nQ, c->pw[p + ".r2r_attn.out_proj"], Q, 128);
    g2.residual = Q; g2.ldr = 128;
    if (dp > 0.f) { g2.dropout_p = dp; g2.seed = f.seed; g2.stream = f.next_stream(); }
    gemm(c, g2, c->pw[p + ".r2r_attn.out_proj"], f.fp32);
    // ---- m2m self attention over the 12 modes: q = k = (h + m_pos) W + b, v = h W + b
    bool fill_mp;
    float* MP = wconst_get(c, p + ".mp", (size_t)M * 384, f.fp32, &fill_mp);
    if (fill_mp) gemm(c, mk(fptr(c, PD + ".m_pos"), 128, M, c->pw[p + ".m2m_attn.qk0"], MP, 384), c->pw[p + ".m2m_attn.qk0"], f.fp32);
    GemmP g3 = mk(Q, 128, nQ, c->pw[p + ".m2m_attn.qkv"], DQKV, 384);
    g3.pro = PRO_LN; g3.pg = fptr(c, p + ".norm2.weight"); g3.pb = fptr(c, p + ".norm2.bias");
    g3.gbias = MP; g3.gb_div = 1; g3.gb_mod = M;
    gemm(c, g3, c->pw[p + ".m2m_attn.qkv"], f.fp32);
    {
      MhaP m; memset(&m, 0, sizeof(m));
      m.Q = DQKV; m.ldq = 384; m.K = DQKV + 128; m.V = DQKV + 256; m.ldkv = 384; m.O = DAO; m.ldo = 128;
      m.nb_outer = nL; m.nb_inner = 1; m.H = 4; m.Lq = M; m.Lk = M;
      m.q_outer = M; m.q_stride = 1; m.kv_outer = M; m.kv_stride = 1; m.o_outer = M; m.o_stride = 1;
      if (dp > 0.f) { m.dropout_p = dp; m.seed = f.seed; m.stream = f.next_stream(); }
      run_mha(c, m, f.fp32);
    }
    GemmP g4 = mk(DAO, 128, nQ, c->pw[p + ".m2m_attn.out_proj"], Q, 128);
    g4.residual = Q; g4.ldr = 128; g4.rowzero = r_kpm; g4.rz_div = M;   // rows of padded ref lines become 0 (:65-72)
    if (dp > 0.f) { g4.dropout_p = dp; g4.seed = f.seed; g4.stream = f.next_stream(); }
    gemm(c, g4, c->pw[p + ".m2m_attn.out_proj"], f.fp32);
    // ---- cross attention: R*12 queries per scene against the encoder tokens
    GemmP g5 = mk(Q, 128, nQ, c->pw[p + ".cross_attn.q"], DQc, 128);
    g5.pro = PRO_LN; g5.pg = fptr(c, p + ".norm3.weight"); g5.pb = fptr(c, p + ".norm3.bias");
    gemm(c, g5, c->pw[p + ".cross_attn.q"], f.fp32);
    gemm(c, mk(ENC, 128, nT, c->pw[p + ".cross_attn.kv"], KVm, 256), c->pw[p + ".cross_attn.kv"], f.fp32);
    {
      MhaP m; memset(&m, 0, sizeof(m));
      m.Q = DQc; m.ldq = 128; m.K = KVm; m.V = KVm + 128; m.ldkv = 256; m.O = DAO; m.ldo = 128;
      m.nb_outer = bs; m.nb_inner = 1; m.H = 4; m.Lq = R * M; m.Lk = N;
      m.q_outer = R * M; m.q_stride = 1; m.kv_outer = N; m.kv_stride = 1; m.o_outer = R * M; m.o_stride = 1;
      m.mask = kpm; m.mask_mod = 1;
      if (dp > 0.f) { m.dropout_p = dp; m.seed = f.seed; m.stream = f.next_stream(); }
      run_mha(c, m, f.fp32);
    }
    GemmP g6 = mk(DAO, 128, nQ, c->pw[p + ".cross_attn.out_proj"], Q, 128);
    g6.residual = Q; g6.ldr = 128;
    if (dp > 0.f) { g6.dropout_p = dp; g6.seed = f.seed; g6.stream = f.next_stream(); }
    gemm(c, g6, c->pw[p + ".cross_attn.out_proj"], f.fp32);
    // ---- FFN
    GemmP g7 = mk(Q, 128, nQ, c->pw[p + ".ffn.0"], DH, 512);
    g7.pro = PRO_LN; g7.pg = fptr(c, p + ".norm4.weight"); g7.pb = fptr(c, p + ".norm4.bias"); g7.act = ACT_RELU;
    if (dp > 0.f) { g7.dropout_p = dp; g7.seed = f.seed; g7.stream = f.next_stream(); }
    gemm(c, g7, c->pw[p + ".ffn.0"], f.fp32);
    GemmP g8 = mk(DH, 512, nQ, c->pw[p + ".ffn.3"], Q, 128);
    g8.residual = Q; g8.ldr = 128;
    if (dp > 0.f) { g8.dropout_p = dp; g8.seed = f.seed; g8.stream = f.next_stream(); }
    gemm(c, g8, c->pw[p + ".ffn.3"], f.fp32);
  }
  }
  tap(c, "dec3", Q, (int64_t)nQ * 128);
  // cat_x_proj(cat[q, enc_emb[:, 0]]) (planning_decoder.py:177-179): ego-token part is a per-scene bias
  float* x0p = enc_x0p;
  if (!x0p) {
    x0p = A_alloc<float>(c, (size_t)bs * 128);
    gemm(c, mk(ENC, N * 128, bs, c->pw[PD + ".cat_x_proj.x"], x0p, 128), c->pw[PD + ".cat_x_proj.x"], f.fp32);
  }
  // ---- the policy head (and the trajectory heads that read its q_final): right here, or -- RIFT_F_DEFER_HEAD -- later through
  // rift_forward_head on a stream of the caller's choice, so that it (and the loss / backward / update behind it) runs beside the next
  // forward's frozen trunk
  RiftCtx::Head hs;
  hs.valid = true; hs.fp32 = f.fp32; hs.need_traj = f.need_traj && out->trajectory != nullptr; hs.Q = Q; hs.x0p = x0p; hs.r_kpm = r_kpm;
  hs.bs = bs; hs.R = R; hs.nQ = nQ; hs.traj = out->trajectory;
  hs.QF = A_alloc<float>(c, (size_t)nQ * 128);
  hs.Hpi = A_alloc<float>(c, (size_t)nQ * 128);
  hs.prob = out->probability ? out->probability : A_alloc<float>(c, nQ);
  if (hs.need_traj && (f.fp32 || !c->heads_fused))      // layer-wise trajectory heads: their buffers come out of this forward's arena
    for (int i = 0; i < 3; ++i) { hs.oT[i] = A_alloc<float>(c, (size_t)nQ * 256); hs.o3[i] = A_alloc<float>(c, (size_t)nQ * 160); }
  hs.dec_pending = dec_deferred; hs.dec = dec_later;
  if (flags & RIFT_F_DEFER_HEAD) { if (!c->dry) c->head[c->parity] = hs; }
  else TRY(head_impl(c, hs));
  // hidden_proj / ref_free_decoder on the ego token (pluto_model.py:173-180)
  if (!f.fp32 && c->heads_fused && (out->hidden || (f.need_traj && out->ref_free_trajectory))) {      // one launch (heads_fused.h: ego_heads_kernel)
    const PW &h0 = c->pw["hidden_proj.0"], &h2 = c->pw["hidden_proj.2"], &r0 = c->pw["ref_free_decoder.mlp.0"], &r3 = c->pw["ref_free_decoder.mlp.3"];
    EgoHeadsP q; memset(&q, 0, sizeof(q));
    q.X = ENC; q.ldx = N * 128; q.rows = bs;
    q.wh0 = (const unsigned short*)h0.bf; q.wh2 = (const unsigned short*)h2.bf; q.wr0 = (const unsigned short*)r0.bf; q.wr3 = (const unsigned short*)r3.bf;
    q.bh0 = h0.bias; q.bh2 = h2.bias; q.br0 = r0.bias; q.br3 = r3.bias;
    q.lng = fptr(c, "ref_free_decoder.mlp.1.weight"); q.lnb = fptr(c, "ref_free_decoder.mlp.1.bias");
    q.hidden = out->hidden; q.ref = (f.need_traj && out->ref_free_trajectory) ? out->ref_free_trajectory : nullptr;
    c->prof_flops = 2.0 * bs * (2.0 * 128 * 128 + 128.0 * 256 + 256.0 * 320);
    launch(c, "ego_heads_kernel", ego_heads_kernel, dim3(cdiv(bs, 16)), dim3(256), 0, q);
    return RIFT_OK;
  }
  if (out->hidden) {
    float* Th = A_alloc<float>(c, (size_t)bs * 128);
    GemmP g = mk(ENC, N * 128, bs, c->pw["hidden_proj.0"], Th, 128);
    g.act = ACT_RELU;
    gemm(c, g, c->pw["hidden_proj.0"], f.fp32);
    gemm(c, mk(Th, 128, bs, c->pw["hidden_proj.2"], out->hidden, 128), c->pw["hidden_proj.2"], f.fp32);
  }
  if (f.need_traj && out->ref_free_trajectory)
    mlp_layer(f, ENC, N * 128, bs, "ref_free_decoder", out->ref_free_trajectory, 320, f.fp32);

#undef RIFT_SET_DS
  return RIFT_OK;
}

}  // namespace

// ============================================================================================
// C-ABI of this build (one per operand format).  The exported `rift_*` symbols of include/rift_hip.h are trampolines (abi.cpp, generated
// from the header) that route a call through the table at the end of this file to the build that owns the context.
// ============================================================================================
namespace RIFT_NS { namespace abi {

int rift_ctx_create(int device, RiftCtx** ctx) {
  if (!ctx) return RIFT_ERR_ARG;
  RiftCtx* c = new RiftCtx();
  c->opfmt = RIFT_OP_F16;
  c->device = device;
  if (hipSetDevice(device) != hipSuccess) { delete c; return RIFT_ERR_HIP; }
  { const char* ev = getenv("RIFT_NAT_UNFUSED"); c->nat_fused = !(ev && ev[0] == '1'); }
  { const char* ev = getenv("RIFT_DEC_UNFUSED"); c->dec_fused = !(ev && ev[0] == '1'); }
  { const char* ev = getenv("RIFT_ENC_UNFUSED"); c->enc_fused = !(ev && ev[0] == '1'); }
  { const char* ev = getenv("RIFT_PE_UNFUSED"); c->pe_fused = !(ev && ev[0] == '1'); }
  { const char* ev = getenv("RIFT_FPN_UNFUSED"); c->fpn_fused = !(ev && ev[0] == '1'); }
  { const char* ev = getenv("RIFT_EGO_UNFUSED"); c->ego_fused = !(ev && ev[0] == '1'); }
  { const char* ev = getenv("RIFT_HEADS_UNFUSED"); c->heads_fused = !(ev && ev[0] == '1'); }
  { const char* ev = getenv("RIFT_PI_UNFUSED"); c->pi_fused = !(ev && ev[0] == '1'); }
  { const char* ev = getenv("RIFT_NAT_GRID"); if (ev && atoi(ev) > 0) c->nat_grid = atoi(ev); }
  { const char* ev = getenv("RIFT_NAT_MAIN"); if (ev) c->nat_on_main = atoi(ev) != 0; }
  { const char* ev = getenv("RIFT_NAT_COMPACT"); if (ev) c->nat_compact = atoi(ev) != 0; }
  { const char* ev = getenv("RIFT_PE_LIVE"); if (ev) c->pe_live = atoi(ev) != 0; }
  { const char* ev = getenv("RIFT_PE_PACK"); if (ev) c->pe_pack = atoi(ev) != 0; }
  // (1: the scene encoder assembles its token rows in its prologue instead of token_kernel.  Measured and NOT kept as the default: one ~10 us
  // launch less between the join and the encoder, but the step is 15 us LONGER (0.659 against 0.643 ms) -- six dependent gathers per
  // thread in the prologue of a kernel that holds every CU whole, where token_kernel's 5000 small blocks ran beside the fronts' tails)
  { const char* ev = getenv("RIFT_TOKEN_FUSED"); if (ev) c->tok_fused = atoi(ev) != 0; }
  { const char* ev = getenv("RIFT_KEEP_TOKENS"); c->keep_tokens = ev && ev[0] == '1'; }
  // (1: preparation and ranking in one launch, front.h.  Measured and NOT kept as the default: at 256 scenes both already run INSIDE the previous
  // step's decoder (<= 64 VGPRs), so the launch saved is not on any chain -- 0.640 against 0.642 ms, noise -- while the ranking block, which then has
  // to derive the marks from the raw validity (344 KB through one workgroup), makes the merged launch 34 us where the two took 24: a loss wherever
  // the forward runs serially (get_action, small batches))
  { const char* ev = getenv("RIFT_FRONT_FUSED"); if (ev) c->front_fused = atoi(ev) != 0; }
  { const char* ev = getenv("RIFT_RANK_IN_PREP"); if (ev) c->rank_in_prep = atoi(ev) != 0; }
  { const char* ev = getenv("RIFT_ENC112"); if (ev) c->enc112 = atoi(ev) != 0; }
  { const char* ev = getenv("RIFT_RANK_FAULT"); if (ev) c->rank_fault = atoi(ev) != 0; }
  { const char* ev = getenv("RIFT_EGO_NOFIT"); c->ego_nofit = ev && ev[0] == '1'; }
  { const char* ev = getenv("RIFT_FRONT_EGO"); if (ev) c->front_ego = atoi(ev) != 0; }        // (1: the ego token as blocks of that launch too -- 116 VGPRs: it no longer fits beside the decoder's workgroups)
  { const char* ev = getenv("RIFT_SIDE_GATE"); if (ev) c->side_gate = atoi(ev); }
  { const char* ev = getenv("RIFT_NAT_ASIDE"); if (ev) c->nat_aside = atoi(ev) != 0; }
  { const char* ev = getenv("RIFT_JOIN_ONCE"); if (ev) c->join_once = atoi(ev); }
  { const char* ev = getenv("RIFT_DEC_DEFER"); if (ev) c->dec_defer_max = atoi(ev); }
  { const char* ev = getenv("RIFT_DEC_SPLIT"); if (ev) c->dec_split = atoi(ev) != 0; }
  { const char* ev = getenv("RIFT_RO8"); if (ev) c->ro8 = atoi(ev) != 0; }
  { const char* ev = getenv("RIFT_POISON_LDS"); if (ev) c->poison_lds = (int)strtol(ev, nullptr, 0) & 0xff; }
  { const char* ev = getenv("RIFT_POISON_ARENA"); if (ev) c->dg.poison_arena = (int)strtol(ev, nullptr, 0) & 0xff; }
  { const char* ev = getenv("RIFT_PE_TS"); if (ev) c->dg.pe_ts = atoi(ev); }
  { const char* ev = getenv("RIFT_DELAY"); const char* col = ev ? strrchr(ev, ':') : nullptr;      // wall_clock64: 100 MHz
    if (col) { c->dg.delay_label.assign(ev, col - ev); c->dg.delay_ticks = (long long)(atof(col + 1) * 100.0); } }
  { const char* ev = getenv("RIFT_PEW_DBG"); if (ev) c->dg.pew_dbg = atoi(ev); }
  { const char* ev = getenv("RIFT_PEW_TS"); c->dg.pew_ts = ev && ev[0] == '1'; }
  { const char* ev = getenv("RIFT_NAT_TS"); if (ev) c->dg.nat_ts = atoi(ev); }
  c->dg.enc_ts = getenv("RIFT_ENC_TS") != nullptr; c->dg.dec_ts = getenv("RIFT_DEC_TS") != nullptr;
  { const char* ev = getenv("RIFT_DEC_DBG"); if (ev) c->dg.dec_dbg = atoi(ev); }
  { const char* ev = getenv("RIFT_FOURIER_UNFUSED"); c->fo_fused = !(ev && ev[0] == '1'); }
  { const char* ev = getenv("RIFT_PE_W"); c->pe_w = !(ev && ev[0] == '0'); }
  { const char* ev = getenv("RIFT_FO_W"); c->fo_w = !(ev && ev[0] == '0'); }
  { const char* ev = getenv("RIFT_TWO_STREAMS"); c->two_streams = !(ev && ev[0] == '0'); }
  { const char* ev = getenv("RIFT_GEMM_DBG"); c->gemm_dbg = ev ? atoi(ev) : 0; }
  if (hipMalloc((void**)&c->nonfinite, sizeof(int)) != hipSuccess || hipMemset(c->nonfinite, 0, sizeof(int)) != hipSuccess) { delete c; return RIFT_ERR_HIP; }
  for (int i = 0; i < RIFT_DEFER_SLOTS; ++i) {      // the in-launch ranking's published counts (kernels.h: rank_scene_body): zero = "never written"
    if (hipMalloc((void**)&c->rk_pub[i], 4096 * sizeof(unsigned long long)) != hipSuccess || hipMemset(c->rk_pub[i], 0, 4096 * sizeof(unsigned long long)) != hipSuccess) { delete c; return RIFT_ERR_HIP; }
    c->rk_cap[i] = 4096;
  }
  if (hipDeviceSynchronize() != hipSuccess) { delete c; return RIFT_ERR_HIP; }      // (the zeros are in place before any stream launches a kernel that reads them)
  int rc = set_lds_attrs(c);
  if (rc != RIFT_OK) { fprintf(stderr, "rift_ctx_create: %s\n", c->err.c_str()); delete c; return rc; }
  *ctx = c;
  return RIFT_OK;
}

void rift_ctx_destroy(RiftCtx* c) {
  if (!c) return;
  if (c->comm) { (void)rccl_api(nullptr).CommDestroy(c->comm); c->comm = nullptr; }
  if (c->tick_scratch) (void)hipFree(c->tick_scratch);
  if (c->ro_raw) (void)hipFree(c->ro_raw);
  for (void* p : c->owned) (void)hipFree(p);
  for (int i = 0; i < RIFT_DEFER_SLOTS; ++i) if (c->arenas[i]) (void)hipFree(c->arenas[i]);
  if (c->l_S) (void)hipFree(c->l_S);
  if (c->l_cnt) (void)hipFree(c->l_cnt);
  if (c->l_dz) (void)hipFree(c->l_dz);
  if (c->l_partial) (void)hipFree(c->l_partial);
  if (c->ego_w) (void)hipFree(c->ego_w);
  if (c->ego_b) (void)hipFree(c->ego_b);
  if (c->enc_idx) (void)hipFree(c->enc_idx);
  if (c->cr_buf) { (void)hipFree(c->cr_buf); (void)hipFree(c->cr_part); }
  if (c->clip_part) (void)hipFree(c->clip_part);
  if (c->nonfinite) (void)hipFree(c->nonfinite);
  for (int i = 0; i < RIFT_DEFER_SLOTS; ++i) if (c->rk_pub[i]) (void)hipFree(c->rk_pub[i]);
  if (c->l0w_img) { (void)hipFree(c->l0w_img); (void)hipFree(c->l0w_par); }
  if (c->l1w_img) { (void)hipFree(c->l1w_img); (void)hipFree(c->l1w_par); }
  if (c->l2w_img) { (void)hipFree(c->l2w_img); (void)hipFree(c->l2w_par); }
  if (c->encw_img) { (void)hipFree(c->encw_img); (void)hipFree(c->encw_par); }
  if (c->decw_img) { (void)hipFree(c->decw_img); (void)hipFree(c->decw_par); }
  for (int i = 0; i < 2; ++i) if (c->pew_img[i]) (void)hipFree(c->pew_img[i]);
  for (int i = 0; i < 3; ++i) if (c->fow_img[i]) { (void)hipFree(c->fow_img[i]); (void)hipFree(c->fow_par[i]); }
  if (c->ev_prep) (void)hipEventDestroy(c->ev_prep);
  if (c->ev_join2) (void)hipEventDestroy(c->ev_join2);
  if (c->side && c->side_owned) (void)hipStreamDestroy(c->side);
  if (c->ev_fork) { (void)hipEventDestroy(c->ev_fork); (void)hipEventDestroy(c->ev_join); }
  for (int i = 0; i < 4; ++i) { if (c->enc_wqkv[i]) (void)hipFree(c->enc_wqkv[i]); if (c->enc_bqkv[i]) (void)hipFree(c->enc_bqkv[i]); }
  delete c;
}

const char* rift_last_error(RiftCtx* c) { return c ? c->err.c_str() : "null context"; }

int rift_model_load(RiftCtx* c, const RiftTensorDesc* params, int n, void* stream) {
  if (!c || !params) return RIFT_ERR_ARG;
  c->err.clear();
  c->stream = (hipStream_t)stream;
  HIPCHK(c, hipSetDevice(c->device));
  for (void* p : c->owned) (void)hipFree(p);
  c->owned.clear(); c->pw.clear(); c->params.clear(); c->wconst.clear();
  c->dry_need = 0;                      // (the first forward on new weights sizes its arena again: it also computes the weight-only products)
  for (int i = 0; i < n; ++i) {
    Param p; p.data = params[i].data; p.numel = params[i].numel; p.ndim = params[i].ndim;
    for (int d = 0; d < 4; ++d) p.shape[d] = params[i].shape[d];
    c->params[params[i].name] = p;
  }
  const std::string HE = "agent_encoder.history_encoder";
  TRY(pack_conv(c, HE + ".embed.proj"));
  for (int lv = 0; lv < 3; ++lv) {
    for (int b = 0; b < 2; ++b) {
      const std::string p = HE + ".levels." + std::to_string(lv) + ".blocks." + std::to_string(b);
      TRY(pack_linear(c, p + ".attn.qkv")); TRY(pack_linear(c, p + ".attn.proj"));
      TRY(pack_linear(c, p + ".mlp.fc1")); TRY(pack_linear(c, p + ".mlp.fc2"));
    }
    if (lv < 2) TRY(pack_conv(c, HE + ".levels." + std::to_string(lv) + ".downsample.reduction"));
    TRY(pack_conv(c, HE + ".lateral_convs." + std::to_string(lv)));
  }
  {  // wave-private level-0 kernel: K-permuted weight fragments + parameter block (nat_l0w.h)
    NatL0WSrc q; memset(&q, 0, sizeof(q));
    q.w_tok = fptr(c, HE + ".embed.proj.weight"); q.b_tok = fptr(c, HE + ".embed.proj.bias");
    for (int b = 0; b < 2; ++b) {
      const std::string p = HE + ".levels.0.blocks." + std::to_string(b);
      NatL0WSrc::Blk& k = q.blk[b];
      k.ln1_g = fptr(c, p + ".norm1.weight"); k.ln1_b = fptr(c, p + ".norm1.bias"); k.wqkv = fptr(c, p + ".attn.qkv.weight"); k.bqkv = fptr(c, p + ".attn.qkv.bias");
      k.rpb = fptr(c, p + ".attn.rpb"); k.wproj = fptr(c, p + ".attn.proj.weight"); k.bproj = fptr(c, p + ".attn.proj.bias");
      k.ln2_g = fptr(c, p + ".norm2.weight"); k.ln2_b = fptr(c, p + ".norm2.bias"); k.w1 = fptr(c, p + ".mlp.fc1.weight"); k.b1 = fptr(c, p + ".mlp.fc1.bias");
      k.w2 = fptr(c, p + ".mlp.fc2.weight"); k.b2 = fptr(c, p + ".mlp.fc2.bias");
    }
    q.fn_g = fptr(c, HE + ".norm0.weight"); q.fn_b = fptr(c, HE + ".norm0.bias");
    q.w_ds = fptr(c, HE + ".levels.0.downsample.reduction.weight"); q.ds_g = fptr(c, HE + ".levels.0.downsample.norm.weight"); q.ds_b = fptr(c, HE + ".levels.0.downsample.norm.bias");
    if (!c->err.empty()) return RIFT_ERR_ARG;
    if (!c->l0w_img) { HIPCHK(c, hipMalloc((void**)&c->l0w_img, (size_t)L0W_NFRAG * 1024)); HIPCHK(c, hipMalloc((void**)&c->l0w_par, (size_t)L0W_NPAR * 4)); }
    l0w_pack(q, c->l0w_img, c->l0w_par, c->stream);
  }
  {  // wave-private level-1 kernel (nat_l1w.h)
    NatL1WSrc q; memset(&q, 0, sizeof(q));
    for (int b = 0; b < 2; ++b) {
      const std::string p = HE + ".levels.1.blocks." + std::to_string(b);
      NatL1WSrc::Blk& k = q.blk[b];
      k.ln1_g = fptr(c, p + ".norm1.weight"); k.ln1_b = fptr(c, p + ".norm1.bias"); k.wqkv = fptr(c, p + ".attn.qkv.weight"); k.bqkv = fptr(c, p + ".attn.qkv.bias");
      k.rpb = fptr(c, p + ".attn.rpb"); k.wproj = fptr(c, p + ".attn.proj.weight"); k.bproj = fptr(c, p + ".attn.proj.bias");
      k.ln2_g = fptr(c, p + ".norm2.weight"); k.ln2_b = fptr(c, p + ".norm2.bias"); k.w1 = fptr(c, p + ".mlp.fc1.weight"); k.b1 = fptr(c, p + ".mlp.fc1.bias");
      k.w2 = fptr(c, p + ".mlp.fc2.weight"); k.b2 = fptr(c, p + ".mlp.fc2.bias");
    }
    q.fn_g = fptr(c, HE + ".norm1.weight"); q.fn_b = fptr(c, HE + ".norm1.bias");
    q.w_ds = fptr(c, HE + ".levels.1.downsample.reduction.weight"); q.ds_g = fptr(c, HE + ".levels.1.downsample.norm.weight"); q.ds_b = fptr(c, HE + ".levels.1.downsample.norm.bias");
    if (!c->err.empty()) return RIFT_ERR_ARG;
    if (!c->l1w_img) { HIPCHK(c, hipMalloc((void**)&c->l1w_img, (size_t)L1W_NFRAG * 1024)); HIPCHK(c, hipMalloc((void**)&c->l1w_par, (size_t)L1W_NPAR * 4)); }
    l1w_pack(q, c->l1w_img, c->l1w_par, c->stream);
  }
  {  // wave-private level-2 kernel (nat_l2w.h)
    NatL2WSrc q; memset(&q, 0, sizeof(q));
    for (int b = 0; b < 2; ++b) {
      const std::string p = HE + ".levels.2.blocks." + std::to_string(b);
      NatL2WSrc::Blk& k = q.blk[b];
      k.ln1_g = fptr(c, p + ".norm1.weight"); k.ln1_b = fptr(c, p + ".norm1.bias"); k.wqkv = fptr(c, p + ".attn.qkv.weight"); k.bqkv = fptr(c, p + ".attn.qkv.bias");
      k.rpb = fptr(c, p + ".attn.rpb"); k.wproj = fptr(c, p + ".attn.proj.weight"); k.bproj = fptr(c, p + ".attn.proj.bias");
      k.ln2_g = fptr(c, p + ".norm2.weight"); k.ln2_b = fptr(c, p + ".norm2.bias"); k.w1 = fptr(c, p + ".mlp.fc1.weight"); k.b1 = fptr(c, p + ".mlp.fc1.bias");
      k.w2 = fptr(c, p + ".mlp.fc2.weight"); k.b2 = fptr(c, p + ".mlp.fc2.bias");
    }
    q.fn_g = fptr(c, HE + ".norm2.weight"); q.fn_b = fptr(c, HE + ".norm2.bias");
    if (!c->err.empty()) return RIFT_ERR_ARG;
    if (!c->l2w_img) { HIPCHK(c, hipMalloc((void**)&c->l2w_img, (size_t)2 * L2W_BLK_FRAGS * 1024)); HIPCHK(c, hipMalloc((void**)&c->l2w_par, (size_t)L2W_NPAR * 4)); }
    l2w_pack(q, c->l2w_img, c->l2w_par, c->stream);
  }
  {  // fpn_conv at the last step: taps 0,1 only -> [128][256]
    const Param* w = find(c, HE + ".fpn_conv.weight");
    const Param* b = find(c, HE + ".fpn_conv.bias");
    if (!w || !b) { c->err = "missing fpn_conv"; return RIFT_ERR_ARG; }
    TRY(pack(c, HE + ".fpn_conv.last", (const float*)w->data, 128, 128, 256, 0, 0, 0, 128, 0, 2, (const float*)b->data));
  }
  const std::string EG = "agent_encoder.ego_state_emb";
  TRY(pack_rows(c, EG + ".attn.q", EG + ".attn.in_proj_weight", EG + ".attn.in_proj_bias", 0, 128, 128));
  TRY(pack_rows(c, EG + ".attn.kv", EG + ".attn.in_proj_weight", EG + ".attn.in_proj_bias", 128, 256, 256));
  TRY(pack_linear(c, EG + ".attn.out_proj"));
  if (!c->ego_w) { HIPCHK(c, hipMalloc((void**)&c->ego_w, 6 * 128 * 4)); HIPCHK(c, hipMalloc((void**)&c->ego_b, 6 * 128 * 4)); }
  for (int i = 0; i < 6; ++i) {
    const float* w = fptr(c, EG + ".linears." + std::to_string(i) + ".weight");
    const float* b = fptr(c, EG + ".linears." + std::to_string(i) + ".bias");
    if (!w || !b) return RIFT_ERR_ARG;
    HIPCHK(c, hipMemcpyAsync(c->ego_w + i * 128, w, 128 * 4, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->ego_b + i * 128, b, 128 * 4, hipMemcpyDeviceToDevice, c->stream));
  }
  TRY(pack_points_encoder(c, "map_encoder.polygon_encoder"));
  TRY(pack_fourier(c, "map_encoder.speed_limit_emb", 1));
  TRY(pack_fourier(c, "static_objects_encoder.obj_encoder", 2));
  TRY(pack_fourier(c, "pos_emb", 3));
  for (int i = 0; i < 4; ++i) {
    const std::string p = "encoder_blocks." + std::to_string(i);
    TRY(pack_self_mha(c, p + ".attn"));
    TRY(pack_linear(c, p + ".mlp.fc1")); TRY(pack_linear(c, p + ".mlp.fc2")); TRY(pack_hid(c, p + ".mlp.fc2"));
  }
  {  // chunked in_proj image for the fused encoder kernel: per 2-head chunk (q_h0 | k_h0 | q_h1 | k_h1 | v_h0 | v_h1)
    int idx[384];
    for (int ch = 0; ch < 2; ++ch)
      for (int seg = 0; seg < 6; ++seg) {
        const int part = seg < 4 ? (seg & 1) : 2;                 // 0 q, 1 k, 2 v
        const int head = 2 * ch + (seg < 4 ? (seg >> 1) : (seg - 4));
        for (int d = 0; d < 32; ++d) idx[ch * 192 + seg * 32 + d] = part * 128 + head * 32 + d;
      }
    if (!c->enc_idx) HIPCHK(c, hipMalloc((void**)&c->enc_idx, sizeof(idx)));
    HIPCHK(c, hipMemcpyAsync(c->enc_idx, idx, sizeof(idx), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));   // idx is a stack buffer
    for (int i = 0; i < 4; ++i) {
      const std::string p = "encoder_blocks." + std::to_string(i) + ".attn";
      const float* w = fptr(c, p + ".in_proj_weight"); const float* bsrc = fptr(c, p + ".in_proj_bias");
      if (!w || !bsrc) return RIFT_ERR_ARG;
      if (!c->enc_wqkv[i]) { HIPCHK(c, hipMalloc((void**)&c->enc_wqkv[i], 384 * 128 * 2)); HIPCHK(c, hipMalloc((void**)&c->enc_bqkv[i], 384 * 4)); }
      hipLaunchKernelGGL(pack_rows_indexed_kernel, dim3(cdiv(384 * 128, 256)), dim3(256), 0, c->stream, w, bsrc, (const int*)c->enc_idx,
                         384, 128, c->enc_wqkv[i], c->enc_bqkv[i]);
    }
  }
  const char* ap[3] = {"loc_predictor", "yaw_predictor", "vel_predictor"};
  for (int i = 0; i < 3; ++i) TRY(pack_mlp_layer(c, std::string("agent_predictor.") + ap[i]));
  const std::string PD = "planning_decoder";
  TRY(pack_points_encoder(c, PD + ".r_encoder"));
  TRY(pack_fourier(c, PD + ".r_pos_emb", 3));
  TRY(pack_cols(c, PD + ".q_proj.r", PD + ".q_proj", 0, 128, true));
  TRY(pack_cols(c, PD + ".q_proj.m", PD + ".q_proj", 128, 128, false));
  for (int i = 0; i < 4; ++i) {
    const std::string p = PD + ".decoder_blocks." + std::to_string(i);
    TRY(pack_self_mha(c, p + ".r2r_attn"));
    TRY(pack_self_mha(c, p + ".m2m_attn"));
    TRY(pack_rows(c, p + ".m2m_attn.qk0", p + ".m2m_attn.in_proj_weight", "", 0, 256, 384));
    TRY(pack_rows(c, p + ".cross_attn.q", p + ".cross_attn.in_proj_weight", p + ".cross_attn.in_proj_bias", 0, 128, 128));
    TRY(pack_rows(c, p + ".cross_attn.kv", p + ".cross_attn.in_proj_weight", p + ".cross_attn.in_proj_bias", 128, 256, 256));
    TRY(pack_linear(c, p + ".cross_attn.out_proj"));
    TRY(pack_linear(c, p + ".ffn.0")); TRY(pack_linear(c, p + ".ffn.3"));
  }
  {
    std::vector<std::string> wn, bn;
    for (int i = 0; i < 4; ++i) {
      wn.push_back(PD + ".decoder_blocks." + std::to_string(i) + ".cross_attn.in_proj_weight");
      bn.push_back(PD + ".decoder_blocks." + std::to_string(i) + ".cross_attn.in_proj_bias");
    }
    TRY(pack_stacked_rows(c, PD + ".kv_all", wn, bn, 128, 256));
  }
  {  // dense-traffic scene encoder (enc_w.h)
    EncWSrc q; memset(&q, 0, sizeof(q));
    for (int i = 0; i < 4; ++i) {
      const std::string p = "encoder_blocks." + std::to_string(i);
      EncWSrc::L& k = q.l[i];
      k.ln1_g = fptr(c, p + ".norm1.weight"); k.ln1_b = fptr(c, p + ".norm1.bias"); k.ln2_g = fptr(c, p + ".norm2.weight"); k.ln2_b = fptr(c, p + ".norm2.bias");
      k.w_in = fptr(c, p + ".attn.in_proj_weight"); k.b_in = fptr(c, p + ".attn.in_proj_bias");
      k.wo = fptr(c, p + ".attn.out_proj.weight"); k.bo = fptr(c, p + ".attn.out_proj.bias");
      k.w1 = fptr(c, p + ".mlp.fc1.weight"); k.b1 = fptr(c, p + ".mlp.fc1.bias"); k.w2 = fptr(c, p + ".mlp.fc2.weight"); k.b2 = fptr(c, p + ".mlp.fc2.bias");
    }
    q.fn_g = fptr(c, "norm.weight"); q.fn_b = fptr(c, "norm.bias");
    if (!c->err.empty()) return RIFT_ERR_ARG;
    for (int i = 0; i < 4; ++i) {
      const std::string p = PD + ".decoder_blocks." + std::to_string(i) + ".cross_attn";
      q.dkv_w[i] = fptr(c, p + ".in_proj_weight"); q.dkv_b[i] = fptr(c, p + ".in_proj_bias");
    }
    if (!c->err.empty()) return RIFT_ERR_ARG;
    if (!c->encw_img) { HIPCHK(c, hipMalloc((void**)&c->encw_img, (size_t)(4 * ENCW_LAYER_FRAGS + ENCW_TAIL_FRAGS) * 1024)); HIPCHK(c, hipMalloc((void**)&c->encw_par, (size_t)ENCW_NPAR * 4)); }
    encw_pack(q, c->encw_img, c->encw_par, c->stream);
  }
  {  // wave-private Fourier embeddings (fo_w.h): weight streams + parameter blocks of the three embeddings of the fused launch
    const std::string emb[3] = {"pos_emb", "map_encoder.speed_limit_emb", PD + ".r_pos_emb"};
    const int dims[3] = {3, 1, 3};
    for (int i = 0; i < 3; ++i) {
      FoWSrc q; memset(&q, 0, sizeof(q));
      q.D = dims[i];
      for (int d = 0; d < q.D; ++d) {
        const std::string m = emb[i] + ".mlps." + std::to_string(d);
        q.w0[d] = fptr(c, m + ".0.weight"); q.b0[d] = fptr(c, m + ".0.bias"); q.lng[d] = fptr(c, m + ".1.weight"); q.lnb[d] = fptr(c, m + ".1.bias");
        q.w3[d] = fptr(c, m + ".3.weight"); q.b3[d] = fptr(c, m + ".3.bias");
      }
      q.og = fptr(c, emb[i] + ".to_out.0.weight"); q.ob = fptr(c, emb[i] + ".to_out.0.bias");
      q.wo = fptr(c, emb[i] + ".to_out.2.weight"); q.bo = fptr(c, emb[i] + ".to_out.2.bias"); q.freqs = fptr(c, emb[i] + ".freqs.weight");
      if (!c->err.empty()) return RIFT_ERR_ARG;
      if (!c->fow_img[i]) { HIPCHK(c, hipMalloc((void**)&c->fow_img[i], (size_t)7 * 32 * 1024)); HIPCHK(c, hipMalloc((void**)&c->fow_par[i], (size_t)FOW_NPAR * 4)); }
      fow_pack(q, c->fow_img[i], c->fow_par[i], c->stream);
    }
  }
  {  // wave-private PointsEncoder pass B (pe_w.h): W1 | W2 | W3a streams of the two encoders
    const std::string enc[2] = {"map_encoder.polygon_encoder", PD + ".r_encoder"};
    const int cin[2] = {10, 6};
    for (int i = 0; i < 2; ++i) {
      PeWSrc q; q.Cin = cin[i];
      q.w1 = fptr(c, enc[i] + ".first_mlp.0.weight"); q.w2 = fptr(c, enc[i] + ".first_mlp.3.weight"); q.w3 = fptr(c, enc[i] + ".second_mlp.0.weight");
      if (!c->err.empty()) return RIFT_ERR_ARG;
      if (!c->pew_img[i]) HIPCHK(c, hipMalloc((void**)&c->pew_img[i], (size_t)PEW_FRAGS * 1024));
      pew_pack(q, c->pew_img[i], c->stream);
    }
  }
  {  // wave-private decoder kernel (dec_w.h): the layers' weight stream and parameter blocks
    DecWSrc q; memset(&q, 0, sizeof(q));
    for (int i = 0; i < 4; ++i) {
      const std::string p = PD + ".decoder_blocks." + std::to_string(i);
      DecWSrc::L& k = q.l[i];
      for (int n = 0; n < 4; ++n) { k.ln[2 * n] = fptr(c, p + ".norm" + std::to_string(n + 1) + ".weight"); k.ln[2 * n + 1] = fptr(c, p + ".norm" + std::to_string(n + 1) + ".bias"); }
      k.r2r_w = fptr(c, p + ".r2r_attn.in_proj_weight"); k.r2r_b = fptr(c, p + ".r2r_attn.in_proj_bias");
      k.r2ro_w = fptr(c, p + ".r2r_attn.out_proj.weight"); k.r2ro_b = fptr(c, p + ".r2r_attn.out_proj.bias");
      k.m2m_w = fptr(c, p + ".m2m_attn.in_proj_weight"); k.m2m_b = fptr(c, p + ".m2m_attn.in_proj_bias");
      k.m2mo_w = fptr(c, p + ".m2m_attn.out_proj.weight"); k.m2mo_b = fptr(c, p + ".m2m_attn.out_proj.bias");
      k.c_w = fptr(c, p + ".cross_attn.in_proj_weight"); k.c_b = fptr(c, p + ".cross_attn.in_proj_bias");
      k.co_w = fptr(c, p + ".cross_attn.out_proj.weight"); k.co_b = fptr(c, p + ".cross_attn.out_proj.bias");
      k.f1_w = fptr(c, p + ".ffn.0.weight"); k.f1_b = fptr(c, p + ".ffn.0.bias"); k.f2_w = fptr(c, p + ".ffn.3.weight"); k.f2_b = fptr(c, p + ".ffn.3.bias");
    }
    q.m_pos = fptr(c, PD + ".m_pos");
    if (!c->err.empty()) return RIFT_ERR_ARG;
    if (!c->decw_img) { HIPCHK(c, hipMalloc((void**)&c->decw_img, (size_t)4 * DECW_LAYER_FRAGS * 1024)); HIPCHK(c, hipMalloc((void**)&c->decw_par, (size_t)4 * DECW_PAR_LAYER * 4)); }
    decw_pack(q, c->decw_img, c->decw_par, c->stream);
  }
  TRY(pack_cols(c, PD + ".cat_x_proj.q", PD + ".cat_x_proj", 0, 128, true));
  TRY(pack_cols(c, PD + ".cat_x_proj.x", PD + ".cat_x_proj", 128, 128, false));
  const char* hd[3] = {"loc_head", "yaw_head", "vel_head"};
  for (int i = 0; i < 3; ++i) TRY(pack_mlp_layer(c, PD + "." + hd[i]));
  TRY(pack_linear(c, "hidden_proj.0")); TRY(pack_linear(c, "hidden_proj.2"));
  TRY(pack_mlp_layer(c, "ref_free_decoder"));
  // required unpacked tensors: fail early and loudly
  const char* need[] = {"norm.weight", "agent_encoder.type_emb.weight", "map_encoder.type_emb.weight",
                        "planning_decoder.m_emb", "planning_decoder.m_pos", "planning_decoder.pi_head.mlp.0.weight",
                        "planning_decoder.pi_head.mlp.3.weight", "agent_encoder.ego_state_emb.query"};
  for (const char* nme : need) if (!fptr(c, nme)) return RIFT_ERR_ARG;
  HIPCHK(c, hipGetLastError());
  c->loaded = true;
  return RIFT_OK;
}

int rift_forward(RiftCtx* c, const RiftFeatureBatch* B, const RiftOutputs* out, int flags, uint32_t seed, void* stream) {
  if (!c || !B || !out) return RIFT_ERR_ARG;
  c->err.clear();
  if (!c->loaded) { c->err = "rift_forward before rift_model_load"; return RIFT_ERR_STATE; }
  if (B->bs <= 0 || B->A <= 0 || B->Mp <= 0 || B->R <= 0 || B->S < 0 || B->T < 21) { c->err = "bad batch dims"; return RIFT_ERR_ARG; }
  HIPCHK(c, hipSetDevice(c->device));
  c->stream = (hipStream_t)stream;
  // deferred-head forwards cycle through the arenas (the caller guarantees that the head / loss of the forward RIFT_DEFER_SLOTS calls back is
  // over, i.e. that this arena is free again); every other forward runs in arena 0
  c->parity = (flags & RIFT_F_DEFER_HEAD) ? (c->parity + 1) % RIFT_DEFER_SLOTS : 0;
  c->arena = c->arenas[c->parity]; c->arena_cap = c->arena_caps[c->parity];
  // pass 1 (dry): size the activation arena; pass 2: launch.  The sizes depend on the batch dimensions, the flags and the data-parallel
  // descriptor only: a forward like the previous one skips pass 1 (it is half of the call's host time, which bounds a small-batch step)
  long long omask = 0;      // which outputs are wanted
  { const void* const* op = reinterpret_cast<const void* const*>(out); for (size_t i = 0; i < sizeof(RiftOutputs) / sizeof(void*); ++i) omask |= (long long)(op[i] != nullptr) << i; }
  // (everything forward_impl's A_alloc sequence depends on: batch dimensions, flags, the data-parallel descriptor, the wanted outputs and the
  // profiler switch; the environment-driven diagnostic taps are read per call and allocate at most a few KB, which the 1/8 slack of the
  // arena covers -- and the launch pass is checked against the capacity below)
  const long long dkey[11] = {B->bs, B->A, B->Mp, B->R, B->S, B->T, flags, c->dp.on ? 1 : 0, c->dp.on ? c->dp.gbs : 0, omask, c->prof_on ? 1 : 0};
  int rc = RIFT_OK;
  if (c->dry_need == 0 || memcmp(dkey, c->dry_key, sizeof(dkey)) != 0) {
    c->dry = true; c->arena_off = 0; c->dry_need = 0;
    rc = forward_impl(c, B, out, flags, seed);
    c->stream = (hipStream_t)stream;              // (forward_impl forks onto its side stream; an early error return leaves it selected)
    if (rc != RIFT_OK) { c->dry = false; return rc; }
    memcpy(c->dry_key, dkey, sizeof(dkey)); c->dry_need = c->arena_off;
  }
  const size_t need = c->dry_need;
  if (need > c->arena_cap) {
    // (the whole device: with the deferred head and the prepare stream the arena's last readers sit on streams other than the caller's)
    HIPCHK(c, hipDeviceSynchronize());
    if (c->arena) HIPCHK(c, hipFree(c->arena));
    c->arena = nullptr; c->arena_cap = 0;
    const size_t cap = need + (need >> 3);
    HIPCHK(c, hipMalloc((void**)&c->arena, cap));
    c->arena_cap = cap;
  }
  c->arenas[c->parity] = c->arena; c->arena_caps[c->parity] = c->arena_cap;
  c->dry = false; c->arena_off = 0; c->taps.clear();
  // diagnostic: RIFT_POISON_ARENA=<byte> fills the scratch arena before every forward (0xFF = NaN pattern), so that a kernel reading
  // scratch it never wrote shows up as NaN / as run-to-run differences instead of depending on what the memory held before
  // (on the prepare stream if the preparation runs there: every other stream of the forward waits for the preparation)
  if (c->dg.poison_arena >= 0 && c->arena) { HIPCHK(c, hipMemsetAsync(c->arena, c->dg.poison_arena, c->arena_cap, c->prep_set && !c->prof_on ? c->prep_stream : c->stream)); }
  rc = forward_impl(c, B, out, flags, seed);
  c->stream = (hipStream_t)stream;
  if (rc != RIFT_OK) return rc;
  if (!c->err.empty()) return RIFT_ERR_ARG;
  if (c->arena_off > c->arena_cap) {     // the launch pass allocated more than the sizing pass it was matched with: kernels were given scratch beyond the arena
    c->dry_need = 0;
    c->err = "activation arena overrun: the forward allocated " + std::to_string(c->arena_off) + " B of a " + std::to_string(c->arena_cap) + " B arena (stale sizing pass)";
    (void)hipDeviceSynchronize();
    return RIFT_ERR_STATE;
  }
  HIPCHK(c, hipGetLastError());
  return RIFT_OK;
}

int rift_forward_head_back(RiftCtx* c, int back, void* stream) {
  if (!c || back < 0 || back >= RIFT_DEFER_SLOTS) return RIFT_ERR_ARG;
  c->err.clear();
  const int slot = (c->parity - back + RIFT_DEFER_SLOTS) % RIFT_DEFER_SLOTS;
  if (!c->head[slot].valid) { c->err = "rift_forward_head without a RIFT_F_DEFER_HEAD forward"; return RIFT_ERR_STATE; }
  HIPCHK(c, hipSetDevice(c->device));
  const hipStream_t trunk_stream = c->stream;
  c->stream = (hipStream_t)stream; c->dry = false;
  if (c->head[slot].dec_pending) {          // (the forward left its planning decoder to this call: dec_defer_max)
    const DecWP dq = c->head[slot].dec;
    launch_call(c, "dec_w_kernel", [&] { decw_launch(dq, c->stream); });
    c->head[slot].dec_pending = false;
  }
  const int rc = head_impl(c, c->head[slot]);
  c->head[slot].valid = false;
  c->stream = trunk_stream;
  if (rc != RIFT_OK) return rc;
  if (!c->err.empty()) return RIFT_ERR_ARG;
  HIPCHK(c, hipGetLastError());
  return RIFT_OK;
}

int rift_forward_head(RiftCtx* c, void* stream) { return abi::rift_forward_head_back(c, 0, stream); }

int rift_loss_backward(RiftCtx* c, int kind, const RiftLossIn* in, const RiftLossOut* out, void* stream) {
  if (!c || !in || !out || !out->stats || !out->flat_grad_sum) return RIFT_ERR_ARG;
  c->err.clear();
  if (!c->last_qfinal) { c->err = "rift_loss_backward before rift_forward"; return RIFT_ERR_STATE; }
  HIPCHK(c, hipSetDevice(c->device));
  c->stream = (hipStream_t)stream; c->dry = false;
  const int bs = c->last_bs, R = c->last_R, M = 12, G = R * M, rows = bs * G;
  if (G > 64 * 16) { c->err = "group too large"; return RIFT_ERR_ARG; }
  const int nwg = cdiv(rows, RIFT_PI_BWD_ROWS);
  if ((size_t)bs > c->l_cap_bs) {
    if (c->l_S) { (void)hipFree(c->l_S); (void)hipFree(c->l_cnt); }
    HIPCHK(c, hipMalloc((void**)&c->l_S, bs * 8)); HIPCHK(c, hipMalloc((void**)&c->l_cnt, bs * 8)); c->l_cap_bs = bs;
  }
  if ((size_t)rows > c->l_cap_rows) {
    if (c->l_dz) (void)hipFree(c->l_dz);
    HIPCHK(c, hipMalloc((void**)&c->l_dz, (size_t)rows * 4)); c->l_cap_rows = rows;
  }
  if ((size_t)nwg > c->l_cap_wg) {
    if (c->l_partial) (void)hipFree(c->l_partial);
    HIPCHK(c, hipMalloc((void**)&c->l_partial, (size_t)nwg * RIFT_PI_NPARAM * 4)); c->l_cap_wg = nwg;
  }
  LossP p; memset(&p, 0, sizeof(p));
  p.kind = kind; p.bs = bs; p.G = G; p.M = M; p.prob = c->last_prob; p.r_kpm = c->last_rkpm;
  p.old_logits = in->old_group_logits; p.ref_logits = in->ref_group_logits; p.adv64 = in->group_advantage;
  p.valid = in->group_valid_mask; p.action_mode = (const long long*)in->action_mode;
  p.scal_a = kind == RIFT_LOSS_PPO ? in->advantage : in->returns; p.old_log_prob = in->old_log_prob;
  p.clip_eps = in->clip_epsilon; p.lambda_entropy = in->lambda_entropy;
  p.S = c->l_S; p.cnt = c->l_cnt; p.dlogit = c->l_dz; p.argmax_rm = (long long*)out->argmax_rm;
  if ((kind == RIFT_LOSS_RIFT || kind == RIFT_LOSS_GRPO) && (!p.old_logits || !p.adv64 || !p.valid)) { c->err = "missing loss inputs"; return RIFT_ERR_ARG; }
  if (kind == RIFT_LOSS_GRPO && !p.ref_logits) { c->err = "missing ref logits"; return RIFT_ERR_ARG; }
  if (kind == RIFT_LOSS_PPO && (!p.action_mode || !p.scal_a || !p.old_log_prob)) { c->err = "missing ppo inputs"; return RIFT_ERR_ARG; }
  if (kind == RIFT_LOSS_REINFORCE && !p.scal_a) { c->err = "missing returns"; return RIFT_ERR_ARG; }
  if (kind == RIFT_LOSS_SFT && !p.action_mode) { c->err = "missing teacher mode (action_mode)"; return RIFT_ERR_ARG; }
  const dim3 lgrid(cdiv(bs, 4)), lblock(256);
  if (G <= 64 * 2) launch(c, "loss_kernel", loss_kernel<2>, lgrid, lblock, 0, p);
  else if (G <= 64 * 4) launch(c, "loss_kernel", loss_kernel<4>, lgrid, lblock, 0, p);
  else launch(c, "loss_kernel", loss_kernel<16>, lgrid, lblock, 0, p);
  const std::string PH = "planning_decoder.pi_head.mlp.";
  launch(c, "pi_backward_kernel", pi_backward_kernel, dim3(nwg), dim3(256), 0, (const float*)c->last_qfinal, (const float*)c->last_hpi,
         (const float*)c->l_dz, rows, fptr(c, PH + "1.weight"), fptr(c, PH + "1.bias"), fptr(c, PH + "3.weight"), 1e-5f,
         c->l_partial);
  launch(c, "loss_reduce_kernel", loss_reduce_kernel, dim3(cdiv(RIFT_PI_NPARAM, 256)), dim3(256), 0, (const float*)c->l_partial, nwg,
         out->flat_grad_sum, (const double*)c->l_S, (const double*)c->l_cnt, bs, out->stats, out->exchange);
  HIPCHK(c, hipGetLastError());
  return RIFT_OK;
}

int rift_loss_finalize(RiftCtx* c, const RiftLossOut* out, int accumulate, void* stream) {
  if (!c || !out || !out->stats || !out->flat_grad_sum) return RIFT_ERR_ARG;
  c->stream = (hipStream_t)stream; c->dry = false;
  launch(c, "loss_finalize_kernel", loss_finalize_kernel, dim3(cdiv(RIFT_PI_NPARAM, 256)), dim3(256), 0, (const float*)out->flat_grad_sum,
         (const double*)out->stats, out->grad_w1, out->grad_b1, out->grad_ln_w, out->grad_ln_b, out->grad_w2, out->grad_b2,
         out->loss, accumulate, (const double*)out->exchange, out->stats);
  HIPCHK(c, hipGetLastError());
  return RIFT_OK;
}

int rift_other_vehicle_rollout(RiftCtx* c, const double* actions, const double* speed, const double* location, const double* yaw_deg,
                               const double* extent, int N, int T, int near_lane_change, double bbox_inflation_ratio, double* vertices,
                               void* stream) {
  if (!c || N < 0 || T <= 0) return RIFT_ERR_ARG;
  if (N == 0) return RIFT_OK;
  if (!actions || !speed || !location || !yaw_deg || !extent || !vertices) return RIFT_ERR_ARG;
  c->err.clear();
  HIPCHK(c, hipSetDevice(c->device));
  hipLaunchKernelGGL(other_vehicle_rollout_kernel, dim3(cdiv(N, 64)), dim3(64), 0, (hipStream_t)stream, actions, speed, location, yaw_deg, extent,
                     N, T, near_lane_change, bbox_inflation_ratio, vertices);
  HIPCHK(c, hipGetLastError());
  return RIFT_OK;
}

int rift_collision_matrix(RiftCtx* c, const float* center_vertices, int G, int Tc, const double* other_vertices, int N, int Ts,
                          uint8_t* collision, void* stream) {
  if (!c || G <= 0 || Ts <= 0 || Tc < Ts || N < 0 || !center_vertices || !collision || (N > 0 && !other_vertices)) return RIFT_ERR_ARG;
  c->err.clear();
  HIPCHK(c, hipSetDevice(c->device));
  hipLaunchKernelGGL(collision_matrix_kernel, dim3(cdiv((long long)G * Ts, 256)), dim3(256), 0, (hipStream_t)stream, center_vertices, G, Tc,
                     other_vertices, N, Ts, collision);
  HIPCHK(c, hipGetLastError());
  return RIFT_OK;
}

int rift_off_road_matrix(RiftCtx* c, const float* rollout_center, int n_points, const uint8_t* off_road_mask, int H, int W,
                         double origin_x, double origin_y, double heading, double res_x, double res_y, double off_x, double off_y,
                         uint8_t* off_road, void* stream) {
  if (!c || n_points <= 0 || H <= 0 || W <= 0 || !rollout_center || !off_road_mask || !off_road || res_x == 0.0 || res_y == 0.0) return RIFT_ERR_ARG;
  c->err.clear();
  HIPCHK(c, hipSetDevice(c->device));
  hipLaunchKernelGGL(off_road_kernel, dim3(cdiv(n_points, 256)), dim3(256), 0, (hipStream_t)stream, rollout_center, n_points, off_road_mask,
                     H, W, origin_x, origin_y, std::cos(heading), std::sin(heading), res_x, res_y, off_x, off_y, off_road);
  HIPCHK(c, hipGetLastError());
  return RIFT_OK;
}

int rift_sft_teacher_mode(RiftCtx* c, const float* trajectory, const float* teacher_infos, int bs, int R, int M, int T, int frame_rate,
                          int64_t* mode_rm, void* stream) {
  if (!c || !trajectory || !teacher_infos || !mode_rm || bs <= 0 || R <= 0 || M <= 0 || T <= 0 || frame_rate <= 0) return RIFT_ERR_ARG;
  c->err.clear();
  HIPCHK(c, hipSetDevice(c->device));
  hipLaunchKernelGGL(sft_teacher_mode_kernel, dim3(bs), dim3(64), 0, (hipStream_t)stream, trajectory, teacher_infos, bs, R * M, M, T, frame_rate,
                     (long long*)mode_rm);
  HIPCHK(c, hipGetLastError());
  return RIFT_OK;
}

int rift_set_dp(RiftCtx* c, const RiftDp* dp) {
  if (!c) return RIFT_ERR_ARG;
  if (!dp || dp->global_bs <= 0) { c->dp = RiftCtx::Dp(); return RIFT_OK; }
  if (!dp->xchg || dp->scene_offset < 0 || dp->xchg_len <= 0) { c->err = "rift_set_dp: bad descriptor"; return RIFT_ERR_ARG; }
  if (!dp->exchange && !c->comm) { c->err = "rift_set_dp: no exchange callback and no library communicator (rift_comm_init)"; return RIFT_ERR_ARG; }
  c->dp.on = true; c->dp.off = dp->scene_offset; c->dp.gbs = dp->global_bs; c->dp.xchg = dp->xchg; c->dp.len = dp->xchg_len;
  c->dp.fn = dp->exchange; c->dp.user = dp->user;
  return RIFT_OK;
}

int rift_comm_unique_id(RiftCtx* c, void* unique_id_out) {
  if (!c || !unique_id_out) return RIFT_ERR_ARG;
  std::string why;
  RcclApi& r = rccl_api(&why);
  if (!r.ok) { c->err = "rift_comm_unique_id: " + why; return RIFT_ERR_STATE; }
  RcclApi::UniqueId id;
  const int rc = r.GetUniqueId(&id);
  if (rc != 0) { c->err = std::string("ncclGetUniqueId: ") + (r.GetErrorString ? r.GetErrorString(rc) : "failed"); return RIFT_ERR_HIP; }
  memcpy(unique_id_out, &id, sizeof(id));
  return RIFT_OK;
}

int rift_comm_init(RiftCtx* c, const void* unique_id, int rank, int world) {
  if (!c || !unique_id || world < 1 || rank < 0 || rank >= world) return RIFT_ERR_ARG;
  if (c->comm) { c->err = "rift_comm_init: the context has a communicator already (rift_comm_destroy first)"; return RIFT_ERR_STATE; }
  std::string why;
  RcclApi& r = rccl_api(&why);
  if (!r.ok) { c->err = "rift_comm_init: " + why; return RIFT_ERR_STATE; }
  HIPCHK(c, hipSetDevice(c->device));
  RcclApi::UniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  void* comm = nullptr;
  const int rc = r.CommInitRank(&comm, world, id, rank);
  if (rc != 0 || !comm) { c->err = std::string("ncclCommInitRank: ") + (r.GetErrorString ? r.GetErrorString(rc) : "failed"); return RIFT_ERR_HIP; }
  c->comm = comm; c->comm_rank = rank; c->comm_world = world;
  return RIFT_OK;
}

int rift_comm_all_reduce(RiftCtx* c, double* buf, int64_t count, void* stream) {
  if (!c || !buf || count <= 0) return RIFT_ERR_ARG;
  if (!c->comm) { c->err = "rift_comm_all_reduce before rift_comm_init"; return RIFT_ERR_STATE; }
  HIPCHK(c, hipSetDevice(c->device));
  const int rc = comm_sum_f64(c, buf, count, (hipStream_t)stream);
  if (rc != 0) { RcclApi& r = rccl_api(nullptr); c->err = std::string("ncclAllReduce: ") + (r.GetErrorString ? r.GetErrorString(rc) : "failed"); return RIFT_ERR_HIP; }
  return RIFT_OK;
}

int rift_comm_destroy(RiftCtx* c) {
  if (!c) return RIFT_ERR_ARG;
  if (c->comm) {
    if (c->dp.on && !c->dp.fn) c->dp = RiftCtx::Dp();      // (forwards must not reach for a communicator that is gone)
    (void)rccl_api(nullptr).CommDestroy(c->comm);
    c->comm = nullptr; c->comm_world = 1; c->comm_rank = 0;
  }
  return RIFT_OK;
}

int rift_set_prepare_stream(RiftCtx* c, void* prepare_stream) {
  if (!c) return RIFT_ERR_ARG;
  c->prep_stream = (hipStream_t)prepare_stream; c->prep_set = prepare_stream != nullptr;
  return RIFT_OK;
}

int rift_set_side_stream(RiftCtx* c, void* side_stream) {
  if (!c) return RIFT_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->side && c->side_owned) { HIPCHK(c, hipStreamSynchronize(c->side)); (void)hipStreamDestroy(c->side); }
  else if (c->side) HIPCHK(c, hipStreamSynchronize(c->side));
  c->side = (hipStream_t)side_stream; c->side_owned = false;
  return RIFT_OK;
}

int rift_check_finite(RiftCtx* c, void* stream) {
  if (!c) return RIFT_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  int flag = 0;
  HIPCHK(c, hipMemcpyAsync(&flag, c->nonfinite, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIPCHK(c, hipStreamSynchronize((hipStream_t)stream));
  if (!flag) return RIFT_OK;
  HIPCHK(c, hipMemsetAsync(c->nonfinite, 0, sizeof(int), (hipStream_t)stream));
  c->err = (flag & 2) ? "the in-launch ranking of the history encoder's sequences gave up waiting for a predecessor (kernels.h: rank_scene_body): the forward's "
                        "results are invalid; non-finite flag raised"
                      : "non-finite decoder queries (the reference asserts torch.isfinite(q).all(), planning_decoder.py:175)";
  return RIFT_ERR_NONFINITE;
}

int rift_set_param_event(RiftCtx* c, void* event) {
  if (!c) return RIFT_ERR_ARG;
  c->param_event = (hipEvent_t)event;
  return RIFT_OK;
}

int rift_loss_finalize_clip(RiftCtx* c, const RiftLossOut* out, int accumulate, float max_norm, float* total_norm, void* stream) {
  if (!c || !out || !out->stats || !out->flat_grad_sum || !(max_norm > 0.f)) return RIFT_ERR_ARG;
  if (!out->grad_w1 || !out->grad_b1 || !out->grad_ln_w || !out->grad_ln_b || !out->grad_w2 || !out->grad_b2) return RIFT_ERR_ARG;
  c->stream = (hipStream_t)stream; c->dry = false;
  launch(c, "loss_finalize_clip_kernel", loss_finalize_clip_kernel, dim3(1), dim3(1024), 0, (const float*)out->flat_grad_sum,
         (const double*)out->stats, out->grad_w1, out->grad_b1, out->grad_ln_w, out->grad_ln_b, out->grad_w2, out->grad_b2,
         out->loss, accumulate, (const double*)out->exchange, out->stats, max_norm, total_norm);
  HIPCHK(c, hipGetLastError());
  return RIFT_OK;
}

int rift_clip_grad_norm(RiftCtx* c, float* const* grads, const int64_t* numels, int n, float max_norm, float* total_norm, void* stream) {
  if (!c || !grads || !numels || n <= 0 || n > 16) return RIFT_ERR_ARG;
  c->err.clear();
  HIPCHK(c, hipSetDevice(c->device));
  c->stream = (hipStream_t)stream; c->dry = false;
  ClipList L; memset(&L, 0, sizeof(L));
  long long tot = 0;
  for (int i = 0; i < n; ++i) { L.g[i] = grads[i]; L.n[i] = numels[i]; tot += numels[i]; }
  L.count = n;
  const int nb = (int)std::min<long long>(64, (tot + 2047) / 2048);
  if (!c->clip_part) HIPCHK(c, hipMalloc((void**)&c->clip_part, 64 * 8));
  launch(c, "clip_norm_partial_kernel", clip_norm_partial_kernel, dim3(nb), dim3(256), 0, L, c->clip_part);
  launch(c, "clip_scale_kernel", clip_scale_kernel, dim3(nb), dim3(256), 0, L, (const double*)c->clip_part, nb, max_norm, total_norm);
  HIPCHK(c, hipGetLastError());
  return RIFT_OK;
}

int rift_adamw_step(RiftCtx* c, int n, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                    float* const* steps, const int64_t* numels, const double* lr, const double* weight_decay, double step_new,
                    double beta1, double beta2, double eps, void* stream) {
  if (!c || n <= 0 || n > 16 || !params || !grads || !exp_avg || !exp_avg_sq || !steps || !numels || !lr || !weight_decay) return RIFT_ERR_ARG;
  if (!(step_new >= 1.0)) return RIFT_ERR_ARG;
  c->err.clear();
  HIPCHK(c, hipSetDevice(c->device));
  c->stream = (hipStream_t)stream; c->dry = false;
  AdamList L; memset(&L, 0, sizeof(L));
  const double bc1 = 1.0 - std::pow(beta1, step_new), bc2 = 1.0 - std::pow(beta2, step_new);
  long long tot = 0;
  for (int i = 0; i < n; ++i) {
    if (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i] || !steps[i] || numels[i] <= 0) return RIFT_ERR_ARG;
    L.p[i] = params[i]; L.g[i] = grads[i]; L.m[i] = exp_avg[i]; L.v[i] = exp_avg_sq[i]; L.step[i] = steps[i];
    L.n[i] = numels[i]; L.step_size[i] = (float)(lr[i] / bc1); L.decay[i] = (float)(1.0 - lr[i] * weight_decay[i]); L.step_new[i] = (float)step_new;
    tot += numels[i];
  }
  L.count = n; L.beta1 = (float)beta1; L.beta2 = (float)beta2; L.omb1 = (float)(1.0 - beta1); L.omb2 = (float)(1.0 - beta2);
  L.bc2_sqrt = (float)std::sqrt(bc2); L.eps = (float)eps;
  const int nb = (int)std::min<long long>(256, (tot + 255) / 256);
  launch(c, "adamw_kernel", adamw_kernel, dim3(nb), dim3(256), 0, L);
  HIPCHK(c, hipGetLastError());
  return RIFT_OK;
}

int rift_update_tail(RiftCtx* c, const RiftLossOut* out, int accumulate, float max_norm, float* total_norm, int n, float* const* params,
                     const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq, float* const* steps, const double* lr,
                     const double* weight_decay, double step_new, double beta1, double beta2, double eps, void* stream) {
  if (!c || !out || !out->stats || !out->flat_grad_sum || !(max_norm > 0.f) || n != 6) return RIFT_ERR_ARG;
  if (!params || !grads || !exp_avg || !exp_avg_sq || !steps || !lr || !weight_decay || !(step_new >= 1.0)) return RIFT_ERR_ARG;
  float* const gseg[6] = {out->grad_w1, out->grad_b1, out->grad_ln_w, out->grad_ln_b, out->grad_w2, out->grad_b2};
  c->err.clear();
  HIPCHK(c, hipSetDevice(c->device));
  c->stream = (hipStream_t)stream; c->dry = false;
  TailAdam A; memset(&A, 0, sizeof(A));
  const double bc1 = 1.0 - std::pow(beta1, step_new), bc2 = 1.0 - std::pow(beta2, step_new);
  for (int sgi = 0; sgi < 6; ++sgi) {
    int j = -1;
    for (int t = 0; t < 6; ++t) if (gseg[sgi] && grads[t] == gseg[sgi]) j = t;
    if (j < 0 || !params[j] || !exp_avg[j] || !exp_avg_sq[j] || !steps[j]) { c->err = "rift_update_tail: the AdamW list does not hold the six pi_head gradient tensors"; return RIFT_ERR_ARG; }
    A.p[sgi] = params[j]; A.m[sgi] = exp_avg[j]; A.v[sgi] = exp_avg_sq[j]; A.step[sgi] = steps[j];
    A.step_size[sgi] = (float)(lr[j] / bc1); A.decay[sgi] = (float)(1.0 - lr[j] * weight_decay[j]);
  }
  A.step_new = (float)step_new; A.beta2 = (float)beta2; A.omb1 = (float)(1.0 - beta1); A.omb2 = (float)(1.0 - beta2);
  A.bc2_sqrt = (float)std::sqrt(bc2); A.eps = (float)eps;
  launch(c, "update_tail_kernel", update_tail_kernel, dim3(1), dim3(1024), 0, (const float*)out->flat_grad_sum, (const double*)out->stats,
         out->grad_w1, out->grad_b1, out->grad_ln_w, out->grad_ln_b, out->grad_w2, out->grad_b2, out->loss, accumulate,
         (const double*)out->exchange, out->stats, max_norm, total_norm, A);
  HIPCHK(c, hipGetLastError());
  return RIFT_OK;
}

// ---- PPO critic ------------------------------------------------------------------------------------
static int critic_scratch(RiftCtx* c, int n) {
  const size_t need = (size_t)((n + 15) / 16 * 16) * (128 + 4 * 256 + 1 + 2 * 128 + 2);
  if (need > c->cr_cap) {
    if (c->cr_buf) { (void)hipFree(c->cr_buf); (void)hipFree(c->cr_part); }
    HIPCHK(c, hipMalloc((void**)&c->cr_buf, need * 4));
    HIPCHK(c, hipMalloc((void**)&c->cr_part, (size_t)((n + 15) / 16) * 8));
    c->cr_cap = need;
  }
  return RIFT_OK;
}

static CriticW critic_w(const RiftCritic* w) {
  CriticW q;
  q.w0 = w->w0; q.b0 = w->b0; q.w1 = w->w1; q.b1 = w->b1; q.w2 = w->w2; q.b2 = w->b2;
  q.savg = w->state_avg; q.sstd = w->state_std; q.vavg = w->value_avg; q.vstd = w->value_std;
  return q;
}

int rift_critic_forward(RiftCtx* c, const RiftCritic* w, const float* state, int n, float* value, void* stream) {
  if (!c || !w || !state || !value || n < 0) return RIFT_ERR_ARG;
  c->err.clear();
  HIPCHK(c, hipSetDevice(c->device));
  c->stream = (hipStream_t)stream; c->dry = false;
  if (n == 0) return RIFT_OK;
  CriticRowsP p; memset(&p, 0, sizeof(p));
  p.w = critic_w(w); p.state = state; p.n = n; p.value = value;
  launch(c, "critic_rows_kernel", critic_rows_kernel, dim3(cdiv(n, 16)), dim3(256), 0, p);
  HIPCHK(c, hipGetLastError());
  return RIFT_OK;
}

int rift_critic_loss_backward(RiftCtx* c, const RiftCritic* w, const float* state, const float* reward_sum, int n, double* stats,
                              float* flat, void* stream) {
  if (!c || !w || !state || !reward_sum || !stats || !flat || n <= 0) return RIFT_ERR_ARG;
  c->err.clear();
  HIPCHK(c, hipSetDevice(c->device));
  c->stream = (hipStream_t)stream; c->dry = false;
  TRY(critic_scratch(c, n));
  const int np = (n + 15) / 16 * 16, nwg = cdiv(n, 16);
  CriticRowsP p; memset(&p, 0, sizeof(p));
  p.w = critic_w(w); p.state = state; p.target = reward_sum; p.n = n;
  p.sn = c->cr_buf; p.h1 = p.sn + (size_t)np * 128; p.h2 = p.h1 + (size_t)np * 256; p.dh1 = p.h2 + (size_t)np * 256;
  p.dh2 = p.dh1 + (size_t)np * 256; p.dout = p.dh2 + (size_t)np * 256; p.sl1_part = c->cr_part;
  p.gsa = p.dout + np; p.gss = p.gsa + (size_t)np * 128; p.gva = p.gss + (size_t)np * 128; p.gvs = p.gva + np;
  launch(c, "critic_rows_kernel", critic_rows_kernel, dim3(nwg), dim3(256), 0, p);
  // flat = -sum_rows d SmoothL1 / d theta: one thread per parameter, rows summed in order
  launch(c, "critic_outer_sum_kernel", critic_outer_sum_kernel, dim3(cdiv(256 * 128, 256)), dim3(256), 0, (const float*)p.dh1, 256, (const float*)p.sn, 128, n, -1.f, flat);
  launch(c, "critic_col_sum_kernel", critic_col_sum_kernel, dim3(1), dim3(256), 0, (const float*)p.dh1, 256, n, -1.f, flat + RIFT_CRITIC_OFF_B0);
  launch(c, "critic_outer_sum_kernel", critic_outer_sum_kernel, dim3(cdiv(256 * 256, 256)), dim3(256), 0, (const float*)p.dh2, 256, (const float*)p.h1, 256, n, -1.f, flat + RIFT_CRITIC_OFF_W1);
  launch(c, "critic_col_sum_kernel", critic_col_sum_kernel, dim3(1), dim3(256), 0, (const float*)p.dh2, 256, n, -1.f, flat + RIFT_CRITIC_OFF_B1);
  launch(c, "critic_outer_sum_kernel", critic_outer_sum_kernel, dim3(1), dim3(256), 0, (const float*)p.dout, 1, (const float*)p.h2, 256, n, -1.f, flat + RIFT_CRITIC_OFF_W2);
  launch(c, "critic_col_sum_kernel", critic_col_sum_kernel, dim3(1), dim3(64), 0, (const float*)p.dout, 1, n, -1.f, flat + RIFT_CRITIC_OFF_B2);
  launch(c, "critic_col_sum_kernel", critic_col_sum_kernel, dim3(1), dim3(128), 0, (const float*)p.gsa, 128, n, -1.f, flat + RIFT_CRITIC_OFF_SAVG);
  launch(c, "critic_col_sum_kernel", critic_col_sum_kernel, dim3(1), dim3(128), 0, (const float*)p.gss, 128, n, -1.f, flat + RIFT_CRITIC_OFF_SSTD);
  launch(c, "critic_col_sum_kernel", critic_col_sum_kernel, dim3(1), dim3(64), 0, (const float*)p.gva, 1, n, -1.f, flat + RIFT_CRITIC_OFF_VAVG);
  launch(c, "critic_col_sum_kernel", critic_col_sum_kernel, dim3(1), dim3(64), 0, (const float*)p.gvs, 1, n, -1.f, flat + RIFT_CRITIC_OFF_VSTD);
  launch(c, "critic_stats_kernel", critic_stats_kernel, dim3(1), dim3(64), 0, (const double*)c->cr_part, nwg, stats);
  HIPCHK(c, hipGetLastError());
  return RIFT_OK;
}

int rift_critic_finalize(RiftCtx* c, const float* flat, const double* stats, float* g_w0, float* g_b0, float* g_w1, float* g_b1,
                         float* g_w2, float* g_b2, float* g_savg, float* g_sstd, float* g_vavg, float* g_vstd, void* stream) {
  if (!c || !flat || !stats) return RIFT_ERR_ARG;
  c->stream = (hipStream_t)stream; c->dry = false;
  launch(c, "critic_finalize_kernel", critic_finalize_kernel, dim3(cdiv(RIFT_CRITIC_NPARAM, 256)), dim3(256), 0, flat, stats, g_w0, g_b0,
         g_w1, g_b1, g_w2, g_b2, g_savg, g_sstd, g_vavg, g_vstd);
  HIPCHK(c, hipGetLastError());
  return RIFT_OK;
}

int rift_prof_enable(RiftCtx* c, int on) {
  if (!c) return RIFT_ERR_ARG;
  c->prof_on = on != 0;
  c->dry_need = 0;                      // the next forward sizes its arena again
  const char* ev = getenv("RIFT_PROF_SHAPES");
  c->prof_shapes = ev && ev[0] == '1';
  if (on) { c->prof_recs.clear(); c->prof_used = 0; }
  return RIFT_OK;
}

// JSON: {"label": {"count": n, "ms": total_ms, "flops": total_flops}, ...}; synchronises the device.
int rift_prof_report(RiftCtx* c, char* buf, int buflen) {
  if (!c || !buf || buflen <= 0) return RIFT_ERR_ARG;
  HIPCHK(c, hipDeviceSynchronize());
  struct Agg { long n = 0; double ms = 0, fl = 0; };
  std::unordered_map<std::string, Agg> agg;
  for (auto& r : c->prof_recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
    Agg& a = agg[r.label]; a.n++; a.ms += ms; a.fl += r.flops;
  }
  std::string js = "{";
  bool first = true;
  for (auto& kv : agg) {
    char tmp[256];
    snprintf(tmp, sizeof(tmp), "%s\"%s\": {\"count\": %ld, \"ms\": %.6f, \"flops\": %.6e}", first ? "" : ", ", kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.fl);
    js += tmp; first = false;
  }
  js += "}";
  if ((int)js.size() + 1 > buflen) { c->err = "prof buffer too small"; return RIFT_ERR_ARG; }
  memcpy(buf, js.c_str(), js.size() + 1);
  return RIFT_OK;
}

int rift_tap(RiftCtx* c, const char* name, float* dst, int64_t* numel, void* stream) {
  if (!c || !name) return RIFT_ERR_ARG;
  auto it = c->taps.find(name);
  if (it == c->taps.end()) { c->err = std::string("unknown tap ") + name; return RIFT_ERR_ARG; }
  if (numel) *numel = it->second.numel;
  if (dst && it->second.numel > 0)
    HIPCHK(c, hipMemcpyAsync(dst, it->second.p, it->second.numel * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return RIFT_OK;
}

int rift_op_linear(RiftCtx* c, const float* X, int M, int K, const float* W, const float* bias, int N,
                   const float* ln_w, const float* ln_b, int act, int fp32, float* Y, void* stream) {
  if (!c || !X || !W || !Y || K > 512) return RIFT_ERR_ARG;
  c->err.clear();
  HIPCHK(c, hipSetDevice(c->device));
  c->stream = (hipStream_t)stream; c->dry = false;
  const std::string key = "__op_linear__";
  // transient pack (test entry point; not on the hot path)
  TRY(pack(c, key, W, N, N, K, 0, 0, K, 0, 0, 0, bias));
  PW w = c->pw[key];
  GemmP g = mk(X, K, M, w, Y, N);
  if (ln_w) { g.pro = PRO_LN; g.pg = ln_w; g.pb = ln_b; }
  g.act = act;
  gemm(c, g, w, fp32 != 0);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  (void)hipFree(w.bf); (void)hipFree(w.f32);
  c->owned.pop_back(); c->owned.pop_back();
  c->pw.erase(key);
  HIPCHK(c, hipGetLastError());
  return RIFT_OK;
}

// Diagnostic: pack once, launch the GEMM `reps` times back to back, return the average milliseconds per launch.
int rift_op_linear_bench(RiftCtx* c, const float* X, int M, int K, const float* W, const float* bias, int N,
                         const float* ln_w, const float* ln_b, int act, const float* residual, float* Y, int reps,
                         float* ms_out, void* stream) {
  if (!c || !X || !W || !Y || K > 512 || reps <= 0) return RIFT_ERR_ARG;
  c->err.clear();
  HIPCHK(c, hipSetDevice(c->device));
  c->stream = (hipStream_t)stream; c->dry = false;
  const std::string key = "__op_linear_bench__";
  TRY(pack(c, key, W, N, N, K, 0, 0, K, 0, 0, 0, bias));
  PW w = c->pw[key];
  GemmP g = mk(X, K, M, w, Y, N);
  if (ln_w) { g.pro = PRO_LN; g.pg = ln_w; g.pb = ln_b; }
  g.act = act;
  if (residual) { g.residual = residual; g.ldr = N; }
  hipEvent_t e0, e1;
  HIPCHK(c, hipEventCreate(&e0)); HIPCHK(c, hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) gemm(c, g, w, false);
  HIPCHK(c, hipEventRecord(e0, c->stream));
  for (int i = 0; i < reps; ++i) gemm(c, g, w, false);
  HIPCHK(c, hipEventRecord(e1, c->stream));
  HIPCHK(c, hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
  if (ms_out) *ms_out = ms / reps;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(w.bf); (void)hipFree(w.f32);
  c->owned.pop_back(); c->owned.pop_back();
  c->pw.erase(key);
  return RIFT_OK;
}

// An empty buffer is a valid (no-op) input of the three scans below, as it is for the reference's loops.
int rift_gae(RiftCtx* c, const double* rewards, const float* undones, const float* values, const float* next_values,
             const float* unterminated, float gamma, float lambda_, int n, float* advantages, void* stream) {
  if (!c || n < 0) return RIFT_ERR_ARG;
  if (n == 0) return RIFT_OK;
  GaeCoef k{rewards, undones, values, next_values, unterminated, gamma, lambda_};
  hipLaunchKernelGGL((affine_scan_reverse_kernel<GaeCoef, float>), dim3(1), dim3(RIFT_SCAN_THREADS), 0, (hipStream_t)stream, k, n, advantages);
  HIPCHK(c, hipGetLastError());
  return RIFT_OK;
}

int rift_discounted_return(RiftCtx* c, const double* rewards, const float* dones, double gamma, int n, double* returns,
                           void* stream) {
  if (!c || n < 0) return RIFT_ERR_ARG;
  if (n == 0) return RIFT_OK;
  ReturnCoef k{rewards, dones, gamma};
  hipLaunchKernelGGL((affine_scan_reverse_kernel<ReturnCoef, double>), dim3(1), dim3(RIFT_SCAN_THREADS), 0, (hipStream_t)stream, k, n, returns);
  HIPCHK(c, hipGetLastError());
  return RIFT_OK;
}

int rift_normalize_advantage(RiftCtx* c, float* x, int n, void* stream) {
  if (!c || n < 0) return RIFT_ERR_ARG;
  if (n == 0) return RIFT_OK;
  hipLaunchKernelGGL(normalize_unbiased_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, n);
  HIPCHK(c, hipGetLastError());
  return RIFT_OK;
}

int rift_group_advantage(RiftCtx* c, const double* returns, int n_groups, int G, double* advantage, void* stream) {
  if (!c || n_groups <= 0 || G <= 0) return RIFT_ERR_ARG;
  hipLaunchKernelGGL(group_zscore_kernel, dim3(cdiv(n_groups, 4)), dim3(256), 0, (hipStream_t)stream, returns, n_groups, G, advantage);
  HIPCHK(c, hipGetLastError());
  return RIFT_OK;
}

int rift_rollout_return(RiftCtx* c, const float* delta_dis, const float* delta_angle, const float* speed, const float* acc,
                        const float* ang_vel, const float* ang_acc, const uint8_t* collision, int collision_ld,
                        const uint8_t* off_road, int off_road_ld, int G, int Ts, double gamma, double* returns, void* stream) {
  if (!c || G <= 0 || Ts <= 0) return RIFT_ERR_ARG;
  hipLaunchKernelGGL(rollout_return_kernel, dim3(cdiv(G, 4)), dim3(256), 0, (hipStream_t)stream, delta_dis, delta_angle, speed,
                     acc, ang_vel, ang_acc, collision, collision_ld, off_road, off_road_ld, G, Ts, gamma, returns, Ts);
  HIPCHK(c, hipGetLastError());
  return RIFT_OK;
}

int rift_ref_line_info(RiftCtx* c, const float* traj, int G, int Tfull, int Ts, int M, const float* ref_pos, const float* ref_ang,
                       const int32_t* ref_len, int Pmax, float* delta_dis, float* delta_angle, int32_t* closest_idx, void* stream) {
  if (!c || !traj || !ref_pos || !ref_ang || !ref_len || G <= 0 || Ts <= 0 || M <= 0) return RIFT_ERR_ARG;
  hipLaunchKernelGGL(ref_line_info_kernel, dim3(cdiv((long long)G * Ts, 128)), dim3(128), 0, (hipStream_t)stream, traj, G, Tfull, Ts, M,
                     ref_pos, ref_ang, (const int*)ref_len, Pmax, delta_dis, delta_angle, (int*)closest_idx);
  HIPCHK(c, hipGetLastError());
  return RIFT_OK;
}

int rift_rollout(RiftCtx* c, const RiftRolloutIO* io, void* stream) {
  if (!c || !io || !io->trajectories || !io->center_state || io->G <= 0 || io->Tfull < 40 || io->G_per_group <= 0) return RIFT_ERR_ARG;
  RolloutP p; memset(&p, 0, sizeof(p));
  p.traj = io->trajectories; p.G = io->G; p.Tfull = io->Tfull; p.Gper = io->G_per_group; p.state = io->center_state;
  p.turn_buf = io->turn_buf; p.turn_ptr = (int*)io->turn_ptr; p.turn_len = (int*)io->turn_len;
  p.speed_buf = io->speed_buf; p.speed_ptr = (int*)io->speed_ptr; p.speed_len = (int*)io->speed_len;
  p.center = io->center; p.angle = io->angle; p.speed = io->speed; p.acc = io->acc; p.ang_vel = io->ang_vel; p.ang_acc = io->ang_acc;
  p.vertices = io->vertices; p.closest_index = (int*)io->closest_index; p.aim_idx = (int*)io->aim_idx;
  if ((size_t)io->G * RIFT_RO_LEN * 4 > c->ro_cap) {          // the raw speed history between the two kernels (grown behind a device-wide wait)
    if (c->ro_raw) { HIPCHK(c, hipDeviceSynchronize()); (void)hipFree(c->ro_raw); c->ro_raw = nullptr; c->ro_cap = 0; }
    const size_t cap = std::max((size_t)io->G * 2, (size_t)256) * RIFT_RO_LEN * 4;
    HIPCHK(c, hipMalloc((void**)&c->ro_raw, cap));
    c->ro_cap = cap;
  }
  p.raw_speed = c->ro_raw;
  if (!p.center || !p.angle || !p.speed || !p.acc || !p.ang_vel || !p.ang_acc || !p.vertices || !p.closest_index || !p.aim_idx) return RIFT_ERR_ARG;
  if (c->ro8) hipLaunchKernelGGL(rollout8_kernel, dim3(cdiv(io->G, 8)), dim3(64), (size_t)RIFT_RO_LDS_BYTES, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(rollout_kernel, dim3(cdiv(io->G, 64)), dim3(64), (size_t)RIFT_RO_LDS_BYTES, (hipStream_t)stream, p);
  hipLaunchKernelGGL(rollout_kinematics_kernel, dim3(cdiv(io->G * RIFT_RO_LEN, 256)), dim3(256), 0, (hipStream_t)stream, p);
  HIPCHK(c, hipGetLastError());
  return RIFT_OK;
}

// One rollout tick's group advantages (rift_hip.h).  Everything but the candidate rollouts is independent between the CBVs: one launch per
// stage serves a chunk of up to RIFT_TICK_CHUNK of them (tick_multi.h) -- reference-line deviations and neighbour forecasts first, then the
// rollouts CBV by CBV in list order (the shared PID state), then kinematics, flags, returns and z-scores: K + 7 launches instead of 9 K.
int rift_group_advantage_tick(RiftCtx* c, const float* trajectory, int Rb, int Tfull, const RiftTickCBV* cbvs, int K,
                              float* turn_buf, int32_t* turn_ptr, int32_t* turn_len, float* speed_buf, int32_t* speed_ptr, int32_t* speed_len,
                              double gamma, double* advantage, void* stream) {
  if (!c || !trajectory || !cbvs || K <= 0 || Rb <= 0 || Tfull < 80 || !advantage || !turn_buf || !turn_ptr || !turn_len || !speed_buf || !speed_ptr || !speed_len) return RIFT_ERR_ARG;
  c->err.clear();
  HIPCHK(c, hipSetDevice(c->device));
  constexpr int M = 12, Ts = 40, TR = RIFT_RO_LEN;
  int Gmax = 0, Nmax = 0;
  for (int k = 0; k < K; ++k) {
    const RiftTickCBV& v = cbvs[k];
    if (v.R <= 0 || v.R > Rb || v.batch_index < 0 || v.Pmax <= 0 || v.n_actors < 0 || !v.center_state || !v.ref_pos || !v.ref_angle || !v.ref_len ||
        (v.n_actors > 0 && !v.actors) || (v.off_road_mask && (v.H <= 0 || v.W <= 0))) { c->err = "rift_group_advantage_tick: bad entry"; return RIFT_ERR_ARG; }
    Gmax = std::max(Gmax, v.R * M); Nmax = std::max(Nmax, v.n_actors);
  }
  // per-CBV scratch (bytes, 256-aligned pieces), one set per CBV of a chunk
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  const size_t G = (size_t)Gmax;
  const size_t o_dd = take(G * Ts * 4), o_da = take(G * Ts * 4), o_ci = take(G * Ts * 4);
  const size_t o_center = take(G * TR * 2 * 4), o_angle = take(G * TR * 4), o_speed = take(G * TR * 4), o_acc = take(G * TR * 4), o_av = take(G * TR * 4),
               o_aa = take(G * TR * 4), o_vert = take(G * TR * 8 * 4), o_cl = take(G * 79 * 4), o_aim = take(G * 79 * 4), o_raw = take(G * TR * 4);
  const size_t o_ov = take((size_t)std::max(Nmax, 1) * Ts * 8 * 8), o_col = take(G * Ts), o_offr = take(G * TR), o_ret = take(G * 8);
  const size_t per = off, need = per * (size_t)std::min(K, RIFT_TICK_CHUNK);
  if (need > c->tick_cap) {
    if (c->tick_scratch) { HIPCHK(c, hipDeviceSynchronize()); (void)hipFree(c->tick_scratch); c->tick_scratch = nullptr; c->tick_cap = 0; }
    HIPCHK(c, hipMalloc((void**)&c->tick_scratch, need * 2));
    c->tick_cap = need * 2;
  }
  const hipStream_t st = (hipStream_t)stream;
  for (int k0 = 0; k0 < K; k0 += RIFT_TICK_CHUNK) {
    TickArr a; memset(&a, 0, sizeof(a));
    a.K = std::min(RIFT_TICK_CHUNK, K - k0); a.gamma = gamma;
    int gmax = 0, nmax = 0;
    for (int j = 0; j < a.K; ++j) {
      const RiftTickCBV& v = cbvs[k0 + j];
      char* S = c->tick_scratch + per * (size_t)j;
      TickK& d = a.d[j];
      d.traj = trajectory + (size_t)v.batch_index * Rb * M * Tfull * 6; d.G = v.R * M; d.Tfull = Tfull; d.Pmax = v.Pmax; d.N = v.n_actors; d.H = v.H; d.W = v.W;
      d.ref_pos = v.ref_pos; d.ref_ang = v.ref_angle; d.ref_len = (const int*)v.ref_len;
      d.dd = (float*)(S + o_dd); d.da = (float*)(S + o_da); d.ci = (int*)(S + o_ci);
      d.actors = v.actors; d.ov = (double*)(S + o_ov);
      RolloutP& p = d.ro;
      p.traj = d.traj; p.G = d.G; p.Tfull = Tfull; p.Gper = d.G; p.state = v.center_state;
      p.turn_buf = turn_buf; p.turn_ptr = (int*)turn_ptr; p.turn_len = (int*)turn_len; p.speed_buf = speed_buf; p.speed_ptr = (int*)speed_ptr; p.speed_len = (int*)speed_len;
      p.center = (float*)(S + o_center); p.angle = (float*)(S + o_angle); p.speed = (float*)(S + o_speed); p.acc = (float*)(S + o_acc);
      p.ang_vel = (float*)(S + o_av); p.ang_acc = (float*)(S + o_aa); p.vertices = (float*)(S + o_vert);
      p.closest_index = (int*)(S + o_cl); p.aim_idx = (int*)(S + o_aim); p.raw_speed = (float*)(S + o_raw);
      d.mask = v.off_road_mask; d.ox = v.pose[0]; d.oy = v.pose[1]; d.ch = std::cos(v.pose[2]); d.sh = std::sin(v.pose[2]);
      d.col = (uint8_t*)(S + o_col); d.offr = (uint8_t*)(S + o_offr); d.ret = (double*)(S + o_ret);
      d.adv = advantage + (size_t)(k0 + j) * Rb * M;
      gmax = std::max(gmax, d.G); nmax = std::max(nmax, d.N);
    }
    const dim3 gy((unsigned)1, (unsigned)a.K);
    hipLaunchKernelGGL(tick_multi_kernel<0>, dim3(cdiv(gmax * Ts, 128), a.K), dim3(128), 0, st, a);
    if (nmax > 0) hipLaunchKernelGGL(tick_multi_kernel<1>, dim3(cdiv(nmax, 64), a.K), dim3(64), 0, st, a);
    for (int j = 0; j < a.K; ++j) {
      if (c->ro8) hipLaunchKernelGGL(rollout8_kernel, dim3(cdiv(a.d[j].G, 8)), dim3(64), (size_t)RIFT_RO_LDS_BYTES, st, a.d[j].ro);
      else hipLaunchKernelGGL(rollout_kernel, dim3(cdiv(a.d[j].G, 64)), dim3(64), (size_t)RIFT_RO_LDS_BYTES, st, a.d[j].ro);
    }
    hipLaunchKernelGGL(tick_multi_kernel<2>, dim3(cdiv(gmax * TR, 256), a.K), dim3(256), 0, st, a);
    hipLaunchKernelGGL(tick_multi_kernel<3>, dim3(cdiv(gmax * Ts, 256), a.K), dim3(256), 0, st, a);
    hipLaunchKernelGGL(tick_multi_kernel<4>, dim3(cdiv(gmax * TR, 256), a.K), dim3(256), 0, st, a);
    hipLaunchKernelGGL(tick_multi_kernel<5>, dim3(cdiv(gmax, 4), a.K), dim3(256), 0, st, a);
    hipLaunchKernelGGL(tick_multi_kernel<6>, dim3(1, a.K), dim3(256), 0, st, a);
    (void)gy;
    HIPCHK(c, hipGetLastError());
  }
  return RIFT_OK;
}

int rift_collate(RiftCtx* c, const RiftReplayArena* ar, const int32_t* scene_idx, int bs, int R_out,
                 const RiftFeatureBatch* ob, float* out_old_logits, float* out_ref_logits, double* out_advantage,
                 uint8_t* out_valid_mask, void* stream) {
  if (!c || !ar || !scene_idx || !ob || bs <= 0 || R_out <= 0 || R_out > ar->Rcap) return RIFT_ERR_ARG;
  // the arena's rows are stored padded to its capacities (ar->A, Mp, Rcap, S); the batch is padded to the dimensions of `ob`, which
  // may be smaller (a host arena that grows by doubling keeps capacities above the largest stored scene): every ragged dimension
  // leads its per-scene block, so the crop is a prefix copy
  if (ob->A > ar->A || ob->Mp > ar->Mp || ob->S > ar->S || ob->A <= 0 || ob->T != ar->T) return RIFT_ERR_ARG;
  const RiftFeatureBatch& s = ar->scenes;
  const int A = ar->A, Mp = ar->Mp, Rc = ar->Rcap, S = ar->S, T = ar->T;
  const int oA = ob->A, oMp = ob->Mp, oS = ob->S;
  CollateP p; memset(&p, 0, sizeof(p));
  int k = 0;
  auto add = [&](const void* src, void* dst, long long src_bytes, long long dst_bytes) {
    if (!src || !dst || dst_bytes <= 0) return;
    p.src[k] = (const unsigned char*)src; p.dst[k] = (unsigned char*)dst; p.src_bytes[k] = (int)src_bytes; p.dst_bytes[k] = (int)dst_bytes; ++k;
  };
#define FIX(field, n_src, n_dst, per) add(s.field, (void*)ob->field, (long long)(n_src) * (per), (long long)(n_dst) * (per))
  FIX(agent_position, A, oA, T * 8); FIX(agent_heading, A, oA, T * 4); FIX(agent_velocity, A, oA, T * 8);
  FIX(agent_shape, A, oA, T * 8); FIX(agent_category, A, oA, 1); FIX(agent_valid_mask, A, oA, T);
  FIX(map_point_position, Mp, oMp, 3 * 20 * 8); FIX(map_point_vector, Mp, oMp, 3 * 20 * 8);
  FIX(map_point_orientation, Mp, oMp, 3 * 20 * 4); FIX(map_polygon_center, Mp, oMp, 12);
  FIX(map_polygon_type, Mp, oMp, 1); FIX(map_polygon_on_route, Mp, oMp, 1); FIX(map_polygon_tl_status, Mp, oMp, 1);
  FIX(map_polygon_has_speed_limit, Mp, oMp, 1); FIX(map_polygon_speed_limit, Mp, oMp, 4); FIX(map_valid_mask, Mp, oMp, 20);
  FIX(static_position, S, oS, 8); FIX(static_heading, S, oS, 4); FIX(static_shape, S, oS, 8);
  FIX(static_category, S, oS, 1); FIX(static_valid_mask, S, oS, 1); FIX(current_state, 1, 1, ar->cs_ld * 4);
#undef FIX
#define RAG(srcp, dstp, per_line) add((srcp), (void*)(dstp), (long long)Rc * (per_line), (long long)R_out * (per_line))
  RAG(s.ref_position, ob->ref_position, 120 * 8); RAG(s.ref_vector, ob->ref_vector, 120 * 8);
  RAG(s.ref_orientation, ob->ref_orientation, 120 * 4); RAG(s.ref_valid_mask, ob->ref_valid_mask, 120);
  RAG(ar->old_group_logits, out_old_logits, 12 * 4); RAG(ar->ref_group_logits, out_ref_logits, 12 * 4);
  RAG(ar->group_advantage, out_advantage, 12 * 8); RAG(ar->group_valid_mask, out_valid_mask, 12);
#undef RAG
  p.nt = k;
  hipLaunchKernelGGL(collate_kernel, dim3(bs, k), dim3(256), 0, (hipStream_t)stream, p, scene_idx);
  HIPCHK(c, hipGetLastError());
  return RIFT_OK;
}

}}  // namespace RIFT_NS::abi

#include "abi_table.h"
#if RIFT_OP_F16
#define RIFT_VT_SYM rift_vtable_fp16
#else
#define RIFT_VT_SYM rift_vtable_bf16
#endif
#if !defined(__HIP_DEVICE_COMPILE__)      // host data: the device pass of this file must not see pointers to host functions
extern "C" const RiftVTable RIFT_VT_SYM = {
#define RIFT_FN(name) &RIFT_NS::abi::name,
#include "abi_list.h"
#undef RIFT_FN
};
#endif
