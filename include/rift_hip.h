/*
 * librift_hip.so -- C-ABI of the MI355X-native RIFT train_cbv policy-update hot path.
 *
 * The reference (CurryChen77/RIFT) is pure Python over PyTorch and has no FFI; these
 * entry points are what a maintainer binds (ctypes) to replace
 * the device work behind the reference's own Python interfaces.  Each entry cites the
 * reference interface it replaces (paths relative to the upstream checkout).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless marked "host"; tensors are caller-owned,
 *     dense row-major with the reference's shapes; bool tensors are 1 byte per element
 *     (torch.bool), categories are int8.
 *   - all work is enqueued on the hipStream_t passed by the caller (void*; NULL = the
 *     default stream); no hidden synchronisation unless stated.
 *   - return value: 0 = ok, <0 = error; message via rift_last_error().
 *   - one context per (process, device); not thread-safe.
 */
#ifndef RIFT_HIP_H
#define RIFT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct RiftCtx RiftCtx;

enum { RIFT_OK = 0, RIFT_ERR_ARG = -1, RIFT_ERR_HIP = -2, RIFT_ERR_STATE = -3, RIFT_ERR_NONFINITE = -4 };

/* forward flags */
enum {
  RIFT_F_TRAIN      = 1,   /* Lightning .train(): dropout / DropPath / state-dropout on, BatchNorm batch statistics */
  RIFT_F_NEED_TRAJ  = 2,   /* also produce trajectory / prediction / ref_free_trajectory (unused by the RLFT losses) */
  RIFT_F_FP32       = 4,   /* exact-fp32 MFMA (v_mfma_f32_16x16x4_f32), layer by layer, instead of the fused 16-bit-operand kernels */
  RIFT_F_NO_DROP    = 8,   /* with TRAIN: every drop probability 0 (BatchNorm batch statistics only) */
  RIFT_F_NO_BN_UPDATE = 16,/* with TRAIN: do not update BatchNorm running statistics */
  RIFT_F_DEFER_HEAD = 32   /* stop in front of the policy head: everything up to the decoder output reads frozen weights only, so the caller
                              may run rift_forward_head + rift_loss_backward + its optimizer step on ANOTHER stream while the next
                              rift_forward (the trunk of the next minibatch) is already under way.  Deferred forwards cycle through
                              RIFT_DEFER_SLOTS activation arenas; before the RIFT_DEFER_SLOTS-th forward after this one the caller must
                              know the deferred head and rift_loss_backward of this one to be finished (an event wait).
                              Below a chip-filling batch (a measured table of sizes up to 192 scenes; RIFT_DEC_DEFER=<n> in the
                              environment: bs <= n, 0 = never) the planning decoder -- frozen weights as well -- or its second half is left
                              to rift_forward_head(_back) too: a step is then the latency of token assembly -> encoder -> decoder layers
                              0 - 1 on `stream` beside decoder layers 2 - 3 -> head -> loss -> update on the other one */
};
/* (four: the front of step k + 1 -- gather, preparation, history and map encoders -- can then run beside step k without waiting for the
 *  head / loss / update of step k - 1: with two arenas that wait closes a cycle of two steps' length through tail -> front -> encoder ->
 *  decoder -> tail, which is what bounds a small-batch step.) */
#define RIFT_DEFER_SLOTS 4

/* loss kinds */
enum { RIFT_LOSS_RIFT = 0, RIFT_LOSS_GRPO = 1, RIFT_LOSS_PPO = 2, RIFT_LOSS_REINFORCE = 3,
       RIFT_LOSS_SFT = 4 /* teacher cross entropy, sft_trainer.py:123-184: in->action_mode[b][1] = teacher mode (rift_sft_teacher_mode) */ };

/* A named view onto one tensor of PlanningModel.state_dict()
 * (rift/cbv/planning/pluto/model/pluto_model.py:22-120; names as in SURVEY.md Appendix B). */
typedef struct RiftTensorDesc {
  const char* name;     /* host string */
  void* data;           /* device pointer (fp32; num_batches_tracked int64) */
  int64_t numel;
  int32_t ndim;
  int64_t shape[4];
} RiftTensorDesc;

/* The collated feature batch that PlanningModel.forward(data) consumes
 * (pluto_model.py:122-138; produced by PlutoFeature.collate, pluto_feature.py:25-96;
 * schema in SURVEY.md Appendix A). */
typedef struct RiftFeatureBatch {
  int32_t bs, A, Mp, R, S;           /* scenes, padded agents / polygons / ref lines / static objects */
  int32_t T;                          /* time extent of the agent tensors (>= 21; first 21 steps are used) */
  const float*   agent_position;      /* (bs,A,T,2) */
  const float*   agent_heading;       /* (bs,A,T)   */
  const float*   agent_velocity;      /* (bs,A,T,2) */
  const float*   agent_shape;         /* (bs,A,T,2) */
  const int8_t*  agent_category;      /* (bs,A)     */
  const uint8_t* agent_valid_mask;    /* (bs,A,T)   */
  const float*   map_point_position;  /* (bs,Mp,3,20,2) */
  const float*   map_point_vector;    /* (bs,Mp,3,20,2) */
  const float*   map_point_orientation; /* (bs,Mp,3,20) */
  const float*   map_polygon_center;  /* (bs,Mp,3) */
  const int8_t*  map_polygon_type;    /* (bs,Mp) */
  const uint8_t* map_polygon_on_route; /* (bs,Mp) */
  const int8_t*  map_polygon_tl_status; /* (bs,Mp) */
  const uint8_t* map_polygon_has_speed_limit; /* (bs,Mp) */
  const float*   map_polygon_speed_limit; /* (bs,Mp) */
  const uint8_t* map_valid_mask;      /* (bs,Mp,20) */
  const float*   ref_position;        /* (bs,R,120,2) */
  const float*   ref_vector;          /* (bs,R,120,2) */
  const float*   ref_orientation;     /* (bs,R,120) */
  const uint8_t* ref_valid_mask;      /* (bs,R,120) */
  const float*   static_position;     /* (bs,S,2)  (may be NULL when S == 0) */
  const float*   static_heading;      /* (bs,S) */
  const float*   static_shape;        /* (bs,S,2) */
  const int8_t*  static_category;     /* (bs,S) */
  const uint8_t* static_valid_mask;   /* (bs,S) */
  const float*   current_state;       /* (bs,cs_ld) first 6 used */
  int32_t cs_ld;
} RiftFeatureBatch;

/* Outputs of PlanningModel.forward (pluto_model.py:167-223).  NULL pointers are skipped. */
typedef struct RiftOutputs {
  float* probability;          /* (bs,R,12) logits, -1e6 on padded reference lines */
  float* hidden;               /* (bs,128) */
  float* trajectory;           /* (bs,R,12,80,6)  [NEED_TRAJ] */
  float* prediction;           /* (bs,A-1,80,6)   [NEED_TRAJ] */
  float* ref_free_trajectory;  /* (bs,80,4)       [NEED_TRAJ] */
} RiftOutputs;

/* Inputs of the RLFT objectives:
 *   RIFT      get_rift_loss      fine_tuner/rlft/rift_pluto/rift_trainer.py:140-182
 *   GRPO      get_grpo_loss      fine_tuner/rlft/grpo_pluto/grpo_trainer.py:140-194
 *   PPO       get_ppo_loss       fine_tuner/rlft/ppo_pluto/ppo_trainer.py:161-183 (actor part)
 *   REINFORCE get_reinforce_loss fine_tuner/rlft/reinforce_pluto/reinforce_trainer.py:125-170 */
typedef struct RiftLossIn {
  const float*   old_group_logits;   /* (bs,R,12) f32   RIFT, GRPO */
  const float*   ref_group_logits;   /* (bs,R,12) f32   GRPO */
  const double*  group_advantage;    /* (bs,R,12) f64   RIFT, GRPO */
  const uint8_t* group_valid_mask;   /* (bs,R,12) bool  RIFT, GRPO */
  const int64_t* action_mode;        /* (bs,2) int64    PPO */
  const float*   advantage;          /* (bs)            PPO */
  const float*   old_log_prob;       /* (bs)            PPO */
  const float*   returns;            /* (bs)            REINFORCE */
  float clip_epsilon;                /* PPO, default 0.2 */
  float lambda_entropy;              /* PPO, default 0.01 */
} RiftLossIn;

/* Gradients of planning_decoder.pi_head (the trainable set of rift_training.yaml:26-27) are
 * written into caller-owned buffers (the torch .grad tensors), so clip_grad_norm_ and
 * torch.optim.AdamW run unchanged (rift_trainer.py:279-362). */
typedef struct RiftLossOut {
  double* loss;            /* device scalar, f64: -(sum objective)/(count) */
  double* stats;           /* device [2] f64: (sum objective, count) -- all-reduce these for DP */
  float*  flat_grad_sum;   /* device [16897] unnormalised d(sum objective)/d(pi_head params): all-reduce for DP */
  float*  grad_w1;  float* grad_b1;      /* mlp.0.weight (128,128), mlp.0.bias (128) */
  float*  grad_ln_w; float* grad_ln_b;   /* mlp.1.weight (128), mlp.1.bias (128) */
  float*  grad_w2;  float* grad_b2;      /* mlp.3.weight (1,128), mlp.3.bias (1) */
  int64_t* argmax_rm;      /* (bs,2) int64 chosen (r,m) -- REINFORCE, bit-exact integer output */
  double* exchange;        /* optional device [16899] f64: flat_grad_sum (as f64) followed by stats -- ONE buffer to all-reduce for DP;
                              when set, rift_loss_finalize reads the sums and the stats from here */
} RiftLossOut;

/* The 16-bit MFMA operand format of a context's fused kernels (RIFT_F_FP32 forwards do not use it).  The reference trains at
 * `precision: 32` (fine_tuner/rlft/config/lightning/custom_lightning.yaml:28); BASELINE's benchmark is quoted on bf16 MFMA.
 *   BF16: 8 significand bits  -- the benchmarked default
 *   FP16: 11 significand bits -- same instruction rate and bytes; holds the 1e-4 loss / advantage tolerance on small batches too */
enum { RIFT_OPERANDS_BF16 = 0, RIFT_OPERANDS_FP16 = 1 };

int  rift_ctx_create(int device, RiftCtx** ctx);     /* operands: bf16, or fp16 when the environment has RIFT_OPERANDS=fp16 */
int  rift_ctx_create_ex(int device, int operand_format, RiftCtx** ctx);
int  rift_ctx_operand_format(RiftCtx* ctx);          /* RIFT_OPERANDS_* of the context */
void rift_ctx_destroy(RiftCtx* ctx);
const char* rift_last_error(RiftCtx* ctx);

/* Bind the model parameters (views onto the torch state_dict storage) and build the packed
 * bf16 / fp32 weight images.  Replaces PlanningModel.load_state_dict for the device side
 * (pluto.py:130-141).  Call again after the frozen trunk changes; pi_head.* is read live. */
int rift_model_load(RiftCtx* ctx, const RiftTensorDesc* params, int n, void* stream);

/* PlanningModel.forward (pluto_model.py:122-225). */
int rift_forward(RiftCtx* ctx, const RiftFeatureBatch* batch, const RiftOutputs* out, int flags,
                 uint32_t seed, void* stream);

/* The policy head of the last RIFT_F_DEFER_HEAD forward (cat_x_proj -> pi_head -> masked logits -> `probability`, and the trajectory heads
 * when they were requested), launched on `stream`; the caller orders it behind that forward (event).  rift_loss_backward then refers to it. */
int rift_forward_head(RiftCtx* ctx, void* stream);
/* The same for the deferred forward `back` rift_forward calls before the latest one (0 = the latest; back < RIFT_DEFER_SLOTS): a host that
 * issues forward k + 1 ahead of the head / loss of step k -- the data-parallel update loop does, so that the BatchNorm all-reduces of step k + 1
 * are not queued behind the loss all-reduce of step k on the communicator -- names the forward it means. */
int rift_forward_head_back(RiftCtx* ctx, int back, void* stream);

/* ---- data parallelism (absent in the reference: one device, custom_lightning.yaml:26,43) ----
 * A minibatch of `global_bs` scenes is split contiguously over the ranks; this rank's rift_forward receives scenes
 * [scene_offset, scene_offset + batch->bs).  Two things in PlanningModel.forward couple the scenes of a minibatch, and both are made
 * to see the GLOBAL minibatch so that the sharded update equals the single-process one:
 *   - train-mode BatchNorm1d batch statistics of the two PointsEncoders (layers/embedding.py:260,266): per-channel (sum, sum of
 *     squares, count) are all-reduced before normalisation (two exchanges per forward on the fused path), running statistics are
 *     updated from the global sums, identically on every rank;
 *   - the r2r attention's `tgt_key_padding_mask.repeat(M, 1)` (modules/planning_decoder.py:56-60): row b*12+m is masked with the
 *     padding row of scene (b*12+m) % bs, of the global minibatch: the padding rows are gathered with the first exchange (in an eval
 *     forward: in an exchange of their own).
 * The library owns no communicator: at each exchange point it fills xchg[offset, offset+count) (device f64, caller-owned) on `stream`
 * and calls `exchange`, which must enqueue an in-place SUM all-reduce of that range over the ranks on `stream` (RCCL through
 * torch.distributed in the host layer; any transport works) and return 0.  Every rank must call rift_forward with the same R and flags.
 * xchg_len >= global_bs * R + 1026.  The loss exchange (RiftLossOut.exchange) stays with the host between rift_loss_backward and
 * rift_loss_finalize.  dp == NULL (or global_bs <= 0) switches data parallelism off.  exchange == NULL is accepted behind rift_comm_init
 * (below): the library's own communicator carries the exchanges. */
typedef int (*RiftExchangeFn)(void* user, int64_t offset, int64_t count, void* stream);
typedef struct RiftDp {
  int32_t scene_offset, global_bs;
  double* xchg; int64_t xchg_len;
  RiftExchangeFn exchange; void* user;
} RiftDp;
int rift_set_dp(RiftCtx* ctx, const RiftDp* dp);

/* A library-owned communicator (`rift_comm_init(ctx, ncclUniqueId, rank, world)`) for hosts that have none of their own: RCCL
 * (librccl.so, opened with dlopen the first time one of these is called -- an RCCL already in the process, e.g. torch's, is reused) over xGMI.
 * rift_comm_unique_id fills 128 bytes (an ncclUniqueId) on ONE rank; the launcher hands them to every rank out of band; rift_comm_init joins
 * the communicator (collective: every rank calls it).  With a communicator in place rift_set_dp accepts `exchange == NULL` -- the forward's
 * exchanges are then ncclAllReduce(sum, f64) on the forward's own streams -- and rift_comm_all_reduce does the same for the caller's loss
 * exchange buffer (RiftLossOut.exchange) between rift_loss_backward and rift_loss_finalize.  A host WITH a communicator (the Python layer:
 * torch.distributed's) keeps passing its callback; the two routes produce the same sums. */
int rift_comm_unique_id(RiftCtx* ctx, void* unique_id_out /*host, 128 bytes*/);
int rift_comm_init(RiftCtx* ctx, const void* unique_id /*host, 128 bytes*/, int rank, int world);
int rift_comm_all_reduce(RiftCtx* ctx, double* buf /*device f64, in place*/, int64_t count, void* stream);
int rift_comm_destroy(RiftCtx* ctx);

/* Input prefetch.  Everything a forward does before its first encoder kernel depends on the batch alone: the caller's gather of the batch
 * (rift_collate) and the forward's own input-only preparation (difference features, masks, positions: one launch).  With a prepare stream
 * set, rift_forward launches that preparation on `prepare_stream` -- behind what the caller queued there, i.e. the gather -- and makes its
 * own streams wait for it: a host that runs ahead of the device (the update loop does, by about a step) gets the next step's inputs built
 * beside the current step's kernels instead of between two steps (12 + 12 us of a 0.7 ms step at 256 scenes, 7 + 6 of 0.39 at 32).  The
 * caller orders `prepare_stream` behind whatever last read the batch buffers and the activation arena of this forward (with
 * RIFT_F_DEFER_HEAD: the head / loss of the forward RIFT_DEFER_SLOTS calls back).  NULL (the default) keeps the preparation on the forward's stream; the
 * per-kernel profile ignores the setting.  Results do not depend on it.  (With rift_set_dp the forward's slots of the exchange buffer are
 * filled on the map encoder's stream, behind the previous forward's exchanges.) */
int rift_set_prepare_stream(RiftCtx* ctx, void* prepare_stream);

/* The forward's second stream (the map-encoder chain beside the history chain) is the library's own by default.  A host that manages
 * the placement of its streams on the device's hardware queues -- streams created in a process share a few queues in creation order, and
 * two streams of one update pipeline on one queue serialise what they are there to overlap (measured 0.65 -> 0.72 - 0.80 ms per step with two or
 * three unrelated streams created first) -- hands the library a stream of its choice instead (RLFTTrainer probes for three streams that run
 * concurrently with the caller's and with each other).  NULL returns to the library's own.  The stream must outlive the context's forwards. */
int rift_set_side_stream(RiftCtx* ctx, void* side_stream);

/* The reference asserts torch.isfinite(q).all() on the decoder queries after every decoder layer (planning_decoder.py:175).  Here the
 * policy-head kernels of rift_forward raise a device flag when the decoder output holds a NaN / Inf (either propagates through the
 * residual stream to the last layer); rift_forward itself stays asynchronous.  rift_check_finite is the sync point: it waits for
 * `stream`, returns RIFT_ERR_NONFINITE (and clears the flag) if any forward since the last check raised it, RIFT_OK otherwise.
 * The host layer calls it wherever it reads results back (once per epoch in the update loop, once per get_action).
 * What it guarantees, precisely (weaker than an assert per layer): the test is an integer exponent compare on the LAST decoder output, so
 * it sees every NaN / Inf that is in the residual stream of the encoder or decoder at some point (a residual add keeps it to the end).
 * The wave-private kernels are built without IEEE NaN handling (-fno-honor-nans -mno-amdgpu-ieee: v_max_f32 returns the non-NaN
 * operand), so a NaN born INSIDE a branch ahead of a ReLU / max -- Inf - Inf in an FFN hidden layer, a PointsEncoder max-pool --
 * can be replaced by a finite value there and never reach the stream; an Inf survives either way.  The reference's assert shares
 * the blind spot for the PointsEncoder (it looks at the decoder queries only) but not for the decoder FFN. */
int rift_check_finite(RiftCtx* ctx, void* stream);

/* SFTTrainer.generate_target_label's teacher side (fine_tuner/sft/sft_trainer.py:186-199): per scene, the (r, m) index of the candidate
 * whose PID target speed (mean spacing of the trajectory sub-sampled every `frame_rate` frames, in the teacher's local frame;
 * sft/utils.py:10-32, pluto/controller/pid_controller.py:108-125) is closest to teacher_infos[b][0].  trajectory: the (bs,R,M,T,6) output
 * of rift_forward; teacher_infos: (bs,5) f32 = target speed, origin x, y, heading, speed; mode_rm: (bs,2) int64.  Only column 1 (the
 * mode) enters the label: RIFT_LOSS_SFT combines it with the policy's own best reference line. */
int rift_sft_teacher_mode(RiftCtx* ctx, const float* trajectory, const float* teacher_infos, int bs, int R, int M, int T, int frame_rate,
                          int64_t* mode_rm, void* stream);

/* Optional: `event` (a hipEvent_t, NULL to clear) marks the point after which the trainable parameters (planning_decoder.pi_head.*,
 * read live by rift_forward) are up to date.  rift_forward then waits for it on ITS stream right before the first kernel that reads
 * pi_head, so a host may run the previous step's exchange + clip + optimizer on a second stream while the frozen trunk of the next
 * step already executes (only pi_head is trainable: rift_trainer.py:78-90).  The event is re-read at every rift_forward. */
int rift_set_param_event(RiftCtx* ctx, void* event);

/* Objective + analytic backward into pi_head, using the activations of the last rift_forward.
 * Two phases so that a data-parallel host can all-reduce (flat_grad_sum, stats) in between:
 *   rift_loss_backward : fills stats, flat_grad_sum (and argmax_rm)
 *   rift_loss_finalize : loss = -S/count, grads = -flat/count into the .grad buffers        */
int rift_loss_backward(RiftCtx* ctx, int kind, const RiftLossIn* in, const RiftLossOut* out, void* stream);
int rift_loss_finalize(RiftCtx* ctx, const RiftLossOut* out, int accumulate, void* stream);
/* rift_loss_finalize followed by clip_grad_norm_(the six pi_head .grad tensors, max_norm) in one launch -- the tail of
 * LightningTrainer.training_step when pi_head is the only trainable module (rift_trainer.py:78-90; gradient_clip_val 0.5,
 * custom_lightning.yaml:40-41).  total_norm: device float, may be NULL.  Not for PPO (the critic's gradients join the norm). */
int rift_loss_finalize_clip(RiftCtx* ctx, const RiftLossOut* out, int accumulate, float max_norm, float* total_norm, void* stream);

/* Per-launch HIP-event profiling of the forward/loss kernels on the caller's stream (bench roofline leg).
 * rift_prof_report synchronises and writes a JSON object {label: {count, ms, flops}} into buf. */
int rift_prof_enable(RiftCtx* ctx, int on);
int rift_prof_report(RiftCtx* ctx, char* buf /*host*/, int buflen);

/* Debug / parity taps: copy a named intermediate of the last forward to `dst` (device, fp32).
 * *numel receives the element count; dst may be NULL to query the size only. */
int rift_tap(RiftCtx* ctx, const char* name, float* dst, int64_t* numel, void* stream);

/* Single GEMM through the same kernel the forward uses (parity tests of the MFMA path):
 * Y[M,N] = act(LN?(X)[M,K] . W[N,K]^T + bias).  W/bias fp32 device pointers. */
int rift_op_linear(RiftCtx* ctx, const float* X, int M, int K, const float* W, const float* bias, int N,
                   const float* ln_w, const float* ln_b, int act, int fp32, float* Y, void* stream);

/* Diagnostic: the same GEMM launched `reps` times back to back; *ms_out = average milliseconds per launch. */
int rift_op_linear_bench(RiftCtx* ctx, const float* X, int M, int K, const float* W, const float* bias, int N,
                         const float* ln_w, const float* ln_b, int act, const float* residual, float* Y, int reps,
                         float* ms_out /*host*/, void* stream);

/* get_advantages_GAE (fine_tuner/rlft/ppo_pluto/ppo_datamodule.py:22-37): reverse scan over n fp32 entries. */
int rift_gae(RiftCtx* ctx, const double* rewards /*f64, as np.stack of python floats*/, const float* undones,
             const float* values, const float* next_values, const float* unterminated, float gamma, float lambda_,
             int n, float* advantages, void* stream);

/* compute_return (fine_tuner/rlft/reinforce_pluto/reinforce_datamodule.py:19-38). */
int rift_discounted_return(RiftCtx* ctx, const double* rewards /*f64*/, const float* dones /*0/1 as f32*/,
                           double gamma, int n, double* returns /*f64*/, void* stream);

/* Buffer-wide advantage normalisation (ppo_datamodule.py:166): (x - mean) / (std_unbiased + 1e-5), in place. */
int rift_normalize_advantage(RiftCtx* ctx, float* x, int n, void* stream);

/* Group z-score of GRPO returns (traj_eval/traj_evaluator.py:467-470): per group of G fp64 returns,
 * adv = (ret - mean) / (std_ddof0 + 1e-5). */
int rift_group_advantage(RiftCtx* ctx, const double* returns, int n_groups, int G, double* advantage, void* stream);

/* TrajEvaluator.get_other_vehicle_rollout (traj_eval/traj_evaluator.py:160-239): the nearby actors' footprints over T future frames under
 * their last control -- KinematicBicycleModel.forecast_other_vehicles (rift/ego/pdm_lite/kinematic_bicycle_model.py:33-62), the speed- and
 * horizon-dependent extent inflation (GlobalConfig, rift/ego/pdm_lite/config.py:186-199) times bbox_inflation_ratio, corners FL RL RR FR in
 * the right-handed global frame.  Inputs are what the reference reads off the CARLA actors, as device fp64 arrays: actions (N,3) = steer,
 * throttle, brake; speed (N) = |velocity|; location (N,3) and yaw_deg (N) in CARLA's left-handed frame; extent (N,2) = bounding-box half
 * extents (x, y).  vertices: (N,T,4,2) f64 -- the `other_vertices` of rift_collision_matrix. */
int rift_other_vehicle_rollout(RiftCtx* ctx, const double* actions, const double* speed, const double* location, const double* yaw_deg,
                               const double* extent, int N, int T, int near_lane_change, double bbox_inflation_ratio, double* vertices,
                               void* stream);

/* TrajEvaluator.get_collision_matrix (traj_eval/traj_evaluator.py:241-275).  The reference queries an STRtree of the other vehicles'
 * footprints with the candidate's footprint and NO predicate, i.e. an envelope test: collision[g][j] = 1 iff the axis-aligned bounding
 * box of center_vertices[g][j] intersects (touching included) that of other_vertices[n][j] for some n.  center_vertices: (G,Tc,4,2) f32
 * (rift_rollout's `vertices`), other_vertices: (N,Ts,4,2) f64 (get_other_vehicle_rollout, :160-239; N may be 0), collision: (G,Ts) u8. */
int rift_collision_matrix(RiftCtx* ctx, const float* center_vertices, int G, int Tc, const double* other_vertices, int N, int Ts,
                          uint8_t* collision, void* stream);

/* TrajEvaluator.get_off_road_matrix, the lookup (traj_evaluator.py:299-318): pixel = round(((p - origin) . R(heading)) / resolution_hw
 * + offset) in fp64 with round-half-even; off_road = inside the (H,W) raster and off_road_mask[py][px] == 1.  The mask itself (1 =
 * not drivable; cv2.fillPoly of the HD-map drivable polygons, :284-297) is the caller's.  rollout_center: (n_points,2) f32. */
int rift_off_road_matrix(RiftCtx* ctx, const float* rollout_center, int n_points, const uint8_t* off_road_mask, int H, int W,
                         double origin_x, double origin_y, double heading, double res_x, double res_y, double off_x, double off_y,
                         uint8_t* off_road, void* stream);

/* Discounted dense-reward return of candidate rollouts (traj_evaluator.py:333-370 with
 * gym_carla/reward/reward_model.py:34-50): inputs (G,Ts) f32, flags (G,*) bool with row strides. */
int rift_rollout_return(RiftCtx* ctx, const float* delta_dis, const float* delta_angle, const float* speed,
                        const float* acc, const float* ang_vel, const float* ang_acc, const uint8_t* collision,
                        int collision_ld, const uint8_t* off_road, int off_road_ld, int G, int Ts, double gamma,
                        double* returns, void* stream);

/* Reference-line deviation of every candidate (TrajEvaluator.get_ref_line_info, traj_eval/traj_evaluator.py:372-420):
 * trajectories (G = R*M, Tfull, 6) -> first Ts frames; ragged reference lines padded to Pmax with lengths ref_len (R).
 * closest_idx (G,Ts) int32 is the bit-exact integer output (argmin over the line's points). */
int rift_ref_line_info(RiftCtx* ctx, const float* trajectories, int G, int Tfull, int Ts, int M, const float* ref_pos /*(R,Pmax,2)*/,
                       const float* ref_angle /*(R,Pmax)*/, const int32_t* ref_len /*(R)*/, int Pmax, float* delta_dis /*(G,Ts)*/,
                       float* delta_angle /*(G,Ts)*/, int32_t* closest_idx /*(G,Ts)*/, void* stream);

/* Candidate closed-loop rollout (TrajEvaluator.get_center_rollout -> TrackPropagate.propagate,
 * traj_eval/traj_evaluator.py:115-158, traj_eval/track_propogate.py:599-780): global transform of the candidate
 * path, 79 PID + kinematic-bicycle steps, Savitzky-Golay kinematics, box corners.  The two PID ring buffers
 * (BatchPIDTorch, :318-400) are persistent caller-owned state, as in the reference (never reset between calls). */
typedef struct RiftRolloutIO {
  const float* trajectories;   /* (G, Tfull, 6) raw model trajectories (x, y, cos, sin, vx, vy); Tfull >= 40 */
  int32_t G, Tfull, G_per_group;
  const float* center_state;   /* (G / G_per_group, 6): x, y, heading, speed, width, length of the centre vehicle */
  float* turn_buf;  int32_t* turn_ptr;  int32_t* turn_len;    /* (G,20) f32, (G) i32, (G) i32 : turn PID state, in/out */
  float* speed_buf; int32_t* speed_ptr; int32_t* speed_len;   /* speed PID state, in/out */
  float* center;    /* (G,80,2) */
  float* angle; float* speed; float* acc; float* ang_vel; float* ang_acc;   /* (G,80) each */
  float* vertices;  /* (G,80,4,2) FL, RL, RR, FR */
  int32_t* closest_index;      /* (G,79) closest reference-path index after each step (bit-exact integer output) */
  int32_t* aim_idx;            /* (G,79) PID aim waypoint index of each step (bit-exact integer output) */
} RiftRolloutIO;
int rift_rollout(RiftCtx* ctx, const RiftRolloutIO* io, void* stream);

/* The group-relative advantage of every CBV of one rollout tick in ONE call (RIFTPluto / GRPOPluto.get_action in train mode,
 * rift_pluto.py:113-135 -> TrajEvaluator.get_grpo_advantage, traj_evaluator.py:422-475): per CBV, in list order (the PID state is shared and
 * never reset, as in the reference), rift_ref_line_info -> rift_rollout -> rift_other_vehicle_rollout -> rift_collision_matrix ->
 * rift_off_road_matrix -> rift_rollout_return -> rift_group_advantage on the library's own scratch, i.e. exactly the launches of the seven
 * calls above without a host round trip between them (the Python tick was ~12 C-ABI calls and ~0.55 ms of host time per CBV).
 * Every pointer inside RiftTickCBV is a DEVICE pointer (the caller stages a tick's readings in one upload). */
typedef struct RiftTickCBV {
  int32_t batch_index;         /* row of `trajectory` (the tick's collated batch) */
  int32_t R;                   /* this CBV's valid reference lines: candidates (r, m), r < R, of that row (the valid lines are a prefix) */
  int32_t Pmax;                /* padded points per line of ref_pos / ref_angle */
  int32_t n_actors;            /* nearby actors (0: none -- no collision flags) */
  const float* center_state;   /* (6) x, y, heading, speed, width, length (rear axle / footprint) */
  const float* ref_pos;        /* (R, Pmax, 2) valid points of each valid line, zero padded */
  const float* ref_angle;      /* (R, Pmax) */
  const int32_t* ref_len;      /* (R) */
  const double* actors;        /* [actions (N,3) | speed (N) | location (N,3) | yaw_deg (N) | extent (N,2)] as blocks of one array; NULL with n_actors 0 */
  const uint8_t* off_road_mask;/* (H, W), 1 = not drivable; NULL: no off-road flags */
  int32_t H, W;
  double pose[3];              /* origin x, y and heading of the raster (footprint centre) */
} RiftTickCBV;
/* trajectory: (bs, Rb, 12, Tfull, 6) raw model output of the tick's forward; pid: the six BatchPIDTorch arrays of RiftRolloutIO with at least
 * 12 * max R rows; advantage: (K, Rb, 12) f64, rows r >= R of a CBV are left untouched.  `cbvs` is a HOST array of K entries. */
int rift_group_advantage_tick(RiftCtx* ctx, const float* trajectory, int Rb, int Tfull, const RiftTickCBV* cbvs, int K,
                              float* turn_buf, int32_t* turn_ptr, int32_t* turn_len, float* speed_buf, int32_t* speed_ptr, int32_t* speed_len,
                              double gamma, double* advantage, void* stream);

/* Device-side collation (PlutoFeature.collate, pluto_feature.py:83-94 + RIFTCollate,
 * rift_datamodule.py:33-49): gather `bs` scenes by index from a replay arena whose tensors
 * are stored padded to (A, Mp, Rcap, S) per scene, writing a batch padded to R = batch max. */
typedef struct RiftReplayArena {
  int32_t n_scenes, A, Mp, Rcap, S, T, cs_ld;
  RiftFeatureBatch scenes;            /* same pointers, leading dim n_scenes, R = Rcap */
  const int32_t* r_count;             /* (n_scenes) valid reference lines per scene */
  const float*   old_group_logits;    /* (n_scenes,Rcap,12) */
  const float*   ref_group_logits;    /* (n_scenes,Rcap,12) or NULL */
  const double*  group_advantage;     /* (n_scenes,Rcap,12) */
  const uint8_t* group_valid_mask;    /* (n_scenes,Rcap,12) */
} RiftReplayArena;

int rift_collate(RiftCtx* ctx, const RiftReplayArena* arena, const int32_t* scene_idx /*device (bs)*/, int bs,
                 int R_out, const RiftFeatureBatch* out_batch /*caller-allocated*/, float* out_old_logits,
                 float* out_ref_logits, double* out_advantage, uint8_t* out_valid_mask, void* stream);

/* Gradient-norm clipping over caller-owned .grad tensors (Lightning's gradient_clip_val = 0.5, custom_lightning.yaml:40-41 ->
 * torch.nn.utils.clip_grad_norm_(params, max_norm), norm_type 2): in place, total norm written to total_norm (device, may be NULL). */
int rift_clip_grad_norm(RiftCtx* ctx, float* const* grads /*host array of device pointers*/, const int64_t* numels /*host*/,
                        int n_tensors /*<= 16*/, float max_norm, float* total_norm, void* stream);

/* torch.optim.AdamW's update (amsgrad False, maximize False; configure_optimizers, rift_trainer.py:279-362) for up to 16 tensors of
 * any parameter groups in one launch, in place on the caller's parameters and on torch's optimizer state (exp_avg, exp_avg_sq and the
 * float32 device step counter of every tensor, as created by AdamW(fused=True)).  lr / weight_decay: per tensor (host arrays);
 * step_new: the step count of THIS update (written to the counters).  Host arrays of device pointers to fp32 tensors; the scalars
 * are doubles because torch forms 1 - beta, the bias corrections and lr / (1 - beta1^t) in double before rounding to fp32. */
int rift_adamw_step(RiftCtx* ctx, int n_tensors /*<= 16*/, float* const* params, const float* const* grads, float* const* exp_avg,
                    float* const* exp_avg_sq, float* const* steps, const int64_t* numels /*host*/, const double* lr /*host*/,
                    const double* weight_decay /*host*/, double step_new, double beta1, double beta2, double eps, void* stream);

/* The step's last three launches in ONE (round 5): rift_loss_finalize_clip (loss = -S / count, gradients = -sums / count into the six pi_head .grad
 * tensors, gradient-norm clip) and rift_adamw_step on exactly those six tensors, in a single single-workgroup launch -- a thread updates the
 * parameters whose gradients it has just formed, so nothing but the clip coefficient crosses threads.  Same arithmetic, bit for bit, as the two
 * calls it replaces (tests/test_gpu_update.py).  The AdamW arrays are those of rift_adamw_step and must list the six tensors whose gradients
 * `out` names (any order, n_tensors == 6): the update of the reference's `trainable_layers: [planning_decoder.pi_head]` configuration
 * (rift_training.yaml:26-27); other trainable sets (PPO's critic) use the separate calls. */
int rift_update_tail(RiftCtx* ctx, const RiftLossOut* out, int accumulate, float max_norm, float* total_norm, int n_tensors /*== 6*/,
                     float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq, float* const* steps,
                     const double* lr /*host*/, const double* weight_decay /*host*/, double step_new, double beta1, double beta2, double eps,
                     void* stream);

/* ---- PPO critic (CriticPPO, rift/gym_carla/utils/net.py:420-431 with CriticBase :355-371; built with dims [256, 256],
 * state_dim 128 by PPOPlutoModel, ppo_pluto.py:37 and planning/config/ppo_pluto.yaml:43-45).  All pointers are device fp32
 * views onto the caller's parameters: net.{0,2,4}.{weight,bias}, state_avg / state_std (128), value_avg / value_std (1). */
typedef struct RiftCritic {
  const float *w0, *b0, *w1, *b1, *w2, *b2;
  const float *state_avg, *state_std, *value_avg, *value_std;
} RiftCritic;
#define RIFT_CRITIC_NPARAM_C 99331   /* w0 256x128 | b0 256 | w1 256x256 | b1 256 | w2 256 | b2 1 | state_avg 128 | state_std 128 | value_avg 1 | value_std 1 */

/* value[i] = value_net(state[i]) -- ppo_datamodule.py:137,149 (buffer sweeps) and ppo_trainer.py:175 */
int rift_critic_forward(RiftCtx* ctx, const RiftCritic* w, const float* state /*(n,128)*/, int n, float* value /*(n)*/, void* stream);

/* The value-loss half of get_ppo_loss (ppo_trainer.py:175-176,183): SmoothL1(value_net(state), reward_sum), mean reduction.
 * Call after rift_loss_backward(kind = PPO) on the same stats: stats[0] -= sum_i SmoothL1_i (so that loss = -stats[0]/stats[1]
 * is value_loss + actor_loss) and flat_grad_sum[RIFT_CRITIC_NPARAM_C] = -sum_i d SmoothL1_i / d theta (all-reduce-able sums). */
int rift_critic_loss_backward(RiftCtx* ctx, const RiftCritic* w, const float* state /*(n,128)*/, const float* reward_sum /*(n)*/,
                              int n, double* stats /*[2]*/, float* flat_grad_sum, void* stream);

/* grads = -flat_grad_sum / stats[1] into the ten caller-owned .grad tensors (NULL = skip).  The four normalisation constants are
 * included because the reference's freeze_parameters (ppo_trainer.py:84-96) makes every parameter of value_net trainable. */
int rift_critic_finalize(RiftCtx* ctx, const float* flat_grad_sum, const double* stats, float* g_w0, float* g_b0, float* g_w1,
                         float* g_b1, float* g_w2, float* g_b2, float* g_state_avg, float* g_state_std, float* g_value_avg,
                         float* g_value_std, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RIFT_HIP_H */
