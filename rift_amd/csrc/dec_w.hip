// Wave-private planning decoder kernel (see dec_w.h).  Its own translation unit: built with `-fno-honor-nans -mno-amdgpu-ieee` (rift_amd/build.py).
#include "common.h"
#include "dec_w.h"
#include "wp_stream.h"
#include <type_traits>

namespace RIFT_NS {

// Weight image: [layer][group][fragment f = ks * 8 + nt][lane][8]: element = W[out = nt*16 + lane&15][in = chan(ks, lane>>4, j)] with the
// K permutation of nat_l0w.h (an n-tile PAIR of one GEMM's C/D output is the next GEMM's k-step).  Softmax scales are folded into q.
__global__ void pack_decw_kernel(DecWSrc s, unsigned short* __restrict__ img, float* __restrict__ par) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const float SC = 0.17677669529663687f * 1.4426950408889634f;   // head_dim^-0.5 (head_dim = 32) x log2 e: the softmaxes run on v_exp_f32 = 2^x
  if (e < 4 * DECW_LAYER_FRAGS * 512) {
    const int j = e & 7, lane = (e >> 3) & 63, fr = e >> 9, li = fr / DECW_LAYER_FRAGS, g = (fr % DECW_LAYER_FRAGS) >> 5, f = fr & 31;
    const int ks = f >> 3, nt = f & 7, l15 = lane & 15, l4 = lane >> 4;
    const int ch = l0w_chan(l4, j, 2 * ks), o = nt * 16 + l15;
    const DecWSrc::L& L = s.l[li];
    float v;
    auto lg = [&](int n) { return RIFT_LN_FOLD ? L.ln[2 * n][ch] : 1.0f; };       // gamma of norm n + 1 folded into the GEMM it feeds (opfmt.h: RIFT_LN_FOLD)
    if (g < 3) v = L.r2r_w[(g * 128 + o) * 128 + ch] * (g == 0 ? SC : 1.0f) * lg(0);
    else if (g == 3) v = L.r2ro_w[o * 128 + ch];
    else if (g < 7) v = L.m2m_w[((g - 4) * 128 + o) * 128 + ch] * (g == 4 ? SC : 1.0f) * lg(1);
    else if (g == 7) v = L.m2mo_w[o * 128 + ch];
    else if (g == 8) v = L.c_w[o * 128 + ch] * SC * lg(2);
    else if (g == 9) v = L.co_w[o * 128 + ch];
    else {
      const int hc = (g - 10) >> 1;
      v = ((g - 10) & 1) ? L.f2_w[o * 512 + hc * 128 + ch] : L.f1_w[(hc * 128 + o) * 128 + ch] * lg(3);
    }
    img[e] = f2h(v);
  }
  if (e < 4 * DECW_PAR_LAYER) {
    const int li = e / DECW_PAR_LAYER, o = e % DECW_PAR_LAYER;
    const DecWSrc::L& L = s.l[li];
    float v = 0.f;
    // (RIFT_LN_FOLD) beta of norm n + 1 through the rows of W into the bias: b + W beta
    auto wb = [&](const float* Wm, int row, int n) { float a = 0.f; if (RIFT_LN_FOLD) for (int k = 0; k < 128; ++k) a += Wm[(size_t)row * 128 + k] * L.ln[2 * n + 1][k]; return a; };
    if (o < DECW_E_N) {
      if (o < 256) v = L.ln[o >> 7][o & 127];
      else if (o < 640) v = (L.r2r_b[o - 256] + wb(L.r2r_w, o - 256, 0)) * (o - 256 < 128 ? SC : 1.0f);
      else if (o < 768) v = L.r2ro_b[o - 640];
      else if (o < 1024) v = L.ln[2 + ((o - 768) >> 7)][o & 127];
      else if (o < DECW_E_BM2MV) {
        const int m = (o - 1024) / DECW_PBS, c = (o - 1024) % DECW_PBS;
        if (c < 256) {
          float acc = 0.f;
          for (int k = 0; k < 128; ++k) acc += s.m_pos[m * 128 + k] * L.m2m_w[c * 128 + k];
          v = (acc + L.m2m_b[c] + wb(L.m2m_w, c, 1)) * (c < 128 ? SC : 1.0f);
        }
      } else if (o < DECW_E_BM2MO) v = L.m2m_b[256 + o - DECW_E_BM2MV] + wb(L.m2m_w, 256 + o - DECW_E_BM2MV, 1);
      else if (o < DECW_E_BM2MO + 128) v = L.m2mo_b[o - DECW_E_BM2MO];
    } else {
      const int q = o - DECW_E_N;
      if (q < 256) v = L.ln[4 + (q >> 7)][q & 127];
      else if (q < 384) v = (L.c_b[q - 256] + wb(L.c_w, q - 256, 2)) * SC;
      else if (q < 512) v = L.co_b[q - 384];
      else if (q < 768) v = L.ln[6 + ((q - 512) >> 7)][q & 127];
      else if (q < 1280) v = L.f1_b[q - 768] + wb(L.f1_w, q - 768, 3);
      else if (q < 1408) v = L.f2_b[q - 1280];
    }
    par[e] = v;
  }
}

// Dropout decisions of this kernel: four per-lane 24-bit linear congruential streams, one FULL-RATE instruction per TWO decisions
// (v_mad_u32_u24: x <- x[23:0] * A + C_i; the 32-bit result is a word of two 16-bit uniforms: bits 31..16 -- the middle of the 42-bit
// product, where every state bit has mixed in -- and bits 15..0, which by themselves are the full-period 16-bit generator x[15:0] * A + C_i).
// History: the counter hash of common.h costs three quarter-rate 32-bit multiplies per two decisions; round 2 drew two decisions from one
// xorshift32 step (six instructions); rounds 3 / 4 one decision per LCG step (upper half only).  Round 5 uses both halves and applies the
// decision where the data already is a PAIR of 16-bit operand words (softmax weights, the FFN hidden layer: 256 of a lane's 384 decisions
// per layer): d = sat_i16((tau - 1) - s) per half (v_pk_sub_i16 clamp; s = the half as a signed integer, tau = p * 65536 - 32768) is
// negative exactly where s >= tau, v_pk_ashrrev_i16 turns that into 0xffff / 0, one v_and_b32 applies both -- 2 instructions per decision
// where the fp32 form (step, compare, select, multiply) took 3.5; the 1/(1-p) factor rides on 1/sum of the softmax and on the FFN's output.
// The residual-branch sites stay fp32 (compare + select on each half of the word: 3 per decision).  What the lower half is worth
// (tools/checks/lcg_halves.py, over the full period, p = 0.1): single rates 0.09999, pairs of one step's two decisions and of successive
// decisions within 0.3 % of p^2, triples within 11 % of p^3, the count of drops among 12 successive draws within 0.3 % absolute of the
// binomial -- first- and second-order statistics of independent Bernoulli(p) decisions; tests/test_gpu_dropstats.py checks rates and
// scaling in the kernel.  The mask is a deterministic function of (seed, stream, scene, lane, draw order).  A = 214013 (a = 1 mod 4: full
// period 2^24 with an odd increment, and A mod 2^16 = 17405 = 1 mod 4 likewise for the lower half); the four states of a lane use four
// different odd increments, so their sequences are different affine images of the cycle, and every lane starts its four at hashed positions.
struct DecWRng { uint32_t x[4]; };
__device__ __forceinline__ uint32_t decw_step(uint32_t x, uint32_t c) {
  return (x & 0xffffffu) * 214013u + c;          // both factors below 2^24: hipcc selects v_mad_u32_u24 and drops the mask (the ISA test of tools/checks counts them)
}
typedef short decw_s16x2 __attribute__((ext_vector_type(2)));
#if RIFT_ATTN_K16
typedef h16x4 DecwVf;            // V^T of a 16-key tile: a K = 16 operand (opfmt.h: RIFT_ATTN_K16)
#else
typedef h16x8 DecwVf;
#endif
// a word of two uniforms -> 0xffff where the half is KEPT (tm1 = the pair (tau - 1, tau - 1))
__device__ __forceinline__ uint32_t decw_keep2(uint32_t w, uint32_t tm1) {
  const decw_s16x2 d = __builtin_elementwise_sub_sat(__builtin_bit_cast(decw_s16x2, tm1), __builtin_bit_cast(decw_s16x2, w));
  return __builtin_bit_cast(uint32_t, d >> (decw_s16x2){15, 15});
}
// ... -> the sign bit of the half set where it is DROPPED (tau2 = the pair (tau, tau)), everything else clear
__device__ __forceinline__ uint32_t decw_drop_sign2(uint32_t w, uint32_t tau2) {
  const decw_s16x2 d = __builtin_elementwise_sub_sat(__builtin_bit_cast(decw_s16x2, w), __builtin_bit_cast(decw_s16x2, tau2));
  return __builtin_bit_cast(uint32_t, d) & 0x80008000u;
}
// ReLU of a pair of operand words: a negative bf16 / fp16 word is a negative 16-bit integer (v_pk_max_i16); relu(round(x)) == round(relu(x))
__device__ __forceinline__ uint32_t decw_relu2(uint32_t w) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(decw_s16x2, w), (decw_s16x2){0, 0}));
}
__device__ __forceinline__ h16x8 decw_words(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return l0w_from_u2(make_uint2(a, b), make_uint2(c, d)); }
// DROP: train mode (dropout 0.1 at eight sites of every layer) -- a template parameter, so that the per-site tests are not sixteen uniform
// branches per epilogue
// DENSE: the dense-traffic shapes (BASELINE configs[4]: up to 16 reference lines, up to 192 tokens).  r2r tiles hold ONE mode x 16 lines
// (12 tiles), m2m / cross / FFN tiles one line each (up to 16): both tilings run in rounds of eight tiles, the sub-block's groups are streamed
// once per round; the residual changes tiling through the query array in global memory (release / acquire fences around the group barrier)
// instead of the 96-row LDS buffer, the parameter blocks of two layers are resident (double-buffered by layer parity), and the scene's
// K | V^T operands come as four per-head groups of 12 key tiles (dec_kv.h writes them).
// NKE (DENSE): key tiles the cross attention walks -- 8 when the batch has N <= 128 token slots, else all 12 of the scene's K | V^T image;
// the tiles beyond the batch's own are zero fragments behind -inf masks (exact zeros in both sums).
// NKE = 8 without DENSE (round 5, "MID"): the standard kernel -- one round of tiles, LDS hand-over, one parameter block -- for batches with
// R <= 8 reference lines and 96 < N <= 128 token slots: what `train_cbv` really collates (49 agents + 60 polygons = 109 slots), which the
// dense variant served at 1.5 x the standard kernel's time.  Eight key tiles make a head pair's K | V^T group exactly 32 fragments, one
// ring slot; they are gathered out of the dense per-head image (dec_kv.h) in runs of four fragments.
template <bool DROP, bool DENSE, int NKE = (DENSE ? 12 : 6)>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void dec_w_kernel(DecWP p) {
  constexpr int M = 12, XS = DECW_XS;
  constexpr int NS = DENSE ? 16 : 8;             // reference-line slots of an r2r tile
  constexpr int NKT = DENSE ? 12 : NKE;          // key tiles of the cross attention
  constexpr bool MID = !DENSE && NKE == 8;
  constexpr int NTA = DENSE ? 12 : 6;            // r2r tiles: one mode each / one mode pair each
  constexpr int RA = DENSE ? 2 : 1;              // rounds of eight tiles in the r2r tiling
  constexpr int GB = DENSE ? 18 : 16;            // groups of the reference-line tiling: m2m 4 | cross q | K|V^T 4 or 2 | cross out | FFN 8
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char* ring = smem_raw;
  float* xs = reinterpret_cast<float*>(smem_raw + 2 * 32768);                  // (not DENSE) [96][XS] residual hand-over
  float* parbase = DENSE ? xs : xs + 96 * XS;                                   // DENSE: [2][DECW_PAR_LAYER] by layer parity
  float* parE = parbase;
  float* parL = parE + DECW_E_N;
  float* smaskf = parbase + (DENSE ? 2 : 1) * DECW_PAR_LAYER;                  // [16 NKT] encoder key mask as 0 / -inf
  float* qmaskf = smaskf + 16 * NKT;                                           // [12][NS] r2r quirk rows as 0 / -inf
  unsigned char* rz = reinterpret_cast<unsigned char*>(qmaskf + 12 * NS);      // [NS] padded reference lines of this scene
  const int tid = threadIdx.x;
  int lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;            // re-derived through an opaque zero every layer (see the loop)
  const int wv0 = __builtin_amdgcn_readfirstlane(tid >> 6);
  int wv = wv0;
  const int b = blockIdx.x, N = p.N, R = p.R, NQ = R * M;
  const bool six_all = N > 80;                   // slot-ordered keys: whether the BATCH has a sixth key tile; compacted keys: per scene, read off the mask below
  const int RB = DENSE ? (R > 8 ? 2 : 1) : 1;    // rounds of eight tiles in the reference-line tiling
  const int GL = 4 * RA + GB * RB;               // groups of a layer (even)
  const size_t qrow0 = (size_t)b * NQ;
  const float dp = p.dropout, dpk = dp > 0.f ? 1.0f / (1.0f - dp) : 1.0f;
  const uint32_t thr16 = drop_thr16(dp);                                     // a half u (unsigned) is dropped where u < thr16 (fp32 sites) ...
  const uint32_t tau16 = (thr16 - 32768u) & 0xffffu, tau2 = tau16 | (tau16 << 16);           // ... or, read as a signed integer, below tau (packed sites)
  const uint32_t tm1 = ((tau16 - 1u) & 0xffffu) | (((tau16 - 1u) & 0xffffu) << 16);
  const uint32_t lds00 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem_raw);
  uint32_t lds0 = lds00;
  uint32_t voff = (uint32_t)lane * 16u;
  const f32x4 Z = {0.f, 0.f, 0.f, 0.f};
  DecWRng rng;
#pragma unroll
  for (int i = 0; i < 4; ++i) rng.x[i] = hash32(p.seed, p.stream + i, (uint32_t)(b * 512 + tid)) | 1u;
  if (DROP && p.l0 > 0 && p.rng_io) {          // the second half of a split launch carries on where the first one stopped
    const uint4 st = *reinterpret_cast<const uint4*>(p.rng_io + ((size_t)b * 512 + tid) * 4);
    rng.x[0] = st.x; rng.x[1] = st.y; rng.x[2] = st.z; rng.x[3] = st.w;
  }
#if RIFT_DROP_STATS
  int dsite = 0;                                 // which of the eight dropout sites of a layer the next keep4() calls belong to
  unsigned int dkept[RIFT_DS_DEC_SITES] = {0, 0, 0, 0, 0, 0, 0, 0}, ddrawn[RIFT_DS_DEC_SITES] = {0, 0, 0, 0, 0, 0, 0, 0};
#define DSITE(s) dsite = (s)
#else
#define DSITE(s)
#endif
  int tsn = 0;
#define DTS() do { if (p.ts && b == 0 && tid == 0 && tsn < 120) p.ts[tsn++] = clock64(); } while (0)
#ifndef RIFT_DEC_ARR
#define RIFT_DEC_ARR 0                           // (diagnostic build, tools/decw_arrivals.py: -DRIFT_DEC_ARR=1 -- ten more live registers)
#endif
#if RIFT_DEC_ARR
  int asn = 0;                                   // when each wave ARRIVES at a group boundary: [128 + wave * 112 + boundary]
#define DTS_ARR() do { if (p.ts && b == 0 && (tid & 63) == 0 && asn < 112) p.ts[128 + (tid >> 6) * 112 + asn++] = clock64(); } while (0)
#else
#define DTS_ARR()
#endif

  // a contiguous source of nfrag fragments -> LDS byte offset dst.  Who loads: with R <= 6 waves 6 and 7 own no tile in either tiling, and
  // waves 2 and 3 are alone on their SIMDs among the working waves (6 tiles on 4 SIMDs): those four carry the stream, the waves of the
  // doubly loaded SIMDs 0 and 1 go from the barrier straight to their MFMAs.  Otherwise all eight waves share it.
  const bool ld_few = !DENSE && R <= 6 && !(p.dbg & 16);     // (dbg 16: all eight waves carry the stream)
  auto dma = [&](const void* src, uint32_t dst, int nfrag) {
    if ((ld_few && !(wv & 2)) || (p.dbg & 8)) return;            // (dbg 8: no stream at all -- compute on whatever LDS holds, timing only)
    const int lw = ld_few ? (wv & 1) + ((wv >> 2) << 1) : wv, ln = ld_few ? 4 : 8;
    decw_dma_share(reinterpret_cast<const unsigned char*>(src), voff, lds0 + dst, nfrag, lw, ln);   // (one rolled loop: every boundary site carries it)
  };
  // (MID) head pair g of a layer's dense K | V^T image (per head: 12 K fragments, 12 V^T fragments (dim tile d) x 6 + pt) -> one ring slot,
  // per head hh 16 fragments: K key tiles 0..7 | V^T (d, pt < 4) at 8 + 4 d + pt -- eight runs of four consecutive fragments
  auto dma_kv_mid = [&](const unsigned char* kvl, int g, uint32_t dst) {
    if ((ld_few && !(wv & 2)) || (p.dbg & 8)) return;
    const int lw = ld_few ? (wv & 1) + ((wv >> 2) << 1) : wv, nr = ld_few ? 2 : 1;
#pragma unroll 1
    for (int r = lw * nr; r < lw * nr + nr; ++r) {
      const int hh = r >> 2, q = r & 3;
      const int so = (2 * g + hh) * 24 + (q < 2 ? q * 4 : 12 + (q - 2) * 6);
      decw_glds4(kvl + (size_t)so * 1024, voff, lds0 + dst + (uint32_t)r * 4096u);
    }
  };
  // group boundary: my share of the next group has landed; after the barrier everybody's has, and nobody reads the other slot any more
  auto sync = [&]() { DTS_ARR(); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); DTS(); };
  auto W = [&](int slot, int f) { return *reinterpret_cast<const h16x8*>(ring + slot * 32768 + f * 1024 + lane * 16); };

  const unsigned char* wimg = reinterpret_cast<const unsigned char*>(p.img);
  constexpr int KVF = (DENSE || MID) ? 96 : DECW_KV_FRAGS;                     // K | V^T fragments per (scene, layer)
  const unsigned char* kvimg = reinterpret_cast<const unsigned char*>(p.KV) + (size_t)b * 4 * KVF * 1024;
  const uint32_t OFF_P = 2 * 32768 + (DENSE ? 0 : 96 * XS * 4), OFF_E = OFF_P, OFF_L = OFF_E + DECW_E_N * 4;
  // prologue: first group + the parameter block of layer 0 in flight, then the queries and masks
  dma(wimg + (size_t)p.l0 * DECW_LAYER_FRAGS * 1024, 0, 32);
  dma(p.par + (size_t)p.l0 * DECW_PAR_LAYER, OFF_E, 18);
  dma(p.par + (size_t)p.l0 * DECW_PAR_LAYER + DECW_E_N, OFF_L, 6);
  if (!DENSE) {                                 // every query row requested before the first LDS store (as a rolled loop this was six dependent global round trips)
    float4 qv[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int i = tid + u * 512, r = i >> 5, c4 = (i & 31) * 4;
      qv[u] = *reinterpret_cast<const float4*>(p.Q + (qrow0 + (r < NQ ? r : 0)) * 128 + c4);
    }
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int i = tid + u * 512, r = i >> 5, c4 = (i & 31) * 4;
      if (r < NQ) *reinterpret_cast<float4*>(xs + r * XS + c4) = qv[u];
    }
  }
  for (int i = tid; i < 16 * NKT; i += 512) smaskf[i] = ((i >= N) || p.kpm[(size_t)b * N + i]) ? -INFINITY : 0.f;
  for (int i = tid; i < 12 * NS; i += 512) {   // quirk: mode m of scene b uses the padding row of scene (b*12+m) % bs (planning_decoder.py:56-60)
    const int m = i / NS, r = i - m * NS;
    qmaskf[i] = ((r >= R) || p.q_kpm[(size_t)(((p.q_off + b) * M + m) % p.q_bs) * R + r]) ? -INFINITY : 0.f;
  }
  if (tid < NS) rz[tid] = (tid >= R) || p.r_kpm[(size_t)b * R + tid];

  bool actA, actB, a_ok, b_ok;
  int a_sub, a_row, b_row, tileA, tileB;
  // tiling A: slot l15 = sub * 8 + r of mode pair tileA (DENSE: slot l15 = r of mode tileA);  tiling B (reference line tileB): slot l15 = mode
  auto derive = [&](int zv, int zs) {
    lane = (tid & 63) + zv; l15 = lane & 15; l4 = lane >> 4; voff = (uint32_t)lane * 16u;
    wv = wv0 + zs; lds0 = lds00 + (uint32_t)zs;
  };
  auto set_tiles = [&](int ra, int rb) {
    tileA = ra * 8 + wv; tileB = rb * 8 + wv;
    actA = tileA < NTA && !(p.dbg & 1); actB = tileB < R && !(p.dbg & 1);
    a_sub = DENSE ? 0 : (l15 >> 3);
    const int a_r = DENSE ? l15 : (l15 & 7);
    a_ok = a_r < R; b_ok = l15 < M;
    a_row = a_ok ? a_r * M + (DENSE ? tileA : 2 * tileA + a_sub) : 0; b_row = b_ok ? tileB * M + l15 : 0;
  };
  derive(0, 0);
  set_tiles(0, 0);

  auto init8 = [&](f32x4 (&a)[8], const float* bias) {        // accumulators start from the bias row (4 channels per n-tile of this lane)
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) { const float4 v = *reinterpret_cast<const float4*>(bias + nt * 16 + l4 * 4); a[nt] = (f32x4){v.x, v.y, v.z, v.w}; }
  };
  auto zero8 = [&](f32x4 (&a)[8]) {
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) a[nt] = Z;
  };
  // residual hand-over between the tilings: the LDS buffer xs, or (DENSE) the query array itself -- written with a release fence ahead of
  // the group barrier, read behind an acquire fence.  WORKGROUP scope: writer and reader waves sit on one CU and share its write-through
  // vector L1, so the fences are waits, not cache maintenance (agent scope writes back / invalidates the XCD's whole L2: measured 40-60 us
  // per hand-over)
  // (a lane row without a query -- reference-line slot >= R, mode slot >= 12 -- reads row 0: `row` is 0 there.  Its keys are masked, its V
  // row meets P = 0 and is finite, its results are never written: selecting zeros instead cost a divergent branch per hand-over)
  auto read_xs = [&](f32x4 (&res)[8], int row, bool) {
    const float* src = DENSE ? p.Q + (qrow0 + row) * 128 + l4 * 4 : xs + row * XS + l4 * 4;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      const float4 v = *reinterpret_cast<const float4*>(src + nt * 16);
      res[nt] = (f32x4){v.x, v.y, v.z, v.w};
    }
  };
  auto write_xs = [&](const f32x4 (&res)[8], int row, bool ok) {
    if (ok) {
      float* dst = DENSE ? p.Q + (qrow0 + row) * 128 + l4 * 4 : xs + row * XS + l4 * 4;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) *reinterpret_cast<float4*>(dst + nt * 16) = make_float4(res[nt][0], res[nt][1], res[nt][2], res[nt][3]);
    }
  };
  auto publish = [&]() { if (DENSE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); };     // after the last write_xs of a tiling, before its barrier
  auto acquire = [&]() { if (DENSE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); };     // after that barrier, before the first read_xs
  // res -> xb (bf16 operands of the four k-steps); g: gamma 128 | beta 128 in LDS.  Two-pass statistics as torch; vector (packed fp32) math.
  auto layer_norm = [&](const f32x4 (&res)[8], h16x8 (&xb)[4], const float* g) {
    f32x4 s4 = (res[0] + res[1]) + (res[2] + res[3]);
    s4 += (res[4] + res[5]) + (res[6] + res[7]);
    const float mean = rows_sum((s4[0] + s4[1]) + (s4[2] + s4[3])) * (1.0f / 128.0f);
    if (RIFT_LN_FOLD && true) {     // (opfmt.h: gamma / beta live in the consuming GEMM's weights and bias; one-pass statistics)
      f32x4 q4 = res[0] * res[0];
#pragma unroll
      for (int nt = 1; nt < 8; ++nt) q4 += res[nt] * res[nt];
      const float ex2 = rows_sum((q4[0] + q4[1]) + (q4[2] + q4[3])) * (1.0f / 128.0f);
      const float m2 = mean * mean;
      float var = ex2 - m2;
      if (__builtin_expect(ln_cancels(m2, var), 0)) {      // (common.h: a row whose mean dwarfs its spread -- the centred form, as torch)
        f32x4 d4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) { const f32x4 d = res[nt] - mean; d4 += d * d; }
        var = rows_sum((d4[0] + d4[1]) + (d4[2] + d4[3])) * (1.0f / 128.0f);
      }
      const float r = rsqrtf(fmaxf(var, 0.f) + 1e-5f), c = -mean * r;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) xb[ks] = l0w_pack8(res[2 * ks] * r + c, res[2 * ks + 1] * r + c);
      return;
    }
    f32x4 d[8];
    f32x4 q4 = Z;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) { d[nt] = res[nt] - mean; q4 += d[nt] * d[nt]; }
    const float r = rsqrtf(rows_sum((q4[0] + q4[1]) + (q4[2] + q4[3])) * (1.0f / 128.0f) + 1e-5f);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      f32x4 y[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int nt = 2 * ks + u;
        const float4 gg = *reinterpret_cast<const float4*>(g + nt * 16 + l4 * 4), bb = *reinterpret_cast<const float4*>(g + 128 + nt * 16 + l4 * 4);
        y[u] = d[nt] * ((f32x4){gg.x, gg.y, gg.z, gg.w} * r) + (f32x4){bb.x, bb.y, bb.z, bb.w};
      }
      xb[ks] = l0w_pack8(y[0], y[1]);
    }
  };
  // 16 rows x K=128 -> 128 output channels against the 32 fragments of a ring slot (swapped operands: lane = 4 channels of its row)
  auto gemm = [&](int slot, const h16x8 (&x)[4], f32x4 (&acc)[8]) {
    decw_gemm<false>((uint32_t)(uintptr_t)ring + (uint32_t)slot * 32768u + voff, x, acc);
  };
  // q / k of the four heads as attention operands: n-tile pair (2h, 2h+1) -> one bf16 fragment (the bias came in through the accumulator)
  auto to_heads = [&](const f32x4 (&acc)[8], h16x8 (&out)[4]) {
#pragma unroll
    for (int h = 0; h < 4; ++h) out[h] = l0w_pack8(acc[2 * h], acc[2 * h + 1]);
  };
  // V of the tile in the plain operand order (A = activations): lane = 4 consecutive KEYS (rows 4*l4..) of dim nt*16 + l15 = the V^T operand
  auto gemm_v = [&](int slot, const h16x8 (&xb)[4], DecwVf (&vf)[8], const float* bias) {
    f32x4 acc[8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) { const float bv = bias[nt * 16 + l15]; acc[nt] = (f32x4){bv, bv, bv, bv}; }
    decw_gemm<true>((uint32_t)(uintptr_t)ring + (uint32_t)slot * 32768u + voff, xb, acc);
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
#if RIFT_ATTN_K16
      vf[nt] = __builtin_bit_cast(h16x4, pack_h4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]));
#else
      vf[nt] = l0w_from_u2(pack_h4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]), make_uint2(0u, 0u));   // k slots 4..7 unused
#endif
    }
  };
  // one step of stream i: a word of two 16-bit uniforms
  auto draw2 = [&](int i) -> uint32_t {
    constexpr uint32_t C[4] = {2531011u, 1013904223u & 0xffffffu, 12345u, 7046029u};
    return rng.x[i] = decw_step(rng.x[i], C[i]);
  };
#if RIFT_DROP_STATS
  auto dcount = [&](uint32_t keepmask, int n) { ddrawn[dsite] += n; dkept[dsite] += __builtin_popcount(keepmask & 0x00010001u); };
#else
  auto dcount = [&](uint32_t, int) {};
#endif
  // 0xffff / 0 per half of the next word of stream i: kept / dropped
  auto keep2 = [&](int i) -> uint32_t { const uint32_t m = decw_keep2(draw2(i), tm1); dcount(m, 2); return m; };
  // fp32 multipliers of four consecutive draws (streams i, i + 1): ks or 0
  auto keep4 = [&](int i, float ks) -> f32x4 {
    const uint32_t h0 = draw2(i), h1 = draw2(i + 1);
    const unsigned short t = (unsigned short)thr16;         // 16-bit compares on the register halves: no extraction instructions
    const bool d0 = (unsigned short)(h0 & 0xffffu) < t, d1 = (unsigned short)(h0 >> 16) < t, d2 = (unsigned short)(h1 & 0xffffu) < t, d3 = (unsigned short)(h1 >> 16) < t;
#if RIFT_DROP_STATS
    ddrawn[dsite] += 4; dkept[dsite] += (d0 ? 0 : 1) + (d1 ? 0 : 1) + (d2 ? 0 : 1) + (d3 ? 0 : 1);
#endif
    return (f32x4){d0 ? 0.f : ks, d1 ? 0.f : ks, d2 ? 0.f : ks, d3 ? 0.f : ks};
  };
  // 16 x 16 self-attention of the tile, per head, entirely in registers: S^T = K Q^T (lane: 4 keys of query l15), O^T = V^T P^T.
  // `mask4`: 0 / -inf of this lane's four keys, entering as the accumulator of the score MFMA; scores are in log2 units (q carries log2 e).
  auto self_attention = [&](const f32x4 mask4, const h16x8 (&qf)[4], const h16x8 (&kf)[4], const DecwVf (&vf)[8], h16x8 (&ao)[4]) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const f32x4 s = mfma_h(kf[h], qf[h], mask4, 0, 0, 0);
      const float m = rows_max(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])));
      f32x4 ev = {__builtin_amdgcn_exp2f(s[0] - m), __builtin_amdgcn_exp2f(s[1] - m), __builtin_amdgcn_exp2f(s[2] - m), __builtin_amdgcn_exp2f(s[3] - m)};
      const float lsum = rows_sum((ev[0] + ev[1]) + (ev[2] + ev[3]));
      uint2 pw = pack_h4(ev[0], ev[1], ev[2], ev[3]);
      if (DROP) { pw.x &= keep2(2 * (h & 1)); pw.y &= keep2(2 * (h & 1) + 1); }      // dropped weights are zero words; 1/(1-p) rides on 1/sum
#if RIFT_ATTN_K16
      const h16x4 pf = __builtin_bit_cast(h16x4, pw);
      const f32x4 o0 = mfma_h16(vf[2 * h], pf, Z);
      const f32x4 o1 = mfma_h16(vf[2 * h + 1], pf, Z);
#else
      const h16x8 pf = l0w_from_u2(pw, make_uint2(0u, 0u));
      const f32x4 o0 = mfma_h(vf[2 * h], pf, Z, 0, 0, 0);
      const f32x4 o1 = mfma_h(vf[2 * h + 1], pf, Z, 0, 0, 0);
#endif
      const float inv = __builtin_amdgcn_rcpf(lsum) * (DROP ? dpk : 1.0f);
      ao[h] = l0w_pack8(o0 * inv, o1 * inv);
    }
  };
  // x += dropout(acc)   (acc already holds the bias)
  auto residual = [&](f32x4 (&res)[8], const f32x4 (&acc)[8], float ks) {       // ks: 1/(1-p) x whatever factor acc still lacks
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      if (DROP) res[nt] += acc[nt] * keep4(2 * (nt & 1), ks);
      else res[nt] += acc[nt];
    }
  };
  // cross attention of `nh` heads starting at h0 against the K | V^T fragments in a ring slot (16 NKT keys); the key-padding mask is the score
  // accumulator.  Fragment order of the slot: K (kt, hh) at kt * nh + hh, V^T (hh, dim tile d, key pair pt) behind them.
  auto cross_heads = [&](int slot, int h0, auto nh_t, const h16x8 (&qf)[4], h16x8 (&ao)[4]) {
    constexpr int nh = decltype(nh_t)::value;
#pragma unroll
    for (int hh = 0; hh < nh; ++hh) {
      const int h = h0 + hh;
      f32x4 s[NKT];
      constexpr int NK0 = (DENSE || MID) ? NKE : 5;         // (standard: keys 80..95 exist only in batches with more than 80 tokens)
      const bool six = (!DENSE && !MID) && (p.compact ? (smaskf[80] == 0.f) : six_all);      // (workgroup-uniform: one scene per workgroup)
      auto KF = [&](int kt) { return MID ? hh * 16 + kt : kt * nh + hh; };                    // fragment of the slot: K (key tile kt, head hh) ...
      auto VF = [&](int d, int pt) { return MID ? hh * 16 + 8 + d * 4 + pt : NKT * nh + (hh * 2 + d) * (NKT / 2) + pt; };     // ... V^T (dim tile d, key pair pt)
#pragma unroll
      for (int kt = 0; kt < NK0; ++kt) {
        const float4 mk = *reinterpret_cast<const float4*>(smaskf + kt * 16 + l4 * 4);
        s[kt] = mfma_h(W(slot, KF(kt)), qf[h], (f32x4){mk.x, mk.y, mk.z, mk.w}, 0, 0, 0);
      }
      float m = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3]));
#pragma unroll
      for (int kt = 1; kt < NK0; ++kt) m = fmaxf(fmaxf(m, fmaxf(s[kt][0], s[kt][1])), fmaxf(s[kt][2], s[kt][3]));
      if (!DENSE && !MID && six) {
        const float4 mk = *reinterpret_cast<const float4*>(smaskf + 80 + l4 * 4);
        s[NKT - 1] = mfma_h(W(slot, KF(5)), qf[h], (f32x4){mk.x, mk.y, mk.z, mk.w}, 0, 0, 0);
        m = fmaxf(fmaxf(m, fmaxf(s[NKT - 1][0], s[NKT - 1][1])), fmaxf(s[NKT - 1][2], s[NKT - 1][3]));
      }
      m = rows_max(m);
      f32x4 l4s = Z;
#pragma unroll
      for (int kt = 0; kt < NK0; ++kt) {
        s[kt] = (f32x4){__builtin_amdgcn_exp2f(s[kt][0] - m), __builtin_amdgcn_exp2f(s[kt][1] - m), __builtin_amdgcn_exp2f(s[kt][2] - m), __builtin_amdgcn_exp2f(s[kt][3] - m)};
        l4s += s[kt];
      }
      if (!DENSE && !MID) {
        if (six) {
          s[5] = (f32x4){__builtin_amdgcn_exp2f(s[5][0] - m), __builtin_amdgcn_exp2f(s[5][1] - m), __builtin_amdgcn_exp2f(s[5][2] - m), __builtin_amdgcn_exp2f(s[5][3] - m)};
          l4s += s[5];
        } else s[5] = Z;
      }
      const float lsum = rows_sum((l4s[0] + l4s[1]) + (l4s[2] + l4s[3]));
      f32x4 o0 = Z, o1 = Z;
#pragma unroll
      for (int pt = 0; pt < ((DENSE || MID) ? NKE : NKT) / 2; ++pt) {
        uint2 pa = pack_h4(s[2 * pt][0], s[2 * pt][1], s[2 * pt][2], s[2 * pt][3]), pb2 = pack_h4(s[2 * pt + 1][0], s[2 * pt + 1][1], s[2 * pt + 1][2], s[2 * pt + 1][3]);
        if (DROP) { pa.x &= keep2(0); pa.y &= keep2(1); pb2.x &= keep2(2); pb2.y &= keep2(3); }
        const h16x8 pf = l0w_from_u2(pa, pb2);
        o0 = mfma_h(W(slot, VF(0, pt)), pf, o0, 0, 0, 0);
        o1 = mfma_h(W(slot, VF(1, pt)), pf, o1, 0, 0, 0);
      }
      const float inv = __builtin_amdgcn_rcpf(lsum) * (DROP ? dpk : 1.0f);
      ao[h] = l0w_pack8(o0 * inv, o1 * inv);
    }
  };

  // ---- the operand stream of a layer: RA rounds of the r2r groups [r2r q | k | v | out], then RB rounds of the reference-line groups
  // [m2m q | k | v | out | cross q | scene K|V^T (two head pairs; DENSE: four heads) | cross out | (ffn.0 chunk c, ffn.3 chunk c) x 4];
  // position `pos` of layer li has sequence number li * GL + pos and lands in ring slot (pos & 1) (GL is even).
  auto request = [&](int li, int pos) {          // the group at (li, pos); pos == GL means the first group of the next layer
    if (pos >= GL) { pos = 0; ++li; }
    if (li >= p.l1) return;
    const unsigned char* wl = wimg + (size_t)li * DECW_LAYER_FRAGS * 1024;
    const unsigned char* kvl = kvimg + (size_t)li * KVF * 1024;
    const uint32_t dst = (uint32_t)(pos & 1) * 32768u;
    if (pos < 4 * RA) { dma(wl + (size_t)(pos & 3) * 32768, dst, 32); return; }
    const int q = (pos - 4 * RA) % GB;
    constexpr int NKV = DENSE ? 4 : 2;           // K | V^T groups, 24 fragments each
    if (q < 5) dma(wl + (size_t)(4 + q) * 32768, dst, 32);
    else if (q < 5 + NKV) { if (MID) dma_kv_mid(kvl, q - 5, dst); else dma(kvl + (size_t)(q - 5) * 24 * 1024, dst, 24); }
    else dma(wl + (size_t)(9 + q - 5 - NKV) * 32768, dst, 32);
  };
  // group boundary that opens position `pos` of layer li: my share of the group has landed; after the barrier everybody's has, and nobody
  // reads the other slot any more: the group after it goes there.  The parameter blocks ride on fixed boundaries.  (`pos` is a literal at
  // the call sites of the standard kernel, so the stream addresses fold; the dense variant adds its round offsets at run time.)
  auto bnd = [&](int li, int pos) {
    sync();
    request(li, pos + 1);
    if (DENSE) {             // both regions of the NEXT layer into the other parity block, any time during this layer
      if (pos == 0 && li + 1 < p.l1) dma(p.par + (size_t)(li + 1) * DECW_PAR_LAYER, OFF_P + (uint32_t)((li + 1) & 1) * DECW_PAR_LAYER * 4, 24);
    } else {
      if (pos == 0 && li > p.l0) dma(p.par + (size_t)li * DECW_PAR_LAYER + DECW_E_N, OFF_L, 6);        // region L of this layer: the previous FFN epilogue is over
      if (pos == 12 && li + 1 < p.l1) dma(p.par + (size_t)(li + 1) * DECW_PAR_LAYER, OFF_E, 18);       // region E of the next layer: r2r / m2m are over
    }
  };

#pragma unroll 1
  for (int li = p.l0; li < p.l1; ++li) {
    {   // the lane / wave indices pass through opaque zeros once per layer: otherwise every LDS and DMA address of the 20 groups is
        // hoisted out of this loop as a loop invariant and spilled around it
      int zv, zs;
      asm volatile("v_mov_b32 %0, 0" : "=v"(zv));
      asm volatile("s_mov_b32 %0, 0" : "=s"(zs));
      derive(zv, zs);
    }
    if (DENSE) { parE = parbase + (li & 1) * DECW_PAR_LAYER; parL = parE + DECW_E_N; }
    // Waves without a tile in a tiling run the same boundaries (barrier + their share of the stream) and nothing else; the working waves'
    // code is straight-line between the boundaries of a tiling, so that register liveness follows the phases.  A group at round-relative
    // position k sits in ring slot k & 1 (the rounds start at even positions).
    // ================= tiling A: r2r over the reference lines of a mode pair (DENSE: of one mode) =================
#pragma unroll 1
    for (int ra = 0; ra < RA; ++ra) {
      set_tiles(ra, 0);
      const int pa = DENSE ? 4 * ra : 0;                          // first position of this round
      if (actA) {
        f32x4 res[8], acc[8];
        h16x8 xb[4], qf[4], kf[4], ao[4];
        DecwVf vf[8];
        bnd(li, pa + 0);                                        // ---- r2r q
        if (ra == 0) acquire();
        read_xs(res, a_row, a_ok);
        layer_norm(res, xb, parE + DECW_E_LN1);
        init8(acc, parE + DECW_E_BR2R); gemm(0, xb, acc);
        to_heads(acc, qf);
        init8(acc, parE + DECW_E_BR2R + 128);                   // (bias rows are requested ahead of the barrier that hides their latency)
        bnd(li, pa + 1);                                        // ---- r2r k
        gemm(1, xb, acc);
        to_heads(acc, kf);
        bnd(li, pa + 2);                                        // ---- r2r v + attention
        gemm_v(0, xb, vf, parE + DECW_E_BR2R + 256);
        if (DENSE) {     // keys 4 l4 .. + 3 = reference lines of mode tileA: the quirk's padded lines are masked
          const float4 mk = *reinterpret_cast<const float4*>(qmaskf + tileA * NS + l4 * 4);
          DSITE(0);
          self_attention((f32x4){mk.x, mk.y, mk.z, mk.w}, qf, kf, vf, ao);
        } else {         // keys 4 l4 .. + 3 = reference lines (l4 & 1) * 4 .. of mode 2 tileA + (l4 >> 1): other-mode keys are masked as well
          const float4 mk = *reinterpret_cast<const float4*>(qmaskf + (2 * tileA + (l4 >> 1)) * NS + (l4 & 1) * 4);
          const bool cross = (l4 >> 1) != a_sub;
          const float ninf = -INFINITY;
          DSITE(0);
          self_attention((f32x4){cross ? ninf : mk.x, cross ? ninf : mk.y, cross ? ninf : mk.z, cross ? ninf : mk.w}, qf, kf, vf, ao);
        }
        init8(acc, parE + DECW_E_BR2RO);
        bnd(li, pa + 3);                                        // ---- r2r out_proj, residual, hand-over to the reference-line tiling
        gemm(1, ao, acc);
        DSITE(1);
        residual(res, acc, dpk);
        write_xs(res, a_row, a_ok);
      } else {
#pragma unroll 1
        for (int k = 0; k < 4; ++k) bnd(li, pa + k);
      }
    }
    publish();
    // ================= tiling B: m2m, cross attention, FFN of one reference line =================
#pragma unroll 1
    for (int rb = 0; rb < RB; ++rb) {
      set_tiles(0, rb);
      const int pb = DENSE ? 4 * RA + GB * rb : 4;                // first position of this round
      constexpr int KO = DENSE ? 9 : 7;                           // round-relative position of cross out_proj (behind the K | V^T groups)
      if (actB) {
        f32x4 res[8], acc[8];
        h16x8 xb[4], qf[4], kf[4], ao[4];
        DecwVf vf[8];
        bnd(li, pb + 0);                                        // ---- m2m q (+ m_pos)
        if (rb == 0) acquire();
        read_xs(res, b_row, b_ok);
        layer_norm(res, xb, parE + DECW_E_LN2);
        init8(acc, parE + DECW_E_PB + (b_ok ? l15 : 0) * DECW_PBS); gemm(0, xb, acc);
        to_heads(acc, qf);
        init8(acc, parE + DECW_E_PB + (b_ok ? l15 : 0) * DECW_PBS + 128);
        bnd(li, pb + 1);                                        // ---- m2m k (+ m_pos)
        gemm(1, xb, acc);
        to_heads(acc, kf);
        bnd(li, pb + 2);                                        // ---- m2m v + attention over the modes
        gemm_v(0, xb, vf, parE + DECW_E_BM2MV);
        DSITE(2);
        { const float mk = l4 == 3 ? -INFINITY : 0.f; self_attention((f32x4){mk, mk, mk, mk}, qf, kf, vf, ao); }   // keys 12..15 are padding slots
        init8(acc, parE + DECW_E_BM2MO);
        bnd(li, pb + 3);                                        // ---- m2m out_proj, residual, padded lines zeroed (:70-72), LayerNorm
        gemm(1, ao, acc);
        DSITE(3);
        residual(res, acc, dpk);
        if (rz[tileB]) zero8(res);
        layer_norm(res, xb, parL + DECW_L_LN3);
        init8(acc, parL + DECW_L_BCQ);
        bnd(li, pb + 4);                                        // ---- cross q
        gemm(0, xb, acc);
        to_heads(acc, qf);
        DSITE(4);
        if (DENSE) {                                            // ---- the scene's K | V^T: one group per head (DENSE) / per head pair
#pragma unroll
          for (int h = 0; h < 4; ++h) { bnd(li, pb + 5 + h); cross_heads((5 + h) & 1, h, std::integral_constant<int, 1>{}, qf, ao); }
        } else {
          bnd(li, pb + 5); cross_heads(1, 0, std::integral_constant<int, 2>{}, qf, ao);
          bnd(li, pb + 6); cross_heads(0, 2, std::integral_constant<int, 2>{}, qf, ao);
        }
        init8(acc, parL + DECW_L_BCO);
        bnd(li, pb + KO);                                       // ---- cross out_proj, residual, LayerNorm for the FFN
        gemm(KO & 1, ao, acc);
        DSITE(5);
        residual(res, acc, dpk);
        layer_norm(res, xb, parL + DECW_L_LN4);
        DSITE(6);
        f32x4 acc2[8];
        init8(acc2, parL + DECW_L_BF2);
        if (DROP) {            // the hidden layer's kept units go into ffn.3 unscaled: acc2 = (b2 + y) / dpk, the branch site multiplies by dpk^2
          const float idpk = 1.0f - dp;
#pragma unroll
          for (int nt = 0; nt < 8; ++nt) acc2[nt] *= idpk;
        }
#pragma unroll 1
        for (int hc = 0; hc < 4; ++hc) {                        // ---- ffn.0 chunk -> ReLU, dropout -> ffn.3 partial
          init8(acc, parL + DECW_L_BF1 + hc * 128);
          bnd(li, pb + KO + 1 + 2 * hc);
          gemm((KO + 1) & 1, xb, acc);
          h16x8 hb[4];
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {     // ReLU (and dropout: a dropped unit gets the sign bit first) on the packed operand words
            uint32_t w[4];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const f32x4 a = acc[2 * ks + q];
              const uint2 u = pack_h4(a[0], a[1], a[2], a[3]);
              w[2 * q] = u.x; w[2 * q + 1] = u.y;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              if (DROP) { const uint32_t sg = decw_drop_sign2(draw2(q), tau2); dcount(~(sg >> 15), 2); w[q] |= sg; }
              w[q] = decw_relu2(w[q]);
            }
            hb[ks] = decw_words(w[0], w[1], w[2], w[3]);
          }
          bnd(li, pb + KO + 2 + 2 * hc);
          gemm(KO & 1, hb, acc2);
        }
        DSITE(7);
        residual(res, acc2, dpk * dpk);
        if (li == 3) {        // the reference asserts torch.isfinite(q).all() behind every block (planning_decoder.py:175); a NaN / Inf stays in
          uint32_t ex = 0;    // the residual stream, so the rows leaving the last block are tested -- by bit pattern (this unit is built -fno-honor-nans)
#pragma unroll
          for (int nt = 0; nt < 8; ++nt) { ex = exp_max(ex, res[nt][0]); ex = exp_max(ex, res[nt][1]); ex = exp_max(ex, res[nt][2]); ex = exp_max(ex, res[nt][3]); }
          if (__builtin_amdgcn_ballot_w64(b_ok && nonfinite_exp(ex)) != 0ull && lane == 0 && p.nonfinite) atomicOr(p.nonfinite, 1);
        }
        if (DENSE || li + 1 < p.l1) write_xs(res, b_row, b_ok);
        else if (b_ok) {
#pragma unroll
          for (int nt = 0; nt < 8; ++nt)
            *reinterpret_cast<float4*>(p.Q + (qrow0 + b_row) * 128 + nt * 16 + l4 * 4) = make_float4(res[nt][0], res[nt][1], res[nt][2], res[nt][3]);
        }
      } else {
#pragma unroll 1
        for (int k = 0; k < GB; ++k) bnd(li, pb + k);
      }
    }
    publish();
  }
  if (DROP && p.l1 < 4 && p.rng_io)
    *reinterpret_cast<uint4*>(p.rng_io + ((size_t)b * 512 + tid) * 4) = make_uint4(rng.x[0], rng.x[1], rng.x[2], rng.x[3]);
  DTS();
#undef DTS
#undef DTS_ARR
#undef DSITE
#if RIFT_DROP_STATS
  if (DROP && p.ds.elem) {
    for (int s = 0; s < RIFT_DS_DEC_SITES; ++s) {
      atomicAdd(&p.ds.elem[2 * s], (unsigned long long)dkept[s]);
      atomicAdd(&p.ds.elem[2 * s + 1], (unsigned long long)ddrawn[s]);
    }
    if (b == 0 && tid == 0) for (int s = 0; s < RIFT_DS_DEC_SITES; ++s) p.ds.scale[RIFT_DS_SITES + s] = dpk;
  }
#endif
}


int decw_set_attributes() {
  int e = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_w_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, DECW_LDS_BYTES);
  if (!e) e = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_w_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, DECW_LDS_BYTES);
  if (!e) e = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_w_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, DECW_LDS_DENSE_BYTES);
  if (!e) e = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_w_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, DECW_LDS_DENSE_BYTES);
  if (!e) e = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_w_kernel<false, true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, DECW_LDS_DENSE_BYTES);
  if (!e) e = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_w_kernel<true, true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, DECW_LDS_DENSE_BYTES);
  if (!e) e = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_w_kernel<false, false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, DECW_LDS_MID_BYTES);
  if (!e) e = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_w_kernel<true, false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, DECW_LDS_MID_BYTES);
  return e;
}
void decw_pack(const DecWSrc& src, unsigned short* img, float* par, hipStream_t stream) {
  const int n = 4 * DECW_LAYER_FRAGS * 512;
  hipLaunchKernelGGL(pack_decw_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, src, img, par);
}
void decw_launch(const DecWP& p, hipStream_t stream) {
  const bool dense = p.R > 8 || p.N > 96;        // the dense-traffic variant: rounds of eight tiles, hand-over through the query array
  if (decw_mid_shape(p.R, p.N) && !(p.dbg & 32)) {      // eight key tiles on the standard kernel (dbg 32: the dense variant instead)
    if (p.dropout > 0.f) hipLaunchKernelGGL((dec_w_kernel<true, false, 8>), dim3(p.bs), dim3(512), (size_t)DECW_LDS_MID_BYTES, stream, p);
    else hipLaunchKernelGGL((dec_w_kernel<false, false, 8>), dim3(p.bs), dim3(512), (size_t)DECW_LDS_MID_BYTES, stream, p);
  } else if (dense && p.N <= 128) {
    if (p.dropout > 0.f) hipLaunchKernelGGL((dec_w_kernel<true, true, 8>), dim3(p.bs), dim3(512), (size_t)DECW_LDS_DENSE_BYTES, stream, p);
    else hipLaunchKernelGGL((dec_w_kernel<false, true, 8>), dim3(p.bs), dim3(512), (size_t)DECW_LDS_DENSE_BYTES, stream, p);
  } else if (dense) {
    if (p.dropout > 0.f) hipLaunchKernelGGL((dec_w_kernel<true, true>), dim3(p.bs), dim3(512), (size_t)DECW_LDS_DENSE_BYTES, stream, p);
    else hipLaunchKernelGGL((dec_w_kernel<false, true>), dim3(p.bs), dim3(512), (size_t)DECW_LDS_DENSE_BYTES, stream, p);
  } else {
    if (p.dropout > 0.f) hipLaunchKernelGGL((dec_w_kernel<true, false>), dim3(p.bs), dim3(512), (size_t)DECW_LDS_BYTES, stream, p);
    else hipLaunchKernelGGL((dec_w_kernel<false, false>), dim3(p.bs), dim3(512), (size_t)DECW_LDS_BYTES, stream, p);
  }
}

}  // namespace RIFT_NS
