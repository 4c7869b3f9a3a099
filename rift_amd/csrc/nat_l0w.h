// Level 0 of the NAT-FPN history encoder (embedding.py:57-99,196-202: ConvTokenizer -> 2 NATLayers (dim 32, 2 heads, kernel 3) ->
// {LayerNorm of the last 3 steps for the FPN, downsample conv + LayerNorm -> level 1}) as WAVE-PRIVATE, REGISTER-RESIDENT tiles.
//
// Why a second design for this level.  nat_level_kernel keeps an 80-row tile in LDS and splits every GEMM's n-tiles over the 8 waves
// of a workgroup, with a workgroup barrier between phases.  At C = 32 a phase is 2..6 MFMA n-tiles and 80 x 32 values of VALU work:
// most waves idle, and the level's ~30 phases per tile are pure barrier latency (137 us for 13.7 GFLOP).  Here one WAVE owns a tile of
// 4 agents x 20 steps = 80 rows for the whole level and nothing is ever exchanged between waves:
//   * row (agent a, step t) lives in MFMA row tile mt = t / 4, lane row l15 = 4 a + t % 4 -- the 4 lanes of a DPP quad are 4
//     consecutive steps of one agent, so the kernel-3 neighbourhood attention reads its neighbours with quad_perm DPP moves (plus the
//     quad edge from the adjacent row tile's register);
//   * the fp32 residual stream is the MFMA C/D layout itself: x[mt][nt] holds, per lane, 4 consecutive channels of one row
//     (40 VGPRs for 80 x 32).  A GEMM output in that layout IS the next GEMM's B operand: the contraction order is free, so the
//     weights are packed with the K order "lane l4 holds channels {4 l4 .. +3} of n-tile 0 and of n-tile 1" -- no LDS round trip,
//     no transposition between LayerNorm -> qkv, attention -> proj, LayerNorm -> fc1 -> GELU -> fc2;
//   * q, k, v stay fp32 (VALU attention); only MFMA operands are bf16;
//   * weights (54 fragments of 1 KiB) and parameters sit in LDS, read as conflict-free 16-byte-per-lane fragment loads; the only other
//     LDS use is a per-wave bf16 staging tile for the stride-2 downsample conv (rows change lanes there).
// No workgroup barrier after the prologue.
#pragma once
#include "common.h"
#include "dropstats.h"
#include "wp_stream.h"

namespace RIFT_NS {

#define L0W_NFRAG 54
#define L0W_F_TOK 0                       // 2 n-tiles, K = 27 window values (tap-major: tap * 9 + cin)
#define L0W_F_BLK(b) (2 + 20 * (b))      // + 0..5 qkv (q_h0 q_h1 k_h0 k_h1 v_h0 v_h1; q pre-scaled) | + 6..7 proj | + 8..13 fc1 | + 14 + 2 ks + nt fc2
#define L0W_F_DS 42                       // + 4 tap + nt: downsample conv, K = 32 channels of one tap (natural order)
// parameter block (fp32): b_tok 32 | per block: ln1_g 32, ln1_b 32, bqkv 96 (q part pre-scaled), rpb 16 (2 x 5 used), bproj 32, ln2_g 32, ln2_b 32,
// b1 96, b2 32 | fn_g 32, fn_b 32 | ds_g 64, ds_b 64
#define L0W_P_BLK(b) (32 + 400 * (b))
#define L0W_PB_LN1G 0
#define L0W_PB_LN1B 32
#define L0W_PB_BQKV 64
#define L0W_PB_RPB 160
#define L0W_PB_BP 176
#define L0W_PB_LN2G 208
#define L0W_PB_LN2B 240
#define L0W_PB_B1 272
#define L0W_PB_B2 368
#define L0W_P_FN 832
#define L0W_P_DS 896
// (RIFT_NAT_MFMA_ATTN) score-accumulator table [block 2][head 2][tile class 3: first / middle / last][row 5: quad lane s of the query, 4 = "another
// agent's keys"][8: prev tile's key 3 | own tile's keys 0..3 | next tile's key 0 | pad 2] = rpb[h][key step - query step + 2] * log2 e for the
// keys inside the query's window (start clamp(t - 1, 0, L - 3)), L0W_NEG for every other key
#define L0W_P_TBL 1024
#define L0W_NEG (-1.0e30f)
#define L0W_NPAR (1024 + 2 * 2 * 3 * 5 * 8)
#define L0W_ST 40                         // staging tile row stride (bf16): 80 B rows, 16-byte aligned fragments
#ifndef L0W_NWV
#define L0W_NWV 8                        // waves per workgroup (= per CU: the LDS image allows one workgroup); 12 -> 168 VGPRs per wave
#endif
#define L0W_LDS (L0W_NFRAG * 1024 + L0W_NPAR * 4 + L0W_NWV * 80 * L0W_ST * 2)

#if RIFT_NAT_MFMA_ATTN
#define L0W_QSCALE (0.25f * 1.4426950408889634f)
#else
#define L0W_QSCALE 0.25f
#endif
// One entry of the score-accumulator table of the MFMA neighbourhood attention (kernel 3) of a level with L steps in row tiles of 4 steps:
// query = quad lane `row` of tile mt (row 4: a lane that holds another agent's keys), slot 0 = key 3 of tile mt - 1, 1..4 = keys 0..3 of
// tile mt, 5 = key 0 of tile mt + 1.  natten's window of step t starts at clamp(t - 1, 0, L - 3); rpb index = key - query + 2.  A query step
// beyond the sequence (level 1's tile 2 has two of them per agent) sees itself only: a row without any key would turn NaN, and a NaN key
// row poisons every query through the score MFMA.
__host__ __device__ __forceinline__ float nat_band_entry(const float* rpb, int L, int mt, int row, int slot, float neg) {
  if (row >= 4 || slot >= 6) return neg;
  const int qt = 4 * mt + row, kt = slot == 0 ? 4 * mt - 1 : slot == 5 ? 4 * mt + 4 : 4 * mt + slot - 1;
  if (qt >= L) return kt == qt ? rpb[2] * 1.4426950408889634f : neg;
  if (kt < 0 || kt >= L) return neg;
  const int w0 = qt - 1 < 0 ? 0 : qt - 1 > L - 3 ? L - 3 : qt - 1;
  return (kt >= w0 && kt <= w0 + 2) ? rpb[kt - qt + 2] * 1.4426950408889634f : neg;
}

struct NatL0WSrc {    // raw fp32 parameters (views onto the state_dict) for pack_l0w_kernel
  const float* w_tok; const float* b_tok;                                   // embed.proj (32, 9, 3), (32)
  struct Blk { const float *ln1_g, *ln1_b, *wqkv, *bqkv, *rpb, *wproj, *bproj, *ln2_g, *ln2_b, *w1, *b1, *w2, *b2; } blk[2];
  const float* fn_g; const float* fn_b;                                     // norm0
  const float* w_ds; const float* ds_g; const float* ds_b;                  // levels.0.downsample.reduction (64, 32, 3), norm (64)
};


#ifdef RIFT_NAT_L01_IMPL
__global__ void pack_l0w_kernel(NatL0WSrc s, unsigned short* __restrict__ img, float* __restrict__ par) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < L0W_NFRAG * 512) {
    const int f = e >> 9, lane = (e >> 3) & 63, j = e & 7, l15 = lane & 15, l4 = lane >> 4;
    float v = 0.f;
    bool hid = false;                              // fc2 fragments: hidden-layer operand words (common.h: f2h_hid)
    if (f < 2) {                                   // tokenizer: window index = tap * 9 + cin
      const int w = l4 * 8 + j, n = f * 16 + l15;
      if (w < 27) v = s.w_tok[(n * 9 + (w % 9)) * 3 + (w / 9)];
    } else if (f < L0W_F_DS) {
      const int b = (f - 2) / 20, g = (f - 2) % 20;
      const NatL0WSrc::Blk& k = s.blk[b];
      const float fg1 = RIFT_LN_FOLD ? k.ln1_g[l0w_chan(l4, j, 0)] : 1.0f, fg2 = RIFT_LN_FOLD ? k.ln2_g[l0w_chan(l4, j, 0)] : 1.0f;      // (opfmt.h: RIFT_LN_FOLD)
      if (g < 6) v = k.wqkv[(g * 16 + l15) * 32 + l0w_chan(l4, j, 0)] * (g < 2 ? L0W_QSCALE : 1.0f) * fg1;  // q scaled by head_dim^-0.5 (x log2 e: MFMA attention)
      else if (g < 8) v = k.wproj[((g - 6) * 16 + l15) * 32 + l0w_chan(l4, j, 0)];
      else if (g < 14) v = k.w1[((g - 8) * 16 + l15) * 32 + l0w_chan(l4, j, 0)] * fg2;
      else { const int ks = (g - 14) >> 1, nt = (g - 14) & 1; v = k.w2[(nt * 16 + l15) * 96 + l0w_chan(l4, j, 2 * ks)]; hid = true; }
    } else {
      const int tap = (f - L0W_F_DS) >> 2, nt = (f - L0W_F_DS) & 3;
      v = s.w_ds[((nt * 16 + l15) * 32 + l4 * 8 + j) * 3 + tap];
    }
    img[e] = hid ? f2h_hid(v) : f2h(v);
  }
  if (e < L0W_NPAR) {
    float v = 0.f;
    if (e < 32) v = s.b_tok[e];
    else if (e < L0W_P_FN) {
      const int b = (e - 32) / 400, o = (e - 32) % 400;
      const NatL0WSrc::Blk& k = s.blk[b];
      if (o < 32) v = k.ln1_g[o];
      else if (o < 64) v = k.ln1_b[o - 32];
      else if (o < 160) {
        v = k.bqkv[o - 64];
        if (RIFT_LN_FOLD) for (int ch = 0; ch < 32; ++ch) v += k.wqkv[(o - 64) * 32 + ch] * k.ln1_b[ch];
        v *= (o - 64 < 32 ? L0W_QSCALE : 1.0f);
      }
      else if (o < 176) v = (o - 160 < 10) ? k.rpb[o - 160] : 0.f;
      else if (o < 208) v = k.bproj[o - 176];
      else if (o < 240) v = k.ln2_g[o - 208];
      else if (o < 272) v = k.ln2_b[o - 240];
      else if (o < 368) {
        v = k.b1[o - 272];
        if (RIFT_LN_FOLD) for (int ch = 0; ch < 32; ++ch) v += k.w1[(o - 272) * 32 + ch] * k.ln2_b[ch];
      }
      else v = k.b2[o - 368];
    } else if (e < L0W_P_DS) v = (e - L0W_P_FN < 32) ? s.fn_g[e - L0W_P_FN] : s.fn_b[e - L0W_P_FN - 32];
    else if (e < L0W_P_TBL) v = (e - L0W_P_DS < 64) ? s.ds_g[e - L0W_P_DS] : s.ds_b[e - L0W_P_DS - 64];
    else {
      const int t = e - L0W_P_TBL, slot = t & 7, row = (t >> 3) % 5, cls = (t / 40) % 3, h = (t / 120) % 2, bi = t / 240;
      v = nat_band_entry(s.blk[bi].rpb + h * 5, 20, cls == 0 ? 0 : cls == 2 ? 4 : 2, row, slot, L0W_NEG);
    }
    par[e] = v;
  }
}
#endif

struct NatL0WP {
  const float* F9; int nseq;               // (nseq * 20, 9) agent features
  const unsigned short* img; const float* par;
  float* Oc;                               // (nseq * 3, 32)  LayerNorm(norm0) of steps 17..19
  unsigned short* Ocb;                     // if set: the same rows as bf16 instead (what fpn_tail_kernel rounds them to anyway: half the bytes both ways)
  float* Xnext;                            // (nseq * 10, 64) downsample conv + LayerNorm
  float droppath[2]; uint32_t seed, stream;
  long long* ts;                           // optional section timestamps of wave 0 of workgroup 0 (diagnostic, RIFT_NAT_TS=1)
  // compacted launch (aidx, cnt from nat_rank_kernel below): the level runs on the sequences the history encoder's output is read of
  // (agent_encoder.py:77-80 runs it on agent_feature[valid_agent_mask]; the ego row is replaced at :87), ranked per residue class of the
  // agent slot (common.h: SeqCount) -- rank r is sequence r of this and the next levels' buffers and reads the features of slot aidx[r]
  const int* aidx; const int* cnt;
  DropStats ds;                            // diagnostic build only (dropstats.h)
};

int l0w_set_attributes();
void l0w_pack(const NatL0WSrc& src, unsigned short* img, float* par, hipStream_t stream);
void l0w_launch(const NatL0WP& p, int grid, hipStream_t stream);

#ifdef RIFT_NAT_L01_IMPL      // the kernels live in nat_l01w.hip (their own translation unit, built like nat_l2w.hip: -fno-honor-nans -mno-amdgpu-ieee)
template <int CTRL>
__device__ __forceinline__ f32x4 l0w_dpp4(const f32x4 v) {
  return (f32x4){dpp_f<CTRL>(v[0]), dpp_f<CTRL>(v[1]), dpp_f<CTRL>(v[2]), dpp_f<CTRL>(v[3])};
}
__device__ __forceinline__ f32x4 l0w_sel(bool c, const f32x4 a, const f32x4 b) {
  return (f32x4){c ? a[0] : b[0], c ? a[1] : b[1], c ? a[2] : b[2], c ? a[3] : b[3]};
}

// LayerNorm over the 32 channels of every row of x (8 per lane, 4 lanes per row) -> bf16 B operands (two-pass statistics, as torch)
// FOLDED (opfmt.h: RIFT_LN_FOLD): gamma / beta live in the consuming layer's weights; the final / downsample norms keep the affine form
template <bool FOLDED = false>
__device__ __forceinline__ void l0w_layer_norm(const f32x4 (&x)[5][2], h16x8 (&xn)[5], const float* g, const float* b, int l4) {
  if (FOLDED) {
    // statistics of all five row tiles first, ONE cancellation test for the call (common.h: ln_cancels -- a branch per tile kept the
    // scheduler from overlapping the tiles' reductions), then the normalisation
    float mean[5], var[5];
    bool bad = false;
#pragma unroll
    for (int mt = 0; mt < 5; ++mt) {
      const f32x4 u = x[mt][0], w = x[mt][1];
      const f32x4 s4 = u + w, q4 = u * u + w * w;
      mean[mt] = rows_sum((s4[0] + s4[1]) + (s4[2] + s4[3])) * (1.0f / 32.0f);
      const float ex2 = rows_sum((q4[0] + q4[1]) + (q4[2] + q4[3])) * (1.0f / 32.0f);
      const float m2 = mean[mt] * mean[mt];
      var[mt] = ex2 - m2;
      bad |= ln_row_cancels(m2, var[mt]);
    }
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(bad) != 0ull, 0)) {      // a row whose mean dwarfs its spread: the centred form, as torch
#pragma unroll
      for (int mt = 0; mt < 5; ++mt) {
        const f32x4 du = x[mt][0] - mean[mt], dw = x[mt][1] - mean[mt], d4 = du * du + dw * dw;
        var[mt] = rows_sum((d4[0] + d4[1]) + (d4[2] + d4[3])) * (1.0f / 32.0f);
      }
    }
#pragma unroll
    for (int mt = 0; mt < 5; ++mt) {
      const float r = rsqrtf(fmaxf(var[mt], 0.f) + 1e-5f), c = -mean[mt] * r;
      xn[mt] = l0w_pack8(x[mt][0] * r + c, x[mt][1] * r + c);
    }
    return;
  }
  const float4 g0 = *reinterpret_cast<const float4*>(g + l4 * 4), g1 = *reinterpret_cast<const float4*>(g + 16 + l4 * 4);
  const float4 b0 = *reinterpret_cast<const float4*>(b + l4 * 4), b1 = *reinterpret_cast<const float4*>(b + 16 + l4 * 4);
#pragma unroll
  for (int mt = 0; mt < 5; ++mt) {
    const f32x4 u = x[mt][0], w = x[mt][1];
    const float mean = rows_sum(((u[0] + u[1]) + (u[2] + u[3])) + ((w[0] + w[1]) + (w[2] + w[3]))) * (1.0f / 32.0f);
    const f32x4 du = u - mean, dw = w - mean;
    const float var = rows_sum(((du[0] * du[0] + du[1] * du[1]) + (du[2] * du[2] + du[3] * du[3])) +
                               ((dw[0] * dw[0] + dw[1] * dw[1]) + (dw[2] * dw[2] + dw[3] * dw[3]))) * (1.0f / 32.0f);
    const float r = rsqrtf(var + 1e-5f);
    const f32x4 yu = {du[0] * r * g0.x + b0.x, du[1] * r * g0.y + b0.y, du[2] * r * g0.z + b0.z, du[3] * r * g0.w + b0.w};
    const f32x4 yw = {dw[0] * r * g1.x + b1.x, dw[1] * r * g1.y + b1.y, dw[2] * r * g1.z + b1.z, dw[3] * r * g1.w + b1.w};
    xn[mt] = l0w_pack8(yu, yw);
  }
}

__global__ __launch_bounds__(64 * L0W_NWV) void nat_l0w_kernel(NatL0WP p) {
  constexpr int NTHR = 64 * L0W_NWV;
  constexpr int L = 20;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* wl = reinterpret_cast<unsigned short*>(smem_raw);            // [54][64][8] weight fragments
  float* par = reinterpret_cast<float*>(wl + L0W_NFRAG * 512);
  unsigned short* stg = reinterpret_cast<unsigned short*>(par + L0W_NPAR);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  // the weight image by LDS-DMA in runs of four fragments (wp_stream.h; lands under the parameter copy below, waited for at the barrier)
  decw_dma_share(reinterpret_cast<const unsigned char*>(p.img), (uint32_t)lane * 16u, __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem_raw), L0W_NFRAG,
                 __builtin_amdgcn_readfirstlane(wave), L0W_NWV);
  for (int i = tid; i < L0W_NPAR / 4; i += NTHR) reinterpret_cast<float4*>(par)[i] = reinterpret_cast<const float4*>(p.par)[i];
  RIFT_SEQ_COUNT(p.cnt, p.nseq);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  unsigned short* st = stg + wave * 80 * L0W_ST;
  auto W = [&](int f) { return *reinterpret_cast<const h16x8*>(wl + ((size_t)f * 64 + lane) * 8); };
  const f32x4 Z = {0.f, 0.f, 0.f, 0.f};
  const int a = l15 >> 2, s = l15 & 3;
  const int ntiles = (sq_n + 3) >> 2;
  int tsn = 0;
#define L0TS() do { if (p.ts && blockIdx.x == 0 && tid == 0 && tsn < 60) p.ts[tsn++] = clock64(); } while (0)
  L0TS();

  for (int tile = blockIdx.x * L0W_NWV + wave; tile < ntiles; tile += gridDim.x * L0W_NWV) {
    const int seq = tile * 4 + a;
    const bool seq_ok = RIFT_SEQ_LIVE(seq);
    f32x4 x[5][2];
    L0TS();
    // ---- ConvTokenizer: one K = 32 MFMA step over the 3 x 9 window, read straight from the feature rows (27 contiguous floats)
    {
      const h16x8 w0 = W(L0W_F_TOK), w1 = W(L0W_F_TOK + 1);
      const float4 b0 = *reinterpret_cast<const float4*>(par + l4 * 4), b1 = *reinterpret_cast<const float4*>(par + 16 + l4 * 4);
      // the window of step t = feature rows t-1, t, t+1 = 27 contiguous floats; lane l4 takes values 8 l4 .. +7 with two UNCONDITIONAL
      // 16-byte loads (4-byte aligned; the engine pads the feature buffer in front) and masks what lies outside the sequence afterwards
      typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
      const int sq = !seq_ok ? 0 : p.aidx ? p.aidx[seq] : seq;      // the agent slot whose features this row reads
      unsigned m_all = 0u, m_tap0 = 0u, m_tap2 = 0u;          // bit j: window value 8 l4 + j exists / belongs to tap 0 / to tap 2
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int w = l4 * 8 + j;
        m_all |= (w < 27 ? 1u : 0u) << j; m_tap0 |= (w < 9 ? 1u : 0u) << j; m_tap2 |= (w >= 18 ? 1u : 0u) << j;
      }
      if (!seq_ok) m_all = 0u;
#pragma unroll
      for (int mt = 0; mt < 5; ++mt) {
        const int t = mt * 4 + s;
        const float* src = p.F9 + ((long long)sq * L + t - 1) * 9 + l4 * 8;
        const f4u lo = *reinterpret_cast<const f4u*>(src), hi = *reinterpret_cast<const f4u*>(src + 4);
        unsigned m = m_all;
        if (mt == 0) m &= (s == 0) ? ~m_tap0 : ~0u;           // t = 0: no step -1
        if (mt == 4) m &= (s == 3) ? ~m_tap2 : ~0u;           // t = 19: no step 20
        const f32x4 va = {(m & 1u) ? lo[0] : 0.f, (m & 2u) ? lo[1] : 0.f, (m & 4u) ? lo[2] : 0.f, (m & 8u) ? lo[3] : 0.f};
        const f32x4 vb = {(m & 16u) ? hi[0] : 0.f, (m & 32u) ? hi[1] : 0.f, (m & 64u) ? hi[2] : 0.f, (m & 128u) ? hi[3] : 0.f};
        const h16x8 bop = l0w_pack8(va, vb);
        x[mt][0] = mfma_h(w0, bop, Z, 0, 0, 0);
        x[mt][1] = mfma_h(w1, bop, Z, 0, 0, 0);
        x[mt][0] += (f32x4){b0.x, b0.y, b0.z, b0.w};
        x[mt][1] += (f32x4){b1.x, b1.y, b1.z, b1.w};
      }
    }
    L0TS();
#pragma unroll 1
    for (int bi = 0; bi < 2; ++bi) {
      const float* pb = par + L0W_P_BLK(bi);
      const int fb = L0W_F_BLK(bi);
      h16x8 xn[5];
      // ================= attention half =================
      l0w_layer_norm<RIFT_LN_FOLD != 0>(x, xn, pb + L0W_PB_LN1G, pb + L0W_PB_LN1B, l4);
      // one head at a time (rolled loop: the live set is the residual, the LayerNorm operands and ONE head's k / v): its proj contribution
      // goes straight into the residual -- proj(concat(o_0, o_1)) = W[:, head 0] o_0 + W[:, head 1] o_1, each as a K = 32 step whose
      // other half is zero
      float dps = 1.f;
      if (p.droppath[bi] > 0.f) dps = (uniform01(p.seed, p.stream + 2 * bi, (uint32_t)seq) < p.droppath[bi]) ? 0.f : 1.0f / (1.0f - p.droppath[bi]);
      if (p.droppath[bi] > 0.f) ds_sample(p.ds, RIFT_DS_NAT(0, bi, 0), seq_ok ? seq : -1, dps);
      const h16x8 wp0 = W(fb + 6), wp1 = W(fb + 7);
#if RIFT_NAT_MFMA_ATTN
      // 1-D neighbourhood attention (kernel 3, window start clamp(t - 1, 0, L - 3)) on the matrix pipe, K = 16 MFMAs (a head is 16 dims: a
      // projection's C/D fragment IS the operand).  A lane row is (agent a, step 4 mt + s): the score tile S^T = K Q^T of row tile mt against
      // key tile mt' holds, in lane (query l15, l4), the four keys of AGENT l4 in tile mt' -- so every key a query may see sits in the ONE
      // lane l4 = a: its own tile's four, key 3 of the tile before, key 0 of the tile behind.  The band mask and rpb are the score
      // accumulator's initial value (a table row picked by (l4 == a ? s : "other agent"); L0W_NEG, not -inf: the other agents' lanes must come
      // out as P = 0, not NaN -- the row maximum is floored at -1e20 and 1 / sum is taken of max(sum, tiny)), so the softmax has no cross-lane
      // step at all; O^T = V^T P^T takes V^T out of a plain-order MFMA (level 2's scheme); DropPath rides on 1 / sum and proj (the head's 16
      // input channels = one half of the K = 32 weight fragment) accumulates straight into the residual.  ~40 VALU instructions per head and row
      // tile against ~150 of the DPP form below.
      const int trow = (l4 == a) ? s : 4;
#pragma unroll 1
      for (int h = 0; h < 2; ++h) {
        h16x4 kop[5], vt[5];
        const h16x8 wq = W(fb + h);
        {
          const h16x8 wk = W(fb + 2 + h), wv = W(fb + 4 + h);
          const float4 bk = *reinterpret_cast<const float4*>(pb + L0W_PB_BQKV + 32 + h * 16 + l4 * 4);
          const float bv = pb[L0W_PB_BQKV + 64 + h * 16 + l15];
#pragma unroll
          for (int mt = 0; mt < 5; ++mt) {
            const f32x4 kk = mfma_h(wk, xn[mt], (f32x4){bk.x, bk.y, bk.z, bk.w}, 0, 0, 0);      // 4 head dims of row l15: the A operand of S^T
            kop[mt] = pack_h16x4(kk[0], kk[1], kk[2], kk[3]);
            const f32x4 vv = mfma_h(xn[mt], wv, (f32x4){bv, bv, bv, bv}, 0, 0, 0);             // plain order: the 4 keys of agent l4 in this tile, head dim l15 = a V^T fragment
            vt[mt] = pack_h16x4(vv[0], vv[1], vv[2], vv[3]);
          }
        }
        const float4 bq = *reinterpret_cast<const float4*>(pb + L0W_PB_BQKV + h * 16 + l4 * 4);
        const float* tb = par + L0W_P_TBL + ((bi * 2 + h) * 15 + trow) * 8;
        const h16x4 wpa = h == 0 ? h16x4_lo(wp0) : h16x4_hi(wp0), wpb = h == 0 ? h16x4_lo(wp1) : h16x4_hi(wp1);
#pragma unroll
        for (int mt = 0; mt < 5; ++mt) {
          const int cls = mt == 0 ? 0 : mt == 4 ? 2 : 1;
          const float4 ca = *reinterpret_cast<const float4*>(tb + cls * 40), cb = *reinterpret_cast<const float4*>(tb + cls * 40 + 4);
          const f32x4 qq = mfma_h(wq, xn[mt], (f32x4){bq.x, bq.y, bq.z, bq.w}, 0, 0, 0);
          const h16x4 qop = pack_h16x4(qq[0], qq[1], qq[2], qq[3]);
          const f32x4 so = mfma_h16(kop[mt], qop, (f32x4){ca.y, ca.z, ca.w, cb.x});
          float sp = L0W_NEG, sn = L0W_NEG;
          if (mt > 0) sp = mfma_h16(kop[mt - 1], qop, (f32x4){L0W_NEG, L0W_NEG, L0W_NEG, ca.x})[3];
          if (mt < 4) sn = mfma_h16(kop[mt + 1], qop, (f32x4){cb.y, L0W_NEG, L0W_NEG, L0W_NEG})[0];
          const float m = fmaxf(fmaxf(fmaxf(so[0], so[1]), fmaxf(so[2], so[3])), fmaxf(fmaxf(sp, sn), -1.0e20f));
          const float e0 = __builtin_amdgcn_exp2f(so[0] - m), e1 = __builtin_amdgcn_exp2f(so[1] - m), e2 = __builtin_amdgcn_exp2f(so[2] - m), e3 = __builtin_amdgcn_exp2f(so[3] - m);
          const float ep = mt > 0 ? __builtin_amdgcn_exp2f(sp - m) : 0.f, en = mt < 4 ? __builtin_amdgcn_exp2f(sn - m) : 0.f;
          const float inv = __builtin_amdgcn_rcpf(fmaxf(((e0 + e1) + (e2 + e3)) + (ep + en), 1.0e-30f)) * dps;
          f32x4 o = mfma_h16(vt[mt], pack_h16x4(e0 * inv, e1 * inv, e2 * inv, e3 * inv), Z);
          if (mt > 0) o = mfma_h16(vt[mt - 1], pack_h16x4(0.f, 0.f, 0.f, ep * inv), o);
          if (mt < 4) o = mfma_h16(vt[mt + 1], pack_h16x4(en * inv, 0.f, 0.f, 0.f), o);
          const h16x4 ao = pack_h16x4(o[0], o[1], o[2], o[3]);
          x[mt][0] = mfma_h16(wpa, ao, x[mt][0]);
          x[mt][1] = mfma_h16(wpb, ao, x[mt][1]);
        }
      }
#else
#pragma unroll 1
      for (int h = 0; h < 2; ++h) {
        f32x4 k[5], v[5];
        const h16x8 wq = W(fb + h);
        const float4 bq = *reinterpret_cast<const float4*>(pb + L0W_PB_BQKV + h * 16 + l4 * 4);
        {
          const h16x8 wk = W(fb + 2 + h), wv = W(fb + 4 + h);
          const float4 bk = *reinterpret_cast<const float4*>(pb + L0W_PB_BQKV + 32 + h * 16 + l4 * 4);
          const float4 bv = *reinterpret_cast<const float4*>(pb + L0W_PB_BQKV + 64 + h * 16 + l4 * 4);
#pragma unroll
          for (int mt = 0; mt < 5; ++mt) {
            k[mt] = mfma_h(wk, xn[mt], Z, 0, 0, 0) + (f32x4){bk.x, bk.y, bk.z, bk.w};
            v[mt] = mfma_h(wv, xn[mt], Z, 0, 0, 0) + (f32x4){bv.x, bv.y, bv.z, bv.w};
          }
        }
        // 1-D neighbourhood attention, kernel 3, window start clamp(t - 1, 0, L - 3); keys of step t: (t-1, t, t+1), (0, 1, 2) at t = 0,
        // (17, 18, 19) at t = 19.  The quad (4 lanes) holds steps 4 mt .. 4 mt + 3 of one agent: neighbours come through quad_perm DPP,
        // the quad edges from the adjacent row tile's register.  Lane l4 holds 4 of the 16 head dims: partial dots summed over the 4 l4 lanes.
        const float* rp = pb + L0W_PB_RPB + h * 5;
#pragma unroll
        for (int mt = 0; mt < 5; ++mt) {
          const f32x4 qq = mfma_h(wq, xn[mt], Z, 0, 0, 0) + (f32x4){bq.x, bq.y, bq.z, bq.w};
          // neighbours at t - 1 and t + 1 (generic case)
          f32x4 km = l0w_dpp4<0x90>(k[mt]), kp = l0w_dpp4<0xF9>(k[mt]);            // quad_perm [0,0,1,2] / [1,2,3,3]
          f32x4 vm = l0w_dpp4<0x90>(v[mt]), vp = l0w_dpp4<0xF9>(v[mt]);
          if (mt > 0) { km = l0w_sel(s == 0, l0w_dpp4<0xFF>(k[mt - 1]), km); vm = l0w_sel(s == 0, l0w_dpp4<0xFF>(v[mt - 1]), vm); }
          if (mt < 4) { kp = l0w_sel(s == 3, l0w_dpp4<0x00>(k[mt + 1]), kp); vp = l0w_sel(s == 3, l0w_dpp4<0x00>(v[mt + 1]), vp); }
          f32x4 k0 = km, k1 = k[mt], k2 = kp, v0 = vm, v1 = v[mt], v2 = vp;
          int shift = 0;                                                          // rpb index of key j: j + 1 + shift
          if (mt == 0) {        // t = 0 (lane s = 0): keys (0, 1, 2) = (own, +1, +2)
            const f32x4 kpp = l0w_dpp4<0xFE>(k[0]), vpp = l0w_dpp4<0xFE>(v[0]);   // quad_perm [2,3,3,3]
            const bool e = s == 0;
            k0 = l0w_sel(e, k[0], km); k1 = l0w_sel(e, kp, k[0]); k2 = l0w_sel(e, kpp, kp);
            v0 = l0w_sel(e, v[0], vm); v1 = l0w_sel(e, vp, v[0]); v2 = l0w_sel(e, vpp, vp);
            shift = e ? 1 : 0;
          }
          if (mt == 4) {        // t = 19 (lane s = 3): keys (17, 18, 19) = (-2, -1, own)
            const f32x4 kmm = l0w_dpp4<0x40>(k[4]), vmm = l0w_dpp4<0x40>(v[4]);   // quad_perm [0,0,0,1]
            const bool e = s == 3;
            k0 = l0w_sel(e, kmm, km); k1 = l0w_sel(e, km, k[4]); k2 = l0w_sel(e, k[4], kp);
            v0 = l0w_sel(e, vmm, vm); v1 = l0w_sel(e, vm, v[4]); v2 = l0w_sel(e, v[4], vp);
            shift = e ? -1 : 0;
          }
          float s0 = (qq[0] * k0[0] + qq[1] * k0[1]) + (qq[2] * k0[2] + qq[3] * k0[3]);
          float s1 = (qq[0] * k1[0] + qq[1] * k1[1]) + (qq[2] * k1[2] + qq[3] * k1[3]);
          float s2 = (qq[0] * k2[0] + qq[1] * k2[1]) + (qq[2] * k2[2] + qq[3] * k2[3]);
          s0 = rows_sum(s0) + rp[1 + shift]; s1 = rows_sum(s1) + rp[2 + shift]; s2 = rows_sum(s2) + rp[3 + shift];
          const float mx = fmaxf(fmaxf(s0, s1), s2);
          const float e0 = __expf(s0 - mx), e1 = __expf(s1 - mx), e2 = __expf(s2 - mx);
          const float inv = __builtin_amdgcn_rcpf((e0 + e1) + e2);
          const float p0 = e0 * inv, p1 = e1 * inv, p2 = e2 * inv;
          const f32x4 oh = {p0 * v0[0] + p1 * v1[0] + p2 * v2[0], p0 * v0[1] + p1 * v1[1] + p2 * v2[1],
                            p0 * v0[2] + p1 * v1[2] + p2 * v2[2], p0 * v0[3] + p1 * v1[3] + p2 * v2[3]};
          const h16x8 ao = h == 0 ? l0w_pack8(oh, Z) : l0w_pack8(Z, oh);
          x[mt][0] += mfma_h(wp0, ao, Z, 0, 0, 0) * dps;
          x[mt][1] += mfma_h(wp1, ao, Z, 0, 0, 0) * dps;
        }
      }
#endif
      {   // proj bias
        const float4 b0 = *reinterpret_cast<const float4*>(pb + L0W_PB_BP + l4 * 4), b1 = *reinterpret_cast<const float4*>(pb + L0W_PB_BP + 16 + l4 * 4);
#pragma unroll
        for (int mt = 0; mt < 5; ++mt) {
          x[mt][0] += (f32x4){b0.x, b0.y, b0.z, b0.w} * dps; x[mt][1] += (f32x4){b1.x, b1.y, b1.z, b1.w} * dps;
        }
      }
      L0TS();
      // ================= MLP half: fc1 (32 -> 96) -> GELU -> fc2 (96 -> 32), the hidden layer 32 channels (one k-step) at a time =================
      l0w_layer_norm<RIFT_LN_FOLD != 0>(x, xn, pb + L0W_PB_LN2G, pb + L0W_PB_LN2B, l4);
      {
        f32x4 acc2[5][2];
#pragma unroll
        for (int mt = 0; mt < 5; ++mt) { acc2[mt][0] = Z; acc2[mt][1] = Z; }
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
          const h16x8 wa = W(fb + 8 + 2 * ks), wb = W(fb + 9 + 2 * ks), u0 = W(fb + 14 + 2 * ks), u1 = W(fb + 15 + 2 * ks);
          const float4 ba = *reinterpret_cast<const float4*>(pb + L0W_PB_B1 + (2 * ks) * 16 + l4 * 4);
          const float4 bb = *reinterpret_cast<const float4*>(pb + L0W_PB_B1 + (2 * ks + 1) * 16 + l4 * 4);
#pragma unroll
          for (int mt = 0; mt < 5; ++mt) {
            const f32x4 ha = mfma_h(wa, xn[mt], hid_init(ba), 0, 0, 0);      // (packed-fp16 GELU: the bias is the accumulator's initial value)
            const f32x4 hb = mfma_h(wb, xn[mt], hid_init(bb), 0, 0, 0);
            const h16x8 hop = l0w_from_u2(gelu4_hid(ha, ba), gelu4_hid(hb, bb));
            acc2[mt][0] = mfma_hid(u0, hop, acc2[mt][0]);
            acc2[mt][1] = mfma_hid(u1, hop, acc2[mt][1]);
          }
        }
        const float4 b0 = *reinterpret_cast<const float4*>(pb + L0W_PB_B2 + l4 * 4), b1 = *reinterpret_cast<const float4*>(pb + L0W_PB_B2 + 16 + l4 * 4);
        float dps = 1.f;
        if (p.droppath[bi] > 0.f) dps = (uniform01(p.seed, p.stream + 2 * bi + 1, (uint32_t)seq) < p.droppath[bi]) ? 0.f : 1.0f / (1.0f - p.droppath[bi]);
        if (p.droppath[bi] > 0.f) ds_sample(p.ds, RIFT_DS_NAT(0, bi, 1), seq_ok ? seq : -1, dps);
#pragma unroll
        for (int mt = 0; mt < 5; ++mt) {
          x[mt][0] += (acc2[mt][0] + (f32x4){b0.x, b0.y, b0.z, b0.w}) * dps;
          x[mt][1] += (acc2[mt][1] + (f32x4){b1.x, b1.y, b1.z, b1.w}) * dps;
        }
      }
      L0TS();
    }
    L0TS();
    // ---- what the FPN reads of this level: LayerNorm(norm0) of steps 17, 18, 19 (row tile 4, quad lanes 1..3)
    {
      const float* g = par + L0W_P_FN;
      const float4 g0 = *reinterpret_cast<const float4*>(g + l4 * 4), g1 = *reinterpret_cast<const float4*>(g + 16 + l4 * 4);
      const float4 b0 = *reinterpret_cast<const float4*>(g + 32 + l4 * 4), b1 = *reinterpret_cast<const float4*>(g + 48 + l4 * 4);
      const f32x4 u = x[4][0], w = x[4][1];
      const float mean = rows_sum(((u[0] + u[1]) + (u[2] + u[3])) + ((w[0] + w[1]) + (w[2] + w[3]))) * (1.0f / 32.0f);
      const f32x4 du = u - mean, dw = w - mean;
      const float var = rows_sum(((du[0] * du[0] + du[1] * du[1]) + (du[2] * du[2] + du[3] * du[3])) +
                                 ((dw[0] * dw[0] + dw[1] * dw[1]) + (dw[2] * dw[2] + dw[3] * dw[3]))) * (1.0f / 32.0f);
      const float r = rsqrtf(var + 1e-5f);
      if (seq_ok && s >= 1) {
        const float4 o0 = make_float4(du[0] * r * g0.x + b0.x, du[1] * r * g0.y + b0.y, du[2] * r * g0.z + b0.z, du[3] * r * g0.w + b0.w);
        const float4 o1 = make_float4(dw[0] * r * g1.x + b1.x, dw[1] * r * g1.y + b1.y, dw[2] * r * g1.z + b1.z, dw[3] * r * g1.w + b1.w);
        const size_t orow = ((size_t)seq * 3 + (s - 1)) * 32;
        if (p.Ocb) {
          *reinterpret_cast<uint2*>(p.Ocb + orow + l4 * 4) = pack_h4(o0.x, o0.y, o0.z, o0.w);
          *reinterpret_cast<uint2*>(p.Ocb + orow + 16 + l4 * 4) = pack_h4(o1.x, o1.y, o1.z, o1.w);
        } else {
          *reinterpret_cast<float4*>(p.Oc + orow + l4 * 4) = o0;
          *reinterpret_cast<float4*>(p.Oc + orow + 16 + l4 * 4) = o1;
        }
      }
    }
    // ---- next level's input: Conv1d(32 -> 64, k = 3, stride 2, pad 1, no bias) + LayerNorm(64).  Output row (a, t') reads steps
    // 2 t' - 1 .. 2 t' + 1 of its agent: rows change lanes, so the tile goes through a per-wave bf16 staging buffer (natural channel order)
    {
#pragma unroll
      for (int mt = 0; mt < 5; ++mt) {
        const int row = a * L + mt * 4 + s;
        *reinterpret_cast<uint2*>(st + row * L0W_ST + l4 * 4) = pack_h4(x[mt][0][0], x[mt][0][1], x[mt][0][2], x[mt][0][3]);
        *reinterpret_cast<uint2*>(st + row * L0W_ST + 16 + l4 * 4) = pack_h4(x[mt][1][0], x[mt][1][1], x[mt][1][2], x[mt][1][3]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // same-wave LDS hand-off: DS operations of a wave execute in order
      f32x4 d[3][4];
#pragma unroll
      for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) d[mt][nt] = Z;
#pragma unroll
      for (int tap = 0; tap < 3; ++tap) {
        h16x8 wd[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wd[nt] = W(L0W_F_DS + 4 * tap + nt);
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) {
          const int m = mt * 16 + l15, oa = m / 10, t = 2 * (m - oa * 10) - 1 + tap;
          h16x8 bop = (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
          if (m < 40 && t >= 0 && t < L) bop = *reinterpret_cast<const h16x8*>(st + (oa * L + t) * L0W_ST + l4 * 8);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) d[mt][nt] = mfma_h(wd[nt], bop, d[mt][nt], 0, 0, 0);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the staging tile is rewritten by this wave's next tile
      L0TS();
      const float* g = par + L0W_P_DS;
#pragma unroll
      for (int mt = 0; mt < 3; ++mt) {
        float sm = 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) sm += (d[mt][nt][0] + d[mt][nt][1]) + (d[mt][nt][2] + d[mt][nt][3]);
        const float mean = rows_sum(sm) * (1.0f / 64.0f);
        float qs = 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          d[mt][nt] = d[mt][nt] - mean;
          qs += (d[mt][nt][0] * d[mt][nt][0] + d[mt][nt][1] * d[mt][nt][1]) + (d[mt][nt][2] * d[mt][nt][2] + d[mt][nt][3] * d[mt][nt][3]);
        }
        const float r = rsqrtf(rows_sum(qs) * (1.0f / 64.0f) + 1e-5f);
        const int m = mt * 16 + l15, oa = m / 10;
        if (m < 40 && RIFT_SEQ_LIVE(tile * 4 + oa)) {
          float* dst = p.Xnext + ((size_t)tile * 40 + m) * 64;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            const float4 gg = *reinterpret_cast<const float4*>(g + nt * 16 + l4 * 4), bb = *reinterpret_cast<const float4*>(g + 64 + nt * 16 + l4 * 4);
            *reinterpret_cast<float4*>(dst + nt * 16 + l4 * 4) =
                make_float4(d[mt][nt][0] * r * gg.x + bb.x, d[mt][nt][1] * r * gg.y + bb.y, d[mt][nt][2] * r * gg.z + bb.z, d[mt][nt][3] * r * gg.w + bb.w);
          }
        }
      }
    }
  }
}

#endif

// Ranks the marked agent slots for the compacted launch: aidx[3 i + c] = the i-th marked slot of residue class c = slot % 3, cnt[c] = the
// class counts (common.h: SeqCount).  ONE workgroup: thread t counts the marks of slots [t K, (t + 1) K) per class (three
// 21-bit fields of one word), a workgroup-wide exclusive scan, then the ranks in order.  Behind the input preparation on whatever stream
// that runs on (the prepare stream of the update loop: beside the previous step).  (First version: every workgroup of nat_l0w_kernel ranked
// the marks itself in its prologue -- no launch, but 14 us on each of 256 CUs.)
#define NAT_RANK_THREADS 256      // four waves of <= 64 VGPRs: a workgroup that fits beside the decoder's on a CU (224 VGPRs x 2 waves per SIMD leave 64), so
                                  // that the ranking of step k + 1 finishes INSIDE the decoder of step k at a chip-filling batch instead of behind it
// `raw` (optional, with Tfull, A): the (agent, time) validity the marks derive from -- mark = any of the first 21 samples && slot % A != 0, the rule of
// agent_feature_body (kernels.h).  With it the ranking does not wait for the preparation: front_kernel (front.h) runs it as a block of the
// preparation's own launch.
__device__ __forceinline__ void nat_rank_body(const uint8_t* __restrict__ hist, int n, int* __restrict__ aidx, int* __restrict__ cnt, unsigned long long* wsum,
                                              const uint8_t* __restrict__ raw = nullptr, int Tfull = 21, int A = 1) {
  constexpr int NW = NAT_RANK_THREADS / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = (((n + NAT_RANK_THREADS - 1) / NAT_RANK_THREADS) + 15) & ~15, s0 = tid * K;      // slots per thread, a multiple of 16
  // sixteen marks at a time: one 16-byte load when they lie inside the array (hist is 16-byte aligned), byte loads otherwise; with `raw`
  // straight from the (agent, time) validity
  auto marks16 = [&](int j0, uint32_t (&mk)[4]) {
    const int p = s0 + j0;
    if (raw) {
      mk[0] = mk[1] = mk[2] = mk[3] = 0u;
      if (Tfull == 21 && p + 16 <= n) {      // four agents' samples are 84 contiguous bytes (4-byte aligned: p is a multiple of 4): 21 dword loads per quad
#pragma unroll
        for (int g = 0; g < 4; ++g) {          // (a quad at a time: 21 live registers -- this body shares a launch with the preparation and must stay small)
          uint32_t w[21];
          const uint32_t* src = reinterpret_cast<const uint32_t*>(raw + (size_t)(p + 4 * g) * 21);
#pragma unroll
          for (int i = 0; i < 21; ++i) w[i] = src[i];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            uint32_t any = 0;
#pragma unroll
            for (int d = (21 * j) / 4; d <= (21 * j + 20) / 4; ++d) {     // bytes 21 j .. 21 j + 20 of the quad
              const int lo = 21 * j - 4 * d, hi = 21 * j + 20 - 4 * d;   // first / last byte of the agent inside dword d (may lie outside 0..3)
              const uint32_t m = (hi >= 3 ? 0xffffffffu : (0xffffffffu >> (8 * (3 - hi)))) & (lo <= 0 ? 0xffffffffu : (0xffffffffu << (8 * lo)));
              any |= w[d] & m;
            }
            if (any && ((p + 4 * g + j) % A) != 0) mk[g] |= 1u << (8 * j);
          }
        }
        return;
      }
      for (int j = 0; j < 16; ++j) {
        const int a = p + j;
        uint32_t any = 0;
        if (a < n) {
          const uint8_t* v = raw + (size_t)a * Tfull;
          for (int i = 0; i < 21; ++i) any |= v[i];
        }
        if (any && (a % A) != 0) mk[j >> 2] |= 1u << (8 * (j & 3));
      }
      return;
    }
    if (p + 16 <= n) { const uint4 v = *reinterpret_cast<const uint4*>(hist + p); mk[0] = v.x; mk[1] = v.y; mk[2] = v.z; mk[3] = v.w; }
    else {
      mk[0] = mk[1] = mk[2] = mk[3] = 0u;
      for (int j = 0; j < 16; ++j) if (p + j < n && hist[p + j]) mk[j >> 2] |= 1u << (8 * (j & 3));
    }
  };
  unsigned long long mine = 0ull;
  unsigned long long seen0 = 0ull, seen1 = 0ull;        // (raw) the marks of this thread's first 128 slots: the second pass does not scan the validity again
  const bool cached = raw != nullptr && K <= 128;
  int c = s0 % 3;
  for (int j0 = 0; j0 < K; j0 += 16) {
    uint32_t mk[4];
    marks16(j0, mk);
#pragma unroll
    for (int j = 0; j < 16; ++j, c = (c == 2) ? 0 : c + 1) {
      const bool on = ((mk[j >> 2] >> (8 * (j & 3))) & 0xffu) != 0;
      mine += on ? (1ull << (21 * c)) : 0ull;
      if (cached && on) { if (j0 < 64) seen0 |= 1ull << ((j0 + j) & 63); else seen1 |= 1ull << ((j0 + j) & 63); }
    }
  }
  unsigned long long incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const unsigned long long up = __shfl_up(incl, d); if (lane >= d) incl += up; }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  unsigned long long base = 0ull, total = 0ull;
#pragma unroll
  for (int w = 0; w < NW; ++w) { const unsigned long long sw = wsum[w]; base += (w < wave) ? sw : 0ull; total += sw; }
  const unsigned long long off = base + incl - mine;
  int nx0 = (int)(off & 0x1fffffu), nx1 = (int)((off >> 21) & 0x1fffffu), nx2 = (int)((off >> 42) & 0x1fffffu);
  c = s0 % 3;
  for (int j0 = 0; j0 < K; j0 += 16) {
    uint32_t mk[4];
    if (cached) {
      const uint32_t b16 = (uint32_t)(((j0 < 64 ? seen0 : seen1) >> (j0 & 63)) & 0xffffull);
#pragma unroll
      for (int q = 0; q < 4; ++q) mk[q] = ((b16 >> (4 * q)) & 1u) | (((b16 >> (4 * q + 1)) & 1u) << 8) | (((b16 >> (4 * q + 2)) & 1u) << 16) | (((b16 >> (4 * q + 3)) & 1u) << 24);
    } else marks16(j0, mk);
#pragma unroll
    for (int j = 0; j < 16; ++j, c = (c == 2) ? 0 : c + 1) {
      if ((mk[j >> 2] >> (8 * (j & 3))) & 0xffu) {
        const int i = (c == 0) ? nx0 : (c == 1) ? nx1 : nx2;
        nx0 += (c == 0); nx1 += (c == 1); nx2 += (c == 2);
        aidx[3 * i + c] = s0 + j0 + j;
      }
    }
  }
  if (tid == 0) { cnt[0] = (int)(total & 0x1fffffu); cnt[1] = (int)((total >> 21) & 0x1fffffu); cnt[2] = (int)((total >> 42) & 0x1fffffu); }
}
#ifndef RIFT_NAT_L01_IMPL      // (engine.hip owns this one)
__global__ __launch_bounds__(NAT_RANK_THREADS) void nat_rank_kernel(const uint8_t* __restrict__ hist, int n, int* __restrict__ aidx, int* __restrict__ cnt) {
  __shared__ unsigned long long wsum[NAT_RANK_THREADS / 64];
  nat_rank_body(hist, n, aidx, cnt, wsum);
}
#endif

}  // namespace RIFT_NS
