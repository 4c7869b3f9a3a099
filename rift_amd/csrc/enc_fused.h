// Fused scene encoder for gfx950: the 4 TransformerEncoderLayers (transformer.py:73-94) and the final LayerNorm
// (pluto_model.py:152-154) of ONE scene in ONE workgroup, one launch for the whole batch.
// The scene's N = A + Mp + S <= 96 tokens stay resident in LDS (fp32 residual stream, bf16 LayerNorm output,
// a q|k chunk of two heads, V^T of those heads, the attention output); per layer the workgroup runs
// LN -> qkv (2 heads at a time) -> MFMA attention -> out_proj -> +res -> LN -> fc1 -> GELU -> fc2 -> +res with
// bf16 MFMA (fp32 accumulate) and never touches HBM in between.  Weights (0.39 MB bf16 per layer) stream from
// L2 as MFMA B fragments requested one phase ahead.  Replaces ~44 launches and ~1 GB of HBM traffic per step.
#pragma once
#include "common.h"
#include "dropstats.h"
#include "token.h"

namespace RIFT_NS {

struct EncBlockW {
  const float* ln1_g; const float* ln1_b; const float* ln2_g; const float* ln2_b;
  const unsigned short* wqkv;   // bf16 [384][128]: per 2-head chunk rows (q_h0 | k_h0 | q_h1 | k_h1 | v_h0 | v_h1), 32 rows each
  const float* bqkv;            // fp32 [384] same order
  const unsigned short* wo;     // bf16 [128][128]
  const float* bo;
  const unsigned short* w1;     // bf16 [512][128]
  const float* b1;
  const unsigned short* w2;     // bf16 [128][512]
  const float* b2;
  float droppath;
};

struct EncFusedP {
  const float* X;               // (bs*N, 128) tokens + positional embedding
  // (round 4, RIFT_TOKEN_FUSED=1; measured slower, off by default -- engine.hip) or: the token assembly itself -- with tok_on the kernel
  // builds its scene's token rows in its prologue from what token_kernel reads (kernels.h: agent_token_value / polygon_token_value, the
  // same arithmetic); Xout (diagnostic, RIFT_KEEP_TOKENS=1): the rows are also written out
  TokenP tok; int tok_on; float* Xout;
  float* Y;                     // (bs*N, 128) encoder output after the final LayerNorm
  const uint8_t* kpm;           // (bs*N) key padding mask (1 = padded)
  int bs, N;
  EncBlockW blk[4];
  const float* norm_g; const float* norm_b;
  uint32_t seed, stream;
  long long* ts;                // optional phase timestamps of workgroup 0 (diagnostic)
  int* nonfinite;               // device flag: raised when a valid token row leaves the encoder with a NaN / Inf
  // ---- optional tail: the planning decoder's cross-attention K | V projections of this scene's encoder output, all four layers
  // (planning_decoder.py:74-79, nn.MultiheadAttention in_proj rows 128:384), written as bf16 operands of the decoder kernel
  const unsigned short* wkv;    // fragment-major bf16 [4 * 256][128]: per layer (k 128 rows | v 128 rows)
  const float* bkv;             // [4 * 256]
  unsigned short* KT;           // (bs, 4, 48, 512) bf16: per layer the 48 K | V^T MFMA operand fragments of dec_w.h (per head pair: 12 K, 12 V^T;
                                // keys >= N undefined, masked by the decoder)
  const unsigned short* wx0;    // cat_x_proj columns 128:256 (planning_decoder.py:177-179), applied to the scene's ego token (row 0)
  float* x0p;                   // (bs, 128)
  uint8_t* kpm_c;               // (RIFT_ENC_COMPACT) out, (bs, 96): the key padding of the COMPACTED rows (row i padded <=> i >= the scene's valid count) --
                                // what the decoder's cross attention masks with when it reads this kernel's K | V^T fragments
  DropStats ds;                 // diagnostic build only (dropstats.h)
};

#ifndef RIFT_ENC112_TU      // (engine.hip owns this one; enc112.hip includes this header for the 112-row kernel only)
// row-gather weight packer: dst[r][k] = bf16(src[idx[r]][k]); bias_out[r] = bias[idx[r]]
__global__ void pack_rows_indexed_kernel(const float* __restrict__ src, const float* __restrict__ bias,
                                         const int* __restrict__ idx, int nrows, int K,
                                         unsigned short* __restrict__ dst, float* __restrict__ bias_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nrows * K) return;
  const int r = i / K, k = i - r * K;
  dst[fm_index(r, k, K)] = f2h(src[(size_t)idx[r] * K + k]);   // fragment-major image
  if (k == 0 && bias) bias_out[r] = bias[idx[r]];
}
#endif

template <int KS, int NTW>
struct EFrags { h16x8 f[KS][NTW]; };

template <int NWV> struct EWaves {};
template <int V> struct EInt { static constexpr int value = V; };      // a compile-time int as a (generic-lambda) argument
// this wave's n-tiles are (j * NW + wave), j < NTW; tiles at or beyond `ntiles` are skipped (zero fragments)
template <int KS, int NTW, int NW>
__device__ __forceinline__ void e_load_b(EFrags<KS, NTW>& B, const unsigned short* W, int ldw, int n0, int k0, int wave,
                                         int l15, int l4, EWaves<NW>, int ntiles = 1 << 30) {
#pragma unroll
  for (int j = 0; j < NTW; ++j)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      B.f[ks][j] = (j * NW + wave < ntiles) ? fm_load(W, ldw, n0 + (j * NW + wave) * 16, k0 + ks * 32, l4 * 16 + l15)
                                            : (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
}
template <int KS, int NTW>
__device__ __forceinline__ void e_load_b(EFrags<KS, NTW>& B, const unsigned short* W, int ldw, int n0, int k0, int wave,
                                         int l15, int l4) {
  e_load_b(B, W, ldw, n0, k0, wave, l15, l4, EWaves<4>());
}

// acc[mt][j] (+)= X[mt-tile] . W[n-tile j]^T.  Operands are issued "swapped" (weight fragment as the MFMA A operand,
// activation fragment as B), so the C/D layout holds, per lane, FOUR CONSECUTIVE OUTPUT COLUMNS of one row:
//   acc[mt][j][r] = out[row = mt*16 + (lane&15)][col = ntile*16 + 4*(lane>>4) + r]
// -> epilogues are 8/16-byte row-contiguous LDS accesses (2 v_cvt_pk + 1 ds_write_b64 per tile) instead of four
// scattered 2-byte stores.  NSWAP trailing n-tiles (j >= NTW - NSWAP... see call sites) may keep the plain order:
//   plain: acc[r] = out[row = mt*16 + 4*(lane>>4) + r][col = ntile*16 + (lane&15)]   (4 consecutive ROWS: transposed stores)
// HID: A rows and weight fragments are hidden-layer operand words (opfmt.h: fp16 in the packed-fp16-GELU build)
template <int MT, int KS, int NTW, int NPLAIN = 0, bool HID = false>
__device__ __forceinline__ void e_mma(f32x4 (&acc)[MT][NTW], const unsigned short* A, int lda, const EFrags<KS, NTW>& B,
                                      int l15, int l4, int mtn = MT) {      // mtn (workgroup-uniform): row tiles that hold a valid row; the rest are skipped
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    h16x8 a[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) if (mt < mtn) a[mt] = *reinterpret_cast<const h16x8*>(A + (mt * 16 + l15) * lda + ks * 32 + l4 * 8);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      if (mt < mtn) {
#pragma unroll
        for (int j = 0; j < NTW; ++j)
          acc[mt][j] = HID ? mfma_hid(B.f[ks][j], a[mt], acc[mt][j])
                           : (j >= NTW - NPLAIN) ? mfma_h(a[mt], B.f[ks][j], acc[mt][j], 0, 0, 0)
                                                 : mfma_h(B.f[ks][j], a[mt], acc[mt][j], 0, 0, 0);
      }
  }
}

#define RIFT_ENC_NPAR 1664   // per-layer vectors kept in LDS: ln1 g,b | ln2 g,b | bqkv 384 | bo 128 | b1 512 | b2 128
// LDS row strides (bf16 elements).  xn -- the operand of six of the eleven GEMM phases of a layer -- uses 144 = 16 (mod 32): its ds_read_b128
// fragment reads are bank-conflict free (4 LDS cycles; 136 costs 8, tools/lds_conflicts.py).  ao / cb keep 136 / 200: the 160 KB are full.
#define RIFT_ENC_XN 144
#define RIFT_ENC_XA 136
#define RIFT_ENC_LDS_BYTES (96 * 132 * 4 + 96 * (RIFT_ENC_XN + RIFT_ENC_XA) * 2 + 96 * 200 * 2 + 2 * 32 * 104 * 2 + 96 * 4 + RIFT_ENC_NPAR * 4 + 128)      // (+ 96 row -> slot bytes of the compacted layout)
// LDS layout of a scene held in R rows.  R = 96: the layout of rounds 2 - 5 (above).  R = 112 (round 6: the shapes train_cbv collates,
// 49 + 60 = 109 token slots, left the two-pass enc_w_kernel for this kernel): seven row tiles do not fit beside the 96-row strides, so
// the q | k / hidden chunk rows shrink to 136 words (they never held more than 128), the attention output keeps only the CURRENT chunk's
// two heads (72-word rows) and out_proj runs per chunk -- W_o (a_h0 | a_h1 | a_h2 | a_h3) = W_o[:, :64] (a_h0 | a_h1) + W_o[:, 64:] (a_h2 | a_h3),
// each half a K = 64 product accumulated straight into the residual rows -- and V^T rows hold four key pairs: 162 624 of 163 840 bytes.
template <int R> struct EncLay {
  static constexpr int ROWS = R, XS = 132, XN = RIFT_ENC_XN;
  static constexpr int CB = R == 96 ? 200 : 136;
  static constexpr int XA = R == 96 ? RIFT_ENC_XA : 72;
  static constexpr int VS = R == 96 ? 104 : 136;
  static constexpr bool AO_HALF = R != 96;
  static constexpr int BYTES = R * XS * 4 + R * (XN + XA) * 2 + R * CB * 2 + 2 * 32 * VS * 2 + R * 4 + RIFT_ENC_NPAR * 4 + 128;
};
static_assert(EncLay<96>::BYTES == RIFT_ENC_LDS_BYTES, "the 96-row layout is the one of rounds 2 - 5");
static_assert(EncLay<112>::BYTES <= 160 * 1024, "the 112-row layout fits the CU's LDS");

// MTC row (= key) tiles of 16 tokens: 6 = the 96-row layout; 5 = a compacted scene with at most 80 valid tokens (its sixth tile would hold
// padded rows only: every loop over row / key tiles is a sixth shorter).  COMPILE-TIME on purpose: the first version of the compaction tested
// a run-time tile count inside the unrolled loops and ran 275 us instead of 97 (each guarded tile became a basic block of its own, so no
// operand load could move above the MFMAs of the tile before it).  enc_fused_kernel below picks the body per scene.
template <int NW, int MTC, int ROWS_ = 96>
__device__ __forceinline__ void enc_fused_body(const EncFusedP& p, const bool ok, const unsigned long long vmask, const int c0, const int nvv, const bool keep_order) {
  typedef EncLay<ROWS_> Lay;
  constexpr int ROWS = ROWS_, MT = MTC, C = 128;
  constexpr int mtn = MT;
  constexpr int NTH = 64 * NW, NTQ = 12 / NW + (12 % NW ? 1 : 0), NTC = 8 / NW;   // n-tiles per wave: qkv chunk (12), 128-wide output (8)
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
  static_assert(16 * MTC <= ROWS_, "row tiles inside the LDS rows");
  constexpr int XS = Lay::XS, XN = Lay::XN, XA = Lay::XA, CB = Lay::CB, VS = Lay::VS, NKT = MT;
  constexpr bool AO_HALF = Lay::AO_HALF;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* xs = reinterpret_cast<float*>(smem_raw);
  unsigned short* xn = reinterpret_cast<unsigned short*>(xs + ROWS * XS);
  unsigned short* cb = xn + ROWS * XN;
  unsigned short* ao = cb + ROWS * CB;
  unsigned short* vt = ao + ROWS * XA;            // [2][32][VS]
  float* smaskf = reinterpret_cast<float*>(vt + 2 * 32 * VS);   // [ROWS] key-padding mask as 0 / -inf: it enters the scores as the MFMA accumulator
  float* par = smaskf + ROWS;                           // [RIFT_ENC_NPAR] this layer's bias / LayerNorm vectors
  constexpr int P_LN1G = 0, P_LN1B = 128, P_LN2G = 256, P_LN2B = 384, P_BQKV = 512, P_BO = 896, P_B1 = 1024, P_B2 = 1536;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int b = blockIdx.x, N = p.N;
  const size_t grow0 = (size_t)b * N;
  int tsn = 0;
#define TS() do { if (p.ts && b == 0 && tid == 0) p.ts[tsn++] = clock64(); } while (0)
  TS();

  // Per-layer vectors are fetched into registers one layer ahead and dropped into LDS at the top of the layer, so
  // that no epilogue / LayerNorm starts with a dependent global load (each costs ~4k cycles at 1 workgroup per CU).
  constexpr int NPRE = (RIFT_ENC_NPAR + NTH - 1) / NTH;
  float pre[NPRE];
  auto par_fetch = [&](const EncBlockW& w) {
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      const int e = tid + NTH * i;
      const float* src = e < 128 ? w.ln1_g + e : e < 256 ? w.ln1_b + (e - 128) : e < 384 ? w.ln2_g + (e - 256)
                       : e < 512 ? w.ln2_b + (e - 384) : e < 896 ? w.bqkv + (e - 512) : e < 1024 ? w.bo + (e - 896)
                       : e < 1536 ? w.b1 + (e - 1024) : w.b2 + (e - 1536);
      pre[i] = e < RIFT_ENC_NPAR ? *src : 0.f;
    }
  };
  auto par_commit = [&]() {
#pragma unroll
    for (int i = 0; i < NPRE; ++i) { const int e = tid + NTH * i; if (e < RIFT_ENC_NPAR) par[e] = pre[i]; }
  };
  par_fetch(p.blk[0]);
  EFrags<4, NTQ> Bqkv;
  EFrags<4, NTC> Bw;     // out_proj / fc1 chunk (128 output columns, K = 128)
  EFrags<4, NTC> B2;     // fc2 partial (128 output columns, K = 128-wide hidden chunk)
  EFrags<2, NTC> Bo;     // (AO_HALF) out_proj, the K = 64 half of the current chunk's two heads
  e_load_b(Bqkv, p.blk[0].wqkv, C, 0, 0, wave, l15, l4, EWaves<NW>(), 12);

  // ---- (RIFT_ENC_COMPACT) the scene's valid tokens first: srow[i] = the token slot LDS row i holds (valid slots in ascending order, then the
  // padded ones), nv = the valid count, mtn = the row (= key) tiles that hold a valid row: every loop over row tiles below stops there.
  // Slot 0 is the ego token, whose row the tail's cat_x_proj half reads as row 0: a scene whose slot 0 is padded (none in practice: the
  // CBV itself) keeps the slot order (nv = -1).
  unsigned char* srow = reinterpret_cast<unsigned char*>(par + RIFT_ENC_NPAR);        // [ROWS <= 128], lives to the end of the kernel (the output rows go back to their slots)
  int nv = -1;
#if RIFT_ENC_COMPACT
  if (tid < ROWS) {
    const int rk = __builtin_popcountll(vmask & ((1ull << lane) - 1ull)) + (wave == 1 ? c0 : 0);      // valid slots before slot tid
    const int row = keep_order ? tid : ok ? rk : (tid < N ? nvv + (tid - rk) : tid);                 // (rows at or beyond N hold no slot: loaded as zeros)
    srow[row] = (unsigned char)tid;
  }
  lds_barrier();
  if (!keep_order) nv = nvv;
#else
  if (tid < ROWS) srow[tid] = (unsigned char)tid;      // slot order (read behind the barrier below)
#endif
  auto slot_padded = [&](int i) { return (i >= N) || p.kpm[grow0 + (i < N ? i : 0)]; };
  for (int i = tid; i < ROWS * 32; i += NTH) {
    const int r = i >> 5, c4 = (i & 31) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < N) {
#if RIFT_ENC_COMPACT
      const int sl = srow[r];
#else
      const int sl = r;
#endif
      if (p.tok_on) {
        const TokenP& t = p.tok;
        v = sl < t.A ? agent_token_value(b, sl, c4, t.nat, t.x_ego, t.valid_agent, t.category, t.a_type_emb, t.A, t.N, t.pe)
                     : polygon_token_value(b, sl - t.A, c4, t.pooled, t.ptype, t.on_route, t.tl, t.has_sl, t.speed_emb, t.p_type_emb, t.route_emb, t.tl_emb, t.unk_emb,
                                           t.A, t.Mp, t.N, t.pe);
        if (p.Xout) *reinterpret_cast<float4*>(p.Xout + (grow0 + sl) * C + c4) = v;
      } else v = *reinterpret_cast<const float4*>(p.X + (grow0 + sl) * C + c4);
    }
    *reinterpret_cast<float4*>(xs + r * XS + c4) = v;
  }
  // key padding of the LDS rows as 0 / -inf (compacted: row i is padded <=> i >= nv); the same mask, as bytes, is what the decoder reads
  for (int i = tid; i < ROWS; i += NTH) {
    const bool pad = nv >= 0 ? i >= nv : slot_padded(i);
    smaskf[i] = pad ? -INFINITY : 0.f;
    if (p.kpm_c) p.kpm_c[(size_t)b * ROWS + i] = pad;
  }
  lds_barrier();
  TS();

  auto layer_norm = [&](const float* g, const float* be) {   // xs -> xn (bf16); 16 lanes per row (DPP reductions only); g/be in LDS
    const float4 g0 = *reinterpret_cast<const float4*>(g + l15 * 4), g1 = *reinterpret_cast<const float4*>(g + 64 + l15 * 4);
    const float4 b0 = *reinterpret_cast<const float4*>(be + l15 * 4), b1 = *reinterpret_cast<const float4*>(be + 64 + l15 * 4);
#pragma unroll
    for (int r = wave * 4 + l4; r < 16 * mtn; r += 4 * NW) ln128_row16(xs + r * XS, xn + r * XN, g0, g1, b0, b1, l15);
  };

  for (int bi = 0; bi < 4; ++bi) {
    const EncBlockW& w = p.blk[bi];
    float dpscale = 1.f, dpscale2 = 1.f;
    if (w.droppath > 0.f) {
      dpscale = (uniform01(p.seed, p.stream + 2 * bi, (uint32_t)b) < w.droppath) ? 0.f : 1.0f / (1.0f - w.droppath);
      dpscale2 = (uniform01(p.seed, p.stream + 2 * bi + 1, (uint32_t)b) < w.droppath) ? 0.f : 1.0f / (1.0f - w.droppath);
      ds_sample(p.ds, RIFT_DS_ENC(bi, 0), b, dpscale);
      ds_sample(p.ds, RIFT_DS_ENC(bi, 1), b, dpscale2);
    }
    par_commit();
    lds_barrier();
    // ======== self attention ========
    layer_norm(par + P_LN1G, par + P_LN1B);
    lds_barrier();
    TS();
    for (int ch = 0; ch < 2; ++ch) {
      {
        // the chunk's q | k | v GEMM over row tiles [M0, M0 + MN): all of them in one pass, or -- seven tiles (the 112-row layout) -- in
        // two (4 + 3): 56 accumulator and 28 operand registers at once spilled, 32 + 16 do not
        auto qkv_pass = [&](auto m0c, auto mnc) {
          constexpr int M0 = decltype(m0c)::value, MN = decltype(mnc)::value;
          f32x4 acc[MN][NTQ];
#pragma unroll
          for (int mt = 0; mt < MN; ++mt)
#pragma unroll
            for (int j = 0; j < NTQ; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
          e_mma<MN, 4, NTQ, 1>(acc, xn + M0 * 16 * XN, XN, Bqkv, l15, l4, MN);   // q|k tiles swapped (row-major stores), V tile (last j) plain (transposed stores)
          if constexpr (M0 + MN == MT) {      // (behind the last pass: the weights of the next phase)
            if (ch == 0) e_load_b(Bqkv, w.wqkv, C, 192, 0, wave, l15, l4, EWaves<NW>(), 12);
            else if constexpr (!AO_HALF) e_load_b(Bw, w.wo, C, 0, 0, wave, l15, l4, EWaves<NW>());
            if constexpr (AO_HALF) e_load_b(Bo, w.wo, C, 0, ch * 64, wave, l15, l4, EWaves<NW>());      // this chunk's K = 64 half of out_proj
          }
#pragma unroll
          for (int j = 0; j < NTQ - 1; ++j) { // n-tiles 0..7: q|k of the two heads -> cb[row][col..col+3]
            const int nt = j * NW + wave;
            const int col = nt * 16 + l4 * 4;
            const float4 b4 = *reinterpret_cast<const float4*>(par + P_BQKV + ch * 192 + col);
            const float sc = ((nt & 3) < 2) ? 0.17677669529663687f * 1.4426950408889634f : 1.0f;   // q pre-scaled by 32^-0.5 log2 e: the softmax runs on v_exp_f32 = 2^x
#pragma unroll
            for (int mt = 0; mt < MN; ++mt)
              *reinterpret_cast<uint2*>(cb + ((M0 + mt) * 16 + l15) * CB + col) =
                  pack_h4((acc[mt][j][0] + b4.x) * sc, (acc[mt][j][1] + b4.y) * sc, (acc[mt][j][2] + b4.z) * sc, (acc[mt][j][3] + b4.w) * sc);
          }
          if (wave < 4) {                     // n-tiles 8..11: V -> vt[head][d][key..key+3]
            const int nt = 8 + wave;
            const int hh = (nt - 8) >> 1, d = ((nt - 8) & 1) * 16 + l15;
            const float bias = par[P_BQKV + ch * 192 + nt * 16 + l15];
#pragma unroll
            for (int mt = 0; mt < MN; ++mt)
              *reinterpret_cast<uint2*>(vt + (hh * 32 + d) * VS + (M0 + mt) * 16 + l4 * 4) =
                  pack_h4(acc[mt][NTQ - 1][0] + bias, acc[mt][NTQ - 1][1] + bias, acc[mt][NTQ - 1][2] + bias, acc[mt][NTQ - 1][3] + bias);
            // (an odd tile count leaves the second half of the last key PAIR's V^T tile unwritten: zeros there -- P is zero, but 0 x stale NaN is not)
            if ((MT & 1) && M0 + MN == MT) *reinterpret_cast<uint2*>(vt + (hh * 32 + d) * VS + MT * 16 + l4 * 4) = make_uint2(0u, 0u);
          }
        };
        if constexpr (MT > 6) { qkv_pass(EInt<0>(), EInt<4>()); qkv_pass(EInt<4>(), EInt<MT - 4>()); }
        else qkv_pass(EInt<0>(), EInt<MT>());
      }
      lds_barrier();
      TS();
      // ---- MFMA attention: 2 heads x 6 query tiles = 12 (head, tile) pairs, 3 per wave (see mha_mfma_kernel)
      constexpr int nkp = (MT + 1) >> 1;                // key tile PAIRS of the P V product
      for (int pr = wave; pr < 2 * MT; pr += NW) {      // (head, query tile) pairs
        const int hh = pr >= MT ? 1 : 0, qt = pr - hh * MT;
        const h16x8 qf = *reinterpret_cast<const h16x8*>(cb + (qt * 16 + l15) * CB + hh * 64 + l4 * 8);
        f32x4 s[2 * nkp];
        if (MT & 1) s[MT] = (f32x4){0.f, 0.f, 0.f, 0.f};      // (the empty half of the last key pair: P = 0)
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
          if (kt < mtn) {
            const h16x8 kf = *reinterpret_cast<const h16x8*>(cb + (kt * 16 + l15) * CB + hh * 64 + 32 + l4 * 8);
            const float4 mk = *reinterpret_cast<const float4*>(smaskf + kt * 16 + l4 * 4);
            s[kt] = mfma_h(kf, qf, (f32x4){mk.x, mk.y, mk.z, mk.w}, 0, 0, 0);
            m = fmaxf(fmaxf(m, fmaxf(s[kt][0], s[kt][1])), fmaxf(s[kt][2], s[kt][3]));
          } else s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};      // (a key tile without a valid key: P = 0)
        }
        m = rows_max(m);
        float lsum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
          if (kt < mtn) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float e = __builtin_amdgcn_exp2f(s[kt][r] - m); lsum += e; s[kt][r] = e; }
          }
        lsum = rows_sum(lsum);
        f32x4 o0 = (f32x4){0.f, 0.f, 0.f, 0.f}, o1 = o0;
#pragma unroll
        for (int pt = 0; pt < nkp; ++pt) {
          h16x8 pf;
          const unsigned int p0 = pack_h2(s[2 * pt][0], s[2 * pt][1]), p1 = pack_h2(s[2 * pt][2], s[2 * pt][3]);
          const unsigned int p2 = pack_h2(s[2 * pt + 1][0], s[2 * pt + 1][1]), p3 = pack_h2(s[2 * pt + 1][2], s[2 * pt + 1][3]);
          pf[0] = (short)(p0 & 0xffff); pf[1] = (short)(p0 >> 16); pf[2] = (short)(p1 & 0xffff); pf[3] = (short)(p1 >> 16);
          pf[4] = (short)(p2 & 0xffff); pf[5] = (short)(p2 >> 16); pf[6] = (short)(p3 & 0xffff); pf[7] = (short)(p3 >> 16);
          const unsigned short* v0 = vt + (hh * 32 + l15) * VS + pt * 32 + l4 * 4;
          const unsigned short* v1 = vt + (hh * 32 + 16 + l15) * VS + pt * 32 + l4 * 4;
          const uint2 x0 = *reinterpret_cast<const uint2*>(v0), x1 = *reinterpret_cast<const uint2*>(v0 + 16);
          const uint2 y0 = *reinterpret_cast<const uint2*>(v1), y1 = *reinterpret_cast<const uint2*>(v1 + 16);
          h16x8 b0, b1;
          b0[0] = (short)(x0.x & 0xffff); b0[1] = (short)(x0.x >> 16); b0[2] = (short)(x0.y & 0xffff); b0[3] = (short)(x0.y >> 16);
          b0[4] = (short)(x1.x & 0xffff); b0[5] = (short)(x1.x >> 16); b0[6] = (short)(x1.y & 0xffff); b0[7] = (short)(x1.y >> 16);
          b1[0] = (short)(y0.x & 0xffff); b1[1] = (short)(y0.x >> 16); b1[2] = (short)(y0.y & 0xffff); b1[3] = (short)(y0.y >> 16);
          b1[4] = (short)(y1.x & 0xffff); b1[5] = (short)(y1.x >> 16); b1[6] = (short)(y1.y & 0xffff); b1[7] = (short)(y1.y >> 16);
          // O^T = V^T . P^T : lane holds out dims 4*(lane>>4)..+3 (+16) of ITS query (lane&15) -> 8-byte stores, own denominator
          o0 = mfma_h(b0, pf, o0, 0, 0, 0);
          o1 = mfma_h(b1, pf, o1, 0, 0, 0);
        }
        const int head = AO_HALF ? hh : ch * 2 + hh;      // (AO_HALF: the rows hold the current chunk's two heads only)
        const float inv = __builtin_amdgcn_rcpf(lsum);
        unsigned short* op = ao + (qt * 16 + l15) * XA + head * 32 + l4 * 4;
        *reinterpret_cast<uint2*>(op) = pack_h4(o0[0] * inv, o0[1] * inv, o0[2] * inv, o0[3] * inv);
        *reinterpret_cast<uint2*>(op + 16) = pack_h4(o1[0] * inv, o1[1] * inv, o1[2] * inv, o1[3] * inv);
      }
      lds_barrier();
      TS();
      if constexpr (AO_HALF) {
        // ---- this chunk's half of out_proj + residual (the next chunk's attention rewrites ao behind the barrier that follows its q | k | v GEMM)
        f32x4 acc[MT][NTC];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int j = 0; j < NTC; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        e_mma<MT, 2, NTC>(acc, ao, XA, Bo, l15, l4, mtn);
        if (ch == 1) e_load_b(Bw, w.w1, C, 0, 0, wave, l15, l4, EWaves<NW>());             // fc1 weights of hidden chunk 0
#pragma unroll
        for (int j = 0; j < NTC; ++j) {
          const int col = (j * NW + wave) * 16 + l4 * 4;
          float4 b4 = *reinterpret_cast<const float4*>(par + P_BO + col);
          if (ch == 1) b4 = make_float4(0.f, 0.f, 0.f, 0.f);                               // (the bias rides on the first half)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            float4* xp = reinterpret_cast<float4*>(xs + (mt * 16 + l15) * XS + col);
            float4 x = *xp;
            x.x += (acc[mt][j][0] + b4.x) * dpscale; x.y += (acc[mt][j][1] + b4.y) * dpscale;
            x.z += (acc[mt][j][2] + b4.z) * dpscale; x.w += (acc[mt][j][3] + b4.w) * dpscale;
            *xp = x;
          }
        }
      }
    }
    // ---- out_proj + residual
    if constexpr (!AO_HALF) {
      f32x4 acc[MT][NTC];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NTC; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      e_mma<MT, 4, NTC>(acc, ao, XA, Bw, l15, l4, mtn);
      e_load_b(Bw, w.w1, C, 0, 0, wave, l15, l4, EWaves<NW>());             // fc1 weights of hidden chunk 0
#pragma unroll
      for (int j = 0; j < NTC; ++j) {
        const int col = (j * NW + wave) * 16 + l4 * 4;
        const float4 b4 = *reinterpret_cast<const float4*>(par + P_BO + col);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          if (mt >= mtn) break;
          float4* xp = reinterpret_cast<float4*>(xs + (mt * 16 + l15) * XS + col);
          float4 x = *xp;
          x.x += (acc[mt][j][0] + b4.x) * dpscale; x.y += (acc[mt][j][1] + b4.y) * dpscale;
          x.z += (acc[mt][j][2] + b4.z) * dpscale; x.w += (acc[mt][j][3] + b4.w) * dpscale;
          *xp = x;
        }
      }
    }
    lds_barrier();
    TS();
    // ======== MLP ========
    layer_norm(par + P_LN2G, par + P_LN2B);
    if (bi + 1 < 4) par_fetch(p.blk[bi + 1]);
    lds_barrier();
    TS();
    {
      f32x4 acc2[MT][NTC];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NTC; ++j) acc2[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int hc = 0; hc < 4; ++hc) {
        {
          f32x4 acc[MT][NTC];
          float4 b4[NTC];
#pragma unroll
          for (int j = 0; j < NTC; ++j) b4[j] = *reinterpret_cast<const float4*>(par + P_B1 + hc * 128 + (j * NW + wave) * 16 + l4 * 4);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < NTC; ++j) acc[mt][j] = hid_init(b4[j]);      // (packed-fp16 GELU: the bias is the accumulator's initial value)
          e_mma<MT, 4, NTC>(acc, xn, XN, Bw, l15, l4, mtn);
          e_load_b(B2, w.w2, 512, 0, hc * 128, wave, l15, l4, EWaves<NW>());
          if (hc > 0) lds_barrier();   // previous chunk's fc2 reads of cb are complete
#pragma unroll
          for (int j = 0; j < NTC; ++j) {
            const int col = (j * NW + wave) * 16 + l4 * 4;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
              if (mt < mtn)
                *reinterpret_cast<uint2*>(cb + (mt * 16 + l15) * CB + col) =
                    gelu4_hid(acc[mt][j], b4[j]);       // hidden-layer operand words (fc2's weight image matches: w.w2 = the `hid` image)
          }
        }
        if (hc + 1 < 4) e_load_b(Bw, w.w1, C, (hc + 1) * 128, 0, wave, l15, l4, EWaves<NW>());
        else if (bi + 1 < 4) e_load_b(Bqkv, p.blk[bi + 1].wqkv, C, 0, 0, wave, l15, l4, EWaves<NW>(), 12);
        lds_barrier();
        TS();
        e_mma<MT, 4, NTC, 0, true>(acc2, cb, CB, B2, l15, l4, mtn);
        TS();
      }
#pragma unroll
      for (int j = 0; j < NTC; ++j) {
        const int col = (j * NW + wave) * 16 + l4 * 4;
        const float4 b4 = *reinterpret_cast<const float4*>(par + P_B2 + col);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          if (mt >= mtn) break;
          float4* xp = reinterpret_cast<float4*>(xs + (mt * 16 + l15) * XS + col);
          float4 x = *xp;
          x.x += (acc2[mt][j][0] + b4.x) * dpscale2; x.y += (acc2[mt][j][1] + b4.y) * dpscale2;
          x.z += (acc2[mt][j][2] + b4.z) * dpscale2; x.w += (acc2[mt][j][3] + b4.w) * dpscale2;
          *xp = x;
        }
      }
    }
    lds_barrier();
    TS();
  }
  // ---- final LayerNorm (fp32 out) -> global
  {
    const int lr = lane & 31, rsub = lane >> 5;
    const float4 g4 = *reinterpret_cast<const float4*>(p.norm_g + lr * 4), b4 = *reinterpret_cast<const float4*>(p.norm_b + lr * 4);
    uint32_t ex = 0;                                   // non-finite flag by bit pattern (common.h: exp_max)
    for (int r = wave * 2 + rsub; r < ROWS; r += 2 * NW) {
      if (r >= 16 * mtn) {      // a row tile that was skipped: its slots (all padded) get zero rows -- the reference computes something there that nobody reads
        if (r < N) *reinterpret_cast<float4*>(p.Y + (grow0 + srow[r]) * C + lr * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        continue;
      }
      const float4 v = *reinterpret_cast<const float4*>(xs + r * XS + lr * 4);
      float s = (v.x + v.y) + (v.z + v.w);
      s = group_sum<32>(s);
      const float mean = s * (1.0f / C);
      const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
      float q = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      q = group_sum<32>(q);
      const float rstd = rsqrtf(q * (1.0f / C) + 1e-5f);
      const float4 o = make_float4(d0 * rstd * g4.x + b4.x, d1 * rstd * g4.y + b4.y, d2 * rstd * g4.z + b4.z, d3 * rstd * g4.w + b4.w);
      if (r < N) *reinterpret_cast<float4*>(p.Y + (grow0 + srow[r]) * C + lr * 4) = o;
      if (r < N) { ex = exp_max(ex, o.x); ex = exp_max(ex, o.y); ex = exp_max(ex, o.z); ex = exp_max(ex, o.w); }
      if (p.KT) *reinterpret_cast<uint2*>(xn + r * XN + lr * 4) = pack_h4(o.x, o.y, o.z, o.w);
    }
    if (__builtin_amdgcn_ballot_w64(nonfinite_exp(ex)) != 0ull && lane == 0 && p.nonfinite) atomicOr(p.nonfinite, 1);
  }
  if (p.KT && p.x0p) {   // the ego-token half of the decoder's cat_x_proj: one row per scene
    EFrags<4, 1> Wx;
    e_load_b(Wx, p.wx0, C, 0, 0, wave, l15, l4, EWaves<NW>());
    lds_barrier();
    f32x4 ax[1][1];
    ax[0][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    e_mma<1, 4, 1>(ax, xn, XN, Wx, l15, l4);
    if (l15 == 0) *reinterpret_cast<float4*>(p.x0p + (size_t)b * C + wave * 16 + l4 * 4) = make_float4(ax[0][0][0], ax[0][0][1], ax[0][0][2], ax[0][0][3]);
  }
  if (p.KT) {
    // ---- decoder K | V projections of the four layers from the bf16 encoder output still in LDS.  n-tiles 0..7 (K) are issued
    // swapped (a lane holds 4 consecutive channels of one key -> row stores), n-tiles 8..15 (V) plain (4 consecutive keys of one
    // channel -> transposed stores): exactly the two operand layouts the decoder's MFMA cross attention reads.
    EFrags<4, 2> Wk;
    e_load_b(Wk, p.wkv, C, 0, 0, wave, l15, l4, EWaves<NW>());
    lds_barrier();
    static_assert(NW == 8, "the K|V tail deals 16 n-tiles to 8 waves");
    if constexpr (ROWS != 96) {
      // the dense per-head image dec_w_kernel<., false, 8> gathers its eight key tiles from (dec_kv.h: per (scene, layer) 4 heads x
      // (12 K fragments by key tile | 12 V^T fragments by (dim tile, key pair))); key tiles without a valid key: zeros (masked scores
      // of finite operands, P = 0 against finite V^T).  Row tiles in two passes (4 + the rest), as the q | k | v GEMM above.
      for (int l = 0; l < 4; ++l) {
        const int h = wave >> 1;
        unsigned short* base = p.KT + (((size_t)b * 4 + l) * 96 + h * 24) * 512 + lane * 8;
        const float4 b4 = *reinterpret_cast<const float4*>(p.bkv + l * 256 + wave * 16 + l4 * 4);
        const float bias = p.bkv[l * 256 + 128 + wave * 16 + l15];
        auto kv_pass = [&](auto m0c, auto mnc) {
          constexpr int M0 = decltype(m0c)::value, MN = decltype(mnc)::value;
          f32x4 acc[MN][2];
#pragma unroll
          for (int mt = 0; mt < MN; ++mt)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
          e_mma<MN, 4, 2, 1>(acc, xn + M0 * 16 * XN, XN, Wk, l15, l4, MN);
          if constexpr (M0 + MN == MT) { if (l + 1 < 4) e_load_b(Wk, p.wkv, C, (l + 1) * 256, 0, wave, l15, l4, EWaves<NW>()); }
#pragma unroll
          for (int mt = 0; mt < MN; ++mt) {
            const int kt = M0 + mt;
            *reinterpret_cast<uint2*>(base + kt * 512 + (wave & 1) * 4) = pack_h4(acc[mt][0][0] + b4.x, acc[mt][0][1] + b4.y, acc[mt][0][2] + b4.z, acc[mt][0][3] + b4.w);
            *reinterpret_cast<uint2*>(base + (12 + (wave & 1) * 6 + (kt >> 1)) * 512 + (kt & 1) * 4) =
                pack_h4(acc[mt][1][0] + bias, acc[mt][1][1] + bias, acc[mt][1][2] + bias, acc[mt][1][3] + bias);
          }
        };
        if constexpr (MT > 4) { kv_pass(EInt<0>(), EInt<4>()); kv_pass(EInt<4>(), EInt<MT - 4>()); }
        else kv_pass(EInt<0>(), EInt<MT>());
#pragma unroll
        for (int kt = MT; kt < 8; ++kt) {
          *reinterpret_cast<uint2*>(base + kt * 512 + (wave & 1) * 4) = make_uint2(0u, 0u);
          *reinterpret_cast<uint2*>(base + (12 + (wave & 1) * 6 + (kt >> 1)) * 512 + (kt & 1) * 4) = make_uint2(0u, 0u);
        }
      }
    } else
    for (int l = 0; l < 4; ++l) {
      f32x4 acc[MT][2];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      e_mma<MT, 4, 2, 1>(acc, xn, XN, Wk, l15, l4, mtn);
      if (l + 1 < 4) e_load_b(Wk, p.wkv, C, (l + 1) * 256, 0, wave, l15, l4, EWaves<NW>());
      {   // K: channels wave*16 + 4*l4 .. +3 of key mt*16 + l15
        const int col = wave * 16 + l4 * 4;
        const float4 b4 = *reinterpret_cast<const float4*>(p.bkv + l * 256 + col);
        // fragment image: head h = wave >> 1 holds dims 32 h + 16 (j / 4) + 4 l4 + j % 4 in k slot j: this wave supplies j / 4 = wave & 1
        unsigned short* kt = p.KT + (((size_t)b * 4 + l) * 48 + (wave >> 2) * 24 + ((wave >> 1) & 1)) * 512 + lane * 8 + (wave & 1) * 4;
        // (key tiles without a valid key: the decoder walks five tiles whatever the scene holds and the sixth when key 80 is valid -- zeros
        // for the ones skipped here: masked scores of finite operands, and P = 0 against finite V^T)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          *reinterpret_cast<uint2*>(kt + mt * 2 * 512) = pack_h4(acc[mt][0][0] + b4.x, acc[mt][0][1] + b4.y, acc[mt][0][2] + b4.z, acc[mt][0][3] + b4.w);
#pragma unroll
        for (int mt = MT; mt < 6; ++mt) *reinterpret_cast<uint2*>(kt + mt * 2 * 512) = make_uint2(0u, 0u);
      }
      {   // V^T: channel wave*16 + l15, keys mt*16 + 4*l4 .. +3
        const int d = wave * 16 + l15;
        const float bias = p.bkv[l * 256 + 128 + d];
        // fragment image: (head pair, head, dim tile wave & 1, key pair pt = mt >> 1), k slot j <-> key 32 pt + 16 (j / 4) + 4 l4 + j % 4
        unsigned short* vt = p.KT + (((size_t)b * 4 + l) * 48 + (wave >> 2) * 24 + 12 + (((wave >> 1) & 1) * 2 + (wave & 1)) * 3) * 512 + lane * 8;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          *reinterpret_cast<uint2*>(vt + (mt >> 1) * 512 + (mt & 1) * 4) = pack_h4(acc[mt][1][0] + bias, acc[mt][1][1] + bias, acc[mt][1][2] + bias, acc[mt][1][3] + bias);
#pragma unroll
        for (int mt = MT; mt < 6; ++mt) *reinterpret_cast<uint2*>(vt + (mt >> 1) * 512 + (mt & 1) * 4) = make_uint2(0u, 0u);
      }
    }
  }
}

// 97 .. 112 token slots per scene (round 6): the same body on the 112-row layout (EncLay<112>), always compacting -- 5, 6 or 7 row tiles by
// the scene's valid-token count; K | V^T for the decoder's eight-key-tile variant (dense per-head image), kpm_c is (bs, 112).
template <int NW>
__global__ __launch_bounds__(64 * NW) void enc_fused112_kernel(EncFusedP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_cnt[];
  int* cnt = reinterpret_cast<int*>(smem_cnt);
  const int tid = threadIdx.x, wave = tid >> 6, N = p.N;
  const size_t grow0 = (size_t)blockIdx.x * N;
  const bool ok = tid < N && !p.kpm[grow0 + (tid < N ? tid : 0)];
  const unsigned long long vmask = __builtin_amdgcn_ballot_w64(ok);
  if (wave < 2 && (tid & 63) == 0) cnt[wave] = __builtin_popcountll(vmask);
  lds_barrier();
  const int c0 = __builtin_amdgcn_readfirstlane(cnt[0]), nvv = __builtin_amdgcn_readfirstlane(cnt[0] + cnt[1]);
  const bool keep_order = p.kpm[grow0] != 0;
  lds_barrier();
  if (!keep_order && nvv <= 80) enc_fused_body<NW, 5, 112>(p, ok, vmask, c0, nvv, false);
  else if (!keep_order && nvv <= 96) enc_fused_body<NW, 6, 112>(p, ok, vmask, c0, nvv, false);
  else enc_fused_body<NW, 7, 112>(p, ok, vmask, c0, nvv, keep_order);
}

int enc112_set_attributes();                                   // enc112.hip
void enc112_launch(const EncFusedP& p, hipStream_t stream);

template <int NW>
__global__ __launch_bounds__(64 * NW) void enc_fused_kernel(EncFusedP p) {
#if RIFT_ENC_COMPACT
  // the scene's valid-token count first (two ballots and one LDS hand-over), then the body built for its tile count
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_cnt[];
  int* cnt = reinterpret_cast<int*>(smem_cnt);          // (the head of the fp32 row buffer: rewritten by the body's row loads behind its first barrier)
  const int tid = threadIdx.x, wave = tid >> 6, N = p.N;
  const size_t grow0 = (size_t)blockIdx.x * N;
  const bool ok = tid < N && !p.kpm[grow0 + (tid < N ? tid : 0)];
  const unsigned long long vmask = __builtin_amdgcn_ballot_w64(ok);      // waves 0 / 1: slots 0..63 / 64..127
  if (wave < 2 && (tid & 63) == 0) cnt[wave] = __builtin_popcountll(vmask);
  lds_barrier();
  const int c0 = __builtin_amdgcn_readfirstlane(cnt[0]), nvv = __builtin_amdgcn_readfirstlane(cnt[0] + cnt[1]);
  const bool keep_order = p.kpm[grow0] != 0;            // padded ego slot (none in practice): slot order, all six tiles
  lds_barrier();                                        // (everybody has read the counts before the body's row loads overwrite them)
  if (!keep_order && nvv <= 80) enc_fused_body<NW, 5>(p, ok, vmask, c0, nvv, false);
  else enc_fused_body<NW, 6>(p, ok, vmask, c0, nvv, keep_order);
#else
  enc_fused_body<NW, 6>(p, false, 0ull, 0, 0, true);
#endif
}

}  // namespace RIFT_NS
