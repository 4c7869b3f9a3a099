// Fused planning decoder for gfx950: the 4 DecoderLayers (planning_decoder.py:42-86) of ONE scene in ONE
// workgroup.  The scene's R*12 <= 80 mode queries stay resident in LDS (fp32 residual stream + bf16 operand
// buffers) across   r2r self-attention over the R reference lines (with the reference's
// `tgt_key_padding_mask.repeat(M, 1)` mask quirk, :56-60)  ->  m2m self-attention over the 12 modes (+m_pos on
// q/k, padded lines zeroed, :62-72)  ->  cross attention against the scene's encoder tokens (MFMA, :74-79)  ->
// FFN (:81-83).  The cross-attention K | V^T operands of the encoder tokens come as bf16 images written by the encoder kernel's tail
// (or as fp32 rows from the projection GEMM when that kernel is not fused) and are streamed into LDS per 2-head chunk.  All GEMMs are
// bf16 MFMA with swapped operands (row-contiguous epilogues); the r2r / m2m self-attentions are MFMA too (one 16x16 score tile per
// group and head, see self_attention).
#pragma once
#include <type_traits>
#include "enc_fused.h"

namespace rift {

struct DecBlockW {
  const float* par;             // RIFT_DEC_NPAR packed fp32 LayerNorm parameters and biases of this layer (layout below)
  const unsigned short* w_r2r;  // bf16 [384][128], per 2-head chunk rows (q|k|v of head a, q|k|v of head b)
  const unsigned short* w_r2ro; // out_proj [128][128]
  const unsigned short* w_m2m;
  const unsigned short* w_m2mo;
  const unsigned short* w_cq;   // cross_attn in_proj rows 0:128
  const unsigned short* w_co;
  const unsigned short* w_f1;   // ffn.0 [512][128]
  const unsigned short* w_f2;   // ffn.3 [128][512]
  const unsigned short* mpx;    // [2 chunks][8 q|k tiles] 32x16 bf16 weight fragments: rows 0..11 hi(m_pos . Wqk^T), 12..23 lo(...), rest 0
  const float* kv;                                     // (bs*N, kv_ld) fp32: cross-attention K | V projections of the encoder tokens (256 columns of this layer)
};

struct DecFusedP {
  float* Q;                     // (bs*R*12, 128) fp32 decoder queries, updated in place
  const uint8_t* kpm;           // (bs*N) encoder key padding
  const uint8_t* r_kpm;         // (bs*R) reference-line padding
  const uint8_t* q_kpm;         // (q_bs*R) padding rows the r2r quirk indexes: r_kpm, or the gathered masks of the global (data-parallel) minibatch
  int q_bs, q_off;              // scenes of that minibatch and this shard's first scene in it
  int bs, N, R;
  int kv_ld;                    // row stride of the kv matrices (1024 when the four layers share one GEMM)
  const unsigned short* KT;     // optional (bs, 4, 96, 128) bf16 K and (bs, 4, 128, 96) bf16 V^T written by the encoder kernel's tail;
  const unsigned short* VT;     // when set they replace blk[].kv (no fp32 round trip, no conversion, no transposing stores)
  DecBlockW blk[4];
  float dropout;                // 0.1 in train mode
  uint32_t seed, stream;
  long long* ts;                // optional phase timestamps of workgroup 0 (diagnostic)
};

// mpx image of one decoder layer: per (chunk, q|k tile) one 16 x 32 weight fragment in lane order (lane = 16*(k/8) + n, 8 k each);
// k slots 0..11 = bf16 hi part of mp[mode][column], 12..23 = bf16(mp - hi), 24..31 = 0.  mp: (12, 384) fp32 in the ORIGINAL in_proj
// column order, idx: image row -> original row of the chunked in_proj image (q_a k_a q_b k_b v_a v_b per chunk).
__global__ void pack_mpx_kernel(const float* __restrict__ mp, const int* __restrict__ idx, unsigned short* __restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 2 * 8 * 512) return;
  const int i = e & 7, lane = (e >> 3) & 63, tile = e >> 9;          // tile = ch * 8 + nt
  const int ch = tile >> 3, nt = tile & 7;
  const int n = lane & 15, k = (lane >> 4) * 8 + i;
  unsigned short v = 0;
  if (k < 24) {
    const float x = mp[(k % 12) * 384 + idx[ch * 192 + nt * 16 + n]];
    const unsigned short hi = f2bf(x);
    v = k < 12 ? hi : f2bf(x - bf2f(hi));
  }
  out[e] = v;
}

#define RIFT_DEC_NPAR 2944   // ln1..4 g,b (1024) | b_r2r 384 | b_r2ro 128 | b_m2m 384 | b_m2mo 128 | b_cq 128 | b_co 128 | b_f1 512 | b_f2 128
#define RIFT_DEC_NFFB 640    // the FFN biases (tail of the block) stay live to the end of a layer: double-buffered by layer parity
// LDS row strides (elements): bf16 operand tiles use strides = 16 (mod 32), i.e. 8 (mod 16) dwords -- the 16 (row, k-chunk) lanes of every
// ds_read_b128 lane group then fall on 16 distinct 4-bank windows (4 LDS cycles per fragment read; 136 / 200 cost 8: tools/lds_conflicts.py)
#define RIFT_DEC_XN 144
#define RIFT_DEC_CB 208
#define RIFT_DEC_LDS_BYTES (80 * 132 * 4 + 80 * RIFT_DEC_XN * 2 * 2 + 80 * RIFT_DEC_CB * 2 + 96 * 72 * 2 + 64 * 104 * 2 + (RIFT_DEC_NPAR + RIFT_DEC_NFFB) * 4 + 96 + 96 + 16)

// NW waves per workgroup: one scene is one workgroup on one CU, so the wave count is the only occupancy lever
template <int NW, int MTT = 5>
__global__ __launch_bounds__(64 * NW) void dec_fused_kernel(DecFusedP p) {
  constexpr int ROWS = 80, MT = MTT, C = 128, M = 12;   // ROWS: LDS row allocation; MT: 16-row tiles actually processed (R * 12 <= 16 * MT)
  constexpr int RUSE = 16 * MT;
  constexpr int NTH = 64 * NW, NTQ = (12 + NW - 1) / NW, NTC = 8 / NW;   // n-tiles per wave: 192-column chunk, 128-column output
  constexpr int XS = 132, XN = RIFT_DEC_XN, CB = RIFT_DEC_CB, KC = 72, VS = 104, NKT = 6;
  constexpr int P_LN = 0, P_BR2R = 1024, P_BR2RO = 1408, P_BM2M = 1536, P_BM2MO = 1920, P_BCQ = 2048, P_BCO = 2176,
                P_BF1 = 2304, P_BF2 = 2816;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* xs = reinterpret_cast<float*>(smem_raw);
  unsigned short* xn = reinterpret_cast<unsigned short*>(xs + ROWS * XS);
  unsigned short* cb = xn + ROWS * XN;
  unsigned short* ao = cb + ROWS * CB;
  unsigned short* kc = ao + ROWS * XN;             // [96][KC]  cross K of two heads (row-major, 64 dims + pad)
  unsigned short* vtc = kc + 96 * KC;              // [64][VS]  cross V^T of two heads
  float* par = reinterpret_cast<float*>(vtc + 64 * VS);
  unsigned char* smask = reinterpret_cast<unsigned char*>(par + RIFT_DEC_NPAR + RIFT_DEC_NFFB);   // [96] encoder key mask
  unsigned char* qmask = smask + 96;               // [12][8] r2r quirk mask rows
  unsigned char* rz = qmask + 96;                  // [8] padded reference lines of this scene
  const int tid0 = threadIdx.x, wave = tid0 >> 6;
  int tid = tid0, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;   // re-derived through an opaque zero every layer (see the loop)
  const int b = blockIdx.x, N = p.N, R = p.R, NQ = R * M;
  const size_t qrow0 = (size_t)b * NQ;
  const float dp = p.dropout, dpk = dp > 0.f ? 1.0f / (1.0f - dp) : 1.0f;
  const uint32_t thr16 = drop_thr16(dp);
  int tsn = 0;
#define DTS() do { if (p.ts && b == 0 && tid == 0 && tsn < 250) p.ts[tsn++] = clock64(); } while (0)
  DTS();

  // layer parameters: one coalesced read of the host-packed block; everything before the FFN biases is dead once LN4
  // has run, so the next layer's block is fetched right after that barrier and committed behind the first FFN MFMAs
  constexpr int NPRE = (RIFT_DEC_NPAR + NTH - 1) / NTH;
  float pre[NPRE];
  auto par_fetch = [&](const float* src) {
#pragma unroll
    for (int i = 0; i < NPRE; ++i) { const int e = tid + NTH * i; pre[i] = e < RIFT_DEC_NPAR ? src[e] : 0.f; }
  };
  auto par_commit = [&](int parity) {                     // FFN biases of odd layers live RIFT_DEC_NFFB further
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      const int e = tid + NTH * i;
      if (e < RIFT_DEC_NPAR) par[e + ((e >= P_BF1 && parity) ? RIFT_DEC_NFFB : 0)] = pre[i];
    }
  };
  par_fetch(p.blk[0].par);
  EFrags<4, NTQ> Bqkv;       // 192-column qkv chunk
  EFrags<4, NTC> Bw;         // 128-column projections / ffn.0 chunk
  EFrags<4, NTC> B2;         // ffn.3 partial
  e_load_b(Bqkv, p.blk[0].w_r2r, C, 0, 0, wave, l15, l4, EWaves<NW>(), 12);

  for (int i = tid; i < RUSE * 32; i += NTH) {
    const int r = i >> 5, c4 = (i & 31) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < NQ) v = *reinterpret_cast<const float4*>(p.Q + (qrow0 + r) * C + c4);
    *reinterpret_cast<float4*>(xs + r * XS + c4) = v;
  }
  for (int i = tid; i < 96; i += NTH) smask[i] = (i >= N) || p.kpm[(size_t)b * N + i];
  for (int i = tid; i < 96; i += NTH) {            // quirk: mode m of scene b uses the padding row of scene (b*12+m) % bs
    const int m = i >> 3, r = i & 7;
    qmask[i] = (r >= R) || p.q_kpm[(size_t)(((p.q_off + b) * M + m) % p.q_bs) * R + r];
  }
  if (tid < 8) rz[tid] = (tid >= R) || p.r_kpm[(size_t)b * R + tid];
  lds_barrier();

  auto layer_norm = [&](const float* g, const float* be) {   // xs -> xn (bf16); 16 lanes per row; g/be in LDS
    const float4 g0 = *reinterpret_cast<const float4*>(g + l15 * 4), g1 = *reinterpret_cast<const float4*>(g + 64 + l15 * 4);
    const float4 b0 = *reinterpret_cast<const float4*>(be + l15 * 4), b1 = *reinterpret_cast<const float4*>(be + 64 + l15 * 4);
#pragma unroll 1
    for (int r = wave * 4 + l4; r < RUSE; r += 4 * NW) ln128_row16(xs + r * XS, xn + r * XN, g0, g1, b0, b1, l15);
  };

  // x += dropout(acc + bias) for a 128-column projection held as acc[MT][2]; optional row zeroing (m2m)
  auto residual_epilogue = [&](f32x4 (&acc)[MT][NTC], const float* bias, uint32_t stream, bool zero_padded) {
#pragma unroll
    for (int j = 0; j < NTC; ++j) {
      const int col = (j * NW + wave) * 16 + l4 * 4;
      const float4 b4 = *reinterpret_cast<const float4*>(bias + col);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int row = mt * 16 + l15;
        float v[4] = {acc[mt][j][0] + b4.x, acc[mt][j][1] + b4.y, acc[mt][j][2] + b4.z, acc[mt][j][3] + b4.w};
        if (dp > 0.f) dropout4(v, p.seed, stream, (uint32_t)((b * ROWS + row) * C + col), thr16, dpk);
        float4* xp = reinterpret_cast<float4*>(xs + row * XS + col);
        float4 x = *xp;
        x.x += v[0]; x.y += v[1]; x.z += v[2]; x.w += v[3];
        if (zero_padded && rz[(row / M) & 7]) x = make_float4(0.f, 0.f, 0.f, 0.f);
        *xp = x;
      }
    }
  };

  // MFMA self-attention, one 16 x 16 score tile per (group, head).  m2m (over_modes): group = reference line g, queries = keys = its
  // 12 modes (rows g*12 + i).  r2r: group = the mode PAIR (2g, 2g+1): tile slots 0..7 / 8..15 are the R <= 8 reference lines of the
  // two modes (rows r*12 + mode), cross-mode scores masked, keys masked by the quirk row of their mode.  S^T = K . Q^T puts 4 keys of
  // ONE query in each lane (query = lane&15), O^T = V^T . P^T puts 4 output dims of that query in the lane: per-lane denominators,
  // 8-byte stores.  q|k of head hh at cb[row][hh*64 + {0, 32}] (q pre-scaled), V^T of head hh at vtc[hh*32 + d][row].  Each wave
  // carries TWO tiles through the (latency-bound) chain at once.
  auto self_attention = [&](int ch, auto over_modes_t, uint32_t stream) {
    constexpr bool over_modes = decltype(over_modes_t)::value;
    const int ntile = (over_modes ? R : M / 2) * 2;
    const int qslot = over_modes ? l15 : (l15 & 7), qsub = over_modes ? 0 : (l15 >> 3);
    const bool rowok = over_modes ? (l15 < M) : (qslot < R);
    constexpr int U = 1;                  // tiles in flight per wave
    for (int pr0 = wave; pr0 < ntile; pr0 += U * NW) {
      f32x4 s[U];
      unsigned short vv[U][2][4];
      int row[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int pr = pr0 + u * NW;
        if (pr >= ntile) { row[u] = 0; continue; }
        const int hh = pr & 1, g = pr >> 1;
        row[u] = rowok ? (over_modes ? g * M + qslot : qslot * M + 2 * g + qsub) : 0;
        const bf16x8 qf = *reinterpret_cast<const bf16x8*>(cb + row[u] * CB + hh * 64 + l4 * 8);
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(cb + row[u] * CB + hh * 64 + 32 + l4 * 8);
        s[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        // V^T fragments: keys l4*4 .. +3 of dims l15 (+16); zero beyond the valid keys (stale LDS there)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int key = l4 * 4 + i;
          const int kslot = over_modes ? key : (key & 7), ksub = over_modes ? 0 : (key >> 3);
          const bool kok = over_modes ? (key < M) : (kslot < R);
          const int krow = kok ? (over_modes ? g * M + kslot : kslot * M + 2 * g + ksub) : 0;
          const unsigned short a0 = vtc[(hh * 32 + l15) * VS + krow], a1 = vtc[(hh * 32 + 16 + l15) * VS + krow];
          vv[u][0][i] = kok ? a0 : (unsigned short)0; vv[u][1][i] = kok ? a1 : (unsigned short)0;
          const bool masked = !kok || (!over_modes && (ksub != qsub || qmask[(2 * g + ksub) * 8 + kslot]));
          if (masked) s[u][i] = -INFINITY;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int pr = pr0 + u * NW;
        if (pr >= ntile) continue;
        const int hh = pr & 1;
        float m = fmaxf(fmaxf(s[u][0], s[u][1]), fmaxf(s[u][2], s[u][3]));
        m = rows_max(m);
        float ev[4];
        float lsum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { ev[i] = __expf(s[u][i] - m); lsum += ev[i]; }
        lsum = rows_sum(lsum);
        if (dp > 0.f) dropout4(ev, p.seed, stream, (uint32_t)((((b * ROWS + row[u]) * 4 + ch * 2 + hh) * 16) + l4 * 4), thr16, dpk);
        const unsigned int p0 = pack_bf16x2(ev[0], ev[1]), p1 = pack_bf16x2(ev[2], ev[3]);
        bf16x8 pf, b0, b1;
        pf[0] = (short)(p0 & 0xffff); pf[1] = (short)(p0 >> 16); pf[2] = (short)(p1 & 0xffff); pf[3] = (short)(p1 >> 16);
        pf[4] = 0; pf[5] = 0; pf[6] = 0; pf[7] = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) { b0[i] = (short)vv[u][0][i]; b1[i] = (short)vv[u][1][i]; b0[4 + i] = 0; b1[4 + i] = 0; }
        const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
        const f32x4 o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, pf, z4, 0, 0, 0);
        const f32x4 o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, pf, z4, 0, 0, 0);
        if (rowok) {
          const float inv = __builtin_amdgcn_rcpf(lsum);
          unsigned short* op = ao + row[u] * XN + (ch * 2 + hh) * 32 + l4 * 4;
          *reinterpret_cast<uint2*>(op) = pack_bf16x4(o0[0] * inv, o0[1] * inv, o0[2] * inv, o0[3] * inv);
          *reinterpret_cast<uint2*>(op + 16) = pack_bf16x4(o1[0] * inv, o1[1] * inv, o1[2] * inv, o1[3] * inv);
        }
      }
    }
  };

  // qkv chunk GEMM of a self-attention (2 heads): n-tiles 0..7 = q_a q_a k_a k_a q_b q_b k_b k_b -> cb[row][nt*16 ..] (q pre-scaled),
  // n-tiles 8..11 = v_a v_a v_b v_b in the plain MFMA order -> V^T rows.  m2m adds m_pos to q and k (planning_decoder.py:62-64):
  // (LN(x) + m_pos) Wqk^T = LN(x) Wqk^T + onehot(mode) . [hi(m_pos Wqk^T); lo(...)], one extra K = 32 MFMA step per q|k tile
  // whose weight fragment `mpx` is the hi/lo bf16 split of the fp32 product (exact to 2^-17) -- no global reads in the epilogue.
  auto qkv_chunk = [&](int ch, const float* bias, const unsigned short* mpx) {
    static_assert(NW == 8, "tile ownership below assumes 8 waves: one q|k tile each, V tiles on waves 0..3");
    bf16x8 mf = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    if (mpx) mf = *reinterpret_cast<const bf16x8*>(mpx + ((size_t)(ch * 8 + wave) * 64 + lane) * 8);
    f32x4 acc[MT][NTQ];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < NTQ; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    e_mma<MT, 4, NTQ, 1>(acc, xn, XN, Bqkv, l15, l4);
    if (mpx) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m = (mt * 16 + l15) % M;
        uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int t = 0; t < 2; ++t) {                        // k slots m (hi part) and 12 + m (lo part)
          const int i = m + 12 * t - l4 * 8;
          const uint32_t one = ((unsigned)i < 8u) ? (0x3F80u << ((i & 1) * 16)) : 0u;
#pragma unroll
          for (int q = 0; q < 4; ++q) w[q] |= ((i >> 1) == q) ? one : 0u;
        }
        bf16x8 oh;
#pragma unroll
        for (int q = 0; q < 4; ++q) { oh[2 * q] = (short)(w[q] & 0xffff); oh[2 * q + 1] = (short)(w[q] >> 16); }
        acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mf, oh, acc[mt][0], 0, 0, 0);
      }
    }
    {
      const int col = wave * 16 + l4 * 4;
      const float4 b4 = *reinterpret_cast<const float4*>(bias + ch * 192 + col);
      const float sc = ((wave & 3) < 2) ? 0.17677669529663687f : 1.0f;   // q pre-scaled by 32^-0.5
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        *reinterpret_cast<uint2*>(cb + (mt * 16 + l15) * CB + col) =
            pack_bf16x4((acc[mt][0][0] + b4.x) * sc, (acc[mt][0][1] + b4.y) * sc, (acc[mt][0][2] + b4.z) * sc, (acc[mt][0][3] + b4.w) * sc);
    }
    if (wave < 4) {                         // V tile (plain order: 4 consecutive ROWS of dim l15) -> vtc[head][d][row .. row+3]
      const int hh = wave >> 1, d = (wave & 1) * 16 + l15;
      const float bv = bias[ch * 192 + 128 + wave * 16 + l15];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        *reinterpret_cast<uint2*>(vtc + (hh * 32 + d) * VS + mt * 16 + l4 * 4) =
            pack_bf16x4(acc[mt][1][0] + bv, acc[mt][1][1] + bv, acc[mt][1][2] + bv, acc[mt][1][3] + bv);
    }
  };

  for (int li = 0; li < 4; ++li) {
    {   // the per-lane indices pass through an opaque zero once per layer: otherwise every LDS address of the ~30 phases is hoisted out
        // of this loop as a loop invariant, and the kernel (already at 256 VGPRs) spills them around the loop
      int zr;
      asm volatile("v_mov_b32 %0, 0" : "=v"(zr));
      tid = tid0 + zr; lane = tid & 63; l15 = lane & 15; l4 = lane >> 4;
    }
    const DecBlockW& w = p.blk[li];
    const uint32_t st = p.stream + 16 * li;
    const float* bf1 = par + P_BF1 + (li & 1) * RIFT_DEC_NFFB;
    const float* bf2 = par + P_BF2 + (li & 1) * RIFT_DEC_NFFB;
    if (li == 0) par_commit(0);
    lds_barrier(); DTS();
    // ================= r2r =================
    layer_norm(par + P_LN + 0, par + P_LN + 128);
    lds_barrier(); DTS();
    for (int ch = 0; ch < 2; ++ch) {
      qkv_chunk(ch, par + P_BR2R, nullptr);
      if (ch == 0) e_load_b(Bqkv, w.w_r2r, C, 192, 0, wave, l15, l4, EWaves<NW>(), 12);
      else e_load_b(Bw, w.w_r2ro, C, 0, 0, wave, l15, l4, EWaves<NW>());
      lds_barrier(); DTS();
      self_attention(ch, std::false_type{}, st + 0);
      lds_barrier(); DTS();
    }
    {
      f32x4 acc[MT][NTC];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NTC; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      e_mma<MT, 4, NTC>(acc, ao, XN, Bw, l15, l4);
      e_load_b(Bqkv, w.w_m2m, C, 0, 0, wave, l15, l4, EWaves<NW>(), 12);
      residual_epilogue(acc, par + P_BR2RO, st + 1, false);
    }
    lds_barrier(); DTS();
    // ================= m2m =================
    layer_norm(par + P_LN + 256, par + P_LN + 384);
    lds_barrier(); DTS();
    for (int ch = 0; ch < 2; ++ch) {
      qkv_chunk(ch, par + P_BM2M, w.mpx);
      if (ch == 0) e_load_b(Bqkv, w.w_m2m, C, 192, 0, wave, l15, l4, EWaves<NW>(), 12);
      else e_load_b(Bw, w.w_m2mo, C, 0, 0, wave, l15, l4, EWaves<NW>());
      lds_barrier(); DTS();
      self_attention(ch, std::true_type{}, st + 2);
      lds_barrier(); DTS();
    }
    {
      f32x4 acc[MT][NTC];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NTC; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      e_mma<MT, 4, NTC>(acc, ao, XN, Bw, l15, l4);
      e_load_b(Bw, w.w_cq, C, 0, 0, wave, l15, l4, EWaves<NW>());
      residual_epilogue(acc, par + P_BM2MO, st + 3, true);
    }
    lds_barrier(); DTS();
    // ================= cross attention =================
    layer_norm(par + P_LN + 512, par + P_LN + 640);
    lds_barrier(); DTS();
    {
      f32x4 acc[MT][NTC];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NTC; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      e_mma<MT, 4, NTC>(acc, xn, XN, Bw, l15, l4);
      e_load_b(Bw, w.w_co, C, 0, 0, wave, l15, l4, EWaves<NW>());
#pragma unroll
      for (int j = 0; j < NTC; ++j) {
        const int col = (j * NW + wave) * 16 + l4 * 4;
        const float4 b4 = *reinterpret_cast<const float4*>(par + P_BCQ + col);
        const float sc = 0.17677669529663687f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          *reinterpret_cast<uint2*>(cb + (mt * 16 + l15) * CB + col) =
              pack_bf16x4((acc[mt][j][0] + b4.x) * sc, (acc[mt][j][1] + b4.y) * sc, (acc[mt][j][2] + b4.z) * sc, (acc[mt][j][3] + b4.w) * sc);
      }
    }
    for (int ch = 0; ch < 2; ++ch) {
      lds_barrier(); DTS();          // previous chunk's reads of kc / vtc (and the q writes) are complete
      // stream K (row-major) and V (transposed) of heads 2ch, 2ch+1 into LDS as bf16
      if (p.KT) {
        const unsigned short* kt = p.KT + ((size_t)b * 4 + li) * 96 * 128 + ch * 64;
        const unsigned short* vt = p.VT + (((size_t)b * 4 + li) * 128 + ch * 64) * 96;
        for (int i = tid; i < 96 * 8 + 64 * 12; i += NTH) {
          if (i < 96 * 8) {
            const int key = i >> 3, c8 = (i & 7) * 8;
            *reinterpret_cast<uint4*>(kc + key * KC + c8) = *reinterpret_cast<const uint4*>(kt + key * 128 + c8);
          } else {
            const int j = i - 96 * 8, d = j / 12, k8 = (j - d * 12) * 8;
            *reinterpret_cast<uint4*>(vtc + d * VS + k8) = *reinterpret_cast<const uint4*>(vt + d * 96 + k8);
          }
        }
      } else {                             // fp32 K|V rows from the projection GEMM (encoder kernel not fused)
        for (int i = tid; i < 96 * 16; i += NTH) {
          const int key = i >> 4, c4 = (i & 15) * 4;
          float4 kq = make_float4(0.f, 0.f, 0.f, 0.f), vq = kq;
          if (key < N) {
            const float* src = w.kv + ((size_t)b * N + key) * p.kv_ld + ch * 64 + c4;
            kq = *reinterpret_cast<const float4*>(src);
            vq = *reinterpret_cast<const float4*>(src + 128);
          }
          *reinterpret_cast<uint2*>(kc + key * KC + c4) = pack_bf16x4(kq.x, kq.y, kq.z, kq.w);
          vtc[(c4 + 0) * VS + key] = f2bf(vq.x); vtc[(c4 + 1) * VS + key] = f2bf(vq.y);
          vtc[(c4 + 2) * VS + key] = f2bf(vq.z); vtc[(c4 + 3) * VS + key] = f2bf(vq.w);
        }
      }
      lds_barrier(); DTS();          // K / V^T chunk visible to every wave
      for (int pr = wave; pr < 2 * MT; pr += NW) {           // (head, query tile) pairs
        const int hh = pr / MT, qt = pr - hh * MT;
        const bf16x8 qf = *reinterpret_cast<const bf16x8*>(cb + (qt * 16 + l15) * CB + (ch * 2 + hh) * 32 + l4 * 8);
        f32x4 s[NKT];
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kc + (kt * 16 + l15) * KC + hh * 32 + l4 * 8);
          s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (smask[kt * 16 + l4 * 4 + r]) s[kt][r] = -INFINITY;
            m = fmaxf(m, s[kt][r]);
          }
        }
        m = rows_max(m);
        float lsum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
          float ev[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) { ev[r] = __expf(s[kt][r] - m); lsum += ev[r]; }
          if (dp > 0.f)
            dropout4(ev, p.seed, st + 4, (uint32_t)((((b * ROWS + qt * 16 + l15) * 4 + ch * 2 + hh) * 96) + kt * 16 + l4 * 4), thr16, dpk);
#pragma unroll
          for (int r = 0; r < 4; ++r) s[kt][r] = ev[r];
        }
        lsum = rows_sum(lsum);
        f32x4 o0 = (f32x4){0.f, 0.f, 0.f, 0.f}, o1 = o0;
#pragma unroll
        for (int pt = 0; pt < NKT / 2; ++pt) {
          bf16x8 pf;
          const unsigned int p0 = pack_bf16x2(s[2 * pt][0], s[2 * pt][1]), p1 = pack_bf16x2(s[2 * pt][2], s[2 * pt][3]);
          const unsigned int p2 = pack_bf16x2(s[2 * pt + 1][0], s[2 * pt + 1][1]), p3 = pack_bf16x2(s[2 * pt + 1][2], s[2 * pt + 1][3]);
          pf[0] = (short)(p0 & 0xffff); pf[1] = (short)(p0 >> 16); pf[2] = (short)(p1 & 0xffff); pf[3] = (short)(p1 >> 16);
          pf[4] = (short)(p2 & 0xffff); pf[5] = (short)(p2 >> 16); pf[6] = (short)(p3 & 0xffff); pf[7] = (short)(p3 >> 16);
          const unsigned short* v0 = vtc + (hh * 32 + l15) * VS + pt * 32 + l4 * 4;
          const unsigned short* v1 = vtc + (hh * 32 + 16 + l15) * VS + pt * 32 + l4 * 4;
          const uint2 x0 = *reinterpret_cast<const uint2*>(v0), x1 = *reinterpret_cast<const uint2*>(v0 + 16);
          const uint2 y0 = *reinterpret_cast<const uint2*>(v1), y1 = *reinterpret_cast<const uint2*>(v1 + 16);
          bf16x8 b0, b1;
          b0[0] = (short)(x0.x & 0xffff); b0[1] = (short)(x0.x >> 16); b0[2] = (short)(x0.y & 0xffff); b0[3] = (short)(x0.y >> 16);
          b0[4] = (short)(x1.x & 0xffff); b0[5] = (short)(x1.x >> 16); b0[6] = (short)(x1.y & 0xffff); b0[7] = (short)(x1.y >> 16);
          b1[0] = (short)(y0.x & 0xffff); b1[1] = (short)(y0.x >> 16); b1[2] = (short)(y0.y & 0xffff); b1[3] = (short)(y0.y >> 16);
          b1[4] = (short)(y1.x & 0xffff); b1[5] = (short)(y1.x >> 16); b1[6] = (short)(y1.y & 0xffff); b1[7] = (short)(y1.y >> 16);
          o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, pf, o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, pf, o1, 0, 0, 0);
        }
        const float inv = __builtin_amdgcn_rcpf(lsum);
        unsigned short* op = ao + (qt * 16 + l15) * XN + (ch * 2 + hh) * 32 + l4 * 4;
        *reinterpret_cast<uint2*>(op) = pack_bf16x4(o0[0] * inv, o0[1] * inv, o0[2] * inv, o0[3] * inv);
        *reinterpret_cast<uint2*>(op + 16) = pack_bf16x4(o1[0] * inv, o1[1] * inv, o1[2] * inv, o1[3] * inv);
      }
    }
    lds_barrier(); DTS();
    {
      f32x4 acc[MT][NTC];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NTC; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      e_mma<MT, 4, NTC>(acc, ao, XN, Bw, l15, l4);
      e_load_b(Bw, w.w_f1, C, 0, 0, wave, l15, l4, EWaves<NW>());
      residual_epilogue(acc, par + P_BCO, st + 5, false);
    }
    lds_barrier(); DTS();
    // ================= FFN =================
    layer_norm(par + P_LN + 768, par + P_LN + 896);
    lds_barrier(); DTS();
    if (li + 1 < 4) par_fetch(p.blk[li + 1].par);
    {
      f32x4 acc2[MT][NTC];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NTC; ++j) acc2[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int hc = 0; hc < 4; ++hc) {
        {
          f32x4 acc[MT][NTC];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < NTC; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
          e_mma<MT, 4, NTC>(acc, xn, XN, Bw, l15, l4);
          e_load_b(B2, w.w_f2, 512, 0, hc * 128, wave, l15, l4, EWaves<NW>());
          if (hc == 0 && li + 1 < 4) par_commit((li + 1) & 1);
          if (hc > 0) lds_barrier(); DTS();
#pragma unroll
          for (int j = 0; j < NTC; ++j) {
            const int col = (j * NW + wave) * 16 + l4 * 4;
            const float4 b4 = *reinterpret_cast<const float4*>(bf1 + hc * 128 + col);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              const int row = mt * 16 + l15;
              float v[4] = {fmaxf(acc[mt][j][0] + b4.x, 0.f), fmaxf(acc[mt][j][1] + b4.y, 0.f), fmaxf(acc[mt][j][2] + b4.z, 0.f),
                            fmaxf(acc[mt][j][3] + b4.w, 0.f)};
              if (dp > 0.f) dropout4(v, p.seed, st + 6, (uint32_t)((b * ROWS + row) * 512 + hc * 128 + col), thr16, dpk);
              *reinterpret_cast<uint2*>(cb + row * CB + col) = pack_bf16x4(v[0], v[1], v[2], v[3]);
            }
          }
        }
        if (hc + 1 < 4) e_load_b(Bw, w.w_f1, C, (hc + 1) * 128, 0, wave, l15, l4, EWaves<NW>());
        else if (li + 1 < 4) e_load_b(Bqkv, p.blk[li + 1].w_r2r, C, 0, 0, wave, l15, l4, EWaves<NW>(), 12);
        lds_barrier(); DTS();
        e_mma<MT, 4, NTC>(acc2, cb, CB, B2, l15, l4);
      }
      residual_epilogue(acc2, bf2, st + 7, false);
    }
    lds_barrier(); DTS();
  }
  for (int i = tid; i < RUSE * 32; i += NTH) {
    const int r = i >> 5, c4 = (i & 31) * 4;
    if (r < NQ) *reinterpret_cast<float4*>(p.Q + (qrow0 + r) * C + c4) = *reinterpret_cast<const float4*>(xs + r * XS + c4);
  }
  DTS();
#undef DTS
}

}  // namespace rift
