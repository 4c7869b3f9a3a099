// Level 1 of the NAT-FPN history encoder (dim 64, 4 heads, kernel 3, L = 10; embedding.py:93-99,196-202) in the wave-private,
// register-resident form of nat_l0w.h: one wave owns 4 agents x 10 steps (3 row tiles of 16: row (a, t) -> tile t / 4, lane row
// 4 a + t % 4; the last two steps' quads are half empty) for both NATLayers, the residual stream is the MFMA C/D layout (48 VGPRs),
// GEMM outputs chain as the next GEMM's operand through K-permuted weight images, neighbourhood attention reads its neighbours with
// quad_perm DPP.  The level's weights (80 KiB per NATLayer + 48 KiB for the downsample conv) do not fit LDS together: the 8 waves of a
// workgroup walk their tiles in step and swap the weight image between the layers (three swaps = six barriers per 32 agents, against
// ~30 barriers per 8 agents in nat_level_kernel).
#pragma once
#include "common.h"
#include "nat_l0w.h"
#include "wp_stream.h"

namespace RIFT_NS {

#define L1W_BLK_FRAGS 80        // per NATLayer: qkv (nt 0..11) x 2 k-steps | proj 4 x 2 | fc1 12 x 2 | fc2 6 k-steps x 4 n-tiles
#define L1W_F_QKV(nt, ks) ((nt) * 2 + (ks))
#define L1W_F_PROJ(nt, ks) (24 + (nt) * 2 + (ks))
#define L1W_F_FC1(nt, ks) (32 + (nt) * 2 + (ks))
#define L1W_F_FC2(ks, nt) (56 + (ks) * 4 + (nt))
#define L1W_DS_FRAGS 48         // downsample conv: (tap, k-step of 32 channels, n-tile 0..7)
#define L1W_F_DS(tap, ks, nt) (((tap) * 2 + (ks)) * 8 + (nt))
#define L1W_NFRAG (2 * L1W_BLK_FRAGS + L1W_DS_FRAGS)
// parameters (fp32): per block 800: ln1_g 64, ln1_b 64, bqkv 192 (q pre-scaled), rpb 32 (4 x 5 used), bproj 64, ln2_g 64, ln2_b 64, b1 192, b2 64 |
// fn_g 64, fn_b 64 | ds_g 128, ds_b 128
#define L1W_P_BLK(b) (800 * (b))
#define L1W_PB_LN1G 0
#define L1W_PB_LN1B 64
#define L1W_PB_BQKV 128
#define L1W_PB_RPB 320
#define L1W_PB_BP 352
#define L1W_PB_LN2G 416
#define L1W_PB_LN2B 480
#define L1W_PB_B1 544
#define L1W_PB_B2 736
#define L1W_P_FN 1600
#define L1W_P_DS 1728
// (RIFT_NAT_MFMA_ATTN) score-accumulator table [block 2][head 4][row tile 3][row 5][8] (nat_l0w.h: nat_band_entry; tile 2 holds steps 8, 9 and two
// rows beyond the sequence per agent, which see themselves only)
#define L1W_P_TBL 2048
#define L1W_NPAR (2048 + 2 * 4 * 3 * 5 * 8)
#define L1W_ST 72               // staging row stride (bf16): 144 B
#ifndef L1W_NWV
#define L1W_NWV 8                // waves per workgroup (one workgroup per CU: the LDS image); 12 = three per SIMD at <= 168 VGPRs
#endif
#define L1W_LDS (L1W_BLK_FRAGS * 1024 + L1W_NPAR * 4 + L1W_NWV * 40 * L1W_ST * 2)

struct NatL1WSrc {
  struct Blk { const float *ln1_g, *ln1_b, *wqkv, *bqkv, *rpb, *wproj, *bproj, *ln2_g, *ln2_b, *w1, *b1, *w2, *b2; } blk[2];
  const float* fn_g; const float* fn_b;                                     // norm1
  const float* w_ds; const float* ds_g; const float* ds_b;                  // levels.1.downsample.reduction (128, 64, 3), norm (128)
};

#ifdef RIFT_NAT_L01_IMPL
__global__ void pack_l1w_kernel(NatL1WSrc s, unsigned short* __restrict__ img, float* __restrict__ par) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < L1W_NFRAG * 512) {
    const int f = e >> 9, lane = (e >> 3) & 63, j = e & 7, l15 = lane & 15, l4 = lane >> 4;
    float v;
    bool hid = false;                              // fc2 fragments: hidden-layer operand words (common.h: f2h_hid)
    if (f < 2 * L1W_BLK_FRAGS) {
      const int b = f / L1W_BLK_FRAGS, g = f % L1W_BLK_FRAGS;
      const NatL1WSrc::Blk& k = s.blk[b];
      if (g < 24) { const int nt = g >> 1, ks = g & 1, ch = l0w_chan(l4, j, 2 * ks); v = k.wqkv[(nt * 16 + l15) * 64 + ch] * (nt < 4 ? L0W_QSCALE : 1.0f) * (RIFT_LN_FOLD ? k.ln1_g[ch] : 1.0f); }
      else if (g < 32) { const int nt = (g - 24) >> 1, ks = (g - 24) & 1; v = k.wproj[(nt * 16 + l15) * 64 + l0w_chan(l4, j, 2 * ks)]; }
      else if (g < 56) { const int nt = (g - 32) >> 1, ks = (g - 32) & 1, ch = l0w_chan(l4, j, 2 * ks); v = k.w1[(nt * 16 + l15) * 64 + ch] * (RIFT_LN_FOLD ? k.ln2_g[ch] : 1.0f); }
      else { const int ks = (g - 56) >> 2, nt = (g - 56) & 3; v = k.w2[(nt * 16 + l15) * 192 + l0w_chan(l4, j, 2 * ks)]; hid = true; }
    } else {
      const int g = f - 2 * L1W_BLK_FRAGS, nt = g & 7, ks = (g >> 3) & 1, tap = g >> 4;
      v = s.w_ds[((nt * 16 + l15) * 64 + ks * 32 + l4 * 8 + j) * 3 + tap];
    }
    img[e] = hid ? f2h_hid(v) : f2h(v);
  }
  if (e < L1W_NPAR) {
    float v = 0.f;
    if (e < 1600) {
      const int b = e / 800, o = e % 800;
      const NatL1WSrc::Blk& k = s.blk[b];
      if (o < 64) v = k.ln1_g[o];
      else if (o < 128) v = k.ln1_b[o - 64];
      else if (o < 320) {             // (opfmt.h: RIFT_LN_FOLD -- beta through the weights into the bias)
        v = k.bqkv[o - 128];
        if (RIFT_LN_FOLD) for (int ch = 0; ch < 64; ++ch) v += k.wqkv[(o - 128) * 64 + ch] * k.ln1_b[ch];
        v *= (o - 128 < 64 ? L0W_QSCALE : 1.0f);
      }
      else if (o < 352) v = (o - 320 < 20) ? k.rpb[o - 320] : 0.f;
      else if (o < 416) v = k.bproj[o - 352];
      else if (o < 480) v = k.ln2_g[o - 416];
      else if (o < 544) v = k.ln2_b[o - 480];
      else if (o < 736) {
        v = k.b1[o - 544];
        if (RIFT_LN_FOLD) for (int ch = 0; ch < 64; ++ch) v += k.w1[(o - 544) * 64 + ch] * k.ln2_b[ch];
      }
      else v = k.b2[o - 736];
    } else if (e < L1W_P_DS) v = (e - L1W_P_FN < 64) ? s.fn_g[e - L1W_P_FN] : s.fn_b[e - L1W_P_FN - 64];
    else if (e < L1W_P_DS + 256) v = (e - L1W_P_DS < 128) ? s.ds_g[e - L1W_P_DS] : s.ds_b[e - L1W_P_DS - 128];
    else if (e >= L1W_P_TBL) {
      const int t = e - L1W_P_TBL, slot = t & 7, row = (t >> 3) % 5, mt = (t / 40) % 3, h = (t / 120) % 4, bi = t / 480;
      v = nat_band_entry(s.blk[bi].rpb + h * 5, 10, mt, row, slot, L0W_NEG);
    }
    par[e] = v;
  }
}
#endif

struct NatL1WP {
  const float* X; int nseq;                // (nseq * 10, 64) level input (level 0's downsample output)
  const unsigned short* img; const float* par;
  float* Oc;                               // (nseq * 3, 64)  LayerNorm(norm1) of steps 7..9
  unsigned short* Ocb;                     // if set: the same rows as bf16 instead (what fpn_tail_kernel rounds them to anyway: half the bytes both ways)
  float* Xnext;                            // (nseq * 5, 128) downsample conv + LayerNorm
  float droppath[2]; uint32_t seed, stream;
  const int* cnt;                          // if set: the three class counts of the compacted launch (nat_l0w.h; common.h: SeqCount); nseq is the bound
  DropStats ds;                            // diagnostic build only (dropstats.h)
};

// LayerNorm over the 64 channels of every row (16 per lane, 4 lanes per row) -> bf16 operands of the two k-steps
int l1w_set_attributes();
void l1w_pack(const NatL1WSrc& src, unsigned short* img, float* par, hipStream_t stream);
void l1w_launch(const NatL1WP& p, int grid, hipStream_t stream);

#ifdef RIFT_NAT_L01_IMPL      // the kernels live in nat_l01w.hip
template <bool FOLDED = false>      // (nat_l0w.h: l0w_layer_norm)
__device__ __forceinline__ void l1w_layer_norm(const f32x4 (&x)[3][4], h16x8 (&xn)[3][2], const float* g, const float* b, int l4) {
  if (FOLDED) {
    // statistics of the three row tiles first, ONE cancellation test for the call (nat_l0w.h: l0w_layer_norm), then the normalisation
    float mean[3], var[3];
    bool bad = false;
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) {
      const f32x4 s4 = (x[mt][0] + x[mt][1]) + (x[mt][2] + x[mt][3]);
      const f32x4 q4 = (x[mt][0] * x[mt][0] + x[mt][1] * x[mt][1]) + (x[mt][2] * x[mt][2] + x[mt][3] * x[mt][3]);
      mean[mt] = rows_sum((s4[0] + s4[1]) + (s4[2] + s4[3])) * (1.0f / 64.0f);
      const float ex2 = rows_sum((q4[0] + q4[1]) + (q4[2] + q4[3])) * (1.0f / 64.0f);
      const float m2 = mean[mt] * mean[mt];
      var[mt] = ex2 - m2;
      bad |= ln_row_cancels(m2, var[mt]);
    }
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(bad) != 0ull, 0)) {      // a row whose mean dwarfs its spread: the centred form, as torch
#pragma unroll
      for (int mt = 0; mt < 3; ++mt) {
        f32x4 d4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) { const f32x4 d = x[mt][nt] - mean[mt]; d4 += d * d; }
        var[mt] = rows_sum((d4[0] + d4[1]) + (d4[2] + d4[3])) * (1.0f / 64.0f);
      }
    }
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) {
      const float r = rsqrtf(fmaxf(var[mt], 0.f) + 1e-5f), c = -mean[mt] * r;
      xn[mt][0] = l0w_pack8(x[mt][0] * r + c, x[mt][1] * r + c);
      xn[mt][1] = l0w_pack8(x[mt][2] * r + c, x[mt][3] * r + c);
    }
    return;
  }
  float4 gg[4], bb[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) { gg[nt] = *reinterpret_cast<const float4*>(g + nt * 16 + l4 * 4); bb[nt] = *reinterpret_cast<const float4*>(b + nt * 16 + l4 * 4); }
#pragma unroll
  for (int mt = 0; mt < 3; ++mt) {
    float sm = 0.f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) sm += (x[mt][nt][0] + x[mt][nt][1]) + (x[mt][nt][2] + x[mt][nt][3]);
    const float mean = rows_sum(sm) * (1.0f / 64.0f);
    f32x4 d[4];
    float qs = 0.f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      d[nt] = x[mt][nt] - mean;
      qs += (d[nt][0] * d[nt][0] + d[nt][1] * d[nt][1]) + (d[nt][2] * d[nt][2] + d[nt][3] * d[nt][3]);
    }
    const float r = rsqrtf(rows_sum(qs) * (1.0f / 64.0f) + 1e-5f);
    f32x4 y[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
      y[nt] = (f32x4){d[nt][0] * r * gg[nt].x + bb[nt].x, d[nt][1] * r * gg[nt].y + bb[nt].y, d[nt][2] * r * gg[nt].z + bb[nt].z, d[nt][3] * r * gg[nt].w + bb[nt].w};
    xn[mt][0] = l0w_pack8(y[0], y[1]);
    xn[mt][1] = l0w_pack8(y[2], y[3]);
  }
}

__global__ __launch_bounds__(64 * L1W_NWV) void nat_l1w_kernel(NatL1WP p) {
  constexpr int L = 10;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* wl = reinterpret_cast<unsigned short*>(smem_raw);            // [80][64][8] weight fragments of the current phase
  float* par = reinterpret_cast<float*>(wl + L1W_BLK_FRAGS * 512);
  unsigned short* stg = reinterpret_cast<unsigned short*>(par + L1W_NPAR);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  for (int i = tid; i < L1W_NPAR / 4; i += 64 * L1W_NWV) reinterpret_cast<float4*>(par)[i] = reinterpret_cast<const float4*>(p.par)[i];
  unsigned short* st = stg + wave * 40 * L1W_ST;
  auto W = [&](int f) { return *reinterpret_cast<const h16x8*>(wl + ((size_t)f * 64 + lane) * 8); };
  // workgroup-wide swap of the weight image (all waves are between phases), by LDS-DMA in runs of four fragments (wp_stream.h): the
  // copy through registers (512 threads x uint4 per pass, a global and an LDS round trip each) cost 18 us of this kernel's 118
  const uint32_t wl_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem_raw);
  auto load_weights = [&](int frag0, int nfrag) {
    __syncthreads();
    decw_dma_share(reinterpret_cast<const unsigned char*>(p.img) + (size_t)frag0 * 1024, (uint32_t)lane * 16u, wl_lds, nfrag,
                   __builtin_amdgcn_readfirstlane(wave), L1W_NWV);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  const f32x4 Z = {0.f, 0.f, 0.f, 0.f};
  const int a = l15 >> 2, s = l15 & 3;
  RIFT_SEQ_COUNT(p.cnt, p.nseq);
  const int ntiles = (sq_n + 3) >> 2;
  const int ngroups = (ntiles + L1W_NWV - 1) / L1W_NWV;

  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int tile = grp * L1W_NWV + wave;
    const int seq = tile * 4 + a;
    const bool seq_ok = RIFT_SEQ_LIVE(seq);
    f32x4 x[3][4];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) {
      const int t = mt * 4 + s;
      const bool ok = seq_ok && t < L;
      const float* src = p.X + ((size_t)(ok ? seq : 0) * L + (ok ? t : 0)) * 64 + l4 * 4;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const float4 v = *reinterpret_cast<const float4*>(src + nt * 16);
        x[mt][nt] = ok ? (f32x4){v.x, v.y, v.z, v.w} : Z;
      }
    }
#pragma unroll 1
    for (int bi = 0; bi < 2; ++bi) {
      load_weights(bi * L1W_BLK_FRAGS, L1W_BLK_FRAGS);
      const float* pb = par + L1W_P_BLK(bi);
      h16x8 xn[3][2];
      // ================= attention half =================
      l1w_layer_norm<RIFT_LN_FOLD != 0>(x, xn, pb + L1W_PB_LN1G, pb + L1W_PB_LN1B, l4);
      float dps = 1.f;
      if (p.droppath[bi] > 0.f) dps = (uniform01(p.seed, p.stream + 2 * bi, (uint32_t)seq) < p.droppath[bi]) ? 0.f : 1.0f / (1.0f - p.droppath[bi]);
      if (p.droppath[bi] > 0.f) ds_sample(p.ds, RIFT_DS_NAT(1, bi, 0), seq_ok ? seq : -1, dps);
#if RIFT_NAT_MFMA_ATTN
      // neighbourhood attention on the matrix pipe (nat_l0w.h describes the scheme): 4 heads x 3 row tiles, K = 16 MFMAs
      const int trow = (l4 == a) ? s : 4;
#pragma unroll 1
      for (int h = 0; h < 4; ++h) {
        h16x4 kop[3], vt[3];
        {
          const h16x8 wk0 = W(L1W_F_QKV(4 + h, 0)), wk1 = W(L1W_F_QKV(4 + h, 1)), wv0 = W(L1W_F_QKV(8 + h, 0)), wv1 = W(L1W_F_QKV(8 + h, 1));
          const float4 bk = *reinterpret_cast<const float4*>(pb + L1W_PB_BQKV + 64 + h * 16 + l4 * 4);
          const float bv = pb[L1W_PB_BQKV + 128 + h * 16 + l15];
#pragma unroll
          for (int mt = 0; mt < 3; ++mt) {
            f32x4 kk = mfma_h(wk0, xn[mt][0], (f32x4){bk.x, bk.y, bk.z, bk.w}, 0, 0, 0);
            kk = mfma_h(wk1, xn[mt][1], kk, 0, 0, 0);
            kop[mt] = pack_h16x4(kk[0], kk[1], kk[2], kk[3]);
            f32x4 vv = mfma_h(xn[mt][0], wv0, (f32x4){bv, bv, bv, bv}, 0, 0, 0);      // plain order: the 4 keys of agent l4 in this tile, head dim l15
            vv = mfma_h(xn[mt][1], wv1, vv, 0, 0, 0);
            vt[mt] = pack_h16x4(vv[0], vv[1], vv[2], vv[3]);
          }
        }
        const h16x8 wq0 = W(L1W_F_QKV(h, 0)), wq1 = W(L1W_F_QKV(h, 1));
        const float4 bq = *reinterpret_cast<const float4*>(pb + L1W_PB_BQKV + h * 16 + l4 * 4);
        const int pks = h >> 1;                                    // proj k-step that holds this head's 16 channels (its lower / upper half)
        h16x4 wp[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) { const h16x8 w = W(L1W_F_PROJ(nt, pks)); wp[nt] = (h & 1) ? h16x4_hi(w) : h16x4_lo(w); }
        const float* tb = par + L1W_P_TBL + ((bi * 4 + h) * 15 + trow) * 8;
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) {
          const float4 ca = *reinterpret_cast<const float4*>(tb + mt * 40), cb = *reinterpret_cast<const float4*>(tb + mt * 40 + 4);
          f32x4 qq = mfma_h(wq0, xn[mt][0], (f32x4){bq.x, bq.y, bq.z, bq.w}, 0, 0, 0);
          qq = mfma_h(wq1, xn[mt][1], qq, 0, 0, 0);
          const h16x4 qop = pack_h16x4(qq[0], qq[1], qq[2], qq[3]);
          const f32x4 so = mfma_h16(kop[mt], qop, (f32x4){ca.y, ca.z, ca.w, cb.x});
          float sp = L0W_NEG, sn = L0W_NEG;
          if (mt > 0) sp = mfma_h16(kop[mt - 1], qop, (f32x4){L0W_NEG, L0W_NEG, L0W_NEG, ca.x})[3];
          if (mt < 2) sn = mfma_h16(kop[mt + 1], qop, (f32x4){cb.y, L0W_NEG, L0W_NEG, L0W_NEG})[0];
          const float m = fmaxf(fmaxf(fmaxf(so[0], so[1]), fmaxf(so[2], so[3])), fmaxf(fmaxf(sp, sn), -1.0e20f));
          const float e0 = __builtin_amdgcn_exp2f(so[0] - m), e1 = __builtin_amdgcn_exp2f(so[1] - m), e2 = __builtin_amdgcn_exp2f(so[2] - m), e3 = __builtin_amdgcn_exp2f(so[3] - m);
          const float ep = mt > 0 ? __builtin_amdgcn_exp2f(sp - m) : 0.f, en = mt < 2 ? __builtin_amdgcn_exp2f(sn - m) : 0.f;
          const float inv = __builtin_amdgcn_rcpf(fmaxf(((e0 + e1) + (e2 + e3)) + (ep + en), 1.0e-30f)) * dps;
          f32x4 o = mfma_h16(vt[mt], pack_h16x4(e0 * inv, e1 * inv, e2 * inv, e3 * inv), Z);
          if (mt > 0) o = mfma_h16(vt[mt - 1], pack_h16x4(0.f, 0.f, 0.f, ep * inv), o);
          if (mt < 2) o = mfma_h16(vt[mt + 1], pack_h16x4(en * inv, 0.f, 0.f, 0.f), o);
          const h16x4 ao = pack_h16x4(o[0], o[1], o[2], o[3]);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) x[mt][nt] = mfma_h16(wp[nt], ao, x[mt][nt]);
        }
      }
#else
#pragma unroll 1
      for (int h = 0; h < 4; ++h) {
        f32x4 k[3], v[3];
        {
          const h16x8 wk0 = W(L1W_F_QKV(4 + h, 0)), wk1 = W(L1W_F_QKV(4 + h, 1)), wv0 = W(L1W_F_QKV(8 + h, 0)), wv1 = W(L1W_F_QKV(8 + h, 1));
          const float4 bk = *reinterpret_cast<const float4*>(pb + L1W_PB_BQKV + 64 + h * 16 + l4 * 4);
          const float4 bv = *reinterpret_cast<const float4*>(pb + L1W_PB_BQKV + 128 + h * 16 + l4 * 4);
#pragma unroll
          for (int mt = 0; mt < 3; ++mt) {
            k[mt] = mfma_h(wk0, xn[mt][0], Z, 0, 0, 0);
            k[mt] = mfma_h(wk1, xn[mt][1], k[mt], 0, 0, 0) + (f32x4){bk.x, bk.y, bk.z, bk.w};
            v[mt] = mfma_h(wv0, xn[mt][0], Z, 0, 0, 0);
            v[mt] = mfma_h(wv1, xn[mt][1], v[mt], 0, 0, 0) + (f32x4){bv.x, bv.y, bv.z, bv.w};
          }
        }
        const h16x8 wq0 = W(L1W_F_QKV(h, 0)), wq1 = W(L1W_F_QKV(h, 1));
        const float4 bq = *reinterpret_cast<const float4*>(pb + L1W_PB_BQKV + h * 16 + l4 * 4);
        const int pks = h >> 1;                                    // proj k-step that holds this head's 16 channels (its lower / upper half)
        h16x8 wp[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wp[nt] = W(L1W_F_PROJ(nt, pks));
        const float* rp = pb + L1W_PB_RPB + h * 5;
        // neighbourhood attention, kernel 3: keys of step t are (t-1, t, t+1), (0, 1, 2) at t = 0, (7, 8, 9) at t = 9 = (tile 2, quad lane 1)
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) {
          f32x4 qq = mfma_h(wq0, xn[mt][0], Z, 0, 0, 0);
          qq = mfma_h(wq1, xn[mt][1], qq, 0, 0, 0) + (f32x4){bq.x, bq.y, bq.z, bq.w};
          f32x4 km = l0w_dpp4<0x90>(k[mt]), kp = l0w_dpp4<0xF9>(k[mt]);
          f32x4 vm = l0w_dpp4<0x90>(v[mt]), vp = l0w_dpp4<0xF9>(v[mt]);
          if (mt > 0) { km = l0w_sel(s == 0, l0w_dpp4<0xFF>(k[mt - 1]), km); vm = l0w_sel(s == 0, l0w_dpp4<0xFF>(v[mt - 1]), vm); }
          if (mt < 2) { kp = l0w_sel(s == 3, l0w_dpp4<0x00>(k[mt + 1]), kp); vp = l0w_sel(s == 3, l0w_dpp4<0x00>(v[mt + 1]), vp); }
          f32x4 k0 = km, k1 = k[mt], k2 = kp, v0 = vm, v1 = v[mt], v2 = vp;
          int shift = 0;
          if (mt == 0) {        // t = 0: keys (own, +1, +2)
            const f32x4 kpp = l0w_dpp4<0xFE>(k[0]), vpp = l0w_dpp4<0xFE>(v[0]);
            const bool e = s == 0;
            k0 = l0w_sel(e, k[0], km); k1 = l0w_sel(e, kp, k[0]); k2 = l0w_sel(e, kpp, kp);
            v0 = l0w_sel(e, v[0], vm); v1 = l0w_sel(e, vp, v[0]); v2 = l0w_sel(e, vpp, vp);
            shift = e ? 1 : 0;
          }
          if (mt == 2) {        // t = 9 (quad lane 1): keys (7, 8, 9) = (previous tile's lane 3, this quad's lane 0, own)
            const f32x4 k7 = l0w_dpp4<0xFF>(k[1]), v7 = l0w_dpp4<0xFF>(v[1]);
            const bool e = s == 1;
            k0 = l0w_sel(e, k7, km); k1 = l0w_sel(e, km, k[2]); k2 = l0w_sel(e, k[2], kp);
            v0 = l0w_sel(e, v7, vm); v1 = l0w_sel(e, vm, v[2]); v2 = l0w_sel(e, v[2], vp);
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
          const h16x8 ao = (h & 1) ? l0w_pack8(Z, oh) : l0w_pack8(oh, Z);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) x[mt][nt] += mfma_h(wp[nt], ao, Z, 0, 0, 0) * dps;
        }
      }
#endif
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const float4 b4 = *reinterpret_cast<const float4*>(pb + L1W_PB_BP + nt * 16 + l4 * 4);
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) x[mt][nt] += (f32x4){b4.x, b4.y, b4.z, b4.w} * dps;
      }
      // ================= MLP half: fc1 (64 -> 192) -> GELU -> fc2 (192 -> 64), 32 hidden channels (one fc2 k-step) at a time =================
      l1w_layer_norm<RIFT_LN_FOLD != 0>(x, xn, pb + L1W_PB_LN2G, pb + L1W_PB_LN2B, l4);
      {
        f32x4 acc2[3][4];
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) acc2[mt][nt] = Z;
#pragma unroll 1
        for (int ks = 0; ks < 6; ++ks) {
          const h16x8 wa0 = W(L1W_F_FC1(2 * ks, 0)), wa1 = W(L1W_F_FC1(2 * ks, 1)), wb0 = W(L1W_F_FC1(2 * ks + 1, 0)), wb1 = W(L1W_F_FC1(2 * ks + 1, 1));
          h16x8 u[4];
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) u[nt] = W(L1W_F_FC2(ks, nt));
          const float4 ba = *reinterpret_cast<const float4*>(pb + L1W_PB_B1 + (2 * ks) * 16 + l4 * 4);
          const float4 bb = *reinterpret_cast<const float4*>(pb + L1W_PB_B1 + (2 * ks + 1) * 16 + l4 * 4);
#pragma unroll
          for (int mt = 0; mt < 3; ++mt) {
            f32x4 ha = mfma_h(wa0, xn[mt][0], hid_init(ba), 0, 0, 0);      // (packed-fp16 GELU: the bias is the accumulator's initial value)
            ha = mfma_h(wa1, xn[mt][1], ha, 0, 0, 0);
            f32x4 hb = mfma_h(wb0, xn[mt][0], hid_init(bb), 0, 0, 0);
            hb = mfma_h(wb1, xn[mt][1], hb, 0, 0, 0);
            const h16x8 hop = l0w_from_u2(gelu4_hid(ha, ba), gelu4_hid(hb, bb));
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc2[mt][nt] = mfma_hid(u[nt], hop, acc2[mt][nt]);
          }
        }
        float dp2 = 1.f;
        if (p.droppath[bi] > 0.f) dp2 = (uniform01(p.seed, p.stream + 2 * bi + 1, (uint32_t)seq) < p.droppath[bi]) ? 0.f : 1.0f / (1.0f - p.droppath[bi]);
        if (p.droppath[bi] > 0.f) ds_sample(p.ds, RIFT_DS_NAT(1, bi, 1), seq_ok ? seq : -1, dp2);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const float4 b4 = *reinterpret_cast<const float4*>(pb + L1W_PB_B2 + nt * 16 + l4 * 4);
#pragma unroll
          for (int mt = 0; mt < 3; ++mt) x[mt][nt] += (acc2[mt][nt] + (f32x4){b4.x, b4.y, b4.z, b4.w}) * dp2;
        }
      }
    }
    // ---- what the FPN reads: LayerNorm(norm1) of steps 7 (tile 1, lane 3), 8, 9 (tile 2, lanes 0, 1)
    {
      const float* g = par + L1W_P_FN;
#pragma unroll
      for (int mt = 1; mt < 3; ++mt) {
        float sm = 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) sm += (x[mt][nt][0] + x[mt][nt][1]) + (x[mt][nt][2] + x[mt][nt][3]);
        const float mean = rows_sum(sm) * (1.0f / 64.0f);
        f32x4 d[4];
        float qs = 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          d[nt] = x[mt][nt] - mean;
          qs += (d[nt][0] * d[nt][0] + d[nt][1] * d[nt][1]) + (d[nt][2] * d[nt][2] + d[nt][3] * d[nt][3]);
        }
        const float r = rsqrtf(rows_sum(qs) * (1.0f / 64.0f) + 1e-5f);
        const int t = mt * 4 + s;
        if (seq_ok && t >= 7 && t < L) {
          const size_t orow = ((size_t)seq * 3 + (t - 7)) * 64;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            const float4 gg = *reinterpret_cast<const float4*>(g + nt * 16 + l4 * 4), bb = *reinterpret_cast<const float4*>(g + 64 + nt * 16 + l4 * 4);
            const float4 o = make_float4(d[nt][0] * r * gg.x + bb.x, d[nt][1] * r * gg.y + bb.y, d[nt][2] * r * gg.z + bb.z, d[nt][3] * r * gg.w + bb.w);
            if (p.Ocb) *reinterpret_cast<uint2*>(p.Ocb + orow + nt * 16 + l4 * 4) = pack_h4(o.x, o.y, o.z, o.w);
            else *reinterpret_cast<float4*>(p.Oc + orow + nt * 16 + l4 * 4) = o;
          }
        }
      }
    }
    // ---- next level's input: Conv1d(64 -> 128, k = 3, stride 2, pad 1, no bias) + LayerNorm(128) through the per-wave staging tile
    load_weights(2 * L1W_BLK_FRAGS, L1W_DS_FRAGS);
    {
#pragma unroll
      for (int mt = 0; mt < 3; ++mt) {
        const int t = mt * 4 + s;
        if (t < L) {
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            *reinterpret_cast<uint2*>(st + (a * L + t) * L1W_ST + nt * 16 + l4 * 4) = pack_h4(x[mt][nt][0], x[mt][nt][1], x[mt][nt][2], x[mt][nt][3]);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      f32x4 d[2][8];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) d[mt][nt] = Z;
#pragma unroll 1
      for (int tk = 0; tk < 6; ++tk) {                       // (tap, k-step)
        const int tap = tk >> 1, ks = tk & 1;
        h16x8 wd[8];
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) wd[nt] = W(tk * 8 + nt);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int m = mt * 16 + l15, oa = m / 5, t = 2 * (m - oa * 5) - 1 + tap;
          h16x8 bop = (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
          if (m < 20 && t >= 0 && t < L) bop = *reinterpret_cast<const h16x8*>(st + (oa * L + t) * L1W_ST + ks * 32 + l4 * 8);
#pragma unroll
          for (int nt = 0; nt < 8; ++nt) d[mt][nt] = mfma_h(wd[nt], bop, d[mt][nt], 0, 0, 0);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const float* g = par + L1W_P_DS;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        float sm = 0.f;
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) sm += (d[mt][nt][0] + d[mt][nt][1]) + (d[mt][nt][2] + d[mt][nt][3]);
        const float mean = rows_sum(sm) * (1.0f / 128.0f);
        float qs = 0.f;
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
          d[mt][nt] = d[mt][nt] - mean;
          qs += (d[mt][nt][0] * d[mt][nt][0] + d[mt][nt][1] * d[mt][nt][1]) + (d[mt][nt][2] * d[mt][nt][2] + d[mt][nt][3] * d[mt][nt][3]);
        }
        const float r = rsqrtf(rows_sum(qs) * (1.0f / 128.0f) + 1e-5f);
        const int m = mt * 16 + l15, oa = m / 5;
        if (m < 20 && RIFT_SEQ_LIVE(tile * 4 + oa)) {
          float* dst = p.Xnext + ((size_t)tile * 20 + m) * 128;
#pragma unroll
          for (int nt = 0; nt < 8; ++nt) {
            const float4 gg = *reinterpret_cast<const float4*>(g + nt * 16 + l4 * 4), bb = *reinterpret_cast<const float4*>(g + 128 + nt * 16 + l4 * 4);
            *reinterpret_cast<float4*>(dst + nt * 16 + l4 * 4) =
                make_float4(d[mt][nt][0] * r * gg.x + bb.x, d[mt][nt][1] * r * gg.y + bb.y, d[mt][nt][2] * r * gg.z + bb.z, d[mt][nt][3] * r * gg.w + bb.w);
          }
        }
      }
    }
  }
}

#endif

}  // namespace RIFT_NS
