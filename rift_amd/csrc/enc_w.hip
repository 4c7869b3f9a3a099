// Wave-private, weight-streaming scene encoder for the dense-traffic shapes (see enc_w.h); its own translation unit like dec_w.hip.
#include "common.h"
#include "enc_w.h"
#include "wp_stream.h"

namespace RIFT_NS {

__global__ void pack_encw_kernel(EncWSrc s, unsigned short* __restrict__ img, float* __restrict__ par) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const float SC = 0.17677669529663687f * 1.4426950408889634f;   // head_dim^-0.5 (head_dim = 32) x log2 e
  if (e >= 4 * ENCW_LAYER_FRAGS * 512 && e < (4 * ENCW_LAYER_FRAGS + ENCW_TAIL_FRAGS) * 512) {   // decoder K | V projection groups (layer l: k, v)
    const int t = e - 4 * ENCW_LAYER_FRAGS * 512;
    const int j = t & 7, lane = (t >> 3) & 63, fr = t >> 9, g = fr >> 5, f = fr & 31;
    const int ks = f >> 3, nt = f & 7, l15 = lane & 15, l4 = lane >> 4;
    img[e] = f2h(s.dkv_w[g >> 1][((size_t)(128 + (g & 1) * 128 + nt * 16 + l15)) * 128 + l0w_chan(l4, j, 2 * ks)]);
  }
  if (e < 4 * ENCW_LAYER_FRAGS * 512) {
    const int j = e & 7, lane = (e >> 3) & 63, fr = e >> 9, li = fr / ENCW_LAYER_FRAGS, g = (fr % ENCW_LAYER_FRAGS) >> 5, f = fr & 31;
    const int ks = f >> 3, nt = f & 7, l15 = lane & 15, l4 = lane >> 4;
    const int ch = l0w_chan(l4, j, 2 * ks), o = nt * 16 + l15;
    const EncWSrc::L& L = s.l[li];
    float v;
    const bool hid = g >= 4 && ((g - 4) & 1);                       // fc2 fragments: hidden-layer operand words (common.h: f2h_hid)
    const float g1 = RIFT_LN_FOLD ? L.ln1_g[ch] : 1.0f, g2 = RIFT_LN_FOLD ? L.ln2_g[ch] : 1.0f;     // (opfmt.h: gamma of the norm in front folded into the weights)
    if (g == 0) v = L.w_in[(128 + o) * 128 + ch] * g1;             // k
    else if (g == 1) v = L.w_in[(256 + o) * 128 + ch] * g1;        // v
    else if (g == 2) v = L.w_in[o * 128 + ch] * SC * g1;           // q
    else if (g == 3) v = L.wo[o * 128 + ch];
    else {
      const int c = (g - 4) >> 1;
      v = ((g - 4) & 1) ? L.w2[o * 512 + c * 128 + ch] : L.w1[(c * 128 + o) * 128 + ch] * g2;
    }
    img[e] = hid ? f2h_hid(v) : f2h(v);
  }
  if (e < ENCW_NPAR) {
    float v;
    if (e < ENCW_P_FN) {
      const int li = e / ENCW_P_LAYER, o = e % ENCW_P_LAYER;
      const EncWSrc::L& L = s.l[li];
      // (RIFT_LN_FOLD) beta of the norm in front through the rows of W into the bias: b + W beta
      auto wb = [&](const float* Wm, int row, const float* beta) { float a = 0.f; if (RIFT_LN_FOLD) for (int k = 0; k < 128; ++k) a += Wm[(size_t)row * 128 + k] * beta[k]; return a; };
      if (o < 128) v = L.ln1_g[o];
      else if (o < 256) v = L.ln1_b[o - 128];
      else if (o < 384) v = (L.b_in[o - 256] + wb(L.w_in, o - 256, L.ln1_b)) * SC;
      else if (o < 512) v = L.b_in[128 + o - 384] + wb(L.w_in, 128 + o - 384, L.ln1_b);
      else if (o < 640) v = L.b_in[256 + o - 512] + wb(L.w_in, 256 + o - 512, L.ln1_b);
      else if (o < 768) v = L.bo[o - 640];
      else if (o < 896) v = L.ln2_g[o - 768];
      else if (o < 1024) v = L.ln2_b[o - 896];
      else if (o < 1536) v = L.b1[o - 1024] + wb(L.w1, o - 1024, L.ln2_b);
      else v = L.b2[o - 1536];
    } else if (e < ENCW_P_BKV) v = (e - ENCW_P_FN < 128) ? s.fn_g[e - ENCW_P_FN] : s.fn_b[e - ENCW_P_FN - 128];
    else { const int q = e - ENCW_P_BKV; v = s.dkv_b[q >> 8][128 + (q & 255)]; }
    par[e] = v;
  }
}

// NKE: key tiles the attention walks -- 8 when the batch has N <= 128 token slots (the CARLA shapes of rift_pluto.yaml:35-36: 109), else all
// 12 of the image; the tiles beyond the batch's own are zero fragments behind -inf masks either way (exact zeros in both sums)
template <int NKE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void enc_w_kernel(EncWP p) {
  constexpr int NKT = 12;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char* ring = smem_raw;
  float* par = reinterpret_cast<float*>(smem_raw + 2 * 32768);
  float* smaskf = par + ENCW_NPAR;                                              // [192] key mask as 0 / -inf
  const int tid = threadIdx.x;
  int lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const int wv0 = __builtin_amdgcn_readfirstlane(tid >> 6);
  int wv = wv0;
  const int b = blockIdx.x, N = p.N;
  const int ntiles = (N + 15) >> 4, RR = (ntiles + 7) >> 3;                     // rounds of eight tiles
  const int GL = 16 * RR;                                                       // groups of a layer: RR x [k | v], then RR x [q | 4 heads | out | 8 FFN]
  const int Q3 = p.DKV ? 22 : 14;                                               // pass-2 groups per round in the last layer (+ 8: the decoder's K | V)
  const size_t row0 = (size_t)b * N;
  const uint32_t lds00 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem_raw);
  uint32_t lds0 = lds00, voff = (uint32_t)lane * 16u;
  const f32x4 Z = {0.f, 0.f, 0.f, 0.f};
  const unsigned char* wimg = reinterpret_cast<const unsigned char*>(p.img);
  unsigned short* kvs = p.KVs + (size_t)b * 96 * 512;

  auto dma = [&](const void* src, uint32_t dst, int nfrag) {
    decw_dma_share(reinterpret_cast<const unsigned char*>(src), voff, lds0 + dst, nfrag, wv, 8);
  };
  auto request = [&](int li, int pos) {
    if (pos >= (li == 3 ? 2 * RR + Q3 * RR : GL)) { pos = 0; ++li; }
    if (li >= 4) return;
    const unsigned char* wl = wimg + (size_t)li * ENCW_LAYER_FRAGS * 1024;
    const uint32_t dst = (uint32_t)(pos & 1) * 32768u;
    if (pos < 2 * RR) { dma(wl + (size_t)(pos & 1) * 32768, dst, 32); return; }
    const int q = (pos - 2 * RR) % (li == 3 ? Q3 : 14);
    if (q >= 14) dma(wimg + (size_t)(4 * ENCW_LAYER_FRAGS + (q - 14) * 32) * 1024, dst, 32);       // last layer's tail: decoder K | V weights
    else if (q == 0) dma(wl + 2 * 32768, dst, 32);
    else if (q < 5) dma(reinterpret_cast<const unsigned char*>(kvs) + (size_t)(q - 1) * 24 * 1024, dst, 24);
    else dma(wl + (size_t)(3 + q - 5) * 32768, dst, 32);
  };
  auto bnd = [&](int li, int pos) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    request(li, pos + 1);
  };
  auto W = [&](int slot, int f) { return *reinterpret_cast<const h16x8*>(ring + slot * 32768 + f * 1024 + lane * 16); };
  auto gemm = [&](int slot, const h16x8 (&x)[4], f32x4 (&acc)[8]) { decw_gemm<false>((uint32_t)(uintptr_t)ring + (uint32_t)slot * 32768u + voff, x, acc); };

  for (int i = tid; i < ENCW_NPAR / 4; i += 512) reinterpret_cast<float4*>(par)[i] = reinterpret_cast<const float4*>(p.par)[i];
  for (int i = tid; i < 192; i += 512) smaskf[i] = ((i >= N) || p.kpm[row0 + i]) ? -INFINITY : 0.f;
  request(0, 0);

  auto init8 = [&](f32x4 (&a)[8], const float* bias) {
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) { const float4 v = *reinterpret_cast<const float4*>(bias + nt * 16 + l4 * 4); a[nt] = (f32x4){v.x, v.y, v.z, v.w}; }
  };
  // LayerNorm of the tile's rows in the C/D layout; OUT = false: bf16 operands of the four k-steps, OUT = true: fp32 values back into res
  auto layer_norm = [&](f32x4 (&res)[8], h16x8 (&xb)[4], const float* g, bool out) {
    f32x4 s4 = (res[0] + res[1]) + (res[2] + res[3]);
    s4 += (res[4] + res[5]) + (res[6] + res[7]);
    const float mean = rows_sum((s4[0] + s4[1]) + (s4[2] + s4[3])) * (1.0f / 128.0f);
    if (RIFT_LN_FOLD && !out) {     // (opfmt.h: gamma / beta live in the consuming GEMM's weights and bias; one-pass statistics.  The final norm keeps its affine part)
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
        if (out) res[nt] = y[u];
      }
      xb[ks] = l0w_pack8(y[0], y[1]);
    }
  };
  int tile = 0, trow = 0; bool act = false, rok = false;
  auto set_tile = [&](int rr) {
    tile = rr * 8 + wv; act = tile < ntiles;
    trow = tile * 16 + l15; rok = act && trow < N;
  };
  auto load_rows = [&](f32x4 (&res)[8], const float* base) {
    const float* src = base + (row0 + (rok ? trow : 0)) * 128 + l4 * 4;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      const float4 v = *reinterpret_cast<const float4*>(src + nt * 16);
      res[nt] = rok ? (f32x4){v.x, v.y, v.z, v.w} : Z;
    }
  };
  auto store_rows = [&](const f32x4 (&res)[8]) {
    if (rok) {
      float* dst = p.Y + (row0 + trow) * 128 + l4 * 4;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) *reinterpret_cast<float4*>(dst + nt * 16) = make_float4(res[nt][0], res[nt][1], res[nt][2], res[nt][3]);
    }
  };
  // attention of head h over all 16 NKT keys against the K | V^T fragments of a ring slot (K (kt) at kt, V^T (d, pt) at 12 + 6 d + pt)
  auto attend = [&](int slot, const h16x8 qh, h16x8& aoh) {
    f32x4 s[NKT];
#pragma unroll
    for (int kt = 0; kt < NKE; ++kt) {
      const float4 mk = *reinterpret_cast<const float4*>(smaskf + kt * 16 + l4 * 4);
      s[kt] = mfma_h(W(slot, kt), qh, (f32x4){mk.x, mk.y, mk.z, mk.w}, 0, 0, 0);
    }
    float m = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3]));
#pragma unroll
    for (int kt = 1; kt < NKE; ++kt) m = fmaxf(fmaxf(m, fmaxf(s[kt][0], s[kt][1])), fmaxf(s[kt][2], s[kt][3]));
    m = rows_max(m);
    f32x4 l4s = Z;
#pragma unroll
    for (int kt = 0; kt < NKE; ++kt) {
      s[kt] = (f32x4){__builtin_amdgcn_exp2f(s[kt][0] - m), __builtin_amdgcn_exp2f(s[kt][1] - m), __builtin_amdgcn_exp2f(s[kt][2] - m), __builtin_amdgcn_exp2f(s[kt][3] - m)};
      l4s += s[kt];
    }
    const float inv = __builtin_amdgcn_rcpf(rows_sum((l4s[0] + l4s[1]) + (l4s[2] + l4s[3])));
    f32x4 o0 = Z, o1 = Z;
#pragma unroll
    for (int pt = 0; pt < NKE / 2; ++pt) {
      const h16x8 pf = l0w_pack8(s[2 * pt], s[2 * pt + 1]);
      o0 = mfma_h(W(slot, 12 + pt), pf, o0, 0, 0, 0);
      o1 = mfma_h(W(slot, 18 + pt), pf, o1, 0, 0, 0);
    }
    aoh = l0w_pack8(o0 * inv, o1 * inv);
  };

#pragma unroll 1
  for (int li = 0; li < 4; ++li) {
    {   // opaque zeros (see dec_w.hip)
      int zv, zs;
      asm volatile("v_mov_b32 %0, 0" : "=v"(zv));
      asm volatile("s_mov_b32 %0, 0" : "=s"(zs));
      lane = (tid & 63) + zv; l15 = lane & 15; l4 = lane >> 4; voff = (uint32_t)lane * 16u; wv = wv0 + zs; lds0 = lds00 + (uint32_t)zs;
    }
    const float* pl = par + li * ENCW_P_LAYER;
    float dp1 = 1.f, dp2 = 1.f;
    if (p.droppath[li] > 0.f) {
      dp1 = (uniform01(p.seed, p.stream + 2 * li, (uint32_t)b) < p.droppath[li]) ? 0.f : 1.0f / (1.0f - p.droppath[li]);
      dp2 = (uniform01(p.seed, p.stream + 2 * li + 1, (uint32_t)b) < p.droppath[li]) ? 0.f : 1.0f / (1.0f - p.droppath[li]);
      ds_sample(p.ds, RIFT_DS_ENC(li, 0), b, dp1);
      ds_sample(p.ds, RIFT_DS_ENC(li, 1), b, dp2);
    }
    const float* rows_in = li == 0 ? p.X : p.Y;
    // ================= pass 1: the layer's K | V^T operand fragments of every tile =================
#pragma unroll 1
    for (int rr = 0; rr < RR; ++rr) {
      set_tile(rr);
      const int p1 = 2 * rr;
      if (act) {
        f32x4 res[8], acc[8];
        h16x8 xb[4];
        bnd(li, p1 + 0);                                        // ---- k
        if (rr == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        load_rows(res, rows_in);
        layer_norm(res, xb, pl + ENCW_P_LN1, false);
        init8(acc, pl + ENCW_P_BK); gemm(0, xb, acc);
#pragma unroll
        for (int h = 0; h < 4; ++h)                             // K fragment (head h, key tile `tile`): this lane's 16 bytes
          *reinterpret_cast<h16x8*>(kvs + ((size_t)(h * 24 + tile) * 64 + lane) * 8) = l0w_pack8(acc[2 * h], acc[2 * h + 1]);
        bnd(li, p1 + 1);                                        // ---- v (plain order: lane = 4 keys of dim nt * 16 + l15)
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) { const float bv = pl[ENCW_P_BV + nt * 16 + l15]; acc[nt] = (f32x4){bv, bv, bv, bv}; }
        decw_gemm<true>((uint32_t)(uintptr_t)ring + 32768u + voff, xb, acc);
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)                          // V^T fragment (head nt / 2, dim tile nt & 1, key pair tile / 2): k slots (tile & 1) * 4 ..
          *reinterpret_cast<uint2*>(kvs + ((size_t)((nt >> 1) * 24 + 12 + (nt & 1) * 6 + (tile >> 1)) * 64 + lane) * 8 + (tile & 1) * 4) =
              pack_h4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]);
      } else {
        // a round's missing tiles: their K rows must still be finite (masked keys multiply V^T by zero) -- tiles >= ntiles of a 12-tile image
        bnd(li, p1 + 0);
        bnd(li, p1 + 1);
      }
    }
    if (li == 0) {   // key tiles beyond the batch's token count (N <= 176): zero fragments, once (the masks are -inf there, the operands finite)
      for (int t = ntiles + wv; t < NKT; t += 8) {
#pragma unroll
        for (int h = 0; h < 4; ++h) *reinterpret_cast<uint4*>(kvs + ((size_t)(h * 24 + t) * 64 + lane) * 8) = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
          *reinterpret_cast<uint2*>(kvs + ((size_t)((nt >> 1) * 24 + 12 + (nt & 1) * 6 + (t >> 1)) * 64 + lane) * 8 + (t & 1) * 4) = make_uint2(0u, 0u);
      }
    }
    if (li == 0 && p.DKV) {   // likewise in the decoder's image (all four layers)
      for (int t = ntiles + wv; t < NKT; t += 8)
        for (int l = 0; l < 4; ++l) {
          unsigned short* dk = p.DKV + ((size_t)b * 4 + l) * 96 * 512;
#pragma unroll
          for (int h = 0; h < 4; ++h) *reinterpret_cast<uint4*>(dk + ((size_t)(h * 24 + t) * 64 + lane) * 8) = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
          for (int nt = 0; nt < 8; ++nt)
            *reinterpret_cast<uint2*>(dk + ((size_t)((nt >> 1) * 24 + 12 + (nt & 1) * 6 + (t >> 1)) * 64 + lane) * 8 + (t & 1) * 4) = make_uint2(0u, 0u);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    // ================= pass 2: attention over all tokens + MLP of every tile =================
#pragma unroll 1
    for (int rr = 0; rr < RR; ++rr) {
      set_tile(rr);
      const int p2 = 2 * RR + (li == 3 ? Q3 : 14) * rr;
      if (act) {
        f32x4 res[8], acc[8];
        h16x8 xb[4], qf[4], ao[4];
        bnd(li, p2 + 0);                                        // ---- q
        if (rr == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        load_rows(res, rows_in);
        layer_norm(res, xb, pl + ENCW_P_LN1, false);
        init8(acc, pl + ENCW_P_BQ); gemm(0, xb, acc);
#pragma unroll
        for (int h = 0; h < 4; ++h) qf[h] = l0w_pack8(acc[2 * h], acc[2 * h + 1]);
#pragma unroll
        for (int h = 0; h < 4; ++h) { bnd(li, p2 + 1 + h); attend((1 + h) & 1, qf[h], ao[h]); }   // ---- the four heads' K | V^T groups
        init8(acc, pl + ENCW_P_BO);
        bnd(li, p2 + 5);                                        // ---- out_proj, DropPath residual, LayerNorm
        gemm(1, ao, acc);
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) res[nt] += acc[nt] * dp1;
        layer_norm(res, xb, pl + ENCW_P_LN2, false);
        f32x4 acc2[8];
        init8(acc2, pl + ENCW_P_B2);
#pragma unroll 1
        for (int c = 0; c < 4; ++c) {                           // ---- fc1 chunk -> GELU -> fc2 partial
          bnd(li, p2 + 6 + 2 * c);
#pragma unroll
          for (int nt = 0; nt < 8; ++nt) acc[nt] = hid_init(*reinterpret_cast<const float4*>(pl + ENCW_P_B1 + c * 128 + nt * 16 + l4 * 4));   // (packed-fp16 GELU: the bias is the initial value)
          gemm(0, xb, acc);
          h16x8 hb[4];
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const float4 ba = *reinterpret_cast<const float4*>(pl + ENCW_P_B1 + c * 128 + (2 * ks) * 16 + l4 * 4);
            const float4 bb = *reinterpret_cast<const float4*>(pl + ENCW_P_B1 + c * 128 + (2 * ks + 1) * 16 + l4 * 4);
            hb[ks] = l0w_from_u2(gelu4_hid(acc[2 * ks], ba), gelu4_hid(acc[2 * ks + 1], bb));
          }
          bnd(li, p2 + 7 + 2 * c);
          decw_gemm<false, true>((uint32_t)(uintptr_t)ring + 32768u + voff, hb, acc2);      // fc2: hidden-layer operand words (opfmt.h)
        }
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) res[nt] += acc2[nt] * dp2;
        if (li == 3) {
          layer_norm(res, xb, par + ENCW_P_FN, true);   // the encoder's final LayerNorm (xb: its bf16 operand rows)
          uint32_t ex = 0;                              // non-finite flag by bit pattern (common.h: exp_max)
#pragma unroll
          for (int nt = 0; nt < 8; ++nt) { ex = exp_max(ex, res[nt][0]); ex = exp_max(ex, res[nt][1]); ex = exp_max(ex, res[nt][2]); ex = exp_max(ex, res[nt][3]); }
          if (__builtin_amdgcn_ballot_w64(rok && nonfinite_exp(ex)) != 0ull && lane == 0 && p.nonfinite) atomicOr(p.nonfinite, 1);
        }
        store_rows(res);
        if (li == 3 && p.DKV) {                                 // ---- the decoder's K | V^T fragments of its four layers from these rows
#pragma unroll 1
          for (int l = 0; l < 4; ++l) {
            unsigned short* dk = p.DKV + ((size_t)b * 4 + l) * 96 * 512;
            const float* bk = par + ENCW_P_BKV + l * 256;
            init8(acc, bk);
            bnd(li, p2 + 14 + 2 * l);
            gemm(0, xb, acc);
#pragma unroll
            for (int h = 0; h < 4; ++h) *reinterpret_cast<h16x8*>(dk + ((size_t)(h * 24 + tile) * 64 + lane) * 8) = l0w_pack8(acc[2 * h], acc[2 * h + 1]);
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) { const float bv = bk[128 + nt * 16 + l15]; acc[nt] = (f32x4){bv, bv, bv, bv}; }
            bnd(li, p2 + 15 + 2 * l);
            decw_gemm<true>((uint32_t)(uintptr_t)ring + 32768u + voff, xb, acc);
#pragma unroll
            for (int nt = 0; nt < 8; ++nt)
              *reinterpret_cast<uint2*>(dk + ((size_t)((nt >> 1) * 24 + 12 + (nt & 1) * 6 + (tile >> 1)) * 64 + lane) * 8 + (tile & 1) * 4) =
                  pack_h4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]);
          }
        }
      } else {
#pragma unroll 1
        for (int k = 0; k < (li == 3 ? Q3 : 14); ++k) bnd(li, p2 + k);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  }
}

int encw_set_attributes() {
  int e = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&enc_w_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, ENCW_LDS);
  if (!e) e = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&enc_w_kernel<12>), hipFuncAttributeMaxDynamicSharedMemorySize, ENCW_LDS);
  return e;
}
void encw_pack(const EncWSrc& src, unsigned short* img, float* par, hipStream_t stream) {
  const int n = (4 * ENCW_LAYER_FRAGS + ENCW_TAIL_FRAGS) * 512;
  hipLaunchKernelGGL(pack_encw_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, src, img, par);
}
void encw_launch(const EncWP& p, hipStream_t stream) {
  if (p.N <= 128) hipLaunchKernelGGL(enc_w_kernel<8>, dim3(p.bs), dim3(512), (size_t)ENCW_LDS, stream, p);
  else hipLaunchKernelGGL(enc_w_kernel<12>, dim3(p.bs), dim3(512), (size_t)ENCW_LDS, stream, p);
}

}  // namespace RIFT_NS
