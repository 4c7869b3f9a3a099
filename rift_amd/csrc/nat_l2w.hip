// Wave-private level-2 NAT kernel (see nat_l2w.h); its own translation unit for the reasons given in dec_w.hip.
#include "common.h"
#include "nat_l2w.h"
#include "wp_stream.h"

namespace RIFT_NS {

// Weight image: [block][group][fragment f = ks * 8 + nt][lane][8] with the K permutation of nat_l0w.h.  natten's qkv rows are (3, H, 16):
// q of head h = rows h*16.., k = 128 + ..., v = 256 + ...; q carries head_dim^-0.5 log2 e.
__global__ void pack_l2w_kernel(NatL2WSrc s, unsigned short* __restrict__ img, float* __restrict__ par) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const float SC = 0.25f * 1.4426950408889634f;
  if (e < 2 * L2W_BLK_FRAGS * 512) {
    const int j = e & 7, lane = (e >> 3) & 63, fr = e >> 9, bi = fr / L2W_BLK_FRAGS, g = (fr % L2W_BLK_FRAGS) >> 5, f = fr & 31;
    const int ks = f >> 3, nt = f & 7, l15 = lane & 15, l4 = lane >> 4;
    const int ch = l0w_chan(l4, j, 2 * ks), o = nt * 16 + l15;
    const NatL2WSrc::Blk& k = s.blk[bi];
    float v;
    const bool hid = g >= 4 && ((g - 4) & 1);      // fc2 fragments: hidden-layer operand words (common.h: f2h_hid)
    if (g < 3) v = k.wqkv[(g * 128 + o) * 128 + ch] * (g == 0 ? SC : 1.0f) * (RIFT_LN_FOLD ? k.ln1_g[ch] : 1.0f);
    else if (g == 3) v = k.wproj[o * 128 + ch];
    else {
      const int c = (g - 4) >> 1;
      v = ((g - 4) & 1) ? k.w2[o * 384 + c * 128 + ch] : k.w1[(c * 128 + o) * 128 + ch] * (RIFT_LN_FOLD ? k.ln2_g[ch] : 1.0f);
    }
    img[e] = hid ? f2h_hid(v) : f2h(v);
  }
  if (e < L2W_NPAR) {
    float v = 0.f;
    if (e < 2 * 1664) {
      const int bi = e / 1664, o = e % 1664;
      const NatL2WSrc::Blk& k = s.blk[bi];
      if (o < 128) v = k.ln1_g[o];
      else if (o < 256) v = k.ln1_b[o - 128];
      else if (o < 640) {             // (opfmt.h: RIFT_LN_FOLD -- beta through the weights into the bias)
        v = k.bqkv[o - 256];
        if (RIFT_LN_FOLD) for (int ch = 0; ch < 128; ++ch) v += k.wqkv[(o - 256) * 128 + ch] * k.ln1_b[ch];
        v *= (o - 256 < 128 ? SC : 1.0f);
      }
      else if (o < 768) {             // [head 8][16]: rpb[h][0..8] x log2 e, then -inf (slot 9 = "a key of another agent": the lanes read it unconditionally)
        const int h = (o - 640) >> 4, r = (o - 640) & 15;
        v = r < 9 ? k.rpb[h * 9 + r] * 1.4426950408889634f : -INFINITY;
      }
      else if (o < 896) v = k.bproj[o - 768];
      else if (o < 1024) v = k.ln2_g[o - 896];
      else if (o < 1152) v = k.ln2_b[o - 1024];
      else if (o < 1536) {
        v = k.b1[o - 1152];
        if (RIFT_LN_FOLD) for (int ch = 0; ch < 128; ++ch) v += k.w1[(o - 1152) * 128 + ch] * k.ln2_b[ch];
      }
      else v = k.b2[o - 1536];
    } else if (e < L2W_P_FN + 256) v = (e - L2W_P_FN < 128) ? s.fn_g[e - L2W_P_FN] : s.fn_b[e - L2W_P_FN - 128];
    par[e] = v;
  }
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void nat_l2w_kernel(NatL2WP p) {
  constexpr int L = 5;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char* ring = smem_raw;
  float* par = reinterpret_cast<float*>(smem_raw + L2W_SLOTS * 32768);
  const int tid = threadIdx.x;
  int lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const int wv0 = __builtin_amdgcn_readfirstlane(tid >> 6);
  int wv = wv0;
  const uint32_t lds00 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem_raw);
  uint32_t lds0 = lds00, voff = (uint32_t)lane * 16u;
  const f32x4 Z = {0.f, 0.f, 0.f, 0.f};
  RIFT_SEQ_COUNT(p.cnt, p.nseq);
  // How the tiles are dealt.  A chip-filling launch (at least one full round of 8 G tiles): round r of workgroup b is tiles (r G + b) 8 + wave,
  // as in rounds 2 - 5 -- the workgroups without a last round END EARLY and hand their CU to the neighbouring queues' kernels.  (Round 6 tried
  // the leftover tiles WAVE-major over all workgroups with the tile-less waves skipping the arithmetic: the kernel alone 93 -> 84.5 us, the
  // pipelined 256-scene step +10 us -- every CU is held to the end of the launch and the step is bound by CU-TIME, not by this kernel's
  // latency: profiles/r06_ab_l2_dealing.txt.)  A launch far from filling the chip (at most 4 G tiles: batches up to ~50 scenes) deals
  // its tiles wave-major -- tile = wave G + b -- so that every CU runs two or three waves, one per SIMD, instead of a few CUs eight: 33 -> 26 us
  // at 32 scenes; waves without a tile only keep the operand stream's protocol.  Same tiles, same arithmetic: bit-identical either way.
  const int ntiles = (sq_n + 2) / 3;
  const int G = (int)gridDim.x, nrounds = (ntiles + 7) >> 3;
#ifdef RIFT_L2_TILE_MAJOR      // (diagnostic build define: tile-major whatever the size)
  const bool wave_major = false;
#else
  const bool wave_major = ntiles <= 4 * G;      // (at most one wave per SIMD; between 4 G and 8 G tiles -- 64 scenes -- wave-major measured slower: 0.195 against 0.185 ms per step)
#endif
  const int full = wave_major ? 0 : ((int)blockIdx.x < nrounds ? (nrounds - 1 - (int)blockIdx.x) / G + 1 : 0);
  const int trem = wave_major ? ntiles : 0;
  const int my_rounds = full + ((int)blockIdx.x < trem ? 1 : 0);
  const int total = 20 * my_rounds;                      // groups this workgroup consumes
  int tsn = 0;
#define L2TS() do { if (p.ts && blockIdx.x == 0 && tid == 0 && tsn < 60) p.ts[tsn++] = clock64(); } while (0)

  // ---- operand stream: group s of this workgroup = group s % 20 of the two-layer sequence; ring slot s % 3; a boundary requests s + 2
  const unsigned char* wimg = reinterpret_cast<const unsigned char*>(p.img);
  auto request = [&](int s) {
    if (s >= total) return;
    const unsigned char* src = wimg + (size_t)(s % 20) * 32768;
    const uint32_t dst = (uint32_t)(s % 3) * 32768u;
    decw_dma_share(src, voff, lds0 + dst, 32, wv, 8);
  };
  // boundary that opens group s: my four requests of it have landed (at most the four of group s + 1 may still fly behind them), then
  // the barrier: everybody's have, and nobody reads slot (s - 1) % 3 any more
  auto boundary = [&](int s) {
    if (s + 1 < total) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    L2TS();
    request(s + 2);
  };
  auto slot_off = [&](int s) -> uint32_t { return (uint32_t)(s % 3) * 32768u; };
  auto gemm = [&](uint32_t so, const h16x8 (&x)[4], f32x4 (&acc)[8]) { decw_gemm<false>((uint32_t)(uintptr_t)ring + so + voff, x, acc); };

  for (int i = tid; i < L2W_NPAR / 4; i += 512) reinterpret_cast<float4*>(par)[i] = reinterpret_cast<const float4*>(p.par)[i];
  request(0);
  request(1);

  auto init8 = [&](f32x4 (&a)[8], const float* bias) {
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) { const float4 v = *reinterpret_cast<const float4*>(bias + nt * 16 + l4 * 4); a[nt] = (f32x4){v.x, v.y, v.z, v.w}; }
  };
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

  int s = 0;                                             // sequence number of the next group to open
#pragma unroll 1
  for (int rr = 0; rr < my_rounds; ++rr) {
    {   // opaque zeros (see dec_w.hip): keep the addresses of the 20 groups from being hoisted out of the loop
      int zv, zs;
      asm volatile("v_mov_b32 %0, 0" : "=v"(zv));
      asm volatile("s_mov_b32 %0, 0" : "=s"(zs));
      lane = (tid & 63) + zv; l15 = lane & 15; l4 = lane >> 4; voff = (uint32_t)lane * 16u; wv = wv0 + zs; lds0 = lds00 + (uint32_t)zs;
    }
    const int last_idx = wv * G + (int)blockIdx.x;
    const int tile = rr < full ? (rr * G + (int)blockIdx.x) * 8 + wv : full * 8 * G + last_idx;
    if (__builtin_amdgcn_readfirstlane((int)(rr >= full && last_idx >= trem))) {      // a wave without a tile in the last round: the stream's protocol only
#pragma unroll 1
      for (int g = 0; g < 20; ++g) { boundary(s); ++s; }
      continue;
    }
    const int qa = l15 / L, qt = l15 - qa * L;           // this lane row: agent qa (3 = the idle row 15), step qt
    const int seq = tile * 3 + qa;
    const bool row_ok = qa < 3 && RIFT_SEQ_LIVE(seq);
    f32x4 res[8];
    {
      const float* src = p.X + ((size_t)(row_ok ? seq : 0) * L + (row_ok ? qt : 0)) * 128 + l4 * 4;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const float4 v = *reinterpret_cast<const float4*>(src + nt * 16);
        res[nt] = row_ok ? (f32x4){v.x, v.y, v.z, v.w} : Z;
      }
    }
    // score accumulator pattern of this lane: keys 4 l4 + i -> same agent ? rpb[key step - query step + 4] : -inf.  The idle row 15 is its
    // own "agent 3" and sees itself: a row without any key would turn NaN, and a NaN KEY row poisons every query through the score MFMA
    int ridx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int key = l4 * 4 + i, ka = key / L, kt = key - ka * L;
      ridx[i] = (ka == qa) ? kt - qt + 4 : 9;           // slot 9 of a head's table row: -inf
    }
#pragma unroll 1
    for (int bi = 0; bi < 2; ++bi) {
      const float* pb = par + L2W_P_BLK(bi);
      f32x4 acc[8];
      h16x8 xb[4], qf[4], kp[4], ao[4];
#if RIFT_ATTN_K16
      h16x4 vf[8];
#else
      h16x8 vf[8];
#endif
      float dps = 1.f, dp2 = 1.f;
      if (p.droppath[bi] > 0.f) {
        dps = (uniform01(p.seed, p.stream + 2 * bi, (uint32_t)seq) < p.droppath[bi]) ? 0.f : 1.0f / (1.0f - p.droppath[bi]);
        dp2 = (uniform01(p.seed, p.stream + 2 * bi + 1, (uint32_t)seq) < p.droppath[bi]) ? 0.f : 1.0f / (1.0f - p.droppath[bi]);
        ds_sample(p.ds, RIFT_DS_NAT(2, bi, 0), row_ok ? seq : -1, dps);
        ds_sample(p.ds, RIFT_DS_NAT(2, bi, 1), row_ok ? seq : -1, dp2);
      }
      // ---- group 0: q
      layer_norm(res, xb, pb + L2W_PB_LN1);
      init8(acc, pb + L2W_PB_BQKV);
      boundary(s); gemm(slot_off(s), xb, acc); ++s;
#pragma unroll
      for (int j = 0; j < 4; ++j) qf[j] = l0w_pack8(acc[2 * j], acc[2 * j + 1]);
      init8(acc, pb + L2W_PB_BQKV + 128);
      // ---- group 1: k
      boundary(s); gemm(slot_off(s), xb, acc); ++s;
#pragma unroll
      for (int j = 0; j < 4; ++j) kp[j] = l0w_pack8(acc[2 * j], acc[2 * j + 1]);
      // ---- group 2: v (plain order: lane = 4 keys of dim l15 of head nt) + attention
      {
        f32x4 av[8];
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) { const float bv = pb[L2W_PB_BQKV + 256 + nt * 16 + l15]; av[nt] = (f32x4){bv, bv, bv, bv}; }
        boundary(s); decw_gemm<true>((uint32_t)(uintptr_t)ring + slot_off(s) + voff, xb, av); ++s;
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
#if RIFT_ATTN_K16
          vf[nt] = __builtin_bit_cast(h16x4, pack_h4(av[nt][0], av[nt][1], av[nt][2], av[nt][3]));
#else
          vf[nt] = l0w_from_u2(pack_h4(av[nt][0], av[nt][1], av[nt][2], av[nt][3]), make_uint2(0u, 0u));
#endif
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 o[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int h = 2 * j + u;
          f32x4 mb;
#pragma unroll
          for (int i = 0; i < 4; ++i) mb[i] = pb[L2W_PB_RPB + h * 16 + ridx[i]];      // (a select here became four exec-masked loads per head)
#if RIFT_ATTN_K16
          // head h's 16 dims are one half of the projection's fragment: a K = 16 operand as it is
          const f32x4 sc = mfma_h16(u == 0 ? h16x4_lo(kp[j]) : h16x4_hi(kp[j]), u == 0 ? h16x4_lo(qf[j]) : h16x4_hi(qf[j]), mb);
#else
          // head h's 16 dims are one half of the k-step: the other half of the K operand is zero
          h16x8 kh = kp[j];
          if (u == 0) { kh[4] = 0; kh[5] = 0; kh[6] = 0; kh[7] = 0; } else { kh[0] = 0; kh[1] = 0; kh[2] = 0; kh[3] = 0; }
          const f32x4 sc = mfma_h(kh, qf[j], mb, 0, 0, 0);
#endif
          const float m = rows_max(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])));
          const f32x4 ev = {__builtin_amdgcn_exp2f(sc[0] - m), __builtin_amdgcn_exp2f(sc[1] - m), __builtin_amdgcn_exp2f(sc[2] - m), __builtin_amdgcn_exp2f(sc[3] - m)};
          const float inv = __builtin_amdgcn_rcpf(rows_sum((ev[0] + ev[1]) + (ev[2] + ev[3])));
#if RIFT_ATTN_K16
          o[u] = mfma_h16(vf[h], __builtin_bit_cast(h16x4, pack_h4(ev[0], ev[1], ev[2], ev[3])), Z) * inv;
#else
          const h16x8 pf = l0w_from_u2(pack_h4(ev[0], ev[1], ev[2], ev[3]), make_uint2(0u, 0u));
          o[u] = mfma_h(vf[h], pf, Z, 0, 0, 0) * inv;
#endif
        }
        ao[j] = l0w_pack8(o[0], o[1]);
      }
      init8(acc, pb + L2W_PB_BP);
      // ---- group 3: proj, DropPath, residual
      boundary(s); gemm(slot_off(s), ao, acc); ++s;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) res[nt] += acc[nt] * dps;
      // ---- groups 4..9: fc1 chunk -> GELU -> fc2 partial
      layer_norm(res, xb, pb + L2W_PB_LN2);
      f32x4 acc2[8];
      init8(acc2, pb + L2W_PB_B2);
#pragma unroll 1
      for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) acc[nt] = hid_init(*reinterpret_cast<const float4*>(pb + L2W_PB_B1 + c * 128 + nt * 16 + l4 * 4));   // (packed-fp16 GELU: the bias is the initial value)
        boundary(s);
        gemm(slot_off(s), xb, acc); ++s;
        h16x8 hb[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const float4 ba = *reinterpret_cast<const float4*>(pb + L2W_PB_B1 + c * 128 + (2 * ks) * 16 + l4 * 4);
          const float4 bb = *reinterpret_cast<const float4*>(pb + L2W_PB_B1 + c * 128 + (2 * ks + 1) * 16 + l4 * 4);
          hb[ks] = l0w_from_u2(gelu4_hid(acc[2 * ks], ba), gelu4_hid(acc[2 * ks + 1], bb));
        }
        boundary(s); decw_gemm<false, true>((uint32_t)(uintptr_t)ring + slot_off(s) + voff, hb, acc2); ++s;
      }
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) res[nt] += acc2[nt] * dp2;
    }
    // ---- what the FPN reads: LayerNorm(norm2) of steps 2, 3, 4
    {
      const float* g = par + L2W_P_FN;
      f32x4 s4 = (res[0] + res[1]) + (res[2] + res[3]);
      s4 += (res[4] + res[5]) + (res[6] + res[7]);
      const float mean = rows_sum((s4[0] + s4[1]) + (s4[2] + s4[3])) * (1.0f / 128.0f);
      f32x4 d[8];
      f32x4 q4 = Z;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) { d[nt] = res[nt] - mean; q4 += d[nt] * d[nt]; }
      const float r = rsqrtf(rows_sum((q4[0] + q4[1]) + (q4[2] + q4[3])) * (1.0f / 128.0f) + 1e-5f);
      if (row_ok && qt >= 2) {
        const size_t orow = ((size_t)seq * 3 + (qt - 2)) * 128;
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
          const float4 gg = *reinterpret_cast<const float4*>(g + nt * 16 + l4 * 4), bb = *reinterpret_cast<const float4*>(g + 128 + nt * 16 + l4 * 4);
          const float4 o = make_float4(d[nt][0] * r * gg.x + bb.x, d[nt][1] * r * gg.y + bb.y, d[nt][2] * r * gg.z + bb.z, d[nt][3] * r * gg.w + bb.w);
          if (p.Ocb) *reinterpret_cast<uint2*>(p.Ocb + orow + nt * 16 + l4 * 4) = pack_h4(o.x, o.y, o.z, o.w);
          else *reinterpret_cast<float4*>(p.Oc + orow + nt * 16 + l4 * 4) = o;
        }
      }
    }
  }
#undef L2TS
}

int l2w_set_attributes() {
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&nat_l2w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, L2W_LDS);
}
void l2w_pack(const NatL2WSrc& src, unsigned short* img, float* par, hipStream_t stream) {
  const int n = 2 * L2W_BLK_FRAGS * 512;
  hipLaunchKernelGGL(pack_l2w_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, src, img, par);
}
void l2w_launch(const NatL2WP& p, int grid, hipStream_t stream) {
  hipLaunchKernelGGL(nat_l2w_kernel, dim3(grid), dim3(512), (size_t)L2W_LDS, stream, p);
}

}  // namespace RIFT_NS
