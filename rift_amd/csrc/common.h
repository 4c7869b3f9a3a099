// Shared device helpers for the RIFT gfx950 kernels (wave64, MFMA fragments, RNG).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "opfmt.h"

namespace RIFT_NS {

typedef __attribute__((ext_vector_type(8))) short h16x8;    // 8 operand words = 4 VGPRs (MFMA 16x16x32 A/B operand)
typedef __attribute__((ext_vector_type(4))) float f32x4;    // MFMA 16x16 accumulator
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 mfma_h(h16x8 a, h16x8 b, f32x4 c, int, int, int) {
#if RIFT_OP_F16
  typedef _Float16 v8 __attribute__((ext_vector_type(8)));
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
#else
  typedef __bf16 v8 __attribute__((ext_vector_type(8)));
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
#endif
}

// K = 16 form (v_mfma_f32_16x16x16_{bf16,f16}): operands are 4 words = 2 VGPRs, lane (l15, l4) holds k = 4 l4 + j -- the natural fit of a
// 16-dim attention head: a projection's C/D fragment (4 consecutive channels of row l15) IS its operand, no zero half, no permutation.
typedef __attribute__((ext_vector_type(4))) short h16x4;
__device__ __forceinline__ f32x4 mfma_h16(h16x4 a, h16x4 b, f32x4 c) {
#if RIFT_OP_F16
  typedef _Float16 v4 __attribute__((ext_vector_type(4)));
  return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(v4, a), __builtin_bit_cast(v4, b), c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
#endif
}

// two fp32 -> packed operand pair (round-to-nearest-even) in ONE instruction (gfx950 v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32).  The `s_nop 0`
// is part of the contract: an MFMA that reads a VALU-written VGPR as an operand needs TWO wait states behind the write
// (tools/ubench/cvt_mfma_hazard.hip on MI355X: 0 or 1 states -> the MFMA reads the register's previous content in 95-100 % of the issues,
// 2 -> never), and behind an asm statement hipcc pads only one.  Without the nop every "convert, then multiply" site was a latent
// stale-operand read.
__device__ __forceinline__ unsigned int pack_h2(float lo, float hi) {
  unsigned int r;
  asm(RIFT_CVT_PK_H_ASM " %0, %1, %2\n\ts_nop 0" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
// (round 5, measured and not kept: hipcc's own conversion -- pack_h2c below -- in place of this statement everywhere.  ~100 s_nop fewer per level-0
// tile, 0.1 % of the step; pe_w_kernel / enc_w_kernel / mha_mfma_kernel start spilling under the freer schedule, and the fp16 build's results move)
__device__ __forceinline__ unsigned short f2h(float f) {   // round-to-nearest-even fp32 -> operand word: the same instruction, one lane used
  return (unsigned short)(pack_h2(f, 0.f) & 0xffffu);
}
// operand word(s) -> fp32: a shift / mask for bf16, v_cvt_f32_f16 (SDWA picks the upper half) for fp16 -- one VALU instruction either way
// (plain C++ so that hipcc, not an opaque asm statement, owns the SDWA hazards)
#if RIFT_OP_F16
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
#endif
__device__ __forceinline__ float h_lo(unsigned int u) {
#if RIFT_OP_F16
  return (float)__builtin_bit_cast(f16x2_t, u)[0];
#else
  return __uint_as_float(u << 16);
#endif
}
__device__ __forceinline__ float h_hi(unsigned int u) {
#if RIFT_OP_F16
  return (float)__builtin_bit_cast(f16x2_t, u)[1];
#else
  return __uint_as_float(u & 0xffff0000u);
#endif
}
__device__ __forceinline__ float h2f(unsigned short h) { return h_lo((unsigned int)h); }
// re-pack two fp32 values that ARE operand-format values (results of max / select over unpacked words): exact, no rounding involved
__device__ __forceinline__ unsigned int h_pair_exact(float lo, float hi) {
#if RIFT_OP_F16
  return pack_h2(lo, hi);
#else
  return (__float_as_uint(hi) & 0xffff0000u) | (__float_as_uint(lo) >> 16);
#endif
}
#if RIFT_OP_F16
#define RIFT_H_NEG_INF2 0xfc00fc00u      // a packed pair of -inf
#else
#define RIFT_H_NEG_INF2 0xff80ff80u
#endif

__device__ __forceinline__ uint2 pack_h4(float a, float b, float c, float d) {
  uint2 u; u.x = pack_h2(a, b); u.y = pack_h2(c, d); return u;
}
// The same conversion as hipcc's OWN instruction (fptrunc <2 x float> lowers to v_cvt_pk_{bf16,f16}_f32 on gfx950).  Use this form whenever
// an input may be a fresh MFMA result: the wait states between an MFMA write and a VALU read are software's to insert, hipcc pads its own
// instructions and never the inside of an asm statement (round 5: an asm conversion fed by an accumulator read stale registers -> NaN).
// The two wait states ahead of a consuming MFMA are hipcc's to pad as well.
__device__ __forceinline__ unsigned int pack_h2c(float lo, float hi) {
  f32x2_t v; v.x = lo; v.y = hi;
#if RIFT_OP_F16
  typedef _Float16 hc2 __attribute__((ext_vector_type(2)));
#else
  typedef __bf16 hc2 __attribute__((ext_vector_type(2)));
#endif
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, hc2));
}
__device__ __forceinline__ h16x4 pack_h16x4(float a, float b, float c, float d) {
  uint2 u; u.x = pack_h2c(a, b); u.y = pack_h2c(c, d);
  return __builtin_bit_cast(h16x4, u);
}
__device__ __forceinline__ h16x4 h16x4_lo(const h16x8 w) { return (h16x4){w[0], w[1], w[2], w[3]}; }
__device__ __forceinline__ h16x4 h16x4_hi(const h16x8 w) { return (h16x4){w[4], w[5], w[6], w[7]}; }

// ---- bf16 weight images are stored FRAGMENT-MAJOR ---------------------------------------------------------------------
// A weight matrix W[N][Kp] (N padded to 16, Kp to 32) is cut into MFMA operand fragments: fragment (nt, ks) = rows
// 16 nt..+15, columns 32 ks..+31, 1 KiB, holding lane l's 8 bf16 (row 16 nt + (l & 15), columns 32 ks + 8 (l >> 4)..+7)
// at byte offset 16 l.  One wave-wide fragment load is then ONE contiguous 1 KiB request instead of 16 rows x 64 B:
// measured 60-65 B/cycle/CU against 16 B/cycle/CU for the row-major image (tools/ubench/l2_stream.hip), which is what
// bounds kernels that stream their weights from L2 once per row tile.
__host__ __device__ __forceinline__ size_t fm_index(int n, int k, int Kp) {
  return ((size_t)((n >> 4) * (Kp >> 5) + (k >> 5)) * 64 + (size_t)(((k >> 3) & 3) * 16 + (n & 15))) * 8 + (k & 7);
}
// fragment (rows n0.., columns k0..) of a fragment-major image with row length Kp; n0 % 16 == 0, k0 % 32 == 0
__device__ __forceinline__ h16x8 fm_load(const unsigned short* W, int Kp, int n0, int k0, int lane) {
  return *reinterpret_cast<const h16x8*>(W + ((size_t)((n0 >> 4) * (Kp >> 5) + (k0 >> 5)) * 64 + lane) * 8);
}

// counter-based RNG for dropout / drop-path / state-dropout: one 32-bit hash per element
// (two rounds of the lowbias32 integer finaliser over (seed, stream, element index); 32-bit multiplies only).
__device__ __forceinline__ uint32_t hash32(uint32_t seed, uint32_t stream, uint32_t idx) {
  uint32_t x = idx * 0x9E3779B1u + (seed ^ (stream * 0x85EBCA6Bu));
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  x += seed * 0xC2B2AE35u + stream;
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}
// Element-wise dropout decisions: ONE lowbias32 round per counter and two 16-bit uniforms out of every hash (probability resolution
// 2^-16); the decision is an integer compare against thr16 = p * 65536.  Four times cheaper per element than uniform01, which the
// decoder kernel's dropout epilogues (attention weights, residual branches, FFN hidden layer) were paying 70 us a step for.
__device__ __forceinline__ uint32_t hash1(uint32_t seed, uint32_t stream, uint32_t idx) {
  uint32_t x = idx * 0x9E3779B1u + (seed ^ (stream * 0x85EBCA6Bu));
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t drop_thr16(float p) { return (uint32_t)(p * 65536.0f); }
// keep / drop four consecutive elements whose first flat index is idx4 (a multiple of 4): scales v[] in place
__device__ __forceinline__ void dropout4(float (&v)[4], uint32_t seed, uint32_t stream, uint32_t idx4, uint32_t thr16, float keep_scale) {
  const uint32_t h0 = hash1(seed, stream, idx4 >> 1), h1 = hash1(seed, stream, (idx4 >> 1) + 1);
  v[0] = ((h0 & 0xffffu) < thr16) ? 0.f : v[0] * keep_scale;
  v[1] = ((h0 >> 16) < thr16) ? 0.f : v[1] * keep_scale;
  v[2] = ((h1 & 0xffffu) < thr16) ? 0.f : v[2] * keep_scale;
  v[3] = ((h1 >> 16) < thr16) ? 0.f : v[3] * keep_scale;
}

__device__ __forceinline__ float uniform01(uint32_t seed, uint32_t stream, uint32_t idx) {
  return (float)(hash32(seed, stream, idx) >> 8) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// erf-form GELU (nn.GELU default) for the bf16 paths: erf(z) = z P(z^2) / Q(z^2) with z clamped to +-3.3 (a (3,3) rational minimax
// fit, |erf error| <= 2.9e-6, |gelu error| <= 7.2e-6 measured over [-10, 10] -- two orders below the bf16 rounding the result gets
// as an MFMA operand).  One transcendental (v_rcp_f32); every other step is a mul / fma that packs two elements per instruction
// (v_pk_mul_f32 / v_pk_fma_f32) in gelu_fast2: 39 cycles per element against 49 for the rcp + exp2 Abramowitz-Stegun 7.1.25 form
// and 2.6e-5 error (tools/ubench/gelu_rate.hip).  The LDS-resident encoder / NAT kernels are VALU-bound exactly in these epilogues.
// fp32 mode keeps erff (gelu_erf).
#define RIFT_GELU_P0 1.12838531f
#define RIFT_GELU_P1 0.153424003f
#define RIFT_GELU_P2 0.0432474986f
#define RIFT_GELU_P3 0.000753648848f
#define RIFT_GELU_Q1 0.469360935f
#define RIFT_GELU_Q2 0.0945981576f
#define RIFT_GELU_Q3 0.00932609519f
__device__ __forceinline__ float gelu_fast(float x) {
  const float z = __builtin_amdgcn_fmed3f(x * 0.70710678118654752440f, -3.3f, 3.3f);
  const float t = z * z;
  const float pn = fmaf(t, fmaf(t, fmaf(t, RIFT_GELU_P3, RIFT_GELU_P2), RIFT_GELU_P1), RIFT_GELU_P0);
  const float qd = fmaf(t, fmaf(t, fmaf(t, RIFT_GELU_Q3, RIFT_GELU_Q2), RIFT_GELU_Q1), 1.0f);
  const float e = z * pn * __builtin_amdgcn_rcpf(qd);
  const float hx = 0.5f * x;
  return fmaf(hx, e, hx);
}
__device__ __forceinline__ f32x2_t gelu_fast2(f32x2_t x) {
  typedef f32x2_t V;
  V z = x * 0.70710678118654752440f;
  z.x = __builtin_amdgcn_fmed3f(z.x, -3.3f, 3.3f); z.y = __builtin_amdgcn_fmed3f(z.y, -3.3f, 3.3f);
  const V t = z * z;
  const V pn = __builtin_elementwise_fma(t, __builtin_elementwise_fma(t, __builtin_elementwise_fma(t, (V)RIFT_GELU_P3, (V)RIFT_GELU_P2), (V)RIFT_GELU_P1), (V)RIFT_GELU_P0);
  const V qd = __builtin_elementwise_fma(t, __builtin_elementwise_fma(t, __builtin_elementwise_fma(t, (V)RIFT_GELU_Q3, (V)RIFT_GELU_Q2), (V)RIFT_GELU_Q1), (V)1.0f);
  V r; r.x = __builtin_amdgcn_rcpf(qd.x); r.y = __builtin_amdgcn_rcpf(qd.y);
  const V e = z * pn * r;
  const V hx = x * 0.5f;
  return __builtin_elementwise_fma(hx, e, hx);
}

// fp32 += dot of two packed operand pairs (v_dot2c_f32_bf16 / v_dot2c_f32_f16): a 2-element q.k step without unpacking either operand
__device__ __forceinline__ float dot2_h(unsigned int a, unsigned int b, float c) {
#if RIFT_OP_F16
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a), __builtin_bit_cast(f16x2_t, b), c, false);
#else
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
#endif
}
// the two values of a packed operand dword as fp32
__device__ __forceinline__ f32x2_t unpack_h2(unsigned int u) {
  f32x2_t r; r.x = h_lo(u); r.y = h_hi(u); return r;
}

// ---- GELU at the precision its consumer keeps (round 5) ---------------------------------------------------------------------------
// The GELU outputs of this model are MFMA operands: a bf16 word keeps 8 significand bits.  gelu(x) = relu(x) - h(|x|) with the bump
// h(a) = a Phi(-a) (<= 0.17, -> 0 beyond a = 4) as a degree-7 polynomial in u = a / 2 - 1, evaluated in PACKED FP16 (v_pk_fma_f16: two
// elements per full-rate instruction, 11 significand bits): both tails keep their relative accuracy (x < 0: the result IS -h; x > 0: x
// minus a small term), the operand word's error against erf GELU in double is 1.76e-3 rms for N(0, 1.5) inputs against 1.70e-3 for the
// correctly rounded bf16 word (tools/ubench/gelu_pk16.hip: max 1.76e-2 / 1.56e-2, rms 4.93e-3 / 4.81e-3 over [-8, 8]), and the epilogue is
// 36 -> 28 issue slots per four elements of which none is a two-pass packed-fp32 instruction (97 -> 81 ticks per element in that
// micro-benchmark).  Two pairs run in lock step: a v_pk_*_f16 result read by the NEXT instruction costs a wait state.
// Round 6: the fp16-operand build evaluates this form too (it kept the fp32 rational form through round 5).  What its 11-bit consumer
// sees on N(0, 1.5) inputs: operand-word error 3.35e-4 rms against 2.12e-4 for the correctly rounded fp16 word -- of which 3.06e-4 is the
// price of rounding the INPUT to fp16 (exact GELU of the fp16-rounded input), not of the polynomial -- and end to end nothing: the
// benchmark batch's four objectives stay at 8e-6 .. 1e-5 of the oracle, the small-batch envelope is unchanged (tests/test_gpu_parity.py
// header), and the build's step time falls from 0.627 to 0.580 ms (profiles/r06_ab_fp16_gelu.txt).  One instruction-level fact behind
// the accounting (profiles/r06_valu_occupancy.txt): a v_pk_*_f16 instruction costs 1.6 plain VALU slots on this chip, not 1.
// RANGE: the conversion to fp16 is not clamped (a clamp is two more packed instructions per four elements, 2 % of the NAT kernels): a
// pre-activation beyond 65504 becomes inf, travels through fc2 into the residual stream and raises the non-finite flag
// (rift_check_finite; tests/test_gpu_parity.py::test_hidden_layer_overflow_raises_the_non_finite_flag) -- the contract of every fp16
// operand (opfmt.h), never silent garbage.
typedef _Float16 gelu_h2 __attribute__((ext_vector_type(2)));
#define RIFT_GELU_H0 0.04531748f
#define RIFT_GELU_H1 -0.17138292f
#define RIFT_GELU_H2 0.22257517f
#define RIFT_GELU_H3 0.01231068f
#define RIFT_GELU_H4 -0.32790318f
#define RIFT_GELU_H5 0.24358234f
#define RIFT_GELU_H6 0.05996109f
#define RIFT_GELU_H7 -0.08454493f
__device__ __forceinline__ void gelu_pk16x2(float x0, float x1, float x2, float x3, gelu_h2& g0, gelu_h2& g1) {
  typedef gelu_h2 V;
  // (the conversion is hipcc's own v_cvt_pk_f16_f32, NOT an asm statement: x may be a fresh MFMA result, and the wait states between an
  // MFMA write and a VALU read are software's to insert -- hipcc pads its own instructions, never the inside of an asm statement; the
  // first version read stale accumulators here and produced NaN)
  f32x2_t fa, fb;
  fa.x = x0; fa.y = x1; fb.x = x2; fb.y = x3;
  const V xa = __builtin_convertvector(fa, V), xb = __builtin_convertvector(fb, V);
  const V HALF = (V)(_Float16)0.5f, M1 = (V)(_Float16)-1.0f;
  V aa, ab, ra, rb;
  // (asm: hipcc's own min / max carry an IEEE canonicalisation -- v_pk_max_f16 x, x, x -- in the units built without -mno-amdgpu-ieee)
  asm("v_and_b32 %0, 0x7fff7fff, %1" : "=v"(aa) : "v"(xa));
  asm("v_and_b32 %0, 0x7fff7fff, %1" : "=v"(ab) : "v"(xb));
  // (inline constants: the f16 constant sits in the low half of the operand, op_sel_hi 0 hands it to both halves -- no register)
  asm("v_pk_min_f16 %0, %1, 4.0 op_sel_hi:[1,0]" : "=v"(aa) : "v"(aa));
  asm("v_pk_min_f16 %0, %1, 4.0 op_sel_hi:[1,0]" : "=v"(ab) : "v"(ab));
  asm("v_pk_max_f16 %0, %1, 0 op_sel_hi:[1,0]" : "=v"(ra) : "v"(xa));
  asm("v_pk_max_f16 %0, %1, 0 op_sel_hi:[1,0]" : "=v"(rb) : "v"(xb));
  const V ua = __builtin_elementwise_fma(aa, HALF, M1), ub = __builtin_elementwise_fma(ab, HALF, M1);
  V pa = __builtin_elementwise_fma(ua, (V)(_Float16)RIFT_GELU_H7, (V)(_Float16)RIFT_GELU_H6);
  V pb = __builtin_elementwise_fma(ub, (V)(_Float16)RIFT_GELU_H7, (V)(_Float16)RIFT_GELU_H6);
#define RIFT_GELU_STEP(C) pa = __builtin_elementwise_fma(pa, ua, (V)(_Float16)C); pb = __builtin_elementwise_fma(pb, ub, (V)(_Float16)C);
  RIFT_GELU_STEP(RIFT_GELU_H5) RIFT_GELU_STEP(RIFT_GELU_H4) RIFT_GELU_STEP(RIFT_GELU_H3)
  RIFT_GELU_STEP(RIFT_GELU_H2) RIFT_GELU_STEP(RIFT_GELU_H1) RIFT_GELU_STEP(RIFT_GELU_H0)
#undef RIFT_GELU_STEP
  g0 = ra - pa; g1 = rb - pb;
}

// ---- MLP hidden-layer operands (opfmt.h: fp16 words whenever GELU is evaluated in packed fp16) ----
__device__ __forceinline__ f32x4 mfma_hid(h16x8 a, h16x8 b, f32x4 c) {
#if RIFT_GELU_PK16
  typedef _Float16 v8 __attribute__((ext_vector_type(8)));
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
#else
  return mfma_h(a, b, c, 0, 0, 0);
#endif
}
// fp32 -> hidden-layer operand word (weight packers: the fc2 fragments)
__device__ __forceinline__ unsigned short f2h_hid(float f) {
#if RIFT_GELU_PK16
  unsigned int r;
  asm("v_cvt_pk_f16_f32 %0, %1, %2\n\ts_nop 0" : "=v"(r) : "v"(f), "v"(0.f));
  return (unsigned short)(r & 0xffffu);
#else
  return f2h(f);
#endif
}
// fc1 accumulator -> hidden-layer operand words (4 consecutive hidden channels of one row).  Packed-fp16 build: the bias was the
// accumulator's initial value (hid_init) and the words are fp16; otherwise the arithmetic of rounds 2 - 4, bit for bit (zero initial value,
// bias added here): the fp16-operand build's results do not move with this round's change.
__device__ __forceinline__ f32x4 hid_init(const float4 b) {
#if RIFT_GELU_PK16
  return (f32x4){b.x, b.y, b.z, b.w};
#else
  return (f32x4){0.f, 0.f, 0.f, 0.f};
#endif
}
__device__ __forceinline__ uint2 gelu4_pack(const f32x4 a, const float4 b);
__device__ __forceinline__ uint2 gelu4_hid(const f32x4 a, const float4 b) {
#if RIFT_GELU_PK16
  gelu_h2 g0, g1;
  gelu_pk16x2(a[0], a[1], a[2], a[3], g0, g1);
  uint2 u; u.x = __builtin_bit_cast(unsigned int, g0); u.y = __builtin_bit_cast(unsigned int, g1);
  return u;
#else
  return gelu4_pack(a, b);
#endif
}

// bias + GELU + operand pack of one MFMA accumulator fragment (4 consecutive output columns)
__device__ __forceinline__ uint2 gelu4_pack(const f32x4 a, const float4 b) {
#if !RIFT_GELU_PK16
  f32x2_t lo, hi, bl, bh;
  lo.x = a[0]; lo.y = a[1]; hi.x = a[2]; hi.y = a[3];
  bl.x = b.x; bl.y = b.y; bh.x = b.z; bh.y = b.w;
  lo = gelu_fast2(lo + bl); hi = gelu_fast2(hi + bh);
  return pack_h4(lo.x, lo.y, hi.x, hi.y);
#else
  gelu_h2 g0, g1;
  gelu_pk16x2(a[0] + b.x, a[1] + b.y, a[2] + b.z, a[3] + b.w, g0, g1);
#if RIFT_OP_F16
  uint2 u; u.x = __builtin_bit_cast(unsigned int, g0); u.y = __builtin_bit_cast(unsigned int, g1);      // (the result words ARE the build's operand format)
  return u;
#else
  return pack_h4((float)g0[0], (float)g0[1], (float)g1[0], (float)g1[1]);
#endif
#endif
}

// Workgroup barrier for phases that hand data over through LDS only.  __syncthreads() makes hipcc drain EVERY outstanding memory
// operation first (s_waitcnt vmcnt(0) lgkmcnt(0)), which ends the flight of the weight / parameter / next-tile loads these kernels
// issue a phase ahead; here only the LDS counter is drained, so global loads stay in flight across the barrier until their first use.
// Not for phases that exchange data through global memory.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Cross-lane sums.  __shfl_xor lowers to ds_bpermute_b32 (an LDS-crossbar round trip, ~100 cycles of dependent
// latency per step); within a 16-lane row the same exchange is a DPP modifier on a VALU op (a few cycles):
// quad_perm[1,0,3,2], quad_perm[2,3,0,1], row_half_mirror, row_mirror.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float sum4(float v) { v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); return v; }
__device__ __forceinline__ float sum8(float v) { v = sum4(v); v += dpp_f<0x141>(v); return v; }
__device__ __forceinline__ float sum16(float v) { v = sum8(v); v += dpp_f<0x140>(v); return v; }
// Across the four 16-lane rows gfx950 has half-exchange instructions: v_permlane16_swap (odd rows of vdst <-> even rows of src)
// and v_permlane32_swap (upper half of vdst <-> lower half of src).  With vdst = src = v the two results hold v[lane] and
// v[lane ^ 16] (resp. v[lane ^ 32]) in some order in every lane, so a commutative combine of them is the xor exchange -- one
// VALU-rate instruction instead of a ds_bpermute round trip.
__device__ __forceinline__ float xadd16(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xadd32(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xmax16(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xmax32(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float rows_sum(float v) { return xadd32(xadd16(v)); }   // over the 4 lanes {l, l^16, l^32, l^48}
// Guard of the one-pass LayerNorm statistics (opfmt.h: RIFT_LN_FOLD).  var = E[x^2] - mean^2 in fp32 loses log2(1 + mean^2 / var) of its 24
// bits: nothing on the rows this model produces (|mean| of the order of the spread), but a row whose mean dwarfs its spread -- a
// checkpoint with an outlier channel offset -- would get a variance of noise where torch's two-pass form is exact.  m2 = mean^2,
// var1 = E[x^2] - m2: true (wave-uniform) when some row of the wave has lost more than 12 bits (|mean| > 64 sigma, or var1 <= 0 beside
// a non-zero mean: at least 12 of fp32's 24 bits are left otherwise, one more than an fp16 operand keeps); the caller then takes the
// variance of that LayerNorm from the centred values instead.  Two instructions and a scalar branch per row tile on the fast path.
#ifdef RIFT_LN_NO_GUARD      // (diagnostic build define: the unguarded one-pass form of round 5, for A/B runs)
__device__ __forceinline__ bool ln_row_cancels(float, float) { return false; }
__device__ __forceinline__ bool ln_cancels(float, float) { return false; }
#else
__device__ __forceinline__ bool ln_row_cancels(float m2, float var1) { return m2 > 4096.0f * var1; }      // (per lane row; the caller ORs over its row tiles and ballots once)
__device__ __forceinline__ bool ln_cancels(float m2, float var1) { return __builtin_amdgcn_ballot_w64(m2 > 4096.0f * var1) != 0ull; }
#endif
__device__ __forceinline__ float rows_max(float v) { return xmax32(xmax16(v)); }
__device__ __forceinline__ float sum32(float v) { v = sum16(v); return xadd16(v); }
// sum over groups of LPR consecutive lanes (LPR = 8, 16, 32, 64), result in every lane of the group
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
  if (LPR == 4) return sum4(v);
  if (LPR == 8) return sum8(v);
  if (LPR == 16) return sum16(v);
  v = sum32(v);
  if (LPR == 64) v = xadd32(v);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); }
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// LayerNorm of one fp32 row of C = 8 * LPR columns held by LPR adjacent lanes (lr = lane % LPR; LPR = 4, 8, 16).  A lane owns columns
// [4 lr, 4 lr + 4) and [C/2 + 4 lr, C/2 + 4 lr + 4): both 16-byte reads are then contiguous across the lanes of the row (a lane-stride of
// 32 bytes would hit every LDS bank twice).  g0/b0 and g1/b1 are the scale / shift values of those two column groups:
// the two reductions are DPP row sums (no ds_bpermute), the element-wise math is packed fp32 (two columns per instruction), the
// result goes out as 2 x 4 bf16.
template <int LPR>
__device__ __forceinline__ void ln_row8(const float* __restrict__ src, unsigned short* __restrict__ dst, const float4& g0, const float4& g1,
                                        const float4& b0, const float4& b1, int lr, bool relu = false) {
  typedef f32x2_t V;
  constexpr float invC = 1.0f / (8 * LPR);
  const float4 v0 = *reinterpret_cast<const float4*>(src + lr * 4), v1 = *reinterpret_cast<const float4*>(src + 4 * LPR + lr * 4);
  V a, b, c, d;
  a.x = v0.x; a.y = v0.y; b.x = v0.z; b.y = v0.w; c.x = v1.x; c.y = v1.y; d.x = v1.z; d.y = v1.w;
  const V s2 = (a + b) + (c + d);
  const V m2 = (V)(group_sum<LPR>(s2.x + s2.y) * invC);
  a -= m2; b -= m2; c -= m2; d -= m2;
  V q2 = a * a;
  q2 = __builtin_elementwise_fma(b, b, q2); q2 = __builtin_elementwise_fma(c, c, q2); q2 = __builtin_elementwise_fma(d, d, q2);
  const V r2 = (V)rsqrtf(group_sum<LPR>(q2.x + q2.y) * invC + 1e-5f);
  V ga, gb, gc, gd, ba, bb, bc, bd;
  ga.x = g0.x; ga.y = g0.y; gb.x = g0.z; gb.y = g0.w; gc.x = g1.x; gc.y = g1.y; gd.x = g1.z; gd.y = g1.w;
  ba.x = b0.x; ba.y = b0.y; bb.x = b0.z; bb.y = b0.w; bc.x = b1.x; bc.y = b1.y; bd.x = b1.z; bd.y = b1.w;
  a = __builtin_elementwise_fma(a * r2, ga, ba); b = __builtin_elementwise_fma(b * r2, gb, bb);
  c = __builtin_elementwise_fma(c * r2, gc, bc); d = __builtin_elementwise_fma(d * r2, gd, bd);
  if (relu) {
    const V z = (V)0.f;
    a = __builtin_elementwise_max(a, z); b = __builtin_elementwise_max(b, z); c = __builtin_elementwise_max(c, z); d = __builtin_elementwise_max(d, z);
  }
  *reinterpret_cast<uint2*>(dst + lr * 4) = make_uint2(pack_h2(a.x, a.y), pack_h2(b.x, b.y));
  *reinterpret_cast<uint2*>(dst + 4 * LPR + lr * 4) = make_uint2(pack_h2(c.x, c.y), pack_h2(d.x, d.y));
}
__device__ __forceinline__ void ln128_row16(const float* __restrict__ src, unsigned short* __restrict__ dst, const float4& g0, const float4& g1,
                                            const float4& b0, const float4& b1, int l15, bool relu = false) {
  ln_row8<16>(src, dst, g0, g1, b0, b1, l15, relu);
}

// ---- register-resident (wave-private) kernels: nat_l0w.h, nat_l1w.h, dec_w.hip ----
// channel a lane's k-slot j (0..7) of chunk l4 stands for when the operand is a GEMM output kept in the C/D layout (two n-tiles)
__host__ __device__ __forceinline__ int l0w_chan(int l4, int j, int nt_lo) { return (j < 4 ? nt_lo : nt_lo + 1) * 16 + l4 * 4 + (j & 3); }
__device__ __forceinline__ h16x8 l0w_pack8(const f32x4 a, const f32x4 b) {
  h16x8 r;
  const unsigned int p0 = pack_h2(a[0], a[1]), p1 = pack_h2(a[2], a[3]), p2 = pack_h2(b[0], b[1]), p3 = pack_h2(b[2], b[3]);
  r[0] = (short)(p0 & 0xffff); r[1] = (short)(p0 >> 16); r[2] = (short)(p1 & 0xffff); r[3] = (short)(p1 >> 16);
  r[4] = (short)(p2 & 0xffff); r[5] = (short)(p2 >> 16); r[6] = (short)(p3 & 0xffff); r[7] = (short)(p3 >> 16);
  return r;
}
__device__ __forceinline__ h16x8 l0w_from_u2(const uint2 a, const uint2 b) {
  h16x8 r;
  r[0] = (short)(a.x & 0xffff); r[1] = (short)(a.x >> 16); r[2] = (short)(a.y & 0xffff); r[3] = (short)(a.y >> 16);
  r[4] = (short)(b.x & 0xffff); r[5] = (short)(b.x >> 16); r[6] = (short)(b.y & 0xffff); r[7] = (short)(b.y >> 16);
  return r;
}

// ---- sequence bookkeeping of the compacted history-encoder launch (nat_l0w.h ranks the sequences; agent_encoder.py:77-87).
// Rank r = 3 i + c is the i-th marked agent slot of residue class c = slot % 3: a sequence keeps its position (r % 3 == slot % 3) in level 2's
// three-agent tiles, which makes the compacted launch bit-identical to the uncompacted one (that level's P V MFMA sums a query's five keys at
// k offset 5 (r % 3) -- an ulp-level dependence on the position, a rare bf16 flip downstream).  Ranks at or beyond their class's count are
// holes (at most a few, at the end); cnt = the three class counts in device memory, nullptr for the plain launch over nseq sequences
// ("SeqCount" below: sq_n = the number of ranks, sq_c0..2 = the class counts).
// (plain ints, not a struct: hipcc kept a struct of them in scratch memory)
#define RIFT_SEQ_COUNT(cnt, nseq)                                                                        \
  const int sq_c0 = (cnt) ? __builtin_amdgcn_readfirstlane((cnt)[0]) : 0x7fffffff;                       \
  const int sq_c1 = (cnt) ? __builtin_amdgcn_readfirstlane((cnt)[1]) : 0x7fffffff;                       \
  const int sq_c2 = (cnt) ? __builtin_amdgcn_readfirstlane((cnt)[2]) : 0x7fffffff;                       \
  const int sq_n = (cnt) ? 3 * max(sq_c0, max(sq_c1, sq_c2)) : (nseq)
#define RIFT_SEQ_LIVE(seq) seq_live(sq_n, sq_c0, sq_c1, sq_c2, (seq))
__device__ __forceinline__ bool seq_live(int n, int c0, int c1, int c2, int seq) {
  const int i = seq / 3, c = seq - 3 * i;
  return seq < n && i < (c == 0 ? c0 : c == 1 ? c1 : c2);
}

// Finiteness by bit pattern: a float is NaN / +-Inf iff its exponent field is all ones.  Integer compares, so the test also holds in the
// translation units built with -fno-honor-nans (where x != x folds to false) -- used for the device flag behind the reference's
// `assert torch.isfinite(q).all()` (planning_decoder.py:175).  exp_or accumulates the largest exponent field seen; nonfinite_exp tests it.
__device__ __forceinline__ uint32_t exp_max(uint32_t acc, float v) { const uint32_t e = __float_as_uint(v) & 0x7f800000u; return acc > e ? acc : e; }
__device__ __forceinline__ bool nonfinite_exp(uint32_t acc) { return acc == 0x7f800000u; }

}  // namespace RIFT_NS
