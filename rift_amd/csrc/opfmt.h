// Build parameter of the engine: the 16-bit MFMA operand format and the namespace its kernels live in.
#pragma once

// ---- the 16-bit MFMA operand format is a BUILD parameter ------------------------------------------------------------------
// Every fused kernel multiplies 16-bit operands with fp32 accumulation (v_mfma_f32_16x16x32_{bf16,f16}: same rate, same bytes).  The
// library carries two builds of the whole engine, one per format, and a context selects one at creation (abi.cpp):
//   RIFT_OP_F16 = 0 : bf16 operands (8 significand bits)  -- BASELINE's "bf16 MFMA", the benchmarked default
//   RIFT_OP_F16 = 1 : fp16 operands (11 significand bits) -- the mode that holds north_star's 1e-4 on losses / advantages with the same
//                     instruction count; range 6e-8 .. 65504 is enough for what the operands are here (post-LayerNorm / post-BatchNorm
//                     activations, attention probabilities, metre-scale geometry, |weights| < 10); an overflow becomes inf and is
//                     reported by the non-finite flag (rift_check_finite).
// "h16" below means "the build's 16-bit operand format"; weight images, LDS tiles and hand-over buffers hold it as raw 16-bit words.
#ifndef RIFT_OP_F16
#define RIFT_OP_F16 0
#endif
#if RIFT_OP_F16
#define RIFT_NS rift_hf
#define RIFT_MFMA_H_ASM "v_mfma_f32_16x16x32_f16"
#define RIFT_CVT_PK_H_ASM "v_cvt_pk_f16_f32"
#else
#define RIFT_NS rift_bf
#define RIFT_MFMA_H_ASM "v_mfma_f32_16x16x32_bf16"
#define RIFT_CVT_PK_H_ASM "v_cvt_pk_bf16_f32"
#endif

// ---- the MLP hidden layer of the history encoder (NAT levels) is an fp16 operand in BOTH builds (round 5) -------------------------------
// GELU is evaluated in packed fp16 in the bf16 build (common.h: gelu_pk16x2) and its result is left as it comes out: an fp16 operand word
// (11 significand bits against bf16's 8; same MFMA rate) -- no conversion back to bf16, and the fc2 weight fragments are packed as fp16 to
// match (|w| < 10: far inside the format's range).  RIFT_GELU_F32 (diagnostic build define) restores the fp32 rational GELU and bf16 words.
// Round 6: the fp16-operand build evaluates the same packed-fp16 form (it kept the fp32 rational GELU through round 5: 36 issue slots per four
// elements against 26, the whole of its +12 % step time over the bf16 build).  RIFT_F16_GELU_F32 (diagnostic define) restores the rational
// form in the fp16 build.
#if !defined(RIFT_GELU_F32) && !(RIFT_OP_F16 && defined(RIFT_F16_GELU_F32))
#define RIFT_GELU_PK16 1
#else
#define RIFT_GELU_PK16 0
#endif
#if RIFT_OP_F16 || RIFT_GELU_PK16
#define RIFT_MFMA_HID_ASM "v_mfma_f32_16x16x32_f16"
#else
#define RIFT_MFMA_HID_ASM "v_mfma_f32_16x16x32_bf16"
#endif

// ---- neighbourhood attention of NAT levels 0 / 1 on the matrix pipe (round 5; nat_l0w.h, nat_l1w.h) ------------------------------------
// q, k, v and the attention weights are 16-bit MFMA operands (as in level 2 and in the scene encoder / decoder).  Both builds since the end
// of round 5: the fp16 build first kept the arithmetic of rounds 2 - 4 bit for bit (fp32 VALU attention, two-pass LayerNorm, slot-ordered
// encoder keys) because its bars were measured on it; they were re-measured on this arithmetic (tests/test_gpu_parity.py header:
// the benchmark batch's four objectives stay inside 1e-4, the small-batch envelope is the same with another objective carrying its
// maximum) and the fp16 step went 0.655 -> 0.621 ms.  RIFT_F16_R4 (diagnostic build define) restores the old arithmetic of the fp16
// build as a whole; RIFT_NAT_VALU_ATTN restores the VALU attention in either build.
#if !(RIFT_OP_F16 && defined(RIFT_F16_R4)) && !defined(RIFT_NAT_VALU_ATTN)
#define RIFT_NAT_MFMA_ATTN 1
#else
#define RIFT_NAT_MFMA_ATTN 0
#endif

// ---- valid-token compaction inside the scene encoder (round 5; enc_fused.h) ------------------------------------------------------------
// A scene's valid tokens are moved to the front of its LDS rows (stable order), row / key tiles behind the last valid token
// are skipped, the output rows go back to their slots.  Another key order = another fp32 summation order (not bit-identical to the slot
// order).  RIFT_ENC_SLOT_ORDER (diagnostic define) keeps the slot order.
#if !(RIFT_OP_F16 && defined(RIFT_F16_R4)) && !defined(RIFT_ENC_SLOT_ORDER)
#define RIFT_ENC_COMPACT 1
#else
#define RIFT_ENC_COMPACT 0
#endif

// ---- LayerNorm in front of a linear layer: affine part folded into that layer, one-pass statistics (round 5) ---------------------------
// Every LayerNorm of the pre-norm blocks (NAT levels, scene encoder, planning decoder) feeds linear layers only, and its gamma / beta are
// frozen (only pi_head trains): W (g * n + b) + c = (W diag g) n + (W b + c), so the packers fold gamma into the weight image and beta into
// the bias, and the kernel computes n = x * r - mean * r with r = rsqrt(E[x^2] - mean^2 + eps) -- both sums in one pass over the row, one
// packed FMA per two elements behind them: ~3.5 issue slots per element where the two-pass form with its affine part took ~5.5, and the
// two cross-lane reductions no longer wait for each other.  E[x^2] - mean^2 in fp32 loses log2(1 + mean^2 / var) of 24 bits -- nothing
// against the 8 / 11 bits the result is rounded to -- and is clamped at 0.  RIFT_LN_TWO_PASS (diagnostic define) restores the two-pass
// form with its affine part.
#if !(RIFT_OP_F16 && defined(RIFT_F16_R4)) && !defined(RIFT_LN_TWO_PASS)
#define RIFT_LN_FOLD 1
#else
#define RIFT_LN_FOLD 0
#endif

// ---- 16-key / 16-dim contractions of the register-resident attentions as K = 16 MFMAs (round 5; nat_l2w.hip, dec_w.hip) -----------------
// Level 2's heads have 16 dims and its tiles (like the decoder's self-attention tiles) 16 keys: as operands of the K = 32 MFMA they were
// half zeros -- two v_mov per fragment, a copy with a zeroed half per head, and twice the registers (V^T: 32 -> 16).  The K = 16 form
// (common.h: mfma_h16) takes the projection's C/D words as they are.  RIFT_ATTN_K32 (diagnostic define) restores the padding.
#if !(RIFT_OP_F16 && defined(RIFT_F16_R4)) && !defined(RIFT_ATTN_K32)
#define RIFT_ATTN_K16 1
#else
#define RIFT_ATTN_K16 0
#endif
