// Shared device helpers for the gfx950 (MI355X, CDNA4) kernels. wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
// second 16-bit storage type: IEEE half (11 significand bits against bf16's 8: ~8x less rounding per stored activation / weight; the
// same MFMA rate, v_mfma_f32_16x16x32_f16).  Every 16-bit kernel is written once over Hh in {bf16, f16}.
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define MAGIC_OK 0
#define MAGIC_ERR_ARG -1
#define MAGIC_ERR_LAUNCH -2
#define MAGIC_ERR_UNSUPPORTED -3

#define DT_F32 0
#define DT_BF16 1
#define DT_F16 2
static inline bool dtype_ok(int dt) { return dt == DT_F32 || dt == DT_BF16 || dt == DT_F16; }
static inline bool dtype_is16(int dt) { return dt == DT_BF16 || dt == DT_F16; }

#define WAVE 64

__device__ __forceinline__ float to_f(float x) { return x; }
__device__ __forceinline__ float to_f(bf16 x) { return (float)x; }
__device__ __forceinline__ float to_f(f16 x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f(float x);
template <> __device__ __forceinline__ float from_f<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float x) { return (bf16)x; }
template <> __device__ __forceinline__ f16 from_f<f16>(float x) { return (f16)x; }

// MFMA operand vectors of a 16-bit type, the 16x16x32 product, and the transposing LDS read (ds_read_b64_tr_b16: each lane of a 16-lane
// group receives element (l & 15) of 4 consecutive rows) -- the only places where bf16 and f16 need different instructions / builtins
template <typename Hh> struct H16;
template <> struct H16<bf16> { typedef bf16x8 v8; typedef bf16x4 v4; };
template <> struct H16<f16> { typedef f16x8 v8; typedef f16x4 v4; };
template <typename Hh> using h16x8 = typename H16<Hh>::v8;
template <typename Hh> using h16x4 = typename H16<Hh>::v4;
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ bf16x4 lds_tr4(const bf16* p) {
  typedef bf16x4 __attribute__((address_space(3))) * lds4;
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)p);
}
__device__ __forceinline__ f16x4 lds_tr4(const f16* p) {
  typedef s16x4 __attribute__((address_space(3))) * lds4;
  return __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)p));
}
// two transposed reads `pitch4` elements (= 4 rows) apart, concatenated: the 8 k-values of one lane's MFMA operand
template <typename Hh> __device__ __forceinline__ h16x8<Hh> lds_tr8(const Hh* p, int pitch4) {
  const h16x4<Hh> lo = lds_tr4(p), hi = lds_tr4(p + pitch4);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Cross-lane reductions as DPP moves (VALU data-parallel primitives: quad permutes and row mirrors inside a 16-lane row), not
// __shfl_xor: hipcc lowers every __shfl_xor to ds_bpermute_b32, a round trip through the LDS crossbar (~100 cycles, each step of
// a butterfly waiting on the previous one) -- measured: the LayerNorm epilogue of the whole-encoder kernel spent 9.4 of its 10.3 us
// in 160 such shuffles.  The 16-lane butterfly below is bit-identical to the xor 1,2,4,8 shuffle sequence (after each step all
// lanes of a group hold the same value, and a + b == b + a).
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
#define DPP_QUAD_XOR1 0xB1         /* quad_perm [1,0,3,2] */
#define DPP_QUAD_XOR2 0x4E         /* quad_perm [2,3,0,1] */
#define DPP_ROW_HALF_MIRROR 0x141
#define DPP_ROW_MIRROR 0x140
__device__ __forceinline__ float row16_sum(float v) {      // every lane of a 16-lane row gets the row's sum
  v += dpp_mov<DPP_QUAD_XOR1>(v);
  v += dpp_mov<DPP_QUAD_XOR2>(v);
  v += dpp_mov<DPP_ROW_HALF_MIRROR>(v);
  v += dpp_mov<DPP_ROW_MIRROR>(v);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_mov<DPP_QUAD_XOR1>(v));
  v = fmaxf(v, dpp_mov<DPP_QUAD_XOR2>(v));
  v = fmaxf(v, dpp_mov<DPP_ROW_HALF_MIRROR>(v));
  v = fmaxf(v, dpp_mov<DPP_ROW_MIRROR>(v));
  return v;
}
__device__ __forceinline__ float lane_bcast(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
  v = row16_max(v);
  return fmaxf(fmaxf(lane_bcast(v, 0), lane_bcast(v, 16)), fmaxf(lane_bcast(v, 32), lane_bcast(v, 48)));
}

// exact (erf) GELU, as HF "gelu" / torch F.gelu default
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_f(float x) {
  float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// Counter-based dropout: keep(site, idx) is a pure function of (seed[0], seed[1], site, logical element index), so the
// forward kernel and the backward kernel regenerate the same mask without ever storing it.  `seed` lives in device
// memory (a replayed HIP graph sees whatever the host-side generator wrote there this step); `site` distinguishes the
// dropout modules of the network; idx is the row-major index in the LOGICAL tensor (no padding columns).
// One full-avalanche 32-bit finaliser (murmur3 fmix32) over (idx ^ ka) + kb.  (Round 2 started with a second, differently keyed mixing round behind
// it: the whole-encoder kernels hash ~33 k elements per sample and layer, and the second round cost 11 of the forward kernel's 305 us and ~30 us
// of the step for no measurable change in the mask statistics -- tests/test_dropout_gpu.py::test_mask_statistics_and_determinism; -DMAGIC_DROP_TWO_ROUNDS
// restores it.)
struct DropDesc { const unsigned* seed; unsigned site; float p; };       // seed == nullptr or p <= 0: off
struct DropState { unsigned ka, kb, thr; float scale; bool on; };
__device__ __forceinline__ DropState drop_init(const DropDesc& d) {
  DropState s;
  s.on = d.seed != nullptr && d.p > 0.f;
  s.ka = 0; s.kb = 0; s.thr = 0; s.scale = 1.f;
  if (s.on) {
    s.ka = d.seed[0] ^ (d.site * 0x9E3779B1u);
    s.kb = d.seed[1] + d.site * 0x85EBCA77u;
    s.thr = (unsigned)((double)d.p * 4294967296.0);
    s.scale = 1.0f / (1.0f - d.p);
  }
  return s;
}
__device__ __forceinline__ float drop_mul(const DropState& s, unsigned idx) {   // 0 (dropped) or 1/(1-p) (kept)
#ifdef MAGIC_DROP_TWO_ROUNDS
  unsigned x = idx ^ s.ka;
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  x += s.kb;
  x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
#else
  unsigned x = (idx ^ s.ka) + s.kb;            // both keys in front of ONE full-avalanche finaliser (murmur3 fmix32)
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
#endif
  return x >= s.thr ? s.scale : 0.f;
}
static inline bool drop_args_ok(const void* seed, float p) { return p >= 0.f && p < 1.f && (p == 0.f || seed != nullptr); }

// the device word every gradient-seeding loss kernel multiplies its gradient coefficient by (loss.hip magic_seed_scale; NULL: none)
const float* seed_scale_get();

static inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MAGIC_OK : MAGIC_ERR_LAUNCH;
}

// dst_j[c] += sum_b part_j[b][c], b < nblk_j, c < H, in block order (reproducible): the finisher of the partial-row LayerNorm gradients of
// magic_rowbwd (csrc/encbwd.hip), as a standalone launch (magic_colsum_add) or as extra workgroups of magic_embed_in_bwd.  One workgroup per
// job; blockDim.x / H row groups take every (blockDim.x / H)-th block, LDS fold.  `red`: blockDim.x floats of LDS.
#define CSJ_MAX 96
struct ColsumJobs { const float* part[CSJ_MAX]; float* dst[CSJ_MAX]; int nblk[CSJ_MAX]; int n; };
__device__ __forceinline__ void colsum_body(const ColsumJobs& js, const int j, const int H, float* red) {
  const int t = threadIdx.x, c = t % H, grp = t / H, ngrp = blockDim.x / H;
  const float* part = js.part[j];
  const int nb = js.nblk[j];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (grp < ngrp) {
    int b = grp;
    for (; b + 3 * ngrp < nb; b += 4 * ngrp) {
      s0 += part[(long long)b * H + c]; s1 += part[(long long)(b + ngrp) * H + c];
      s2 += part[(long long)(b + 2 * ngrp) * H + c]; s3 += part[(long long)(b + 3 * ngrp) * H + c];
    }
    for (; b < nb; b += ngrp) s0 += part[(long long)b * H + c];
  }
  red[t] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (grp == 0) {
    float v = 0.f;
    for (int g2 = 0; g2 < ngrp; ++g2) v += red[g2 * H + c];
    js.dst[j][c] += v;
  }
}
