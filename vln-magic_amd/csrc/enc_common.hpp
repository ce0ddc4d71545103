// Shared device helpers of the whole-encoder kernels (encoder.hip: forward; encbwd.hip: the backward's per-token chain).
#pragma once
#include "common.hpp"

#define EH 128
#define EI 512
#define ENH 2
#define EHD 64
#define XS 136        // row pitch (elements) of the [rows][128] LDS images: 272 B = 17 16-byte slots, rows land on distinct slots
#define QS 392        // [rows][384] Q|K|V image
#define GS 520        // [rows][512] GELU output image
#define PSW 104       // per-wave probability tile [16][<= 96 keys]
#define MAXROWS 80
#define KROWS 96      // key rows the Q|K|V image provides for (PV consumes keys in steps of 32)
#define NWAVE 8

// the GEMM stages are fully unrolled (their weight fragments are register arrays with static indices); without a fence per k-step the
// scheduler hoists every LDS fragment read of a stage to its top and spills hundreds of registers
#define KSTEP_FENCE() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ f32x4 emma(bf16x8 a, bf16x8 b, f32x4 c) { return mfma16(a, b, c); }
__device__ __forceinline__ f32x4 emma(f16x8 a, f16x8 b, f32x4 c) { return mfma16(a, b, c); }
// MFMA operand fragment (A or B) of 16 rows x 32 k from a k-contiguous image: lane l holds X[row0 + (l&15)][k0 + 8*(l>>4) .. +7]
template <typename Hh> __device__ __forceinline__ h16x8<Hh> lfrag(const Hh* s, int pitch, int row0, int k0, int lane) {
  return *(const h16x8<Hh>*)(s + (row0 + (lane & 15)) * pitch + k0 + 8 * (lane >> 4));
}
// B fragment of a WEIGHT matrix [N, ldw] held in MFMA-FRAGMENT ORDER (magic_pack_frag_spans, csrc/chain.hip): fragment (row0 / 16, k0 / 32) is
// one contiguous KB, lane l's 16 bytes at l.  Round 3: the row-major form (lane l reads 16 bytes of weight row l & 15: one cache line per
// row, half of it used) held the whole-encoder forward at 249 us; in fragment order the same launch takes 200 us (profiles/README.md).
// ENC_ROWMAJOR_W restores the row-major read for that comparison (profiles/micro/enc_rs_timing.hip).
template <typename Hh> __device__ __forceinline__ h16x8<Hh> gfrag(const Hh* __restrict__ W, int ldw, int row0, int k0, int lane) {
#ifdef ENC_ROWMAJOR_W
  return *(const h16x8<Hh>*)(W + (long long)(row0 + (lane & 15)) * ldw + k0 + 8 * (lane >> 4));
#else
  return *(const h16x8<Hh>*)(W + ((long long)((row0 >> 4) * (ldw >> 5) + (k0 >> 5)) * 64 + lane) * 8);
#endif
}
// B fragment from a [k][n] image (V: keys x head dims), transposed on the way out of LDS
template <typename Hh> __device__ __forceinline__ h16x8<Hh> tfrag(const Hh* s, int pitch, int n0, int k0, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  return lds_tr8(s + (k0 + 8 * g + q) * pitch + n0 + 4 * pp, 4 * pitch);
}
// GELU on the 40 K elements a text workgroup produces per layer: libm's erff is ~40 instructions; this rational form (Abramowitz &
// Stegun 7.1.26, |error| <= 1.5e-7) is ~15 and indistinguishable after the bf16 rounding of the result
__device__ __forceinline__ float gelu_fast(float x) {
  const float ax = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.0f - poly * __expf(-ax * ax);
  const float erfv = x < 0.f ? -erf_abs : erf_abs;
  return 0.5f * x * (1.0f + erfv);
}
// LDS hand-off between the lanes of ONE wave (wave-private tile): order the wave's own LDS writes before its reads
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ float g16_sum(float v) { return row16_sum(v); }
__device__ __forceinline__ float g16_max(float v) { return row16_max(v); }

// cooperative copy of `rows` x `cols` bf16 from an LDS image to global rows (16-byte vectors)
template <typename Hh> __device__ __forceinline__ void copy_out(const Hh* s, int pitch, Hh* g, long long ldg, int rows, int cols, int tid) {
  const int cpr = cols / 8;
  for (int id = tid; id < rows * cpr; id += NWAVE * 64) {
    const int r = id / cpr, c = (id % cpr) * 8;
    *(h16x8<Hh>*)(g + (long long)r * ldg + c) = *(const h16x8<Hh>*)(s + r * pitch + c);
  }
}

