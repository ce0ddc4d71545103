// MFMA GEMM for gfx950: C = epilogue(alpha * op(A) @ op(B) + bias) [+ residual]
//
// One kernel family covers every dense contraction on the path (linear fwd = NT, dX = NN,
// dW = TN with split-K + fp32 atomics and fused bias-gradient, QK^T / PV / attention backward as
// batched instances with (batch, head) strides).  64x64 block tile, 4 waves (2x2), each wave a 32x32
// sub-tile = 2x2 MFMA 16x16 tiles; bf16 uses v_mfma_f32_16x16x32_bf16, fp32 ("parity mode") uses
// the exact v_mfma_f32_16x16x4_f32.  Both operand tiles are staged into LDS as [out-dim][k]
// (k contiguous, padded rows) whatever their global orientation, so fragments are plain 16-byte
// (bf16) / 4-byte (f32) LDS reads; the next tile's global loads are in flight during the MFMAs.
#include "common.hpp"
#include "group.hpp"
#include <cstdlib>

#define BM 64
#define BN 64

struct GemmParams {
  const void* A; const void* B; void* C; void* C2;
  const float* bias; const void* aux; const void* residual; float* bias_grad;
  int M, N, K;
  int lda, ldb, ldc, ldc2, ldaux, ldr;
  long long sAb, sAh, sBb, sBh, sCb, sCh;
  int nh, splitk, epilogue, c_f32, accumulate;
  float alpha;
  int batch;      // host-side only (grid z = batch * splitk)
  int big;        // host-side only: 128x128 block tile (single launches; grouped / paired launches use 64x64)
  // deterministic split-K seam of the weight-gradient launch (gemm_dw_batch_kernel; NULL: fp32 atomics): see dw_seam()
  float* ws; unsigned* cnt; int ws_slot0, cnt0, ws_total, ws_first;
  // deterministic split-K of a plain GEMM (magic_gemm with splitk < 0): split s STORES its partial into slab s of C ([splits][M][ldc] fp32, `slab` elements
  // apart) instead of adding it with fp32 atomics; the consumer sums the slabs in slab order (magic_ln_bwd_tail)
  long long slab;
};

template <typename T> struct TT;
// STRIDE: row stride of a k-contiguous operand's LDS image [out][k]; SN: row stride of an out-contiguous operand's
// natural image [k][out] (read back transposed: ds_read_b64_tr_b16 for bf16, plain dword reads for f32).
template <> struct TT<bf16> {
  static constexpr int VE = 8, BK = 64, STRIDE = 72, SN = 72;
  typedef bf16x8 vec;
};
template <> struct TT<f16> {
  static constexpr int VE = 8, BK = 64, STRIDE = 72, SN = 72;
  typedef f16x8 vec;
};
template <> struct TT<float> {
  static constexpr int VE = 4, BK = 32, STRIDE = 34, SN = 68;
  typedef f32x4 vec;
};

// TM = out rows of the tile (64, or 128 for the big-tile kernels); the natural [k][out] image of an out-contiguous operand
// then has rows of TM + (SN - 64) elements
template <typename T, bool KC, int TM = 64> struct TileLoader {
  typedef typename TT<T>::vec vec;
  static constexpr int VE = TT<T>::VE, BK = TT<T>::BK, STRIDE = TT<T>::STRIDE, SN = TT<T>::SN + (TM - 64);
  static constexpr int NV = TM / 32;
  vec v[NV];
  // tile = TM out rows x BK k's.  KC: source is [OUT][K] (k contiguous); else [K][OUT].
  __device__ __forceinline__ void load(const T* __restrict__ base, int ld, int out0, int k0, int OUT, int kend) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int id = (threadIdx.x & 255) + 256 * i;
      vec z;
#pragma unroll
      for (int e = 0; e < VE; ++e) z[e] = (T)0.0f;
      if (KC) {
        int row = id >> 3, cv = id & 7;
        int o = out0 + row, k = k0 + cv * VE;
        if (o < OUT && k < kend) z = *(const vec*)(base + (long long)o * ld + k);
      } else {
        constexpr int VPR = TM / VE;
        int r = id / VPR, ov = id % VPR;
        int k = k0 + r, o = out0 + ov * VE;
        if (k < kend && o < OUT) z = *(const vec*)(base + (long long)k * ld + o);
      }
      v[i] = z;
    }
  }
  __device__ __forceinline__ void store(T* s) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int id = (threadIdx.x & 255) + 256 * i;
      if (KC) {
        int row = id >> 3, cv = id & 7;
        T* d = s + row * STRIDE + cv * VE;
        if constexpr (sizeof(T) == 2) {
          *(vec*)d = v[i];
        } else {   // f32 rows are 8-byte aligned only
          ((float2*)d)[0] = make_float2(v[i][0], v[i][1]);
          ((float2*)d)[1] = make_float2(v[i][2], v[i][3]);
        }
      } else {
        constexpr int VPR = TM / VE;
        int r = id / VPR, ov = id % VPR;
        *(vec*)(s + r * SN + ov * VE) = v[i];          // natural [k][out] image, 16-byte store
      }
    }
  }
};

// one MFMA operand fragment of the 16 out-rows starting at out0, k-step ks.  Lane map (A and B alike):
// lane l holds X[out0 + (l&15)][k = kbase + 8*(l>>4) + j] (bf16, j<8) / X[out0 + (l&15)][k = kbase + (l>>4)] (f32).
template <bool KC, int SN_, typename Hh>
__device__ __forceinline__ h16x8<Hh> frag16(const Hh* s, int out0, int ks, int lane) {
  if constexpr (KC) {
    return *(const h16x8<Hh>*)(s + (out0 + (lane & 15)) * TT<Hh>::STRIDE + ks * 32 + 8 * (lane >> 4));
  } else {
    // natural [k][out] image: each 16-lane group g transposes rows k = 8g..8g+3 (+4) x 16 outs with ds_read_b64_tr_b16;
    // lane 4q+p of the group addresses row q, outs 4p..4p+3 and receives out (l&15) of the 4 rows
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    return lds_tr8(s + (ks * 32 + 8 * g + q) * SN_ + out0 + 4 * pp, 4 * SN_);
  }
}
template <bool KC, int SN_ = 72> __device__ __forceinline__ bf16x8 frag(const bf16* s, int out0, int ks, int lane) { return frag16<KC, SN_>(s, out0, ks, lane); }
template <bool KC, int SN_ = 72> __device__ __forceinline__ f16x8 frag(const f16* s, int out0, int ks, int lane) { return frag16<KC, SN_>(s, out0, ks, lane); }
template <bool KC, int SN_ = TT<float>::SN>
__device__ __forceinline__ float frag(const float* s, int out0, int ks, int lane) {
  if constexpr (KC) return s[(out0 + (lane & 15)) * TT<float>::STRIDE + ks * 4 + (lane >> 4)];
  else return s[(ks * 4 + (lane >> 4)) * SN_ + out0 + (lane & 15)];
}

// NT_ x NT_ MFMA tiles of 16x16 per wave (2: 64x64 block tile, 4: 128x128)
template <bool A_KC, bool B_KC, int NT_, typename Hh>
__device__ __forceinline__ void mma_tile16(const Hh* sA, const Hh* sB, int wr, int wc, int lane, f32x4 (&acc)[NT_][NT_]) {
  constexpr int SN_ = TT<Hh>::SN + (32 * NT_ - 64);
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    h16x8<Hh> a[NT_], b[NT_];
#pragma unroll
    for (int i = 0; i < NT_; ++i) {
      a[i] = frag<A_KC, SN_>(sA, (wr * NT_ + i) * 16, ks, lane);
      b[i] = frag<B_KC, SN_>(sB, (wc * NT_ + i) * 16, ks, lane);
    }
#pragma unroll
    for (int i = 0; i < NT_; ++i)
#pragma unroll
      for (int j = 0; j < NT_; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
  }
}
// ---- conflict-free LDS images of the 64 x 64 tile (round 5; 16-bit operands) ------------------------------------------------------------------
// The padded images above cost every gemm_block kernel a constant 1/3 of its LDS cycles in bank conflicts (profiles/r04_pmc_kernels.json): the
// k-contiguous [out][k] image is read with ds_read_b128, whose 16-lane groups are not contiguous (MI355X_MICROARCH.md, LDS) -- with a 144-byte
// pitch one pair of lanes per group shares a slot; the natural [k][out] image is read with ds_read_b64_tr_b16, where a 32-lane half reads rows
// {8g .. 8g+3} of two groups: windows 36 banks apart that overlap pairwise.  MAGIC_GEMM_LDS_SW (default 1, compile time):
//  * k-contiguous images: rows of exactly 64 elements, the 16-byte chunk c of row r stored at c ^ ((r >> 1) & 7) (the wide-tile kernel's image:
//    0.000 conflict share measured);
//  * TN (both operands natural): the contraction's k order is free, so a group g takes rows 4g .. 4g+3 and 16+4g .. 16+4g+3 of a 32-row step -- a half
//    reads eight CONSECUTIVE rows -- at a pitch of 80 elements (40 banks: eight distinct multiples of 8 mod 64);
//  * NN's natural B keeps the padded image (its k order is tied to A's ds_read_b128 fragment).
// Measured (profiles/micro/r05_ab_lds_swizzle.txt, same box, three rounds each): the conflict share of the gemm_block kernels drops from 0.30-0.33 to
// 0.00, the M ~ 600 / K >= 768 navigator GEMMs (K-group kernel) get 3-6 % faster per launch, the concatenated dW launch 5 % -- and the headline step
// (K = 128-512, 1-8 k-tiles per launch) gets 6-8 us SLOWER with the swizzled k-contiguous images (the extra address arithmetic of a fragment read
// sits on a chain that is latency-bound, not LDS-bound) and does not move with the permuted natural images.  Hence the default, mode 4: swizzled
// k-contiguous images in the K-group kernel only (its launches all have K >= 768), permuted natural images for every TN launch.
// 0: padded images everywhere; 1: both forms everywhere; 2: swizzled k-contiguous images everywhere; 3: permuted natural images only.
#ifndef DW_EXP
#define DW_EXP 0          // timing experiments of the weight-gradient loop (profiles/micro/dw_launch_probe.py): 1 no MFMA / LDS reads, 2 no global re-loads, 3 no bias gradient
#endif
#ifndef DW_PD
#define DW_PD 0           // (experiment) register-staged tiles in flight of the TN loop; 0: the default rule
#endif
#ifndef DW_WAVES
#define DW_WAVES 1        // (experiment) amdgpu_waves_per_eu lower bound of gemm_dw_batch_kernel
#endif
#ifndef MAGIC_GEMM_LDS_SW
#define MAGIC_GEMM_LDS_SW 4
#endif
#define GB_NATP 80          // pitch of the permuted natural image of a 64-wide tile
template <typename Hh, bool KC, bool PERM, bool SWK>
__device__ __forceinline__ void store_sw(const TileLoader<Hh, KC, 64>& l, Hh* s) {
  constexpr int VE = 8;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = (threadIdx.x & 255) + 256 * i;
    if constexpr (KC) {
      const int row = id >> 3, cv = id & 7;
      if constexpr (SWK) *(typename TT<Hh>::vec*)(s + row * 64 + ((cv ^ ((row >> 1) & 7)) * VE)) = l.v[i];
      else *(typename TT<Hh>::vec*)(s + row * TT<Hh>::STRIDE + cv * VE) = l.v[i];
    } else {
      constexpr int P = PERM ? GB_NATP : TT<Hh>::SN;
      const int r = id >> 3, ov = id & 7;
      *(typename TT<Hh>::vec*)(s + r * P + ov * VE) = l.v[i];
    }
  }
}
// bg (block-uniform; TN only): the wc = 0 waves also multiply their A fragments with a fragment of ones -- every column of accb[i] then holds the
// column sums of dY over this k-tile (the bias gradient of the tile's 16 rows), 4 MFMAs per k-tile instead of one wave reading 64 LDS rows
template <bool A_KC, bool B_KC, bool SWK, bool SWP_, typename Hh>
__device__ __forceinline__ void mma_tile_sw(const Hh* sA, const Hh* sB, int wr, int wc, int lane, f32x4 (&acc)[2][2], bool bg, f32x4 (&accb)[2]) {
  constexpr bool PERM = !A_KC && !B_KC && SWP_;
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  auto rd = [&](const Hh* s, bool kc, int out0, int ks) -> h16x8<Hh> {
    if (kc) {
      const int row = out0 + (lane & 15);
      if constexpr (SWK) return *(const h16x8<Hh>*)(s + row * 64 + (((ks * 4 + g) ^ ((row >> 1) & 7)) * 8));
      else return *(const h16x8<Hh>*)(s + row * TT<Hh>::STRIDE + ks * 32 + 8 * g);
    }
    if (PERM) return lds_tr8(s + (ks * 32 + 4 * g + q) * GB_NATP + out0 + 4 * pp, 16 * GB_NATP);
    return lds_tr8(s + (ks * 32 + 8 * g + q) * TT<Hh>::SN + out0 + 4 * pp, 4 * TT<Hh>::SN);
  };
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    h16x8<Hh> a[2], b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      a[i] = rd(sA, A_KC, (wr * 2 + i) * 16, ks);
      b[i] = rd(sB, B_KC, (wc * 2 + i) * 16, ks);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
    if (bg) {
      const Hh one = from_f<Hh>(1.f);
      const h16x8<Hh> ones = {one, one, one, one, one, one, one, one};
#pragma unroll
      for (int i = 0; i < 2; ++i) accb[i] = mfma16(a[i], ones, accb[i]);
    }
  }
}

template <bool A_KC, bool B_KC, int NT_>
__device__ __forceinline__ void mma_tile(const bf16* sA, const bf16* sB, int wr, int wc, int lane, f32x4 (&acc)[NT_][NT_]) { mma_tile16<A_KC, B_KC, NT_>(sA, sB, wr, wc, lane, acc); }
template <bool A_KC, bool B_KC, int NT_>
__device__ __forceinline__ void mma_tile(const f16* sA, const f16* sB, int wr, int wc, int lane, f32x4 (&acc)[NT_][NT_]) { mma_tile16<A_KC, B_KC, NT_>(sA, sB, wr, wc, lane, acc); }

template <bool A_KC, bool B_KC, int NT_>
__device__ __forceinline__ void mma_tile(const float* sA, const float* sB, int wr, int wc, int lane, f32x4 (&acc)[NT_][NT_]) {
  constexpr int SN_ = TT<float>::SN + (32 * NT_ - 64);
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    float a[NT_], b[NT_];
#pragma unroll
    for (int i = 0; i < NT_; ++i) {
      a[i] = frag<A_KC, SN_>(sA, (wr * NT_ + i) * 16, ks, lane);
      b[i] = frag<B_KC, SN_>(sB, (wc * NT_ + i) * 16, ks, lane);
    }
#pragma unroll
    for (int i = 0; i < NT_; ++i)
#pragma unroll
      for (int j = 0; j < NT_; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
  }
}

// ---- fp32 operands on the bf16 matrix cores ("bf16x3") --------------------------------------------------------------------------
// An fp32 value splits exactly into hi = bf16(x) and lo = bf16(x - hi) (+ a remainder below 2^-17 |x|); a product of two such values is
// a_hi b_hi + a_hi b_lo + a_lo b_hi (+ a_lo b_lo ~ 2^-18, dropped): three v_mfma_f32_16x16x32_bf16 with fp32 accumulation instead of
// eight v_mfma_f32_16x16x4_f32 per 32-deep k-step -- 3/16 of the matrix-pipe time of the exact fp32 instruction at ~2^-17 relative
// error per product (the exact fp32 MFMA: 2^-24; plain bf16 operands: 2^-9).  Selected at run time for the fp32 ("parity") storage
// mode by magic_set_f32_mfma(1): activations, weights and every epilogue stay fp32, only the contraction changes.
__device__ int g_f32_x3 = 0;
static int g_f32_x3_host = 0;
extern "C" int magic_set_f32_mfma(int mode) {
  if (mode != 0 && mode != 1) return MAGIC_ERR_ARG;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_f32_x3), &mode, sizeof(int)) != hipSuccess) return MAGIC_ERR_LAUNCH;
  g_f32_x3_host = mode;
  return MAGIC_OK;
}
extern "C" int magic_get_f32_mfma() { return g_f32_x3_host; }

__device__ __forceinline__ void split_x3(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const bf16 h = (bf16)v[j];
    hi[j] = h;
    lo[j] = (bf16)(v[j] - (float)h);
  }
}
// 16 out rows x 32 k of an fp32 LDS image as (hi, lo) bf16 fragments in the 16x16x32 operand layout (lane: row l&15, k = 8(l>>4) + j)
template <bool KC, int SN_ = TT<float>::SN>
__device__ __forceinline__ void frag_x3(const float* s, int out0, int lane, bf16x8& hi, bf16x8& lo) {
  float v[8];
  if constexpr (KC) {
    const float2* q = (const float2*)(s + (out0 + (lane & 15)) * TT<float>::STRIDE + 8 * (lane >> 4));     // rows are 8-byte aligned
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float2 t = q[j]; v[2 * j] = t.x; v[2 * j + 1] = t.y; }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = s[(8 * (lane >> 4) + j) * SN_ + out0 + (lane & 15)];
  }
  split_x3(v, hi, lo);
}
__device__ __forceinline__ f32x4 mma_x3(const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
}
template <bool A_KC, bool B_KC, int NT_>
__device__ __forceinline__ void mma_tile_x3(const float* sA, const float* sB, int wr, int wc, int lane, f32x4 (&acc)[NT_][NT_]) {
  constexpr int SN_ = TT<float>::SN + (32 * NT_ - 64);
  bf16x8 ah[NT_], al[NT_], bh[NT_], bl[NT_];
#pragma unroll
  for (int i = 0; i < NT_; ++i) {
    frag_x3<A_KC, SN_>(sA, (wr * NT_ + i) * 16, lane, ah[i], al[i]);
    frag_x3<B_KC, SN_>(sB, (wc * NT_ + i) * 16, lane, bh[i], bl[i]);
  }
#pragma unroll
  for (int i = 0; i < NT_; ++i)
#pragma unroll
    for (int j = 0; j < NT_; ++j) acc[i][j] = mma_x3(ah[i], al[i], bh[j], bl[j], acc[i][j]);
}

// ---- deterministic split-K seam of the weight gradients --------------------------------------------------------------------------
// dW tile = sum over the K-splits of every problem that targets this dW (one Linear used T times in a step queues T problems).  With fp32
// atomics the sum depends on arrival order; here every split STORES its 64 x 64 partial (+ the 64 bias-gradient partials of a bx = 0 tile)
// into its own slot of a workspace, write-through (`sc1`), every storing wave drains, the workgroup's barrier, ONE agent-scope add on the
// tile's arrival counter; the workgroup that draws the last ticket reads the slots IN SLOT ORDER (sc1 loads), adds them in that fixed order
// and does the one read-modify-write of dW (no other workgroup of the launch touches that tile).  The counter is put back to 0 by that
// workgroup, so the counters need zeroing only once, when they are allocated.  (cdna_hip_programming.md section 6 G16, form R1 with
// every consumer load sc1.)  Slot layout: 16-byte chunk (accumulator tile 2 i + j) of thread t at chunk [(2 i + j) * 256 + t], bias partial of row r at float [4096 + r].
#define DW_SLOT (64 * 64 + 64)
template <int TM, int NE>
__device__ __forceinline__ void dw_seam(const GemmParams& p, const f32x4 (&acc)[2][2], float bsum, bool do_bgrad, int bx, int by, int sk,
                                        int tid, int m0, int n0, int wr, int wc, int cr, int cc, int* flag) {
  static_assert(TM == 64 && NE == 16, "the seam is written for the 64 x 64 tile");
  if (p.ws_total == 1) {
    // ONE contributor to this tile in the whole launch (no K-split, the dW queued once): nothing to order -- plain read-modify-write
    float* C = (float*)p.C;
    float old[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int col = n0 + (wc * 2 + ((e >> 2) & 1)) * 16 + cc, row = m0 + (wr * 2 + (e >> 3)) * 16 + cr + (e & 3);
      old[e] = (col < p.N && row < p.M) ? C[(long long)row * p.ldc + col] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int col = n0 + (wc * 2 + ((e >> 2) & 1)) * 16 + cc, row = m0 + (wr * 2 + (e >> 3)) * 16 + cr + (e & 3);
      if (col < p.N && row < p.M) C[(long long)row * p.ldc + col] = old[e] + acc[e >> 3][(e >> 2) & 1][e & 3] * p.alpha;
    }
    if (do_bgrad && tid < TM && m0 + tid < p.M) p.bias_grad[m0 + tid] += bsum;
    return;
  }
  const int nxt = (p.N + TM - 1) / TM;
  const int tile = by * nxt + bx;
  float* const base = p.ws + ((long long)p.ws_slot0 + (long long)tile * p.ws_total) * DW_SLOT;
  const int mine = p.ws_first + sk;
  // 16 bytes per lane and access: accumulator tile (i, j) of thread t at byte ((2 i + j) 256 + t) 16 of the slot; write-through / sc1 both ways
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_;
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, p.ws_total * DW_SLOT * 4, 0x00020000);
  const int my_off = mine * DW_SLOT * 4;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const f32x4 v = acc[i][j] * p.alpha;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), rsrc, my_off + ((2 * i + j) * 256 + tid) * 16, 0, 16);
    }
  if (bx == 0 && tid < TM) __hip_atomic_store(base + (long long)mine * DW_SLOT + 4096 + tid, do_bgrad ? bsum : 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its write-through stores ...
  __syncthreads();                                            // ... before one lane signals for the workgroup
  if (tid == 0) {
    unsigned* c = p.cnt + p.cnt0 + tile;
    const unsigned t = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (t == (unsigned)p.ws_total - 1u);
    if (last) __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
    *flag = last;
  }
  __syncthreads();
  if (!*flag) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // (compiler only: keeps the loads below the ticket)
  f32x4 sum[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) sum[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int q = 0; q < p.ws_total; ++q) {                      // FIXED order: slot 0, 1, 2, ... (this workgroup's own slot comes from its registers)
    f32x4 v[2][2];
    if (q == mine) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) v[i][j] = acc[i][j] * p.alpha;
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          v[i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, q * DW_SLOT * 4 + ((2 * i + j) * 256 + tid) * 16, 0, 16));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) sum[i][j] += v[i][j];
  }
  float s[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) s[e] = sum[e >> 3][(e >> 2) & 1][e & 3];
  float* C = (float*)p.C;
  float old[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int col = n0 + (wc * 2 + ((e >> 2) & 1)) * 16 + cc, row = m0 + (wr * 2 + (e >> 3)) * 16 + cr + (e & 3);
    old[e] = (col < p.N && row < p.M) ? C[(long long)row * p.ldc + col] : 0.f;
  }
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int col = n0 + (wc * 2 + ((e >> 2) & 1)) * 16 + cc, row = m0 + (wr * 2 + (e >> 3)) * 16 + cr + (e & 3);
    if (col < p.N && row < p.M) C[(long long)row * p.ldc + col] = old[e] + s[e];
  }
  if (bx == 0 && p.bias_grad && tid < TM && m0 + tid < p.M) {
    float b = 0.f;
    for (int q = 0; q < p.ws_total; ++q) b += __hip_atomic_load(base + (long long)q * DW_SLOT + 4096 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    p.bias_grad[m0 + tid] += b;
  }
}

// LAYOUT 0: NT (A[M,K], B[N,K]); 1: NN (A[M,K], B[K,N]); 2: TN (A[K,M], B[K,N])
// NT_: 16x16 MFMA tiles per wave per dimension.  2 -> the 64x64 block tile every small problem uses; 4 -> a 128x128 block tile
// (64x64 per wave, 16 accumulator tiles) for problems with enough rows and columns: each workgroup then streams half the
// operand bytes per output element from L2 and issues half the LDS reads per MFMA, which is what bounds the H >= 256 shapes.
// KG: K-groups per workgroup (1, or 4 for `gemm_kg_kernel`): KG x 256 threads, group g multiplies the K-tiles g, g + KG, ... into its own
// accumulators from its own LDS images -- one barrier round advances KG tiles -- and the partial sums are added through LDS at the end.
// For problems with fewer tiles than CUs and a long K (the M ~ 600 navigator-step GEMMs at H = 768: 120 tiles, 48 K-tiles) the time of
// a launch is (K-tiles) x (latency of one barrier round); the groups divide the rounds by KG without an fp32 round trip through HBM.
// elements of one operand image of the 64 x 64 tile: the padded form (64 x 72) or the permuted natural form (64 x 80)
template <typename T> struct GBI { static constexpr int N = (sizeof(T) == 2 && TT<T>::BK * GB_NATP > BM * TT<T>::STRIDE) ? TT<T>::BK * GB_NATP : BM * TT<T>::STRIDE; };
template <typename T, int LAYOUT, int NT_ = 2, int KG = 1>
__device__ __forceinline__ void gemm_block(const GemmParams& p, const int bx, const int by, const int bzz, T* sA, T* sB) {
  constexpr int BK = TT<T>::BK, STRIDE = TT<T>::STRIDE;
  constexpr int TM = 32 * NT_;                       // block tile is TM x TM
  constexpr int SN_ = TT<T>::SN + (TM - 64);
  static_assert(TM * TT<T>::STRIDE >= TT<T>::BK * SN_, "LDS image sizes");
  constexpr bool A_KC = (LAYOUT != 2), B_KC = (LAYOUT == 0);
  constexpr int NE = NT_ * NT_ * 4;                  // accumulator elements per lane
  constexpr int SWM = MAGIC_GEMM_LDS_SW;
  constexpr bool SW16 = sizeof(T) == 2 && NT_ == 2;
  constexpr bool SWK = SW16 && (SWM == 1 || SWM == 2 || (SWM == 4 && KG > 1));                     // swizzled k-contiguous images
  constexpr bool SWP = SW16 && (SWM == 1 || SWM == 3 || SWM == 4) && LAYOUT == 2 && KG == 1;     // permuted natural images (both operands natural)
  constexpr bool SW = SWK || SWP;                                                                     // (every caller's images hold GBI<T>::N elements)

  const int tid = threadIdx.x & 255, lane = tid & 63, wid = tid >> 6, kg = threadIdx.x >> 8;
  const int wr = wid >> 1, wc = wid & 1;
  const int n0 = bx * TM, m0 = by * TM;
  const bool x3 = (sizeof(T) == 4) && g_f32_x3 != 0;       // block-uniform (one scalar load)
  T* const sA0 = sA;
  if constexpr (KG > 1) { sA += kg * (TM * TT<T>::STRIDE); sB += kg * (TM * TT<T>::STRIDE); }
  const int bz = bzz / p.splitk, sk = bzz % p.splitk;
  const int bb = bz / p.nh, bh = bz % p.nh;

  const T* A = (const T*)p.A + bb * p.sAb + bh * p.sAh;
  const T* B = (const T*)p.B + bb * p.sBb + bh * p.sBh;
  const long long coff = bb * p.sCb + bh * p.sCh;

  // split-K range, in whole BK tiles
  const int ktiles = (p.K + BK - 1) / BK;
  const int per = (ktiles + p.splitk - 1) / p.splitk;
  const int kt0 = sk * per, kt1 = min(ktiles, kt0 + per);
  if (kt0 >= kt1 && p.splitk > 1 && !p.slab) return;          // (slab mode: an empty split still stores its zeros)

  // Register-staged software pipeline, PD tiles deep: the global loads of tiles kt+1 .. kt+PD-1 are in flight while
  // tile kt is written to LDS and multiplied.  These GEMMs run ~1 block per CU (grids of 100-600 blocks), so a
  // block has to hide HBM/L2 latency itself; hipcc turns the in-order loads into counted s_waitcnt vmcnt(N).
  constexpr int PD = (LAYOUT == 2 && DW_PD) ? DW_PD : (NT_ == 2 && KG == 1) ? 3 : 2;      // K-group kernel: 16 waves per CU hide latency, and 1024 threads cap the VGPRs at 128
  TileLoader<T, A_KC, TM> la[PD];
  TileLoader<T, B_KC, TM> lb[PD];
  f32x4 acc[NT_][NT_];
#pragma unroll
  for (int i = 0; i < NT_; ++i)
#pragma unroll
    for (int j = 0; j < NT_; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const bool do_bgrad = (LAYOUT == 2) && p.bias_grad != nullptr && bx == 0;
  constexpr bool BG_MFMA = SW && LAYOUT == 2;              // bias gradient on the matrix cores (see mma_tile_sw)
  f32x4 accb[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};

  // round r of group kg works on K-tile kt0 + r * KG + kg; a tile index past the end loads zeros (the loader's k < kend test), so
  // every group executes the same barriers
  const int kend = min(p.K, kt1 * BK);
  const int rounds = (kt1 - kt0 + KG - 1) / KG;
#pragma unroll
  for (int d = 0; d < PD; ++d)
    if (d < rounds) {
      la[d].load(A, p.lda, m0, (kt0 + d * KG + kg) * BK, p.M, kend);
      lb[d].load(B, p.ldb, n0, (kt0 + d * KG + kg) * BK, p.N, kend);
    }
  // element e of a lane: tile (i, j) = (e / (4 NT_), (e / 4) % NT_), register r = e % 4.
  // C/D map of 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg.
  const int cr = (lane >> 4) * 4, cc = lane & 15;
  auto col_of = [&](int e) { return n0 + (wc * NT_ + ((e >> 2) % NT_)) * 16 + cc; };
  auto row_of = [&](int e) { return m0 + (wr * NT_ + (e / (4 * NT_))) * 16 + cr + (e & 3); };
  // Epilogue operands (bias / activation-derivative input / residual).  64x64 tile: fetched HERE, before the K loop -- these
  // GEMMs are 1-3 k-tiles long, so a second dependent round trip to memory after the loop was a visible share of each launch.
  // 128x128 tile (long K loops, 64 elements per lane): fetched after the loop.
  // Structured as {all loads} -> {math} -> {all stores}: gfx950's vmcnt counts stores too, so interleaving
  // per-element loads and stores serialises the memory round trips of a thread (measured: +4 us per launch).
  constexpr bool PRE = (NT_ == 2);
  float ax[NE], rs[NE];
  float bv[NT_];
  const bool has_aux = (p.epilogue == 3 || p.epilogue == 4);
#pragma unroll
  for (int j = 0; j < NT_; ++j) {
    const int col = n0 + (wc * NT_ + j) * 16 + cc;
    bv[j] = (p.bias && sk == 0 && col < p.N) ? p.bias[col] : 0.f;
  }
  auto load_epi = [&]() {
#pragma unroll
    for (int e = 0; e < NE; ++e) { ax[e] = 0.f; rs[e] = 0.f; }
    if (has_aux) {
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int col = col_of(e), row = row_of(e);
        if (col < p.N && row < p.M) ax[e] = to_f(((const T*)p.aux)[coff + (long long)row * p.ldaux + col]);
      }
    }
    if (p.residual) {
      if (p.c_f32) {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const int col = col_of(e), row = row_of(e);
          if (col < p.N && row < p.M) rs[e] = ((const float*)p.residual)[coff + (long long)row * p.ldr + col];
        }
      } else {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const int col = col_of(e), row = row_of(e);
          if (col < p.N && row < p.M) rs[e] = to_f(((const T*)p.residual)[coff + (long long)row * p.ldr + col]);
        }
      }
    }
  };
  if constexpr (PRE) { if (KG == 1 || kg == 0) load_epi(); }
  for (int r = 0; r < rounds; r += PD) {
#pragma unroll
    for (int d = 0; d < PD; ++d) {
      if (r + d < rounds) {           // block-uniform
        if constexpr (SW) {
          store_sw<T, A_KC, SWP, SWK>(la[d], sA);
          store_sw<T, B_KC, SWP, SWK>(lb[d], sB);
        } else {
          la[d].store(sA);
          lb[d].store(sB);
        }
        __syncthreads();
#if DW_EXP == 2
        if (LAYOUT != 2)
#endif
        if (r + d + PD < rounds) {
          la[d].load(A, p.lda, m0, (kt0 + (r + d + PD) * KG + kg) * BK, p.M, kend);
          lb[d].load(B, p.ldb, n0, (kt0 + (r + d + PD) * KG + kg) * BK, p.N, kend);
        }
#if DW_EXP == 1
        if (LAYOUT != 2)
#endif
        {
        if constexpr (sizeof(T) == 4) {
          if (x3) mma_tile_x3<A_KC, B_KC, NT_>((const float*)sA, (const float*)sB, wr, wc, lane, acc);
          else mma_tile<A_KC, B_KC, NT_>(sA, sB, wr, wc, lane, acc);
        } else if constexpr (SW) {
          mma_tile_sw<A_KC, B_KC, SWK, SWP>(sA, sB, wr, wc, lane, acc, BG_MFMA && do_bgrad && wc == 0, accb);
        } else {
          mma_tile<A_KC, B_KC, NT_>(sA, sB, wr, wc, lane, acc);
        }
        }
#if DW_EXP == 3
        if (false)
#endif
        if (!BG_MFMA && do_bgrad && tid < TM) {
          float s = 0.f;
          constexpr int NP = SWP ? GB_NATP : SN_;
#pragma unroll 8
          for (int k = 0; k < BK; ++k) s += to_f(sA[k * NP + tid]);      // TN: A is held as the natural [k][out] image
          bsum += s;
        }
        __syncthreads();
      }
    }
  }
  if constexpr (BG_MFMA) {
    if (do_bgrad) {                    // column 0 of accb[i] -> bsum of thread (row) tid < 64, through LDS (the images are dead after the loop's last barrier)
      float* red = (float*)sA;
      if (wc == 0 && (lane & 15) == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) red[(wr * 2 + i) * 16 + (lane >> 4) * 4 + r4] = accb[i][r4];
      }
      __syncthreads();
      if (tid < TM) bsum = red[tid];
      __syncthreads();                 // (the seam reuses sA as its flag word)
    }
  }
  if constexpr (LAYOUT == 2 && NT_ == 2 && KG == 1) {
    if (p.ws) { dw_seam<TM, NE>(p, acc, bsum, do_bgrad, bx, by, sk, tid, m0, n0, wr, wc, cr, cc, (int*)sA); return; }
  }
  if (do_bgrad && tid < TM && m0 + tid < p.M) atomicAdd(p.bias_grad + m0 + tid, bsum);
  if constexpr (KG > 1) {
    // partial sums of groups 1 .. KG-1 -> LDS (the operand images are dead after the loop's last barrier), group 0 adds them
    float* red = (float*)sA0;
    if (kg > 0) {
#pragma unroll
      for (int e = 0; e < NE; ++e) red[((kg - 1) * NE + e) * 256 + tid] = acc[e / (4 * NT_)][(e >> 2) % NT_][e & 3];
    }
    __syncthreads();
    if (kg > 0) return;
#pragma unroll
    for (int g = 0; g < KG - 1; ++g)
#pragma unroll
      for (int e = 0; e < NE; ++e) acc[e / (4 * NT_)][(e >> 2) % NT_][e & 3] += red[(g * NE + e) * 256 + tid];
  }
  if constexpr (!PRE) load_epi();

  // epilogue math + stores
  float v[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) v[e] = acc[e / (4 * NT_)][(e >> 2) % NT_][e & 3] * p.alpha + bv[(e >> 2) % NT_];
  if (p.C2) {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int col = col_of(e), row = row_of(e);
      if (col < p.N && row < p.M) ((T*)p.C2)[coff + (long long)row * p.ldc2 + col] = from_f<T>(v[e]);
    }
  }
  if (p.epilogue == 1) {
#pragma unroll
    for (int e = 0; e < NE; ++e) v[e] = gelu_f(v[e]);
  } else if (p.epilogue == 2) {
#pragma unroll
    for (int e = 0; e < NE; ++e) v[e] = fmaxf(v[e], 0.f);
  } else if (p.epilogue == 3) {
#pragma unroll
    for (int e = 0; e < NE; ++e) v[e] *= dgelu_f(ax[e]);
  } else if (p.epilogue == 4) {
#pragma unroll
    for (int e = 0; e < NE; ++e) v[e] = ax[e] > 0.f ? v[e] : 0.f;
  }
#pragma unroll
  for (int e = 0; e < NE; ++e) v[e] += rs[e];
  if (p.c_f32) {
    if (p.accumulate) {
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int col = col_of(e), row = row_of(e);
        if (col < p.N && row < p.M) atomicAdd((float*)p.C + coff + (long long)row * p.ldc + col, v[e]);
      }
    } else {
      float* Cs = (float*)p.C + (long long)sk * p.slab;        // slab 0 when p.slab == 0
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int col = col_of(e), row = row_of(e);
        if (col < p.N && row < p.M) Cs[coff + (long long)row * p.ldc + col] = v[e];
      }
    }
  } else {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int col = col_of(e), row = row_of(e);
      if (col < p.N && row < p.M) ((T*)p.C)[coff + (long long)row * p.ldc + col] = from_f<T>(v[e]);
    }
  }
}

template <typename T, int LAYOUT>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) T sA[GBI<T>::N];
  __shared__ __attribute__((aligned(16))) T sB[GBI<T>::N];
  gemm_block<T, LAYOUT>(p, blockIdx.x, blockIdx.y, blockIdx.z, sA, sB);
}

// XCD-aware variant of the same kernel (1-D grid).  Workgroups are dealt round-robin over the 8 XCDs (ids b and b+8 share an
// XCD) and each XCD has its own L2: with the plain (x, y) grid the N/64 column tiles of one 64-row A panel land on N/64
// different XCDs and every one of them pulls that panel over the fabric.  Here row tile r belongs to XCD r % 8 and its column
// tiles occupy consecutive slots of that XCD, so the panel crosses the fabric once and the siblings hit L2.  Row tiles are
// padded to a multiple of 8 (the surplus workgroups exit at once).
template <typename T, int LAYOUT>
__global__ __launch_bounds__(256) void gemm_xcd_kernel(GemmParams p, int nx, int ny, int ny8) {
  __shared__ __attribute__((aligned(16))) T sA[GBI<T>::N];
  __shared__ __attribute__((aligned(16))) T sB[GBI<T>::N];
  const int per_z = nx * ny8;
  const int z = blockIdx.x / per_z, l2 = blockIdx.x - z * per_z;
  const int xcd = l2 & 7, slot = l2 >> 3;
  const int lr = slot / nx, bx = slot - lr * nx;
  const int by = lr * 8 + xcd;
  if (by >= ny) return;
  gemm_block<T, LAYOUT>(p, bx, by, z, sA, sB);
}

// K-group variant of gemm_kernel: 1024 threads = 4 K-groups of 4 waves (see gemm_block).  LDS: 4 x (A image + B image) = 72 KB.
#define KGROUPS 4
template <typename T, int LAYOUT>
__global__ __launch_bounds__(1024) void gemm_kg_kernel(GemmParams p) {
  // ONE array: after the K loop the whole of it is the reduction buffer ((KGROUPS-1) x 16 floats x 256 lanes = 48 KB)
  constexpr int IMG = BM * TT<T>::STRIDE;
  static_assert(2 * KGROUPS * IMG * sizeof(T) >= (KGROUPS - 1) * 16 * 256 * sizeof(float), "reduction buffer");
  __shared__ __attribute__((aligned(16))) T smem[2 * KGROUPS * IMG];
  gemm_block<T, LAYOUT, 2, KGROUPS>(p, blockIdx.x, blockIdx.y, blockIdx.z, smem, smem + KGROUPS * IMG);
}

// ---------------------------------------------------------------------------------------------------------------
// Wide tile for the H >= 768 shapes (MAGIC-L; bf16 only): 128x128 block tile, BK = 64, 4 waves as 2x2 with a 64x64 sub-tile
// each (4x4 MFMA tiles).  Both operand tiles go global -> LDS by LDS-DMA (`global_load_lds`, 16 B per lane, no VGPR staging,
// no ds_write pass) into a two-stage ring; two workgroups share a CU.
// An LDS-DMA wave-instruction writes 1 KiB contiguously (wave-uniform base + lane * 16), so the LDS images cannot be padded;
// bank conflicts are removed by permuting the 16-byte chunks of a row instead, applied on the per-lane SOURCE address when
// filling and on the fragment address when reading:
//   k-contiguous operand  [128 out][64 k]   (128 B rows): chunk c of row r sits at c ^ ((r >> 1) & 7)  -> the 16 rows of one
//       b128 fragment read cover 16 distinct 16-byte slots of the 256-byte bank row;
//   out-contiguous operand [64 k][128 out]  (256 B rows, read transposed with ds_read_b64_tr_b16): chunk c of k-row r sits at
//       c ^ (2 (r & 3) + 8 ((r >> 3) & 1)) -> the 8 k-rows x 32 B a half-wave reads are 16 distinct slots.
// Out-of-range chunks (tile edges in M / N / K) are fetched from a 16-byte zero page; the contiguous dimension of each operand must be
// a multiple of 8 elements (a chunk is wholly inside or wholly outside), every other extent is free.
__device__ __attribute__((aligned(16))) const unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ int wide_sw_nat(int kr) { return 2 * (kr & 3) + 8 * ((kr >> 3) & 1); }

template <bool KC, typename Hh>
__device__ __forceinline__ void wide_stage(const Hh* __restrict__ base, int ld, int out0, int k0, int OUT, int kend, Hh* s, int tid) {
  // 1024 chunks of 16 B per operand tile: chunk idx = j * 256 + tid (wave w, lane l: LDS bytes (4 j + w) * 1024 + 16 l)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = j * 256 + tid;
    const Hh* src;
    if (KC) {
      const int row = idx >> 3, c = (idx & 7) ^ ((row >> 1) & 7);
      const int o = out0 + row, k = k0 + c * 8;
      src = (o < OUT && k < kend) ? base + (long long)o * ld + k : (const Hh*)g_zero16;
    } else {
      const int kr = idx >> 4, c = (idx & 15) ^ wide_sw_nat(kr);
      const int k = k0 + kr, o = out0 + c * 8;
      src = (k < kend && o < OUT) ? base + (long long)k * ld + o : (const Hh*)g_zero16;
    }
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(s + (j * 256 + (tid & ~63)) * 8), 16, 0, 0);
  }
}

template <bool KC, typename Hh>
__device__ __forceinline__ h16x8<Hh> wide_frag(const Hh* s, int out0, int ks, int lane) {
  if constexpr (KC) {
    const int row = out0 + (lane & 15);
    const int c = (ks * 4 + (lane >> 4)) ^ ((row >> 1) & 7);
    return *(const h16x8<Hh>*)(s + row * 64 + c * 8);
  } else {
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int kr = ks * 32 + 8 * g + q;                       // (kr + 4) has the same swizzle: q < 4
    const int c = ((out0 >> 3) + (pp >> 1)) ^ wide_sw_nat(kr);
    return lds_tr8(s + kr * 128 + c * 8 + (pp & 1) * 4, 4 * 128);
  }
}

template <typename Hh, int LAYOUT>
__device__ __forceinline__ void gemm_wide_block(const GemmParams& p, const int bx, const int by, const int bzz, Hh* sA, Hh* sB) {
  typedef Hh T;
  constexpr bool A_KC = (LAYOUT != 2), B_KC = (LAYOUT == 0);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int n0 = bx * 128, m0 = by * 128;
  const int bz = bzz / p.splitk, sk = bzz % p.splitk;
  const int bb = bz / p.nh, bh = bz % p.nh;
  const T* A = (const T*)p.A + bb * p.sAb + bh * p.sAh;
  const T* B = (const T*)p.B + bb * p.sBb + bh * p.sBh;
  const long long coff = bb * p.sCb + bh * p.sCh;
  const int ktiles = (p.K + 63) / 64;
  const int per = (ktiles + p.splitk - 1) / p.splitk;
  const int kt0 = sk * per, kt1 = min(ktiles, kt0 + per);
  if (kt0 >= kt1 && p.splitk > 1) return;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const bool do_bgrad = (LAYOUT == 2) && p.bias_grad != nullptr && bx == 0;

  // two LDS stages of 32 KB (A tile, B tile): the fill of tile kt+1 is issued right after the barrier that publishes tile kt and lands
  // while tile kt is multiplied -- one barrier per K-step; 64 KB of LDS, two workgroups per CU.  (One stage with four workgroups per CU
  // measured 15-30 % slower: 8192x3072x768 64 vs 55 us.)
  wide_stage<A_KC>(A, p.lda, m0, kt0 * 64, p.M, p.K, sA, tid);
  wide_stage<B_KC>(B, p.ldb, n0, kt0 * 64, p.N, p.K, sB, tid);
  int cur = 0;
  for (int kt = kt0; kt < kt1; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();              // tile kt has landed for every wave, and every wave is past the multiply of tile kt-1
    const Hh* tA = sA + cur * (2 * 128 * 64);
    const Hh* tB = sB + cur * (2 * 128 * 64);
    if (kt + 1 < kt1) {
      wide_stage<A_KC>(A, p.lda, m0, (kt + 1) * 64, p.M, p.K, sA + (cur ^ 1) * (2 * 128 * 64), tid);
      wide_stage<B_KC>(B, p.ldb, n0, (kt + 1) * 64, p.N, p.K, sB + (cur ^ 1) * (2 * 128 * 64), tid);
    }
    cur ^= 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      h16x8<Hh> a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = wide_frag<A_KC>(tA, (wr * 4 + i) * 16, ks, lane);
        b[i] = wide_frag<B_KC>(tB, (wc * 4 + i) * 16, ks, lane);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
    }
    if (do_bgrad && tid < 128) {                     // TN: A is the natural [k][out] image; column `tid` summed over the tile's k
      float s = 0.f;
      const int c = tid >> 3, e = tid & 7;
#pragma unroll 8
      for (int k = 0; k < 64; ++k) s += to_f(tA[k * 128 + ((c ^ wide_sw_nat(k)) << 3) + e]);
      bsum += s;
    }
  }
  if (do_bgrad && tid < 128 && m0 + tid < p.M) atomicAdd(p.bias_grad + m0 + tid, bsum);

  // epilogue, one 16-row band of the wave's sub-tile at a time (16 values per lane live, not 64)
  const int cr = (lane >> 4) * 4, cc = lane & 15;
  const bool has_aux = (p.epilogue == 3 || p.epilogue == 4);
  float bv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = n0 + (wc * 4 + j) * 16 + cc;
    bv[j] = (p.bias && sk == 0 && col < p.N) ? p.bias[col] : 0.f;
  }
  // every runtime switch is block-uniform and sits OUTSIDE the element loops: {all loads} -> {math} -> {all stores} per band, so a
  // lane's memory round trips overlap (a conditional load per element compiles to load / wait / use chains: +14 us per launch)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float v[16], ax[16], rs[16];
    auto col_of = [&](int e) { return n0 + (wc * 4 + (e >> 2)) * 16 + cc; };
    auto row_of = [&](int e) { return m0 + (wr * 4 + i) * 16 + cr + (e & 3); };
#pragma unroll
    for (int e = 0; e < 16; ++e) { ax[e] = 0.f; rs[e] = 0.f; }
    if (has_aux) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int col = col_of(e), row = row_of(e);
        if (col < p.N && row < p.M) ax[e] = to_f(((const T*)p.aux)[coff + (long long)row * p.ldaux + col]);
      }
    }
    if (p.residual) {
      if (p.c_f32) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int col = col_of(e), row = row_of(e);
          if (col < p.N && row < p.M) rs[e] = ((const float*)p.residual)[coff + (long long)row * p.ldr + col];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int col = col_of(e), row = row_of(e);
          if (col < p.N && row < p.M) rs[e] = to_f(((const T*)p.residual)[coff + (long long)row * p.ldr + col]);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = acc[i][e >> 2][e & 3] * p.alpha + bv[e >> 2];
    if (p.C2) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int col = col_of(e), row = row_of(e);
        if (col < p.N && row < p.M) ((T*)p.C2)[coff + (long long)row * p.ldc2 + col] = from_f<T>(v[e]);
      }
    }
    if (p.epilogue == 1) {
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = gelu_f(v[e]);
    } else if (p.epilogue == 2) {
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = fmaxf(v[e], 0.f);
    } else if (p.epilogue == 3) {
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] *= dgelu_f(ax[e]);
    } else if (p.epilogue == 4) {
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = ax[e] > 0.f ? v[e] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] += rs[e];
    if (p.c_f32) {
      if (p.accumulate) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int col = col_of(e), row = row_of(e);
          if (col < p.N && row < p.M) atomicAdd((float*)p.C + coff + (long long)row * p.ldc + col, v[e]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int col = col_of(e), row = row_of(e);
          if (col < p.N && row < p.M) ((float*)p.C)[coff + (long long)row * p.ldc + col] = v[e];
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int col = col_of(e), row = row_of(e);
        if (col < p.N && row < p.M) ((T*)p.C)[coff + (long long)row * p.ldc + col] = from_f<T>(v[e]);
      }
    }
  }
}

// XCD-aware 1-D tile map as in gemm_xcd_kernel, nx / ny counted in 128-wide tiles
__device__ __forceinline__ bool wide_tile_of(int nx, int ny, int ny8, int& bx, int& by, int& z) {
  const int per_z = nx * (ny8 > 0 ? ny8 : ny);
  z = blockIdx.x / per_z;
  const int l2 = blockIdx.x - z * per_z;
  if (ny8 > 0) {                      // row tile r on XCD r % 8, its column tiles in consecutive slots of that XCD
    const int xcd = l2 & 7, slot = l2 >> 3;
    const int lr = slot / nx;
    bx = slot - lr * nx;
    by = lr * 8 + xcd;
    return by < ny;
  }
  by = l2 / nx;                       // few row tiles: plain row-major tile order
  bx = l2 - by * nx;
  return true;
}
template <typename Hh, int LAYOUT>
__global__ __launch_bounds__(256, 2) void gemm_wide_kernel(GemmParams p, int nx, int ny, int ny8) {
  __shared__ __attribute__((aligned(1024))) Hh sAB[4 * 128 * 64];          // stage s: A at s * 32 KB, B 16 KB behind it
  int bx, by, z;
  if (!wide_tile_of(nx, ny, ny8, bx, by, z)) return;
  gemm_wide_block<Hh, LAYOUT>(p, bx, by, z, sAB, sAB + 128 * 64);
}

// Grouped weight-gradient GEMM: up to GROUP_MAX independent TN problems (dW[N,K] += dY^T X, split-K, fp32 atomics, fused
// bias gradient) in ONE launch.  Nothing on the backward chain depends on dW, so the engine defers all of a step's
// weight-gradient GEMMs and issues them ~8 at a time: ~80 launches become ~10.
#define GROUP_MAX 8
// start[i]: first workgroup of problem i (always a multiple of 8, so `local % 8` is the XCD slot of the workgroup);
// cnt[i]: workgroups that have a tile (the rest of the padded range exits); ny8[i] > 0: XCD-aware tile map as in gemm_xcd_kernel.
struct GroupedParams { GemmParams p[GROUP_MAX]; int start[GROUP_MAX + 1]; int cnt[GROUP_MAX]; int ny8[GROUP_MAX]; int n; };

template <typename T, int LAYOUT>
__global__ __launch_bounds__(256) void gemm_grouped_kernel(GroupedParams gp) {
  __shared__ __attribute__((aligned(16))) T sA[GBI<T>::N];
  __shared__ __attribute__((aligned(16))) T sB[GBI<T>::N];
  const int id = blockIdx.x;
  int g = 0;
#pragma unroll
  for (int i = 1; i < GROUP_MAX; ++i) g += (i < gp.n && id >= gp.start[i]) ? 1 : 0;      // block-uniform
  const GemmParams& p = gp.p[g];
  const int local = id - gp.start[g];
  if (local >= gp.cnt[g]) return;
  const int nx = (p.N + BN - 1) / BN, ny = (p.M + BM - 1) / BM;
  const int ny8 = gp.ny8[g];
  if (ny8 < 0) {
    // split-K problem (weight gradient): all nx*ny tiles of one K-split share that split's dY / X panels -> keep a split's
    // tiles on ONE XCD (split z on XCD z % 8, its tiles in consecutive slots) so each panel crosses the fabric once
    const int tiles = nx * ny, nz = p.batch * p.splitk;
    const int xcd = local & 7, slot = local >> 3;
    const int zr = slot / tiles, t = slot - zr * tiles;
    const int z = zr * 8 + xcd;
    if (z >= nz) return;
    gemm_block<T, LAYOUT>(p, t % nx, t / nx, z, sA, sB);
  } else if (ny8 > 0) {
    const int per_z = nx * ny8;
    const int z = local / per_z, l2 = local - z * per_z;
    const int xcd = l2 & 7, slot = l2 >> 3;
    const int lr = slot / nx, bx = slot - lr * nx;
    const int by = lr * 8 + xcd;
    if (by >= ny) return;
    gemm_block<T, LAYOUT>(p, bx, by, z, sA, sB);
  } else {
    gemm_block<T, LAYOUT>(p, local % nx, (local / nx) % ny, local / (nx * ny), sA, sB);
  }
}

// K-group form of the grouped launch (the navigator step's PAIRED GEMMs: the map branch's and the viewpoint branch's twin Linears, a few hundred rows
// each at H = 768): as gemm_kg_kernel, 4 K-groups per workgroup, for launches whose problems all have a long K and together fewer tiles than the
// chip holds workgroups of this size.  No split-K problems (they take the plain grouped kernel).
template <typename T, int LAYOUT>
__global__ __launch_bounds__(1024) void gemm_grouped_kg_kernel(GroupedParams gp) {
  constexpr int IMG = BM * TT<T>::STRIDE;
  static_assert(2 * KGROUPS * IMG * sizeof(T) >= (KGROUPS - 1) * 16 * 256 * sizeof(float), "reduction buffer");
  __shared__ __attribute__((aligned(16))) T smem[2 * KGROUPS * IMG];
  const int id = blockIdx.x;
  int g = 0;
#pragma unroll
  for (int i = 1; i < GROUP_MAX; ++i) g += (i < gp.n && id >= gp.start[i]) ? 1 : 0;      // block-uniform
  const GemmParams& p = gp.p[g];
  const int local = id - gp.start[g];
  if (local >= gp.cnt[g]) return;
  const int nx = (p.N + BN - 1) / BN, ny = (p.M + BM - 1) / BM;
  const int ny8 = gp.ny8[g];
  int bx, by, z;
  if (ny8 > 0) {
    const int per_z = nx * ny8;
    z = local / per_z;
    const int l2 = local - z * per_z, xcd = l2 & 7, slot = l2 >> 3, lr = slot / nx;
    bx = slot - lr * nx;
    by = lr * 8 + xcd;
    if (by >= ny) return;
  } else {
    bx = local % nx; by = (local / nx) % ny; z = local / (nx * ny);
  }
  gemm_block<T, LAYOUT, 2, KGROUPS>(p, bx, by, z, smem, smem + KGROUPS * IMG);
}

// Weight-gradient form with COMPACT descriptors (64 B per problem instead of a 184 B GemmParams): up to DW_MAX problems per launch fit
// the kernel-argument block.  Every dependent launch of the replayed step costs ~9 us whatever its work (8 -> 16 problems per launch:
// 2.93 -> 2.88 ms per step; 48 compact: 2.79; 96: 2.77), so the step's ~80 weight gradients go out in ONE launch (6.9 KB of kernel arguments).
#define DW_MAX 96
#ifndef DW_EXP
#define DW_EXP 0
#endif
struct DwProblem { const void* A; const void* B; float* C; float* bias_grad; int M, N, K, lda, ldb, ldc, splitk, ny8;
                   int ws_slot0, cnt0, ws_total, ws_first; };       // deterministic seam (dw_seam): slots / counters of this dW's tile 0
struct DwBatch { DwProblem p[DW_MAX]; int start[DW_MAX + 1]; int cnt[DW_MAX]; int n; float* ws; unsigned* counters; };

template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DW_WAVES))) void gemm_dw_batch_kernel(DwBatch gp) {
  __shared__ __attribute__((aligned(16))) T sA[GBI<T>::N];
  __shared__ __attribute__((aligned(16))) T sB[GBI<T>::N];
  const int id = blockIdx.x;
  int lo = 0, hi = gp.n - 1;                                   // last problem whose first workgroup is <= id (block-uniform)
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (id >= gp.start[mid]) lo = mid; else hi = mid - 1; }
  const DwProblem& d = gp.p[lo];
  const int local = id - gp.start[lo];
  if (local >= gp.cnt[lo]) return;
  GemmParams p = {};
  p.A = d.A; p.B = d.B; p.C = d.C; p.bias_grad = d.bias_grad;
  p.M = d.M; p.N = d.N; p.K = d.K; p.lda = d.lda; p.ldb = d.ldb; p.ldc = d.ldc;
  p.nh = 1; p.batch = 1; p.splitk = d.splitk; p.c_f32 = 1; p.accumulate = 1; p.alpha = 1.f;
  p.ws = gp.ws; p.cnt = gp.counters; p.ws_slot0 = d.ws_slot0; p.cnt0 = d.cnt0; p.ws_total = d.ws_total; p.ws_first = d.ws_first;
  const int nx = (p.N + BN - 1) / BN, ny = (p.M + BM - 1) / BM;
  if (d.ny8 == -1) {                 // one K-split per XCD slot group (see gemm_grouped_kernel)
    const int tiles = nx * ny;
    const int xcd = local & 7, slot = local >> 3;
    const int zr = slot / tiles, t = slot - zr * tiles;
    const int z = zr * 8 + xcd;
    if (z >= p.splitk) return;
    gemm_block<T, 2>(p, t % nx, t / nx, z, sA, sB);
  } else if (d.ny8 == -2) {          // 1, 2 or 4 K-splits: a split's tiles on 8 / splitk neighbouring XCDs, each taking a contiguous run of tile
    const int tiles = nx * ny;       // columns (or rows) -- see dw_xcd_groups()
    const int G = 8 / p.splitk, per = (tiles + G - 1) / G;
    const int xcd = local & 7, slot = local >> 3;
    const int z = xcd / G, g = xcd - z * G, t = g * per + slot;
    if (t >= tiles) return;
    int bx, by;
    if (nx >= ny) { bx = t / ny; by = t - bx * ny; } else { by = t / nx; bx = t - by * nx; }
    gemm_block<T, 2>(p, bx, by, z, sA, sB);
  } else if (d.ny8 > 0) {
    const int per_z = nx * d.ny8;
    const int z = local / per_z, l2 = local - z * per_z;
    const int xcd = l2 & 7, slot = l2 >> 3;
    const int lr = slot / nx, bx = slot - lr * nx;
    const int by = lr * 8 + xcd;
    if (by >= ny) return;
    gemm_block<T, 2>(p, bx, by, z, sA, sB);
  } else {
    gemm_block<T, 2>(p, local % nx, (local / nx) % ny, local / (nx * ny), sA, sB);
  }
}

// Weight gradients of ONE Linear over MANY row segments in one pass (round 5): dW[N, K] += sum_s dY_s[M_s, N]^T X_s[M_s, K].
// The navigator iteration calls every Linear once per step and rollout (~38 times): the per-step grouped launch above then read-modify-writes
// the whole fp32 gradient of the model once per step (528 MB per step at MAGIC-L: the launch's time was that traffic, 207 us x 75 launches
// per iteration).  Here the step instances keep their dY / X operands (host/step_graphs.py) and ONE launch at the end of the backward pass
// walks all segments of a problem per output tile with the accumulators in registers: one read-modify-write of dW per iteration, no
// workspace, no atomics, one workgroup per 64 x 64 tile (a MAGIC-L model has ~20 k of them: the chip is full without splitting K) --
// segment order = the order the host lists them in, so the sums are reproducible.  Segment tables live in device memory
// ([n_prob][n_seg] operand pointers and row counts); rows = 0 skips a segment.
struct magic_dwcat_prob { float* dW; float* db; int N, K, lda, ldb, ldc; };
struct DwCatProblem { float* C; float* bias_grad; int M, N, lda, ldb, ldc; };        // M = out features (dW rows), N = in features (dW columns)
struct DwCatBatch { DwCatProblem p[DW_MAX]; int start[DW_MAX + 1]; int n, n_seg; const void* const* dy_tab; const void* const* x_tab; const int* m_tab; };

// NT_ = 2: 64 x 64 tiles; NT_ = 4: 128 x 128 tiles (problems with both dimensions >= 128: twice the MFMA work per byte staged through LDS -- the 64 x 64
// form ran the MAGIC-L iteration's 3.8 TFLOP at 425 TFLOP/s)
template <typename T, int NT_>
__global__ __launch_bounds__(256) void gemm_dw_cat_kernel(DwCatBatch gp) {
  // Both operands are natural [k][out] images read through ds_read_b64_tr_b16.  With gemm_block's pitch (72 / 136 elements) and k order (a 16-lane
  // group g reads rows 8g .. 8g+3, then 8g+4 .. 8g+7) the eight 32-byte row windows of a 32-lane half sit 36 banks apart and overlap pairwise: every
  // such read pays half a cycle extra (the constant 1/3 conflict share of every gemm_block kernel, profiles/r04_pmc_kernels.json) and no padding removes
  // it (rows 0 and 8 of a half collide for every pitch that keeps the windows aligned).  Here BOTH operands are natural, so the contraction's k order
  // is free: group g takes rows 4g .. 4g+3, then 16+4g .. 16+4g+3 -- a half reads eight CONSECUTIVE rows -- and the pitch is TM + 16 elements
  // (40 / 72 banks: r x pitch mod 64 = eight distinct multiples of 8): conflict-free.  fp32 operands keep the plain form (dword reads).
  constexpr bool PERM = sizeof(T) == 2;
  constexpr int BK = TT<T>::BK, TM = 32 * NT_, SN_ = PERM ? TM + 16 : TT<T>::SN + (TM - 64), NE = NT_ * NT_ * 4, PD = 3;      // (register-staged prefetch depth; 2 -> 3 on the wide tile: 755 -> 702 us per launch of the MAGIC-L iteration, 228 VGPRs; double-buffered LDS images on top of it: 704, not kept)
  constexpr int IMG = (TM * TT<T>::STRIDE > BK * SN_) ? TM * TT<T>::STRIDE : BK * SN_;
  __shared__ __attribute__((aligned(16))) T sA[IMG];
  __shared__ __attribute__((aligned(16))) T sB[IMG];
  const int id = blockIdx.x;
  int lo = 0, hi = gp.n - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (id >= gp.start[mid]) lo = mid; else hi = mid - 1; }
  const DwCatProblem& d = gp.p[lo];
  const int local = id - gp.start[lo];
  const int nx = (d.N + TM - 1) / TM, ny = (d.M + TM - 1) / TM;
  // XCD-aware placement: consecutive workgroup ids go round the 8 XCDs (a problem's first id is a multiple of 8), so XCD x takes the x-th
  // eighth of the problem's tiles in row-major order -- the tiles of one row block of dW (same dY panel) and of neighbouring row blocks
  // (same X panels) meet in ONE L2 instead of eight
  const int per = (nx * ny + 7) >> 3;
  const int tile = (local & 7) * per + (local >> 3);
  if ((local >> 3) >= per || tile >= nx * ny) return;
  const int bx = tile % nx, by = tile / nx;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1;
  const int n0 = bx * TM, m0 = by * TM;
  const void* const* dyt = gp.dy_tab + (long long)lo * gp.n_seg;
  const void* const* xt = gp.x_tab + (long long)lo * gp.n_seg;
  const int* mt = gp.m_tab + (long long)lo * gp.n_seg;
  TileLoader<T, false, TM> la[PD], lb[PD];
  f32x4 acc[NT_][NT_];
#pragma unroll
  for (int i = 0; i < NT_; ++i)
#pragma unroll
    for (int j = 0; j < NT_; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const bool do_bgrad = d.bias_grad != nullptr && bx == 0;
  // the load stream walks (segment, k-tile) pairs in order; `ls` / `lk` = the pair the NEXT load takes
  int ls = 0, lk = 0;
  auto skip_empty = [&]() { while (ls < gp.n_seg && lk * BK >= mt[ls]) { ++ls; lk = 0; } };
  auto issue = [&](int slot) {          // returns false when the stream is exhausted
    skip_empty();
    if (ls >= gp.n_seg) return false;
    const int rows = mt[ls];
    la[slot].load((const T*)dyt[ls], d.lda, m0, lk * BK, d.M, rows);
    lb[slot].load((const T*)xt[ls], d.ldb, n0, lk * BK, d.N, rows);
    ++lk;
    return true;
  };
  bool live[PD];
#pragma unroll
  for (int q = 0; q < PD; ++q) live[q] = issue(q);
  while (live[0]) {
#pragma unroll
    for (int q = 0; q < PD; ++q) {
      if (live[q]) {                    // block-uniform
        if constexpr (PERM) {
          constexpr int VE = TT<T>::VE, VPR = TM / VE;
#pragma unroll
          for (int i = 0; i < TM / 32; ++i) {
            const int id2 = tid + 256 * i, r = id2 / VPR, ov = id2 % VPR;
            *(typename TT<T>::vec*)(sA + r * SN_ + ov * VE) = la[q].v[i];
            *(typename TT<T>::vec*)(sB + r * SN_ + ov * VE) = lb[q].v[i];
          }
        } else {
          la[q].store(sA);
          lb[q].store(sB);
        }
        __syncthreads();
        live[q] = issue(q);
        if constexpr (PERM) {
          const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            h16x8<T> a[NT_], b[NT_];
            const int r0 = (ks * 32 + 4 * g + qq) * SN_ + 4 * pp;
#pragma unroll
            for (int i = 0; i < NT_; ++i) {
              a[i] = lds_tr8(sA + r0 + (wr * NT_ + i) * 16, 16 * SN_);
              b[i] = lds_tr8(sB + r0 + (wc * NT_ + i) * 16, 16 * SN_);
            }
#pragma unroll
            for (int i = 0; i < NT_; ++i)
#pragma unroll
              for (int j = 0; j < NT_; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
          }
        } else {
          mma_tile<false, false, NT_>(sA, sB, wr, wc, lane, acc);
        }
        if (do_bgrad && tid < TM) {
          float s = 0.f;
#pragma unroll 8
          for (int k = 0; k < BK; ++k) s += to_f(sA[k * SN_ + tid]);
          bsum += s;
        }
        __syncthreads();
      }
    }
    // slots are consumed in order 0, 1, 2, 0, ...: once slot q is dead every later slot of this round and every slot of the next is dead too
  }
  const int cr = (lane >> 4) * 4, cc = lane & 15;
  // element e of a lane: tile (i, j) = (e / (4 NT_), (e / 4) % NT_), register r = e % 4 (gemm_block's map)
  float old[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int col = n0 + (wc * NT_ + ((e >> 2) % NT_)) * 16 + cc, row = m0 + (wr * NT_ + (e / (4 * NT_))) * 16 + cr + (e & 3);
    old[e] = (col < d.N && row < d.M) ? d.C[(long long)row * d.ldc + col] : 0.f;
  }
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int col = n0 + (wc * NT_ + ((e >> 2) % NT_)) * 16 + cc, row = m0 + (wr * NT_ + (e / (4 * NT_))) * 16 + cr + (e & 3);
    if (col < d.N && row < d.M) d.C[(long long)row * d.ldc + col] = old[e] + acc[e / (4 * NT_)][(e >> 2) % NT_][e & 3];
  }
  if (do_bgrad && tid < TM && m0 + tid < d.M) d.bias_grad[m0 + tid] += bsum;
}

extern "C" int magic_gemm_dw_cat(int dtype, int n_prob, const magic_dwcat_prob* probs, int n_seg, const void* const* dy_tab, const void* const* x_tab,
                                 const int* m_tab, void* stream) {
  if (n_prob <= 0 || n_prob > DW_MAX || !probs || n_seg <= 0 || !dy_tab || !x_tab || !m_tab || !dtype_ok(dtype)) return MAGIC_ERR_ARG;
  const int ve = dtype_is16(dtype) ? 8 : 4;
  DwCatBatch gp;
  gp.n = n_prob; gp.n_seg = n_seg; gp.dy_tab = dy_tab; gp.x_tab = x_tab; gp.m_tab = m_tab;
  // one tile size per launch: 128 x 128 when every problem has both dimensions >= 128 (the host sorts the wide problems into launches of their own)
  bool wide = dtype_is16(dtype);
  for (int i = 0; i < n_prob; ++i) wide = wide && probs[i].N >= 128 && probs[i].K >= 128;
  const int tm = wide ? 128 : 64;
  int total = 0;
  for (int i = 0; i < n_prob; ++i) {
    const magic_dwcat_prob& q = probs[i];
    if (!q.dW || q.N <= 0 || q.K <= 0 || q.lda % ve || q.ldb % ve || q.ldc < q.K) return MAGIC_ERR_ARG;
    DwCatProblem& p = gp.p[i];
    p.C = q.dW; p.bias_grad = q.db; p.M = q.N; p.N = q.K; p.lda = q.lda; p.ldb = q.ldb; p.ldc = q.ldc;
    gp.start[i] = total;
    total += (((p.N + tm - 1) / tm) * ((p.M + tm - 1) / tm) + 7) / 8 * 8;
  }
  for (int i = n_prob; i <= DW_MAX; ++i) gp.start[i] = total;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_BF16) { if (wide) hipLaunchKernelGGL((gemm_dw_cat_kernel<bf16, 4>), dim3(total), dim3(256), 0, st, gp); else hipLaunchKernelGGL((gemm_dw_cat_kernel<bf16, 2>), dim3(total), dim3(256), 0, st, gp); }
  else if (dtype == DT_F16) { if (wide) hipLaunchKernelGGL((gemm_dw_cat_kernel<f16, 4>), dim3(total), dim3(256), 0, st, gp); else hipLaunchKernelGGL((gemm_dw_cat_kernel<f16, 2>), dim3(total), dim3(256), 0, st, gp); }
  else hipLaunchKernelGGL((gemm_dw_cat_kernel<float, 2>), dim3(total), dim3(256), 0, st, gp);
  return launch_status();
}

// Wide-tile selection (bf16; M, N >= 128; the contiguous dimension of each operand a multiple of 8).  Measured on MI355X
// (profiles/micro/gemm_tile_sweep.py -> profiles/micro/r01_gemm_tile_sweep.txt): the wide tile wins on the forward / input-gradient GEMMs
// once the output has >= ~192 tiles of 128x128 and K >= 512 (M = 8192 rows at H = 768: 1.4-1.9x over the 64x64 tile, 1.2-1.3x behind
// hipBLASLt); the 64x64 tile keeps every MAGIC-S / MAGIC-M shape and every M ~ 600 navigator-step shape.  The weight gradient (TN) stays on
// the 64x64 tile: it is bound by its fp32 atomics, and the host's split-K choice (ops._splitk) is what matters there.
// mode 0: never; mode 1 (default): the rule above; mode 2: whenever eligible, all three layouts (tests).
// Set from the environment (MAGIC_GEMM_BIG) or magic_gemm_set_big() (tests, tuning).
static int g_big_mode = -1, g_big_min_tiles = 192, g_big_min_k = 512;
extern "C" int magic_gemm_set_big(int mode) {
  if (mode < 0 || mode > 2) return MAGIC_ERR_ARG;
  g_big_mode = mode;
  return MAGIC_OK;
}
static int gemm_big_tile(int dtype, int layout, int M, int N, int K, int nz) {
  if (g_big_mode < 0) {
    const char* e = getenv("MAGIC_GEMM_BIG"); g_big_mode = e ? atoi(e) : 1;
    const char* t = getenv("MAGIC_GEMM_BIG_MIN_TILES"); if (t) g_big_min_tiles = atoi(t);
    const char* k = getenv("MAGIC_GEMM_BIG_MIN_K"); if (k) g_big_min_k = atoi(k);
  }
  if (g_big_mode == 0 || !dtype_is16(dtype) || M < 128 || N < 128) return 0;
  // 16-byte chunks must lie wholly inside or outside the logical extent of each operand's contiguous dimension
  if (layout == 0 ? (K % 8) : layout == 1 ? ((K % 8) || (N % 8)) : ((M % 8) || (N % 8))) return 0;
  if (g_big_mode == 2) return 1;
  if (layout == 2) return 0;
  const long long tiles = (long long)((M + 127) / 128) * ((N + 127) / 128) * nz;
  return (tiles >= g_big_min_tiles && K >= g_big_min_k) ? 1 : 0;
}

extern "C" int magic_gemm(int dtype, int layout, int batch, int nh, int M, int N, int K,
                          const void* A, int lda, long long sAb, long long sAh,
                          const void* B, int ldb, long long sBb, long long sBh,
                          void* C, int ldc, long long sCb, long long sCh, int c_f32, int accumulate,
                          const float* bias, int epilogue, const void* aux, int ldaux,
                          const void* residual, int ldr, void* C2, int ldc2,
                          float alpha, int splitk, float* bias_grad, void* stream) {
  // splitk < 0: |splitk| K-splits, each STORING its partial into its own slab of C ([|splitk|][M][ldc] fp32, no atomics: the consumer adds the slabs in
  // order -- deterministic); plain product only (no epilogue / residual / bias gradient), one batch
  long long slab = 0;
  if (splitk < 0) {
    if (!c_f32 || epilogue != 0 || residual || C2 || bias_grad || batch != 1 || nh != 1 || M <= 0 || ldc <= 0) return MAGIC_ERR_ARG;
    splitk = -splitk;
    slab = (long long)M * ldc;
    accumulate = 0;
  }
  if (M <= 0 || N <= 0 || K <= 0 || batch <= 0 || nh <= 0 || splitk <= 0) return MAGIC_ERR_ARG;
  if (!dtype_ok(dtype)) return MAGIC_ERR_ARG;
  if (layout < 0 || layout > 2) return MAGIC_ERR_ARG;
  const int ve = dtype_is16(dtype) ? 8 : 4;
  if (lda % ve || ldb % ve || (sAb % ve) || (sAh % ve) || (sBb % ve) || (sBh % ve)) return MAGIC_ERR_ARG;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return MAGIC_ERR_ARG;
  if (dtype == DT_F32 && !c_f32) return MAGIC_ERR_ARG;
  if (!slab && (accumulate || splitk > 1) && !(c_f32 && accumulate)) return MAGIC_ERR_ARG;
  if (splitk > 1 && (epilogue != 0 || residual || C2)) return MAGIC_ERR_ARG;
  if (bias_grad && layout != 2) return MAGIC_ERR_ARG;
  if (batch % nh) return MAGIC_ERR_ARG;
  GemmParams p = {};
  p.A = A; p.B = B; p.C = C; p.C2 = C2; p.bias = bias; p.aux = aux; p.residual = residual; p.bias_grad = bias_grad;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldc2 = ldc2; p.ldaux = ldaux; p.ldr = ldr;
  p.sAb = sAb; p.sAh = sAh; p.sBb = sBb; p.sBh = sBh; p.sCb = sCb; p.sCh = sCh;
  p.nh = nh; p.splitk = splitk; p.epilogue = epilogue; p.c_f32 = c_f32; p.accumulate = accumulate; p.alpha = alpha;
  p.batch = batch;
  p.slab = slab;
  p.big = slab ? 0 : gemm_big_tile(dtype, layout, M, N, K, batch * splitk);
  if (group_record(KIND_GEMM, dtype, layout, &p, sizeof(p))) return MAGIC_OK;
  return launch_gemm(dtype, layout, &p, nullptr, (hipStream_t)stream);
}

static bool gemm_xcd_on() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("MAGIC_GEMM_XCD"); v = e ? atoi(e) : 1; }
  return v != 0;
}
// lays problem i into the grouped launch: returns the padded workgroup count (multiple of 8)
static inline int group_place(GroupedParams& gp, int i, int total) {
  const GemmParams& p = gp.p[i];
  const int nx = (p.N + BN - 1) / BN, ny = (p.M + BM - 1) / BM, nz = p.batch * p.splitk;
  gp.start[i] = total;
  if (gemm_xcd_on() && p.splitk >= 8 && nx * ny >= 2) {
    gp.ny8[i] = -1;
    gp.cnt[i] = nx * ny * ((nz + 7) / 8 * 8);
  } else if (gemm_xcd_on() && nx >= 2 && ny >= 16) {
    gp.ny8[i] = (ny + 7) / 8 * 8;
    gp.cnt[i] = nx * gp.ny8[i] * nz;
  } else {
    gp.ny8[i] = 0;
    gp.cnt[i] = nx * ny * nz;
  }
  return (gp.cnt[i] + 7) / 8 * 8;
}
static inline int gemm_blocks(const GemmParams& p) { return ((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM) * p.batch * p.splitk; }

int launch_gemm(int dtype, int layout, const void* pa, const void* pb, hipStream_t st) {
  const GemmParams& a = *(const GemmParams*)pa;
  dim3 block(256);
  if (!pb && a.big) {
    const int nx = (a.N + 127) / 128, ny = (a.M + 127) / 128, nz = a.batch * a.splitk;
    const int ny8 = (gemm_xcd_on() && nx >= 2 && ny >= 16) ? (ny + 7) / 8 * 8 : 0;
    dim3 g1((unsigned)(nx * (ny8 > 0 ? ny8 : ny) * nz));
#define LAUNCHB(TY, L) hipLaunchKernelGGL((gemm_wide_kernel<TY, L>), g1, block, 0, st, a, nx, ny, ny8)
    if (!dtype_is16(dtype)) return MAGIC_ERR_ARG;
    if (dtype == DT_BF16) { if (layout == 0) LAUNCHB(bf16, 0); else if (layout == 1) LAUNCHB(bf16, 1); else LAUNCHB(bf16, 2); }
    else { if (layout == 0) LAUNCHB(f16, 0); else if (layout == 1) LAUNCHB(f16, 1); else LAUNCHB(f16, 2); }
#undef LAUNCHB
    return launch_status();
  }
  if (!pb) {
    const int nx = (a.N + BN - 1) / BN, ny = (a.M + BM - 1) / BM, nz = a.batch * a.splitk;
    // fewer tiles than CUs and a long K: K-group kernel (4 K-tiles per barrier round).  Measured (profiles/micro/r01_gemm_tile_sweep.txt)
    static int kg_on = -1;
    if (kg_on < 0) { const char* e = getenv("MAGIC_GEMM_KG"); kg_on = e ? atoi(e) : 1; }
    if (kg_on && layout != 2 && a.splitk == 1 && (long long)nx * ny * nz <= 224 && a.K >= 768) {
      dim3 gk(nx, ny, nz), bk(1024);
#define LAUNCHK(TY, L) hipLaunchKernelGGL((gemm_kg_kernel<TY, L>), gk, bk, 0, st, a)
      if (dtype == DT_BF16) { if (layout == 0) LAUNCHK(bf16, 0); else LAUNCHK(bf16, 1); }
      else if (dtype == DT_F16) { if (layout == 0) LAUNCHK(f16, 0); else LAUNCHK(f16, 1); }
      else { if (layout == 0) LAUNCHK(float, 0); else LAUNCHK(float, 1); }
#undef LAUNCHK
      return launch_status();
    }
    if (gemm_xcd_on() && nx >= 2 && ny >= 16) {         // enough row tiles that the padding to a multiple of 8 is small
      const int ny8 = (ny + 7) / 8 * 8;
      dim3 g1((unsigned)(nx * ny8 * nz));
#define LAUNCHX(TY, L) hipLaunchKernelGGL((gemm_xcd_kernel<TY, L>), g1, block, 0, st, a, nx, ny, ny8)
      if (dtype == DT_BF16) {
        if (layout == 0) LAUNCHX(bf16, 0); else if (layout == 1) LAUNCHX(bf16, 1); else LAUNCHX(bf16, 2);
      } else if (dtype == DT_F16) {
        if (layout == 0) LAUNCHX(f16, 0); else if (layout == 1) LAUNCHX(f16, 1); else LAUNCHX(f16, 2);
      } else {
        if (layout == 0) LAUNCHX(float, 0); else if (layout == 1) LAUNCHX(float, 1); else LAUNCHX(float, 2);
      }
#undef LAUNCHX
      return launch_status();
    }
    dim3 grid(nx, ny, nz);
#define LAUNCH(TY, L) hipLaunchKernelGGL((gemm_kernel<TY, L>), grid, block, 0, st, a)
    if (dtype == DT_BF16) {
      if (layout == 0) LAUNCH(bf16, 0); else if (layout == 1) LAUNCH(bf16, 1); else LAUNCH(bf16, 2);
    } else if (dtype == DT_F16) {
      if (layout == 0) LAUNCH(f16, 0); else if (layout == 1) LAUNCH(f16, 1); else LAUNCH(f16, 2);
    } else {
      if (layout == 0) LAUNCH(float, 0); else if (layout == 1) LAUNCH(float, 1); else LAUNCH(float, 2);
    }
#undef LAUNCH
    return launch_status();
  }
  const void* ps[2] = {pa, pb};
  return launch_gemm_n(dtype, layout, ps, 2, st);
}

int launch_gemm_n(int dtype, int layout, const void* const* ps, int n, hipStream_t st) {
  if (n == 1) return launch_gemm(dtype, layout, ps[0], nullptr, st);
  if (n <= 0 || n > GROUP_MAX) return MAGIC_ERR_ARG;
  dim3 block(256);
  GroupedParams gp;
  gp.n = n;
  int total = 0;
  for (int i = 0; i < n; ++i) { gp.p[i] = *(const GemmParams*)ps[i]; total += group_place(gp, i, total); }
  for (int i = n; i <= GROUP_MAX; ++i) gp.start[i] = total;
  dim3 grid(total);
  // every problem with a long K, no split-K, and few tiles in all: the K-group form.  For every K >= 768 (MAGIC_GEMM_KG_GROUP=2) it was measured and rejected: back to
  // back it wins where the single-problem form does (profiles/micro/r05_pair_gemm_probe.txt: 624 + 512 rows x 768 x 768 9.2 vs 10.8 us, K = 3072
  // 20.5 vs 30.5; past ~224 tiles it loses, 16.2 vs 12.5 us), but the navigator iteration runs two rollout lanes side by side and a launch of
  // 1024-thread / 72 KB workgroups leaves the other lane's kernels no room on the CUs: 142.0 vs 135.5 ms per iteration
  // What stays on by default: launches whose contraction is LONG (K >= 2048: the FFN's second projection forward, its first projection's input
  // gradient -- 2 of a layer's 12 paired launches), where the K-group form saves a third (30.5 -> 20.5 us) and the crowding is brief.
  // MAGIC_GEMM_KG_GROUP=0 off, =2 every K >= 768 (the form measured above); _MIN_K / _TILES move the bounds.
  static int kgg = -1, kgg_tiles = 224, kgg_min_k = 2048;
  if (kgg < 0) {
    const char* e = getenv("MAGIC_GEMM_KG_GROUP"); kgg = e ? atoi(e) : 1;
    const char* t = getenv("MAGIC_GEMM_KG_GROUP_TILES"); if (t) kgg_tiles = atoi(t);
    const char* k = getenv("MAGIC_GEMM_KG_GROUP_MIN_K"); if (k) kgg_min_k = atoi(k);
    if (kgg == 2) kgg_min_k = 768;
  }
  bool kg_ok = kgg && layout != 2 && total <= kgg_tiles;
  for (int i = 0; i < n && kg_ok; ++i) kg_ok = gp.p[i].splitk == 1 && gp.p[i].K >= kgg_min_k && gp.ny8[i] >= 0;
  if (kg_ok) {
    dim3 bk(1024);
#define LAUNCHGK(TY, L) hipLaunchKernelGGL((gemm_grouped_kg_kernel<TY, L>), grid, bk, 0, st, gp)
    if (dtype == DT_BF16) { if (layout == 0) LAUNCHGK(bf16, 0); else LAUNCHGK(bf16, 1); }
    else if (dtype == DT_F16) { if (layout == 0) LAUNCHGK(f16, 0); else LAUNCHGK(f16, 1); }
    else { if (layout == 0) LAUNCHGK(float, 0); else LAUNCHGK(float, 1); }
#undef LAUNCHGK
    return launch_status();
  }
#define LAUNCHG(TY, L) hipLaunchKernelGGL((gemm_grouped_kernel<TY, L>), grid, block, 0, st, gp)
  if (dtype == DT_BF16) {
    if (layout == 0) LAUNCHG(bf16, 0); else if (layout == 1) LAUNCHG(bf16, 1); else LAUNCHG(bf16, 2);
  } else if (dtype == DT_F16) {
    if (layout == 0) LAUNCHG(f16, 0); else if (layout == 1) LAUNCHG(f16, 1); else LAUNCHG(f16, 2);
  } else {
    if (layout == 0) LAUNCHG(float, 0); else if (layout == 1) LAUNCHG(float, 1); else LAUNCHG(float, 2);
  }
#undef LAUNCHG
  return launch_status();
}

// ---------------------------------------------------------------------------------------------------------------
// Linear + bias + residual + LayerNorm in one launch (the BertSelfOutput / BertOutput tails: dense -> add -> LayerNorm).
// A workgroup owns 32 full rows (BN = H), so the row statistics are a cross-wave LDS reduction in the epilogue and the
// separate LayerNorm launch (and its pass over [M,H]) disappears.  4 waves side by side, each 32 rows x H/4 columns.
struct LlnParams {
  int M, K; const void* X; int lda; const void* W; int ldb; const float* bias; const void* R; int ldr;
  const float* gamma; const float* beta; float eps; void* out; float* rstd_out;
  DropDesc drop;      // hidden dropout between the dense and the residual add (BertSelfOutput / BertOutput)
  int act;            // magic_linear_act_ln: activation between the dense and the LayerNorm (1 erf gelu, 2 relu; 0 none)
  void* pre_out;      // ... and where the pre-activation goes (storage dtype, pitch H), for the backward's act'
};

template <typename T, int HT>      // HT = H / 64 column tiles of 16 per wave (per wave H/4 = 16*HT columns)
__device__ __forceinline__ void linear_ln_body(const LlnParams& pp, const int bid, unsigned char* lds_raw) {
  const int M = pp.M, K = pp.K, lda = pp.lda, ldb = pp.ldb, ldr = pp.ldr;
  const T* __restrict__ X = (const T*)pp.X; const T* __restrict__ W = (const T*)pp.W; const T* __restrict__ R = (const T*)pp.R;
  const float* __restrict__ bias = pp.bias; const float* __restrict__ gamma = pp.gamma; const float* __restrict__ beta = pp.beta;
  const float eps = pp.eps; T* __restrict__ out = (T*)pp.out; float* __restrict__ rstd_out = pp.rstd_out;
  typedef typename TT<T>::vec vec;
  constexpr int VE = TT<T>::VE, BK = TT<T>::BK, STRIDE = TT<T>::STRIDE;
  constexpr int H = 64 * HT, WC = 16 * HT;          // columns per wave
  constexpr int KS = (sizeof(T) == 2) ? 2 : 8;      // MFMA k-steps per BK tile
  T* sA = (T*)lds_raw;                 // [32][STRIDE]
  T* sB = sA + 32 * STRIDE;            // [H][STRIDE]
  float* red = (float*)(sB + H * STRIDE);   // [4][32]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c16 = lane & 15;
  const int m0 = bid * 32;
  f32x4 acc[2][HT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < HT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int VPT_B = H * (BK / VE) / 256;        // B-tile vectors per thread
  vec va, vb[VPT_B];
  auto load = [&](int k0) {
    {
      const int row = tid >> 3, cv = tid & 7;
      vec z;
#pragma unroll
      for (int e = 0; e < VE; ++e) z[e] = (T)0.0f;
      if (m0 + row < M && k0 + cv * VE < K) z = *(const vec*)(X + (long long)(m0 + row) * lda + k0 + cv * VE);
      va = z;
    }
#pragma unroll
    for (int i = 0; i < VPT_B; ++i) {
      const int id = tid + 256 * i, row = id >> 3, cv = id & 7;
      vec z;
#pragma unroll
      for (int e = 0; e < VE; ++e) z[e] = (T)0.0f;
      if (k0 + cv * VE < K) z = *(const vec*)(W + (long long)row * ldb + k0 + cv * VE);
      vb[i] = z;
    }
  };
  auto store = [&]() {
    auto st = [&](T* d, const vec& v) {
      if constexpr (sizeof(T) == 2) { *(vec*)d = v; }
      else { ((float2*)d)[0] = make_float2(v[0], v[1]); ((float2*)d)[1] = make_float2(v[2], v[3]); }
    };
    st(sA + (tid >> 3) * STRIDE + (tid & 7) * VE, va);
#pragma unroll
    for (int i = 0; i < VPT_B; ++i) {
      const int id = tid + 256 * i;
      st(sB + (id >> 3) * STRIDE + (id & 7) * VE, vb[i]);
    }
  };
  const int ktiles = (K + BK - 1) / BK;
  load(0);
  // epilogue operands (bias, gamma, beta, residual tile) are fetched before the K loop: their latency hides under it
  float bv[HT], gv[HT], btv[HT];
#pragma unroll
  for (int j = 0; j < HT; ++j) {
    const int col = w * WC + j * 16 + c16;
    bv[j] = bias ? bias[col] : 0.f; gv[j] = gamma[col]; btv[j] = beta[col];
  }
  float rsd[2][HT][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < HT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + i * 16 + 4 * g + r, col = w * WC + j * 16 + c16;
        rsd[i][j][r] = (R && row < M) ? to_f(R[(long long)row * ldr + col]) : 0.f;
      }
  for (int kt = 0; kt < ktiles; ++kt) {
    store();
    __syncthreads();
    if (kt + 1 < ktiles) load((kt + 1) * BK);
    bool done_x3 = false;
    if constexpr (sizeof(T) == 4) {
      if (g_f32_x3 != 0) {          // fp32 storage, split-bf16 contraction (one 32-deep step per tile)
        bf16x8 a0h, a0l, a1h, a1l;
        frag_x3<true>((const float*)sA, 0, lane, a0h, a0l);
        frag_x3<true>((const float*)sA, 16, lane, a1h, a1l);
#pragma unroll
        for (int j = 0; j < HT; ++j) {
          bf16x8 bh, bl;
          frag_x3<true>((const float*)sB, w * WC + j * 16, lane, bh, bl);
          acc[0][j] = mma_x3(a0h, a0l, bh, bl, acc[0][j]);
          acc[1][j] = mma_x3(a1h, a1l, bh, bl, acc[1][j]);
        }
        done_x3 = true;
      }
    }
    if (!done_x3) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const auto a0 = frag<true>(sA, 0, ks, lane), a1 = frag<true>(sA, 16, ks, lane);
#pragma unroll
      for (int j = 0; j < HT; ++j) {
        const auto b = frag<true>(sB, w * WC + j * 16, ks, lane);
        if constexpr (sizeof(T) == 2) {
          acc[0][j] = mfma16(a0, b, acc[0][j]);
          acc[1][j] = mfma16(a1, b, acc[1][j]);
        } else {
          acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, acc[0][j], 0, 0, 0);
          acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, acc[1][j], 0, 0, 0);
        }
      }
    }
    }
    __syncthreads();
  }
  // ---- epilogue: v = dropout(acc + bias) + residual ; LayerNorm over the full row (cross-wave) ; store
  const DropState dsn = drop_init(pp.drop);
  float s[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < HT; ++j) {
        float v = acc[i][j][r] + bv[j];
        if (pp.act) {                         // (block-uniform) prediction-head transform: dense -> activation -> LayerNorm
          const int row = m0 + i * 16 + 4 * g + r;
          if (pp.pre_out && row < M) ((T*)pp.pre_out)[(long long)row * H + w * WC + j * 16 + c16] = from_f<T>(v);
          v = pp.act == 1 ? gelu_f(v) : fmaxf(v, 0.f);
        }
        if (dsn.on) v *= drop_mul(dsn, (unsigned)((m0 + i * 16 + 4 * g + r) * H + w * WC + j * 16 + c16));
        acc[i][j][r] = v + rsd[i][j][r]; t += acc[i][j][r];
      }
      t = row16_sum(t);
      s[i][r] = t;
    }
  if (c16 == 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[w * 32 + i * 16 + 4 * g + r] = s[i][r];
  }
  __syncthreads();
  float mean[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = i * 16 + 4 * g + r;
      mean[i][r] = (red[rr] + red[32 + rr] + red[64 + rr] + red[96 + rr]) / H;
    }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < HT; ++j) { const float d = acc[i][j][r] - mean[i][r]; t += d * d; }
      t = row16_sum(t);
      s[i][r] = t;
    }
  if (c16 == 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[w * 32 + i * 16 + 4 * g + r] = s[i][r];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = i * 16 + 4 * g + r, row = m0 + rr;
      const float rstd = rsqrtf((red[rr] + red[32 + rr] + red[64 + rr] + red[96 + rr]) / H + eps);
      if (row < M) {
        if (rstd_out && w == 0 && c16 == 0) rstd_out[row] = rstd;
#pragma unroll
        for (int j = 0; j < HT; ++j)
          out[(long long)row * H + w * WC + j * 16 + c16] = from_f<T>((acc[i][j][r] - mean[i][r]) * rstd * gv[j] + btv[j]);
      }
    }
}

template <typename T, int HT>
__global__ __launch_bounds__(256) void linear_ln_kernel(LlnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_dyn[];
  linear_ln_body<T, HT>(p, blockIdx.x, lds_dyn);
}
template <typename T, int HT>
__global__ __launch_bounds__(256) void linear_ln_pair_kernel(LlnParams a, LlnParams b, int nA) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_dyn[];
  if ((int)blockIdx.x < nA) linear_ln_body<T, HT>(a, blockIdx.x, lds_dyn);
  else linear_ln_body<T, HT>(b, blockIdx.x - nA, lds_dyn);
}

extern "C" int magic_linear_ln(int dtype, int M, int H, int K, const void* x, int lda, const void* W, int ldb, const float* bias,
                               const void* residual, int ldr, const float* gamma, const float* beta, float eps,
                               void* out, float* rstd, const void* drop_seed, float drop_p, unsigned drop_site, void* stream) {
  if (M <= 0 || K <= 0 || !gamma || !beta || !out) return MAGIC_ERR_ARG;
  if (!drop_args_ok(drop_seed, drop_p) || (long long)M * H > 0xFFFFFFFFll) return MAGIC_ERR_ARG;
  if (!dtype_ok(dtype)) return MAGIC_ERR_ARG;
  const int ve = dtype_is16(dtype) ? 8 : 4;
  if (lda % ve || ldb % ve || ((uintptr_t)x & 15) || ((uintptr_t)W & 15)) return MAGIC_ERR_ARG;
  if (H != 128 && H != 256 && H != 384) return MAGIC_ERR_UNSUPPORTED;
  LlnParams p{M, K, x, lda, W, ldb, bias, residual, ldr, gamma, beta, eps, out, rstd,
              DropDesc{drop_p > 0.f ? (const unsigned*)drop_seed : nullptr, drop_site, drop_p}};
  const int ht = H / 64;
  if (group_record(KIND_LLN, dtype, ht, &p, sizeof(p))) return MAGIC_OK;
  return launch_lln(dtype, ht, &p, nullptr, (hipStream_t)stream);
}

// dense -> activation -> LayerNorm in one launch (BertPredictionHeadTransform of the MLM head; the region classifier's Linear / ReLU / LayerNorm):
// magic_linear_ln's kernel with the activation applied between the bias and the row statistics, the pre-activation optionally kept for the backward
extern "C" int magic_linear_act_ln(int dtype, int M, int H, int K, const void* x, int lda, const void* W, int ldb, const float* bias, int act,
                                   void* pre_out, const float* gamma, const float* beta, float eps, void* out, float* rstd, void* stream) {
  if (M <= 0 || K <= 0 || !gamma || !beta || !out || (act != 1 && act != 2) || (long long)M * H > 0xFFFFFFFFll) return MAGIC_ERR_ARG;
  if (!dtype_ok(dtype)) return MAGIC_ERR_ARG;
  const int ve = dtype_is16(dtype) ? 8 : 4;
  if (lda % ve || ldb % ve || ((uintptr_t)x & 15) || ((uintptr_t)W & 15)) return MAGIC_ERR_ARG;
  if (H != 128 && H != 256 && H != 384) return MAGIC_ERR_UNSUPPORTED;
  LlnParams p{M, K, x, lda, W, ldb, bias, nullptr, 0, gamma, beta, eps, out, rstd, DropDesc{nullptr, 0u, 0.f}};
  p.act = act; p.pre_out = pre_out;
  return launch_lln(dtype, H / 64, &p, nullptr, (hipStream_t)stream);
}

int launch_lln(int dtype, int ht, const void* pa, const void* pb, hipStream_t st) {
  const LlnParams& a = *(const LlnParams*)pa;
  dim3 block(256);
  const int nA = (a.M + 31) / 32;
#define LLN1(TY, HT)                                                                                                      \
  do {                                                                                                                    \
    const size_t shm = (size_t)(32 + 64 * HT) * TT<TY>::STRIDE * sizeof(TY) + 128 * sizeof(float);                        \
    if (!pb) {                                                                                                            \
      if (shm > 64 * 1024) (void)hipFuncSetAttribute((const void*)linear_ln_kernel<TY, HT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
      hipLaunchKernelGGL((linear_ln_kernel<TY, HT>), dim3(nA), block, shm, st, a);                                        \
    } else {                                                                                                              \
      const LlnParams& b = *(const LlnParams*)pb;                                                                         \
      if (shm > 64 * 1024) (void)hipFuncSetAttribute((const void*)linear_ln_pair_kernel<TY, HT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
      hipLaunchKernelGGL((linear_ln_pair_kernel<TY, HT>), dim3(nA + (b.M + 31) / 32), block, shm, st, a, b, nA);          \
    }                                                                                                                     \
  } while (0)
  if (dtype == DT_BF16) { if (ht == 2) LLN1(bf16, 2); else if (ht == 4) LLN1(bf16, 4); else LLN1(bf16, 6); }
  else if (dtype == DT_F16) { if (ht == 2) LLN1(f16, 2); else if (ht == 4) LLN1(f16, 4); else LLN1(f16, 6); }
  else { if (ht == 2) LLN1(float, 2); else if (ht == 4) LLN1(float, 4); else LLN1(float, 6); }
#undef LLN1
  return launch_status();
}

// ---------------------------------------------------------------------------------------------------------------
// Backward twin of linear_ln: input-gradient GEMM (NN) + residual + LayerNorm BACKWARD in one launch.
//     v  = X[M,K] @ W[K,H] + R                 (gradient arriving at the OUTPUT of an upstream LayerNorm: e.g. dz @ W1 + d_fo)
//     dx = rstd * (v*gamma - mean(v*gamma) - xhat * mean(v*gamma*xhat)),  xhat = (y - beta) / gamma
//     dgamma += sum_rows v*xhat ; dbeta += sum_rows v ;  dxm = dx * dropout-mask (gradient of the dropped dense branch)
// In the backward chain of a BERT block every input-gradient GEMM that lands on a block boundary is followed by exactly
// such a LayerNorm backward (FFN dX -> attention-output LN; QKV dX -> the previous block's output LN), so the separate
// ln_bwd launch and the round trip of v through memory disappear.  A workgroup owns 32 full rows (as linear_ln), the weight
// is staged in its natural [k][out] orientation and read back transposed (ds_read_b64_tr_b16 / dword reads in fp32).
struct LlbParams {
  int M, K; const void* X; int lda; const void* W; int ldb; const void* R; int ldr;
  const void* y; const float* gamma; const float* beta; const float* rstd;
  void* dx; void* dxm; float* dgamma; float* dbeta; DropDesc drop;
};

template <int SN> __device__ __forceinline__ bf16x8 frag_oc(const bf16* s, int out0, int ks, int lane) { return frag16<false, SN>(s, out0, ks, lane); }
template <int SN> __device__ __forceinline__ f16x8 frag_oc(const f16* s, int out0, int ks, int lane) { return frag16<false, SN>(s, out0, ks, lane); }
template <int SN>
__device__ __forceinline__ float frag_oc(const float* s, int out0, int ks, int lane) {
  return s[(ks * 4 + (lane >> 4)) * SN + out0 + (lane & 15)];
}

template <typename T, int HT>
__device__ __forceinline__ void linear_lnb_body(const LlbParams& pp, const int bid, unsigned char* lds_raw) {
  typedef typename TT<T>::vec vec;
  constexpr int VE = TT<T>::VE, BK = TT<T>::BK, STRIDE = TT<T>::STRIDE;
  constexpr int H = 64 * HT, WC = 16 * HT, SN = H + (sizeof(T) == 2 ? 8 : 4);
  constexpr int KS = (sizeof(T) == 2) ? 2 : 8;
  const int M = pp.M, K = pp.K, lda = pp.lda, ldb = pp.ldb;
  const T* __restrict__ X = (const T*)pp.X; const T* __restrict__ W = (const T*)pp.W;
  T* sA = (T*)lds_raw;                       // [32][STRIDE]  (k contiguous)
  T* sB = sA + 32 * STRIDE;                  // [BK][SN]      (natural: out contiguous)
  float* red = (float*)(sB + BK * SN);       // [2][4][32]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c16 = lane & 15;
  const int m0 = bid * 32;
  f32x4 acc[2][HT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < HT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int VPR_B = H / VE;                       // vectors per k-row of the weight tile
  constexpr int VPT_B = BK * VPR_B / 256;             // per thread
  vec va, vb[VPT_B];
  auto load = [&](int k0) {
    {
      const int row = tid >> 3, cv = tid & 7;
      vec z;
#pragma unroll
      for (int e = 0; e < VE; ++e) z[e] = (T)0.0f;
      if (m0 + row < M && k0 + cv * VE < K) z = *(const vec*)(X + (long long)(m0 + row) * lda + k0 + cv * VE);
      va = z;
    }
#pragma unroll
    for (int i = 0; i < VPT_B; ++i) {
      const int id = tid + 256 * i, kr = id / VPR_B, ov = id % VPR_B;
      vec z;
#pragma unroll
      for (int e = 0; e < VE; ++e) z[e] = (T)0.0f;
      if (k0 + kr < K) z = *(const vec*)(W + (long long)(k0 + kr) * ldb + ov * VE);
      vb[i] = z;
    }
  };
  auto store = [&]() {
    {
      T* d = sA + (tid >> 3) * STRIDE + (tid & 7) * VE;
      if constexpr (sizeof(T) == 2) { *(vec*)d = va; }
      else { ((float2*)d)[0] = make_float2(va[0], va[1]); ((float2*)d)[1] = make_float2(va[2], va[3]); }
    }
#pragma unroll
    for (int i = 0; i < VPT_B; ++i) {
      const int id = tid + 256 * i, kr = id / VPR_B, ov = id % VPR_B;
      *(vec*)(sB + kr * SN + ov * VE) = vb[i];          // SN * sizeof(T) is a multiple of 16
    }
  };
  const int ktiles = (K + BK - 1) / BK;
  load(0);
  // epilogue operands (residual gradient R, the LayerNorm output y, rstd) are fetched before the K loop: after it they were two
  // more dependent round trips to memory per workgroup
  const T* __restrict__ R = (const T*)pp.R; const T* __restrict__ Y = (const T*)pp.y;
  float rv[2][HT][4], yv[2][HT][4], rsv[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + i * 16 + 4 * g + r;
      const bool live = row < M;
      rsv[i][r] = live ? pp.rstd[row] : 0.f;
#pragma unroll
      for (int j = 0; j < HT; ++j) {
        const int col = w * WC + j * 16 + c16;
        rv[i][j][r] = (live && R) ? to_f(R[(long long)row * pp.ldr + col]) : 0.f;
        yv[i][j][r] = live ? to_f(Y[(long long)row * H + col]) : 0.f;
      }
    }
  for (int kt = 0; kt < ktiles; ++kt) {
    store();
    __syncthreads();
    if (kt + 1 < ktiles) load((kt + 1) * BK);
    bool done_x3 = false;
    if constexpr (sizeof(T) == 4) {
      if (g_f32_x3 != 0) {          // fp32 storage, split-bf16 contraction; W tile is the natural [k][out] image
        bf16x8 a0h, a0l, a1h, a1l;
        frag_x3<true>((const float*)sA, 0, lane, a0h, a0l);
        frag_x3<true>((const float*)sA, 16, lane, a1h, a1l);
#pragma unroll
        for (int j = 0; j < HT; ++j) {
          bf16x8 bh, bl;
          frag_x3<false, SN>((const float*)sB, w * WC + j * 16, lane, bh, bl);
          acc[0][j] = mma_x3(a0h, a0l, bh, bl, acc[0][j]);
          acc[1][j] = mma_x3(a1h, a1l, bh, bl, acc[1][j]);
        }
        done_x3 = true;
      }
    }
    if (!done_x3) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const auto a0 = frag<true>(sA, 0, ks, lane), a1 = frag<true>(sA, 16, ks, lane);
#pragma unroll
      for (int j = 0; j < HT; ++j) {
        const auto b = frag_oc<SN>(sB, w * WC + j * 16, ks, lane);
        if constexpr (sizeof(T) == 2) {
          acc[0][j] = mfma16(a0, b, acc[0][j]);
          acc[1][j] = mfma16(a1, b, acc[1][j]);
        } else {
          acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, acc[0][j], 0, 0, 0);
          acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, acc[1][j], 0, 0, 0);
        }
      }
    }
    }
    __syncthreads();
  }
  // ---- epilogue: v = acc + R ; LayerNorm backward over the full row (cross-wave) ; parameter gradients
  float gm[HT], bt[HT], ig[HT];
#pragma unroll
  for (int j = 0; j < HT; ++j) {
    const int col = w * WC + j * 16 + c16;
    gm[j] = pp.gamma[col]; bt[j] = pp.beta[col]; ig[j] = gm[j] != 0.f ? 1.f / gm[j] : 0.f;
  }
  float xh[2][HT][4];
  float s1[2][4], s2[2][4];
  float pg[HT], pb[HT];
#pragma unroll
  for (int j = 0; j < HT; ++j) { pg[j] = 0.f; pb[j] = 0.f; }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + i * 16 + 4 * g + r;
      const bool live = row < M;
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int j = 0; j < HT; ++j) {
        float v = 0.f, x = 0.f;
        if (live) {
          v = acc[i][j][r] + rv[i][j][r];
          x = (yv[i][j][r] - bt[j]) * ig[j];
        }
        pg[j] += v * x; pb[j] += v;
        const float ga = v * gm[j];
        acc[i][j][r] = ga; xh[i][j][r] = x;
        t1 += ga; t2 += ga * x;
      }
      t1 = row16_sum(t1); t2 = row16_sum(t2);
      s1[i][r] = t1; s2[i][r] = t2;
    }
  // gamma / beta gradients: this thread's 8 rows are summed already; fold the 4 row groups (g) and issue one atomic per column.
  // Issued HERE, before the row-statistics barrier and the dx stores: ~120 workgroups hit the same 2H addresses, and the 3-6 us that
  // takes at L2 (profiles/micro/lnb_probe.py) then overlaps the rest of the epilogue instead of trailing it.
  if (pp.dgamma) {
#pragma unroll
    for (int j = 0; j < HT; ++j) {
      float a = pg[j], b = pb[j];
      a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
      b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
      if (g == 0) {
        const int col = w * WC + j * 16 + c16;
        atomicAdd(pp.dgamma + col, a);
        atomicAdd(pp.dbeta + col, b);
      }
    }
  }
  if (c16 == 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        red[w * 32 + i * 16 + 4 * g + r] = s1[i][r];
        red[128 + w * 32 + i * 16 + 4 * g + r] = s2[i][r];
      }
  }
  __syncthreads();
  const DropState dsn = drop_init(pp.drop);
  T* __restrict__ dx = (T*)pp.dx; T* __restrict__ dxm = (T*)pp.dxm;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = i * 16 + 4 * g + r, row = m0 + rr;
      if (row < M) {
        const float m1 = (red[rr] + red[32 + rr] + red[64 + rr] + red[96 + rr]) / H;
        const float m2 = (red[128 + rr] + red[160 + rr] + red[192 + rr] + red[224 + rr]) / H;
        const float rs = rsv[i][r];
#pragma unroll
        for (int j = 0; j < HT; ++j) {
          const int col = w * WC + j * 16 + c16;
          const float d = rs * (acc[i][j][r] - m1 - xh[i][j][r] * m2);
          dx[(long long)row * H + col] = from_f<T>(d);
          if (dxm) dxm[(long long)row * H + col] = from_f<T>(dsn.on ? d * drop_mul(dsn, (unsigned)(row * H + col)) : d);
        }
      }
    }
}

template <typename T, int HT>
__global__ __launch_bounds__(256) void linear_lnb_kernel(LlbParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_dyn[];
  linear_lnb_body<T, HT>(p, blockIdx.x, lds_dyn);
}
template <typename T, int HT>
__global__ __launch_bounds__(256) void linear_lnb_pair_kernel(LlbParams a, LlbParams b, int nA) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_dyn[];
  if ((int)blockIdx.x < nA) linear_lnb_body<T, HT>(a, blockIdx.x, lds_dyn);
  else linear_lnb_body<T, HT>(b, blockIdx.x - nA, lds_dyn);
}

extern "C" int magic_linear_lnbwd(int dtype, int M, int H, int K, const void* x, int lda, const void* W, int ldb,
                                  const void* residual, int ldr, const void* y, const float* gamma, const float* beta, const float* rstd,
                                  void* dx, void* dxm, float* dgamma, float* dbeta,
                                  const void* drop_seed, float drop_p, unsigned drop_site, void* stream) {
  if (M <= 0 || K <= 0 || !x || !W || !y || !gamma || !beta || !rstd || !dx) return MAGIC_ERR_ARG;
  if (!dtype_ok(dtype)) return MAGIC_ERR_ARG;
  if ((dgamma == nullptr) != (dbeta == nullptr)) return MAGIC_ERR_ARG;
  if (!drop_args_ok(drop_seed, drop_p) || (long long)M * H > 0xFFFFFFFFll) return MAGIC_ERR_ARG;
  const int ve = dtype_is16(dtype) ? 8 : 4;
  if (lda % ve || ldb % ve || ldb < H || ((uintptr_t)x & 15) || ((uintptr_t)W & 15)) return MAGIC_ERR_ARG;
  if (H != 128 && H != 256) return MAGIC_ERR_UNSUPPORTED;
  const bool don = drop_p > 0.f && drop_site != 0;
  if (don && !dxm) return MAGIC_ERR_ARG;
  LlbParams p{M, K, x, lda, W, ldb, residual, ldr, y, gamma, beta, rstd, dx, don ? dxm : nullptr, dgamma, dbeta,
              DropDesc{don ? (const unsigned*)drop_seed : nullptr, drop_site, drop_p}};
  const int ht = H / 64;
  if (group_record(KIND_LLB, dtype, ht, &p, sizeof(p))) return MAGIC_OK;
  return launch_llb(dtype, ht, &p, nullptr, (hipStream_t)stream);
}

int launch_llb(int dtype, int ht, const void* pa, const void* pb, hipStream_t st) {
  const LlbParams& a = *(const LlbParams*)pa;
  dim3 block(256);
  const int nA = (a.M + 31) / 32;
#define LLB1(TY, HT)                                                                                                      \
  do {                                                                                                                    \
    constexpr int SN_ = 64 * HT + (sizeof(TY) == 2 ? 8 : 4);                                                              \
    const size_t shm = (size_t)(32 * TT<TY>::STRIDE + TT<TY>::BK * SN_) * sizeof(TY) + 256 * sizeof(float);               \
    if (!pb) {                                                                                                            \
      if (shm > 64 * 1024) (void)hipFuncSetAttribute((const void*)linear_lnb_kernel<TY, HT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
      hipLaunchKernelGGL((linear_lnb_kernel<TY, HT>), dim3(nA), block, shm, st, a);                                       \
    } else {                                                                                                              \
      const LlbParams& b = *(const LlbParams*)pb;                                                                         \
      if (shm > 64 * 1024) (void)hipFuncSetAttribute((const void*)linear_lnb_pair_kernel<TY, HT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
      hipLaunchKernelGGL((linear_lnb_pair_kernel<TY, HT>), dim3(nA + (b.M + 31) / 32), block, shm, st, a, b, nA);         \
    }                                                                                                                     \
  } while (0)
  if (dtype == DT_BF16) { if (ht == 2) LLB1(bf16, 2); else LLB1(bf16, 4); }
  else if (dtype == DT_F16) { if (ht == 2) LLB1(f16, 2); else LLB1(f16, 4); }
  else { if (ht == 2) LLB1(float, 2); else LLB1(float, 4); }
#undef LLB1
  return launch_status();
}

// host-visible descriptor of one weight-gradient problem: dW[N,K] (fp32, ldc) += dY[M,N]^T (lda) @ X[M,K] (ldb); db[N] += colsum(dY)
struct magic_dw_desc { const void* dY; const void* X; float* dW; float* db; int M, N, K, lda, ldb, ldc, splitk; };

// non-empty K-splits of a problem (gemm_block returns early for a split without k-tiles: such a split never arrives at the seam)
static int dw_eff_splits(int dtype, int Kred, int splitk) {
  const int bk = dtype_is16(dtype) ? TT<bf16>::BK : TT<float>::BK;
  const int ktiles = (Kred + bk - 1) / bk, per = (ktiles + splitk - 1) / splitk;
  return (ktiles + per - 1) / per;
}
// Workspace the deterministic form needs for these problems: `floats` fp32 words of partial slots and `counters` 32-bit arrival counters
// (zero them ONCE; every launch leaves them zero).  Problems with the same dW pointer form one group: their splits share that dW's slots.
extern "C" int magic_gemm_dw_ws_need(int dtype, int n, const magic_dw_desc* d, long long* floats, int* counters) {
  if (n <= 0 || n > DW_MAX || !d || !floats || !counters || !dtype_ok(dtype)) return MAGIC_ERR_ARG;
  long long slots = 0;
  int cnt = 0;
  for (int i = 0; i < n; ++i) {
    int leader = i;
    for (int j = 0; j < i; ++j) if (d[j].dW == d[i].dW) { leader = j; break; }
    if (leader != i) continue;
    int total = 0;
    for (int j = i; j < n; ++j) if (d[j].dW == d[i].dW) total += dw_eff_splits(dtype, d[j].M, d[j].splitk);
    const int tiles = ((d[i].K + BN - 1) / BN) * ((d[i].N + BM - 1) / BM);
    slots += (long long)tiles * total;
    cnt += tiles;
  }
  *floats = slots * DW_SLOT;
  *counters = cnt;
  return MAGIC_OK;
}

// Placement of a weight gradient with 1, 2 or 4 K-splits (round 6).  Workgroup id & 7 is the XCD a workgroup lands on, and each XCD has its own L2:
// the plain (tile, split) order deals the 64 x 64 tiles of one split round-robin over all eight, so every XCD fetches nearly every dY / X panel of
// the split (the launch requested 2.8x its algorithmic bytes from L2 and missed on 1.9x).  Here split z owns XCDs [z G, (z + 1) G), G = 8 / splitk,
// and each of them takes a contiguous run of the tiles ordered along the WIDER side of dW: the panels of the narrower operand are fetched G times,
// those of the wider one once.  MAGIC_DW_XCD_GROUPS=0 restores the old order.
static bool dw_xcd_groups() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("MAGIC_DW_XCD_GROUPS"); v = e ? atoi(e) : 1; }
  return v != 0 && gemm_xcd_on();
}

extern "C" int magic_gemm_dw_grouped(int dtype, int n, const magic_dw_desc* d, float* ws, long long ws_floats, unsigned* counters, int n_counters,
                                     void* stream) {
  if (n <= 0 || n > DW_MAX || !d) return MAGIC_ERR_ARG;
  if (!dtype_ok(dtype)) return MAGIC_ERR_ARG;
  if ((ws == nullptr) != (counters == nullptr)) return MAGIC_ERR_ARG;
  const int ve = dtype_is16(dtype) ? 8 : 4;
  DwBatch gp;
  gp.n = n;
  gp.ws = ws; gp.counters = counters;
  if (ws) {
    long long need_f = 0; int need_c = 0;
    if (magic_gemm_dw_ws_need(dtype, n, d, &need_f, &need_c) != MAGIC_OK || need_f > ws_floats || need_c > n_counters) return MAGIC_ERR_ARG;
    if (need_f / DW_SLOT > 0x7fffffffLL) return MAGIC_ERR_ARG;
  }
  int total = 0;
  long long slot_next = 0;
  int cnt_next = 0;
  for (int i = 0; i < n; ++i) {
    if (d[i].M <= 0 || d[i].N <= 0 || d[i].K <= 0 || d[i].splitk <= 0 || !d[i].dY || !d[i].X || !d[i].dW) return MAGIC_ERR_ARG;
    if (d[i].lda % ve || d[i].ldb % ve || ((uintptr_t)d[i].dY & 15) || ((uintptr_t)d[i].X & 15)) return MAGIC_ERR_ARG;
    // TN: A = dY stored [Kred = M][Mout = N], B = X stored [Kred = M][Nout = K], C = dW [N, K]
    DwProblem& p = gp.p[i];
    p.A = d[i].dY; p.B = d[i].X; p.C = d[i].dW; p.bias_grad = d[i].db;
    p.M = d[i].N; p.N = d[i].K; p.K = d[i].M; p.lda = d[i].lda; p.ldb = d[i].ldb; p.ldc = d[i].ldc; p.splitk = d[i].splitk;
    p.ws_slot0 = p.cnt0 = p.ws_total = p.ws_first = 0;
    if (ws) {
      // group = every problem with this dW: the leader (first of them) owns the slots, a member's splits follow its predecessors'
      int leader = i;
      for (int j = 0; j < i; ++j) if (d[j].dW == d[i].dW) { leader = j; break; }
      if (leader == i) {
        int tot = 0;
        for (int j = i; j < n; ++j) if (d[j].dW == d[i].dW) {
          if (d[j].N != d[i].N || d[j].K != d[i].K || d[j].ldc != d[i].ldc || d[j].db != d[i].db) return MAGIC_ERR_ARG;
          tot += dw_eff_splits(dtype, d[j].M, d[j].splitk);
        }
        const int tiles = ((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM);
        p.ws_slot0 = (int)slot_next; p.cnt0 = cnt_next; p.ws_total = tot; p.ws_first = 0;
        slot_next += (long long)tiles * tot;
        cnt_next += tiles;
      } else {
        const DwProblem& l = gp.p[leader];
        p.ws_slot0 = l.ws_slot0; p.cnt0 = l.cnt0; p.ws_total = l.ws_total;
        int first = 0;
        for (int j = leader; j < i; ++j) if (d[j].dW == d[i].dW) first += dw_eff_splits(dtype, d[j].M, d[j].splitk);
        p.ws_first = first;
      }
    }
    const int nx = (p.N + BN - 1) / BN, ny = (p.M + BM - 1) / BM, nz = p.splitk;          // same placement rules as group_place()
    gp.start[i] = total;
    if (gemm_xcd_on() && p.splitk >= 8 && nx * ny >= 2) { p.ny8 = -1; gp.cnt[i] = nx * ny * ((nz + 7) / 8 * 8); }
    else if (dw_xcd_groups() && (nz == 1 || nz == 2 || nz == 4) && nx * ny >= 2) { p.ny8 = -2; gp.cnt[i] = 8 * ((nx * ny + 8 / nz - 1) / (8 / nz)); }
    else if (gemm_xcd_on() && nx >= 2 && ny >= 16) { p.ny8 = (ny + 7) / 8 * 8; gp.cnt[i] = nx * p.ny8 * nz; }
    else { p.ny8 = 0; gp.cnt[i] = nx * ny * nz; }
    total += (gp.cnt[i] + 7) / 8 * 8;
  }
  for (int i = n; i <= DW_MAX; ++i) gp.start[i] = total;
  dim3 grid(total), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_BF16) hipLaunchKernelGGL((gemm_dw_batch_kernel<bf16>), grid, block, 0, st, gp);
  else if (dtype == DT_F16) hipLaunchKernelGGL((gemm_dw_batch_kernel<f16>), grid, block, 0, st, gp);
  else hipLaunchKernelGGL((gemm_dw_batch_kernel<float>), grid, block, 0, st, gp);
  return launch_status();
}
