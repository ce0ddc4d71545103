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
};

template <typename T> struct TT;
template <> struct TT<bf16> {
  static constexpr int VE = 8, BK = 64, STRIDE = 72;
  typedef bf16x8 vec;
};
template <> struct TT<float> {
  static constexpr int VE = 4, BK = 32, STRIDE = 34;
  typedef f32x4 vec;
};

template <typename T, bool KC> struct TileLoader {
  typedef typename TT<T>::vec vec;
  static constexpr int VE = TT<T>::VE, BK = TT<T>::BK, STRIDE = TT<T>::STRIDE;
  vec v[2];
  // tile = 64 out rows x BK k's.  KC: source is [OUT][K] (k contiguous); else [K][OUT].
  __device__ __forceinline__ void load(const T* __restrict__ base, int ld, int out0, int k0, int OUT, int kend) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int id = threadIdx.x + 256 * i;
      vec z;
#pragma unroll
      for (int e = 0; e < VE; ++e) z[e] = (T)0.0f;
      if (KC) {
        int row = id >> 3, cv = id & 7;
        int o = out0 + row, k = k0 + cv * VE;
        if (o < OUT && k < kend) z = *(const vec*)(base + (long long)o * ld + k);
      } else {
        constexpr int VPR = 64 / VE;
        int r = id / VPR, ov = id % VPR;
        int k = k0 + r, o = out0 + ov * VE;
        if (k < kend && o < OUT) z = *(const vec*)(base + (long long)k * ld + o);
      }
      v[i] = z;
    }
  }
  __device__ __forceinline__ void store(T* s) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int id = threadIdx.x + 256 * i;
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
        constexpr int VPR = 64 / VE;
        int r = id / VPR, ov = id % VPR;
#pragma unroll
        for (int e = 0; e < VE; ++e) s[(ov * VE + e) * STRIDE + r] = v[i][e];
      }
    }
  }
};

template <typename T>
__device__ __forceinline__ void mma_tile(const T* sA, const T* sB, int wr, int wc, int lane, f32x4 (&acc)[2][2]);

template <>
__device__ __forceinline__ void mma_tile<bf16>(const bf16* sA, const bf16* sB, int wr, int wc, int lane, f32x4 (&acc)[2][2]) {
  constexpr int S = TT<bf16>::STRIDE;
  const int r = lane & 15, g = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    bf16x8 a[2], b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      a[i] = *(const bf16x8*)(sA + (wr * 32 + i * 16 + r) * S + ks * 32 + 8 * g);
      b[i] = *(const bf16x8*)(sB + (wc * 32 + i * 16 + r) * S + ks * 32 + 8 * g);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
}

template <>
__device__ __forceinline__ void mma_tile<float>(const float* sA, const float* sB, int wr, int wc, int lane, f32x4 (&acc)[2][2]) {
  constexpr int S = TT<float>::STRIDE;
  const int r = lane & 15, g = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    float a[2], b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      a[i] = sA[(wr * 32 + i * 16 + r) * S + ks * 4 + g];
      b[i] = sB[(wc * 32 + i * 16 + r) * S + ks * 4 + g];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
  }
}

// LAYOUT 0: NT (A[M,K], B[N,K]); 1: NN (A[M,K], B[K,N]); 2: TN (A[K,M], B[K,N])
template <typename T, int LAYOUT>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
  constexpr int BK = TT<T>::BK, STRIDE = TT<T>::STRIDE;
  constexpr bool A_KC = (LAYOUT != 2), B_KC = (LAYOUT == 0);
  __shared__ __attribute__((aligned(16))) T sA[BM * STRIDE];
  __shared__ __attribute__((aligned(16))) T sB[BN * STRIDE];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
  const int bz = blockIdx.z / p.splitk, sk = blockIdx.z % p.splitk;
  const int bb = bz / p.nh, bh = bz % p.nh;

  const T* A = (const T*)p.A + bb * p.sAb + bh * p.sAh;
  const T* B = (const T*)p.B + bb * p.sBb + bh * p.sBh;
  const long long coff = bb * p.sCb + bh * p.sCh;

  // split-K range, in whole BK tiles
  const int ktiles = (p.K + BK - 1) / BK;
  const int per = (ktiles + p.splitk - 1) / p.splitk;
  const int kt0 = sk * per, kt1 = min(ktiles, kt0 + per);
  if (kt0 >= kt1 && p.splitk > 1) return;

  TileLoader<T, A_KC> la;
  TileLoader<T, B_KC> lb;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const bool do_bgrad = (LAYOUT == 2) && p.bias_grad != nullptr && blockIdx.x == 0;

  if (kt0 < kt1) {
    la.load(A, p.lda, m0, kt0 * BK, p.M, p.K);
    lb.load(B, p.ldb, n0, kt0 * BK, p.N, p.K);
  }
  for (int kt = kt0; kt < kt1; ++kt) {
    la.store(sA);
    lb.store(sB);
    __syncthreads();
    if (kt + 1 < kt1) {
      la.load(A, p.lda, m0, (kt + 1) * BK, p.M, p.K);
      lb.load(B, p.ldb, n0, (kt + 1) * BK, p.N, p.K);
    }
    mma_tile<T>(sA, sB, wr, wc, lane, acc);
    if (do_bgrad && tid < BM) {
      float s = 0.f;
#pragma unroll 8
      for (int k = 0; k < BK; ++k) s += to_f(sA[tid * STRIDE + k]);
      bsum += s;
    }
    __syncthreads();
  }
  if (do_bgrad && tid < BM && m0 + tid < p.M) atomicAdd(p.bias_grad + m0 + tid, bsum);

  // epilogue: C/D map of 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg
  const int cr = (lane >> 4) * 4, cc = lane & 15;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wc * 32 + j * 16 + cc;
      if (col >= p.N) continue;
      const float bv = (p.bias && sk == 0) ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wr * 32 + i * 16 + cr + r;
        if (row >= p.M) continue;
        float v = acc[i][j][r] * p.alpha + bv;
        if (p.C2) ((T*)p.C2)[coff + (long long)row * p.ldc2 + col] = from_f<T>(v);
        if (p.epilogue == 1) v = gelu_f(v);
        else if (p.epilogue == 2) v = fmaxf(v, 0.f);
        else if (p.epilogue == 3) v *= dgelu_f(to_f(((const T*)p.aux)[coff + (long long)row * p.ldaux + col]));
        else if (p.epilogue == 4) v = to_f(((const T*)p.aux)[coff + (long long)row * p.ldaux + col]) > 0.f ? v : 0.f;
        const long long ci = coff + (long long)row * p.ldc + col;
        if (p.c_f32) {
          if (p.residual) v += ((const float*)p.residual)[coff + (long long)row * p.ldr + col];
          if (p.accumulate) atomicAdd((float*)p.C + ci, v);
          else ((float*)p.C)[ci] = v;
        } else {
          if (p.residual) v += to_f(((const T*)p.residual)[coff + (long long)row * p.ldr + col]);
          ((T*)p.C)[ci] = from_f<T>(v);
        }
      }
    }
}

extern "C" int magic_gemm(int dtype, int layout, int batch, int nh, int M, int N, int K,
                          const void* A, int lda, long long sAb, long long sAh,
                          const void* B, int ldb, long long sBb, long long sBh,
                          void* C, int ldc, long long sCb, long long sCh, int c_f32, int accumulate,
                          const float* bias, int epilogue, const void* aux, int ldaux,
                          const void* residual, int ldr, void* C2, int ldc2,
                          float alpha, int splitk, float* bias_grad, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || batch <= 0 || nh <= 0 || splitk <= 0) return MAGIC_ERR_ARG;
  if (dtype != DT_F32 && dtype != DT_BF16) return MAGIC_ERR_ARG;
  if (layout < 0 || layout > 2) return MAGIC_ERR_ARG;
  const int ve = dtype == DT_BF16 ? 8 : 4;
  if (lda % ve || ldb % ve || (sAb % ve) || (sAh % ve) || (sBb % ve) || (sBh % ve)) return MAGIC_ERR_ARG;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return MAGIC_ERR_ARG;
  if (dtype == DT_F32 && !c_f32) return MAGIC_ERR_ARG;
  if ((accumulate || splitk > 1) && !(c_f32 && accumulate)) return MAGIC_ERR_ARG;
  if (splitk > 1 && (epilogue != 0 || residual || C2)) return MAGIC_ERR_ARG;
  if (bias_grad && layout != 2) return MAGIC_ERR_ARG;
  if (batch % nh) return MAGIC_ERR_ARG;
  GemmParams p;
  p.A = A; p.B = B; p.C = C; p.C2 = C2; p.bias = bias; p.aux = aux; p.residual = residual; p.bias_grad = bias_grad;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldc2 = ldc2; p.ldaux = ldaux; p.ldr = ldr;
  p.sAb = sAb; p.sAh = sAh; p.sBb = sBb; p.sBh = sBh; p.sCb = sCb; p.sCh = sCh;
  p.nh = nh; p.splitk = splitk; p.epilogue = epilogue; p.c_f32 = c_f32; p.accumulate = accumulate; p.alpha = alpha;
  dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, batch * splitk), block(256);
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH(TY, L) hipLaunchKernelGGL((gemm_kernel<TY, L>), grid, block, 0, st, p)
  if (dtype == DT_BF16) {
    if (layout == 0) LAUNCH(bf16, 0); else if (layout == 1) LAUNCH(bf16, 1); else LAUNCH(bf16, 2);
  } else {
    if (layout == 0) LAUNCH(float, 0); else if (layout == 1) LAUNCH(float, 1); else LAUNCH(float, 2);
  }
#undef LAUNCH
  return launch_status();
}
