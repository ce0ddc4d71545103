// Fused attention for the short sequences of this model (text <= 80 tokens, 36/37/38-view panoramas, map nodes):
//   forward : S = Q K^T -> scale + key mask + graph-distance bias -> softmax -> P (kept: backward + attention
//             distillation) -> O = P V, one launch instead of GEMM + softmax + GEMM;
//   backward: dV = P^T dO, dP = dO V^T (+ distillation gradient), dS = softmax', dQ = dS K, dK = dS^T Q, one launch
//             instead of five.  One workgroup owns a whole (batch, head): every operand of the five products is an
//             LDS image read either row-wise (16-byte ds_read) or transposed (ds_read_b64_tr_b16), no global round trips.
// Limits: head dim 64; forward: Nk <= 128 with all keys resident, 128 < Nk <= 512 through the K/V-tiled two-pass kernel
// (attn_fwd_tiled_body); backward: Nq, Nk <= 128 bf16 / <= 64 fp32 (LDS) -- longer sequences (RxR, 512 tokens) take the engine's
// GEMM + softmax backward, which reads the P this kernel saved.
#include "common.hpp"
#include <cstdlib>
#include "group.hpp"

#define HD 64

template <typename T> struct AT;
template <> struct AT<bf16> {
  static constexpr int VE = 8, KSTEP = 32, DS = 72, PPAD = 8;
  typedef bf16x8 vec;
  typedef bf16x8 frag_t;
};
template <> struct AT<f16> {
  static constexpr int VE = 8, KSTEP = 32, DS = 72, PPAD = 8;
  typedef f16x8 vec;
  typedef f16x8 frag_t;
};
template <> struct AT<float> {
  static constexpr int VE = 4, KSTEP = 4, DS = 68, PPAD = 4;
  typedef f32x4 vec;
  typedef float frag_t;
};

// fragment of 16 "out" rows starting at out0, k-slice starting at k0.  KC image: [out][k]; OC image: [k][out].
template <typename Hh> __device__ __forceinline__ h16x8<Hh> fragKC16(const Hh* s, int stride, int out0, int k0, int lane) {
  return *(const h16x8<Hh>*)(s + (out0 + (lane & 15)) * stride + k0 + 8 * (lane >> 4));
}
template <typename Hh> __device__ __forceinline__ h16x8<Hh> fragOC16(const Hh* s, int stride, int out0, int k0, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  return lds_tr8(s + (k0 + 8 * g + q) * stride + out0 + 4 * pp, 4 * stride);
}
__device__ __forceinline__ bf16x8 fragKC(const bf16* s, int stride, int out0, int k0, int lane) { return fragKC16(s, stride, out0, k0, lane); }
__device__ __forceinline__ f16x8 fragKC(const f16* s, int stride, int out0, int k0, int lane) { return fragKC16(s, stride, out0, k0, lane); }
__device__ __forceinline__ bf16x8 fragOC(const bf16* s, int stride, int out0, int k0, int lane) { return fragOC16(s, stride, out0, k0, lane); }
__device__ __forceinline__ f16x8 fragOC(const f16* s, int stride, int out0, int k0, int lane) { return fragOC16(s, stride, out0, k0, lane); }
__device__ __forceinline__ float fragKC(const float* s, int stride, int out0, int k0, int lane) {
  return s[(out0 + (lane & 15)) * stride + k0 + (lane >> 4)];
}
__device__ __forceinline__ float fragOC(const float* s, int stride, int out0, int k0, int lane) {
  return s[(k0 + (lane >> 4)) * stride + out0 + (lane & 15)];
}
__device__ __forceinline__ f32x4 mma(bf16x8 a, bf16x8 b, f32x4 c) { return mfma16(a, b, c); }
__device__ __forceinline__ f32x4 mma(f16x8 a, f16x8 b, f32x4 c) { return mfma16(a, b, c); }
__device__ __forceinline__ f32x4 mma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

__device__ __forceinline__ float group16_max(float v) { return row16_max(v); }
__device__ __forceinline__ float group16_sum(float v) { return row16_sum(v); }

// cooperative load of `rows` x 64 elements (row r of the source at src + r*ld) into an LDS image [rows_pad][DS]; rows >= nvalid -> 0
template <typename T>
__device__ __forceinline__ void load_rows(T* s, const T* src, long long ld, int nvalid, int rows_pad) {
  typedef typename AT<T>::vec vec;
  constexpr int VE = AT<T>::VE, DS = AT<T>::DS, VPR = HD / VE;
  for (int id = threadIdx.x; id < rows_pad * VPR; id += blockDim.x) {
    const int r = id / VPR, c = (id % VPR) * VE;
    vec z;
#pragma unroll
    for (int e = 0; e < VE; ++e) z[e] = (T)0.0f;
    if (r < nvalid) z = *(const vec*)(src + (long long)r * ld + c);
    *(vec*)(s + r * DS + c) = z;
  }
}

struct AttnParams {
  const void *q, *k, *v;
  void *P, *ctx;
  const unsigned char* kmask; const float* dist; const float* sprel_w; const float* sprel_b;
  int B, nh, Nq, Nk, ldq, ldkv, ldp, H;
  float scale;
  // backward only
  const void* dctx; const float* dP_init; void *dq, *dk, *dv; int lddq, lddkv; float* dsprel_w; float* dsprel_b;
  // attention-probability dropout (BertSelfAttention.dropout): P stays the clean softmax (the backward needs it), the
  // product uses P*mask/(1-p); Pd (optional) receives the dropped probabilities -- what HF/METER return as the attention map
  DropDesc drop; void* Pd;
  int acc_kv;        // key-split backward: dK / dV are added to the buffers
};

template <typename T>
__device__ __forceinline__ void attn_fwd_body(const AttnParams& p, const int bqt, const int h, const int b, unsigned char* smem_raw) {
  typedef typename AT<T>::vec vec;
  constexpr int VE = AT<T>::VE, KSTEP = AT<T>::KSTEP, DS = AT<T>::DS;
  const int NKP = (p.Nk + 31) / 32 * 32, PS = NKP + AT<T>::PPAD;
  T* sQ = (T*)smem_raw;            // [64][DS]   (re-used to stage O)
  T* sK = sQ + 64 * DS;            // [NKP][DS]
  T* sV = sK + NKP * DS;           // [NKP][DS]
  T* sP = sV + NKP * DS;           // [64][PS]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c16 = lane & 15;
  const int q0 = bqt * 64;
  const int nq = min(64, p.Nq - q0);
  load_rows<T>(sQ, (const T*)p.q + ((long long)b * p.Nq + q0) * p.ldq + h * HD, p.ldq, nq, 64);
  load_rows<T>(sK, (const T*)p.k + (long long)b * p.Nk * p.ldkv + h * HD, p.ldkv, p.Nk, NKP);
  load_rows<T>(sV, (const T*)p.v + (long long)b * p.Nk * p.ldkv + h * HD, p.ldkv, p.Nk, NKP);
  __syncthreads();
  // ---- S = Q K^T for this wave's 16 query rows
  const int NT = NKP / 16;
  f32x4 acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < HD / KSTEP; ++ks) {
    const auto a = fragKC(sQ, DS, w * 16, ks * KSTEP, lane);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < NT) acc[j] = mma(a, fragKC(sK, DS, j * 16, ks * KSTEP, lane), acc[j]);
  }
  // ---- softmax over keys (row = 4g + r of the wave's tile, key = 16j + c16)
  const float sw = p.dist ? p.sprel_w[0] : 0.f, sb = p.dist ? p.sprel_b[0] : 0.f;
  float mx[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (j < NT) {
      const int key = j * 16 + c16;
      const bool kv = key < p.Nk;
      const float mb = (kv && p.kmask && !p.kmask[(long long)b * p.Nk + key]) ? -10000.0f : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qrow = min(q0 + w * 16 + 4 * g + r, p.Nq - 1);
        float x = acc[j][r] * p.scale + mb;
        if (p.dist && kv) x += sw * p.dist[((long long)b * p.Nq + qrow) * p.Nk + key] + sb;
        x = kv ? x : -3.0e38f;
        acc[j][r] = x; mx[r] = fmaxf(mx[r], x);
      }
    }
  float sum[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { mx[r] = group16_max(mx[r]); sum[r] = 0.f; }
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (j < NT) {
      const bool kv = (j * 16 + c16) < p.Nk;
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float e = kv ? __expf(acc[j][r] - mx[r]) : 0.f; acc[j][r] = e; sum[r] += e; }
    }
#pragma unroll
  for (int r = 0; r < 4; ++r) sum[r] = 1.0f / group16_sum(sum[r]);
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (j < NT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) sP[(w * 16 + 4 * g + r) * PS + j * 16 + c16] = from_f<T>(acc[j][r] * sum[r]);
    }
  __syncthreads();
  // ---- P -> global (16-byte chunks; row pitch ldp), kept for the backward and for attention distillation
  {
    T* Pg = (T*)p.P + (((long long)b * p.nh + h) * p.Nq + q0) * p.ldp;
    const int cpr = p.ldp / VE;
    for (int id = tid; id < nq * cpr; id += 256) {
      const int r = id / cpr, c = (id % cpr) * VE;
      *(vec*)(Pg + (long long)r * p.ldp + c) = *(const vec*)(sP + r * PS + c);
    }
  }
  const DropState ds_ = drop_init(p.drop);
  if (ds_.on) {
    __syncthreads();              // the cooperative P store above has read every row of sP
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < NT) {
        const int key = j * 16 + c16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ql = w * 16 + 4 * g + r;
          const unsigned idx = (unsigned)(((((long long)b * p.nh + h) * p.Nq + q0 + ql) * p.Nk) + key);
          const float m = (ql < nq && key < p.Nk) ? drop_mul(ds_, idx) : 0.f;
          sP[ql * PS + key] = from_f<T>(acc[j][r] * sum[r] * m);
        }
      }
    __syncthreads();
    if (p.Pd) {
      T* Pg = (T*)p.Pd + (((long long)b * p.nh + h) * p.Nq + q0) * p.ldp;
      const int cpr = p.ldp / VE;
      for (int id = tid; id < nq * cpr; id += 256) {
        const int r = id / cpr, c = (id % cpr) * VE;
        *(vec*)(Pg + (long long)r * p.ldp + c) = *(const vec*)(sP + r * PS + c);
      }
    }
  }
  // ---- O = P V
  f32x4 o[4];
#pragma unroll
  for (int jd = 0; jd < 4; ++jd) o[jd] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int ks = 0; ks < NKP / KSTEP; ++ks) {
    const auto a = fragKC(sP, PS, w * 16, ks * KSTEP, lane);
#pragma unroll
    for (int jd = 0; jd < 4; ++jd) o[jd] = mma(a, fragOC(sV, DS, jd * 16, ks * KSTEP, lane), o[jd]);
  }
  // stage O through LDS (sQ is dead: every wave only ever read its own 16 rows of it) for 16-byte global stores
#pragma unroll
  for (int jd = 0; jd < 4; ++jd)
#pragma unroll
    for (int r = 0; r < 4; ++r) sQ[(w * 16 + 4 * g + r) * DS + jd * 16 + c16] = from_f<T>(o[jd][r]);
  __syncthreads();
  {
    T* Og = (T*)p.ctx + ((long long)b * p.Nq + q0) * p.H + h * HD;
    constexpr int VPR = HD / VE;
    for (int id = tid; id < nq * VPR; id += 256) {
      const int r = id / VPR, c = (id % VPR) * VE;
      *(vec*)(Og + (long long)r * p.H + c) = *(const vec*)(sQ + r * DS + c);
    }
  }
}

// ---- K/V-tiled forward for long key sequences (128 < Nk <= 512: RxR-length instructions, BASELINE config 5) ------------------------
// Same contract as attn_fwd_body (P clean [+ dropped Pd], ctx), one workgroup per (batch, head, 64-query tile), but K and V pass
// through LDS in tiles of 128 keys and the softmax is two-pass: pass 1 walks the key tiles keeping only the running row maximum and
// row sum (rescaled tile by tile), pass 2 walks them again, recomputes the scores, normalises with the final statistics, writes the
// probabilities and accumulates P V.  The scores never exist in memory (the unfused path wrote them as fp32 [B, h, Nq, Nk] and read
// them back twice); QK^T is computed twice, which at head dim 64 is cheaper than one round trip of the scores.
#define KTILE 128
template <typename T>
__device__ __forceinline__ void attn_fwd_tiled_body(const AttnParams& p, const int bqt, const int h, const int b, unsigned char* smem_raw) {
  typedef typename AT<T>::vec vec;
  constexpr int VE = AT<T>::VE, KSTEP = AT<T>::KSTEP, DS = AT<T>::DS, PS = KTILE + AT<T>::PPAD;
  T* sQ = (T*)smem_raw;            // [64][DS]   (re-used to stage O)
  T* sK = sQ + 64 * DS;            // [128][DS]
  T* sV = sK + KTILE * DS;         // [128][DS]
  T* sP = sV + KTILE * DS;         // [64][PS]   this tile's probabilities, 16 rows per wave
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c16 = lane & 15;
  const int q0 = bqt * 64;
  const int nq = min(64, p.Nq - q0);
  const int ntiles = (p.Nk + KTILE - 1) / KTILE;
  const T* Kg = (const T*)p.k + (long long)b * p.Nk * p.ldkv + h * HD;
  const T* Vg = (const T*)p.v + (long long)b * p.Nk * p.ldkv + h * HD;
  load_rows<T>(sQ, (const T*)p.q + ((long long)b * p.Nq + q0) * p.ldq + h * HD, p.ldq, nq, 64);
  const float sw = p.dist ? p.sprel_w[0] : 0.f, sb = p.dist ? p.sprel_b[0] : 0.f;
  // scores of this wave's 16 query rows against the key tile in sK -> acc (scaled, masked; -3e38 on keys past Nk)
  auto scores = [&](const int k0, f32x4 (&acc)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < HD / KSTEP; ++ks) {
      const auto a = fragKC(sQ, DS, w * 16, ks * KSTEP, lane);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = mma(a, fragKC(sK, DS, j * 16, ks * KSTEP, lane), acc[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int key = k0 + j * 16 + c16;
      const bool kv = key < p.Nk;
      const float mb = (kv && p.kmask && !p.kmask[(long long)b * p.Nk + key]) ? -10000.0f : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qrow = min(q0 + w * 16 + 4 * g + r, p.Nq - 1);
        float x = acc[j][r] * p.scale + mb;
        if (p.dist && kv) x += sw * p.dist[((long long)b * p.Nq + qrow) * p.Nk + key] + sb;
        acc[j][r] = kv ? x : -3.0e38f;
      }
    }
  };
  // ---- pass 1: running row maximum and row sum over the key tiles
  float mx[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f}, sum[4] = {0.f, 0.f, 0.f, 0.f};
  for (int kt = 0; kt < ntiles; ++kt) {
    const int k0 = kt * KTILE;
    __syncthreads();                                   // previous tile's readers are done with sK
    load_rows<T>(sK, Kg + (long long)k0 * p.ldkv, p.ldkv, min(KTILE, p.Nk - k0), KTILE);
    __syncthreads();
    f32x4 acc[8];
    scores(k0, acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float t = -3.0e38f;
#pragma unroll
      for (int j = 0; j < 8; ++j) t = fmaxf(t, acc[j][r]);
      const float nm = fmaxf(mx[r], group16_max(t));
      float e = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) e += (acc[j][r] > -1.0e38f) ? __expf(acc[j][r] - nm) : 0.f;
      sum[r] = sum[r] * __expf(mx[r] - nm) + group16_sum(e);
      mx[r] = nm;
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) sum[r] = 1.0f / sum[r];
  // ---- pass 2: probabilities (written tile by tile) and O = P V
  const DropState ds_ = drop_init(p.drop);
  f32x4 o[4];
#pragma unroll
  for (int jd = 0; jd < 4; ++jd) o[jd] = (f32x4){0.f, 0.f, 0.f, 0.f};
  T* sPw = sP + w * 16 * PS;
  const int nqw = min(16, nq - w * 16);                 // valid query rows of this wave's tile (may be <= 0)
  for (int kt = 0; kt < ntiles; ++kt) {
    const int k0 = kt * KTILE;
    __syncthreads();
    load_rows<T>(sK, Kg + (long long)k0 * p.ldkv, p.ldkv, min(KTILE, p.Nk - k0), KTILE);
    load_rows<T>(sV, Vg + (long long)k0 * p.ldkv, p.ldkv, min(KTILE, p.Nk - k0), KTILE);
    __syncthreads();
    f32x4 acc[8];
    scores(k0, acc);
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pr = (acc[j][r] > -1.0e38f) ? __expf(acc[j][r] - mx[r]) * sum[r] : 0.f;
        acc[j][r] = pr;
        sPw[(4 * g + r) * PS + j * 16 + c16] = from_f<T>(pr);
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int ncol = min(KTILE, p.ldp - k0);            // columns of this tile inside the row pitch (pad columns hold zeros)
    if (nqw > 0) {
      T* Pg = (T*)p.P + (((long long)b * p.nh + h) * p.Nq + q0 + w * 16) * p.ldp + k0;
      const int cpr = ncol / VE;
      for (int id = lane; id < nqw * cpr; id += 64) {
        const int r = id / cpr, c = (id % cpr) * VE;
        *(vec*)(Pg + (long long)r * p.ldp + c) = *(const vec*)(sPw + r * PS + c);
      }
    }
    if (ds_.on) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int key = k0 + j * 16 + c16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ql = w * 16 + 4 * g + r;
          const unsigned idx = (unsigned)(((((long long)b * p.nh + h) * p.Nq + q0 + ql) * p.Nk) + key);
          const float m = (ql < nq && key < p.Nk) ? drop_mul(ds_, idx) : 0.f;
          sPw[(4 * g + r) * PS + j * 16 + c16] = from_f<T>(acc[j][r] * m);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (p.Pd && nqw > 0) {
        T* Pg = (T*)p.Pd + (((long long)b * p.nh + h) * p.Nq + q0 + w * 16) * p.ldp + k0;
        const int cpr = ncol / VE;
        for (int id = lane; id < nqw * cpr; id += 64) {
          const int r = id / cpr, c = (id % cpr) * VE;
          *(vec*)(Pg + (long long)r * p.ldp + c) = *(const vec*)(sPw + r * PS + c);
        }
      }
    }
    for (int ks = 0; ks < KTILE / KSTEP; ++ks) {
      const auto a = fragKC(sPw, PS, 0, ks * KSTEP, lane);
#pragma unroll
      for (int jd = 0; jd < 4; ++jd) o[jd] = mma(a, fragOC(sV, DS, jd * 16, ks * KSTEP, lane), o[jd]);
    }
  }
  __syncthreads();
#pragma unroll
  for (int jd = 0; jd < 4; ++jd)
#pragma unroll
    for (int r = 0; r < 4; ++r) sQ[(w * 16 + 4 * g + r) * DS + jd * 16 + c16] = from_f<T>(o[jd][r]);
  __syncthreads();
  {
    T* Og = (T*)p.ctx + ((long long)b * p.Nq + q0) * p.H + h * HD;
    constexpr int VPR = HD / VE;
    for (int id = tid; id < nq * VPR; id += 256) {
      const int r = id / VPR, c = (id % VPR) * VE;
      *(vec*)(Og + (long long)r * p.H + c) = *(const vec*)(sQ + r * DS + c);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_tiled_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];
  attn_fwd_tiled_body<T>(p, blockIdx.x, blockIdx.y, blockIdx.z, smem_dyn);
}

template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];
  attn_fwd_body<T>(p, blockIdx.x, blockIdx.y, blockIdx.z, smem_dyn);
}
// two problems in one launch: blocks [0, nA) serve problem a, the rest problem b
template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_pair_kernel(AttnParams a, AttnParams b, int nA) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];
  const bool first = (int)blockIdx.x < nA;
  const AttnParams& p = first ? a : b;
  const int local = first ? blockIdx.x : blockIdx.x - nA;
  const int nqt = (p.Nq + 63) / 64;
  attn_fwd_body<T>(p, local % nqt, (local / nqt) % p.nh, local / (nqt * p.nh), smem_dyn);
}


// ---- key-split kernels for long key sequences (128 < Nk <= 512: RxR-length instructions, BASELINE config 5), 16-bit storage ---------
// The K/V-tiled forward above walks the key tiles one after the other, twice (two-pass softmax): eight dependent load -> barrier -> product
// rounds per workgroup, 62 us per launch at Nk = 486 for 1 GFLOP (profiles/micro/nav_kernel_breakdown.py), and the backward for such keys
// was the unfused chain (three batched GEMMs + softmax backward + two dropout launches per attention).  Here ONE workgroup of 8 waves owns a
// (batch, head, 64-query tile) and EVERY WAVE OWNS A 64-KEY SLAB: all slabs are processed at the same time, the waves meet once to combine
// their softmax statistics (local max / sum per query, merged as in online softmax) and once to add their partial P V.
//   forward : S^T = K_w Q^T (a lane owns 4 consecutive keys of one query: 8-byte LDS traffic) -> local stats -> exchange -> P -> the wave's
//             own K slab becomes its P image -> P (and dropped P) to global in 16-byte rows -> O_w = P_w V_w -> 8-way sum through LDS.
//   backward: one workgroup per (batch, head), loop over 64-query tiles.  rowsum_q(P dP) = dO_q . O_q (the forward's output, kept for the
//             output projection's weight gradient) -- also under dropout, since O was formed with the dropped P -- so a slab needs nothing
//             from the other slabs: dP^T = V_w dO^T with V_w fragments straight from global, dS, dV_w = Pd_w^T dO and dK_w = dS_w^T Q
//             complete per wave, only dQ = sum_w dS_w K_w is added across the waves.  The P / dS slab image is XOR-swizzled instead of
//             padded (that is what makes eight 64 x 64 slabs + eight K slabs + Q + dO fit 160 KB).
//             acc_kv: dK / dV are ADDED to the buffers (the navigator's per-episode K/V cache collects every step's gradient in place).
#define KS_DS 72
__device__ __forceinline__ int sw64(int row, int col) { return row * 64 + ((((col >> 3) ^ (row & 7)) << 3) | (col & 7)); }
template <typename Hh> __device__ __forceinline__ h16x8<Hh> fragKC_sw(const Hh* s, int out0, int k0, int lane) {
  return *(const h16x8<Hh>*)(s + sw64(out0 + (lane & 15), k0 + 8 * (lane >> 4)));
}
template <typename Hh> __device__ __forceinline__ h16x8<Hh> fragOC_sw(const Hh* s, int out0, int k0, int lane) {
  const int r0 = k0 + 8 * (lane >> 4) + ((lane & 15) >> 2), col = out0 + 4 * (lane & 3);
  const h16x4<Hh> lo = lds_tr4(s + sw64(r0, col)), hi = lds_tr4(s + sw64(r0 + 4, col));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
#define WAVE_FENCE() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

template <typename T>
__global__ __launch_bounds__(512) void attn_fwd_ks_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];
  typedef typename AT<T>::vec vec;
  typedef T tv4 __attribute__((ext_vector_type(4)));
  constexpr int DS = KS_DS;
  const int NKP = (p.Nk + 63) / 64 * 64, nact = NKP / 64;
  T* sQ = (T*)smem_dyn;                       // [64][DS]
  T* sK = sQ + 64 * DS;                       // [NKP][DS]; wave w's 64 rows double as its P image [64 queries][DS]
  T* sV = sK + NKP * DS;                      // [NKP][DS]
  float* sst = (float*)(sV + NKP * DS);       // [2][8][64] local max / local sum per (wave, query)
  float* sred = (float*)sK;                   // [nact][64][64] partial outputs once the products are done
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * 64, nq = min(64, p.Nq - q0);
  const bool act = w < nact;
  const int kbase = w * 64;
  {
    // every load of the workgroup is issued before the first LDS store: the Q tile (one 16-byte chunk per thread) and, per wave, its OWN 64-key
    // slabs of K and V (8 + 8 chunks per lane) -- a load -> store -> next-load loop costs one L2 round trip per 8 KB (34 us per launch at Nk = 486)
    vec zq, zk[8], zv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) zq[e] = (T)0.0f;
    const int qr = tid >> 3, qc = (tid & 7) * 8;
    if (qr < nq) zq = *(const vec*)((const T*)p.q + ((long long)b * p.Nq + q0 + qr) * p.ldq + h * HD + qc);
    const T* Kg = (const T*)p.k + (long long)b * p.Nk * p.ldkv + h * HD;
    const T* Vg = (const T*)p.v + (long long)b * p.Nk * p.ldkv + h * HD;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = kbase + i * 8 + (lane >> 3), cc = (lane & 7) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) { zk[i][e] = (T)0.0f; zv[i][e] = (T)0.0f; }
      if (act && r < p.Nk) { zk[i] = *(const vec*)(Kg + (long long)r * p.ldkv + cc); zv[i] = *(const vec*)(Vg + (long long)r * p.ldkv + cc); }
    }
    *(vec*)(sQ + qr * DS + qc) = zq;
    if (act) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = kbase + i * 8 + (lane >> 3), cc = (lane & 7) * 8;
        *(vec*)(sK + r * DS + cc) = zk[i];
        *(vec*)(sV + r * DS + cc) = zv[i];
      }
    }
  }
  __syncthreads();
  f32x4 acc[4][4];                            // [key tile of the slab][query tile]: rows = keys 16 jk + 4 g + r, column = query 16 jq + c
  float mloc[4], sloc[4];
#pragma unroll
  for (int jq = 0; jq < 4; ++jq) { mloc[jq] = -3.0e38f; sloc[jq] = 0.f; }
  if (act) {
#pragma unroll
    for (int jk = 0; jk < 4; ++jk)
#pragma unroll
      for (int jq = 0; jq < 4; ++jq) acc[jk][jq] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      h16x8<T> bq[4];
#pragma unroll
      for (int jq = 0; jq < 4; ++jq) bq[jq] = fragKC(sQ, DS, jq * 16, ks * 32, lane);
#pragma unroll
      for (int jk = 0; jk < 4; ++jk) {
        const auto a = fragKC(sK, DS, kbase + jk * 16, ks * 32, lane);
#pragma unroll
        for (int jq = 0; jq < 4; ++jq) acc[jk][jq] = mma(a, bq[jq], acc[jk][jq]);
      }
    }
#pragma unroll
    for (int jk = 0; jk < 4; ++jk)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kbase + jk * 16 + 4 * g + r;
        const bool kv = key < p.Nk;
        const float mb = (kv && p.kmask && !p.kmask[(long long)b * p.Nk + key]) ? -10000.0f : 0.f;
#pragma unroll
        for (int jq = 0; jq < 4; ++jq) {
          const float x = acc[jk][jq][r] * p.scale + mb;
          acc[jk][jq][r] = kv ? x : -3.0e38f;
        }
      }
#pragma unroll
    for (int jq = 0; jq < 4; ++jq) {
      float m = -3.0e38f;
#pragma unroll
      for (int jk = 0; jk < 4; ++jk)
#pragma unroll
        for (int r = 0; r < 4; ++r) m = fmaxf(m, acc[jk][jq][r]);
      m = fmaxf(m, __shfl_xor(m, 16, 64));           // the four key groups of a query sit 16 lanes apart
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      float e = 0.f;
#pragma unroll
      for (int jk = 0; jk < 4; ++jk)
#pragma unroll
        for (int r = 0; r < 4; ++r) e += (acc[jk][jq][r] > -1.0e38f) ? __expf(acc[jk][jq][r] - m) : 0.f;
      e += __shfl_xor(e, 16, 64);
      e += __shfl_xor(e, 32, 64);
      mloc[jq] = m; sloc[jq] = e;
    }
  }
  if (g == 0) {
#pragma unroll
    for (int jq = 0; jq < 4; ++jq) { sst[w * 64 + jq * 16 + c] = mloc[jq]; sst[512 + w * 64 + jq * 16 + c] = sloc[jq]; }
  }
  __syncthreads();
  f32x4 o[4][4];                              // [query tile][16 head dims]
  if (act) {
    float M[4], inv[4];
#pragma unroll
    for (int jq = 0; jq < 4; ++jq) {
      const int q = jq * 16 + c;
      float mm = -3.0e38f;
#pragma unroll
      for (int ww = 0; ww < 8; ++ww) mm = fmaxf(mm, sst[ww * 64 + q]);
      float t = 0.f;
#pragma unroll
      for (int ww = 0; ww < 8; ++ww) t += sst[512 + ww * 64 + q] * __expf(sst[ww * 64 + q] - mm);
      M[jq] = mm; inv[jq] = 1.0f / t;
    }
    T* sPw = sK + kbase * DS;                 // this wave's K slab is dead (its S^T products are done; no other wave reads it)
#pragma unroll
    for (int jk = 0; jk < 4; ++jk)
#pragma unroll
      for (int jq = 0; jq < 4; ++jq) {
        tv4 o4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float x = acc[jk][jq][r];
          const float pr = (x > -1.0e38f) ? __expf(x - M[jq]) * inv[jq] : 0.f;
          acc[jk][jq][r] = pr; o4[r] = from_f<T>(pr);
        }
        *(tv4*)(sPw + (jq * 16 + c) * DS + jk * 16 + 4 * g) = o4;
      }
    WAVE_FENCE();
    const long long prow0 = ((long long)b * p.nh + h) * p.Nq + q0;
    const int ncolw = min(64, p.ldp - kbase);             // columns of this slab inside the row pitch (pad columns hold zeros)
    if (ncolw > 0) {
      T* Pg = (T*)p.P + prow0 * p.ldp + kbase;
      const int cpr = ncolw / 8;
      for (int id = lane; id < nq * cpr; id += 64) {
        const int r = id / cpr, cc = (id % cpr) * 8;
        *(vec*)(Pg + (long long)r * p.ldp + cc) = *(const vec*)(sPw + r * DS + cc);
      }
    }
    const DropState ds_ = drop_init(p.drop);
    if (ds_.on) {
      WAVE_FENCE();                                       // the row stores above have read the clean image
#pragma unroll
      for (int jk = 0; jk < 4; ++jk)
#pragma unroll
        for (int jq = 0; jq < 4; ++jq) {
          const int ql = jq * 16 + c;
          tv4 o4;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kbase + jk * 16 + 4 * g + r;
            const unsigned idx = (unsigned)((prow0 + ql) * p.Nk + key);
            const float m = (ql < nq && key < p.Nk) ? drop_mul(ds_, idx) : 0.f;
            o4[r] = from_f<T>(acc[jk][jq][r] * m);
          }
          *(tv4*)(sPw + ql * DS + jk * 16 + 4 * g) = o4;
        }
      WAVE_FENCE();
      if (p.Pd && ncolw > 0) {
        T* Pg = (T*)p.Pd + prow0 * p.ldp + kbase;
        const int cpr = ncolw / 8;
        for (int id = lane; id < nq * cpr; id += 64) {
          const int r = id / cpr, cc = (id % cpr) * 8;
          *(vec*)(Pg + (long long)r * p.ldp + cc) = *(const vec*)(sPw + r * DS + cc);
        }
      }
    }
#pragma unroll
    for (int jq = 0; jq < 4; ++jq)
#pragma unroll
      for (int jd = 0; jd < 4; ++jd) o[jq][jd] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      h16x8<T> bv[4];
#pragma unroll
      for (int jd = 0; jd < 4; ++jd) bv[jd] = fragOC(sV, DS, jd * 16, kbase + ks * 32, lane);
#pragma unroll
      for (int jq = 0; jq < 4; ++jq) {
        const auto a = fragKC(sPw, DS, jq * 16, ks * 32, lane);
#pragma unroll
        for (int jd = 0; jd < 4; ++jd) o[jq][jd] = mma(a, bv[jd], o[jq][jd]);
      }
    }
  }
  __syncthreads();                            // every wave is done with the K / V / P images: they become the partial outputs
  if (act) {
#pragma unroll
    for (int jq = 0; jq < 4; ++jq)
#pragma unroll
      for (int jd = 0; jd < 4; ++jd)
#pragma unroll
        for (int r = 0; r < 4; ++r) sred[(w * 64 + jq * 16 + 4 * g + r) * 64 + jd * 16 + c] = o[jq][jd][r];
  }
  __syncthreads();
  {
    const int row = tid >> 3, d0 = (tid & 7) * 8;
    f32x4 s0 = (f32x4){0.f, 0.f, 0.f, 0.f}, s1 = s0;
    for (int ww = 0; ww < nact; ++ww) {
      const float* pr = sred + (ww * 64 + row) * 64 + d0;
      s0 += *(const f32x4*)pr; s1 += *(const f32x4*)(pr + 4);
    }
    if (row < nq) {
      vec ov;
#pragma unroll
      for (int e = 0; e < 4; ++e) { ov[e] = from_f<T>(s0[e]); ov[4 + e] = from_f<T>(s1[e]); }
      *(vec*)((T*)p.ctx + ((long long)b * p.Nq + q0 + row) * p.H + h * HD + d0) = ov;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(512) void attn_bwd_ks_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];
  typedef typename AT<T>::vec vec;
  typedef T tv4 __attribute__((ext_vector_type(4)));
  constexpr int DS = KS_DS, WSLAB = 64 * DS + 64 * 64;     // per wave: K slab [64][DS] + P / dS slab [64 x 64] swizzled (17 408 B >= a 64 x 64 fp32 partial)
  const int NKP = (p.Nk + 63) / 64 * 64, nact = NKP / 64;
  T* sQ = (T*)smem_dyn;                       // [64][DS]
  T* sdO = sQ + 64 * DS;                      // [64][DS]
  T* slabs = sdO + 64 * DS;                   // [8][WSLAB]
  float* srs = (float*)(slabs + 8 * WSLAB);   // [64] rowsum(P dP) = dO . O
  const int tid = threadIdx.x, lane0 = tid & 63, w = tid >> 6;
  const int h = blockIdx.x, b = blockIdx.y;
  const bool act = w < nact;
  const int kbase = w * 64;
  T* sKw = slabs + w * WSLAB;
  T* sSw = sKw + 64 * DS;
  const T* Kg = (const T*)p.k + (long long)b * p.Nk * p.ldkv + h * HD;
  const T* Vg = (const T*)p.v + (long long)b * p.Nk * p.ldkv + h * HD;
  const DropState ds_ = drop_init(p.drop);
  for (int q0 = 0; q0 < p.Nq; q0 += 64) {
    // (the lane id is laundered per tile: hipcc otherwise hoists every per-lane LDS address of the loop body out of the loop and spills ~130 registers)
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int g = lane >> 4, c = lane & 15;
    const int nq = min(64, p.Nq - q0);
    const long long prow0 = ((long long)b * p.nh + h) * p.Nq + q0;
    if (q0) __syncthreads();                  // the previous tile's dQ sum has read the slabs
    load_rows<T>(sQ, (const T*)p.q + ((long long)b * p.Nq + q0) * p.ldq + h * HD, p.ldq, nq, 64);
    const T* dOg = (const T*)p.dctx + ((long long)b * p.Nq + q0) * p.H + h * HD;
    load_rows<T>(sdO, dOg, p.H, nq, 64);
    // a slab whose probabilities are ALL zero for this query tile (keys behind the instruction's padding: exp(-10000 - max) = 0 exactly in fp32)
    // contributes nothing to dQ, dK or dV: the wave skips its products and -- when accumulating -- the read-modify-write of its 64 keys' gradient
    // rows (RxR batches pad ~40 % of the 512 cached keys; that traffic was the largest part of a launch)
    bool live = false;
    if (act) {
      const T* Pg = (const T*)p.P + prow0 * p.ldp;
      vec zk[8], zp[8];                        // all 16 loads of the lane in flight before the first LDS store
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = i * 8 + (lane >> 3), ch = lane & 7;
#pragma unroll
        for (int e = 0; e < 8; ++e) { zk[i][e] = (T)0.0f; zp[i][e] = (T)0.0f; }
        if (kbase + r < p.Nk) zk[i] = *(const vec*)(Kg + (long long)(kbase + r) * p.ldkv + ch * 8);
        if (r < nq && kbase + ch * 8 < p.ldp) zp[i] = *(const vec*)(Pg + (long long)r * p.ldp + kbase + ch * 8);
      }
      typedef unsigned uv4 __attribute__((ext_vector_type(4)));
      unsigned nzb = 0u;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = i * 8 + (lane >> 3), ch = lane & 7;
        *(vec*)(sKw + r * DS + ch * 8) = zk[i];
        *(vec*)(sSw + sw64(r, ch * 8)) = zp[i];
        const uv4 u = __builtin_bit_cast(uv4, zp[i]);
        nzb |= (u[0] | u[1] | u[2] | u[3]) & 0x7FFF7FFFu;          // (-0 counts as zero)
      }
      live = __ballot(nzb != 0u) != 0ull;
    }
    {
      const int r = tid >> 3, ch = tid & 7;
      float s = 0.f;
      if (r < nq) {
        const vec a = *(const vec*)(dOg + (long long)r * p.H + ch * 8);
        const vec o8 = *(const vec*)((const T*)p.ctx + ((long long)b * p.Nq + q0 + r) * p.H + h * HD + ch * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += to_f(a[e]) * to_f(o8[e]);
      }
      s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
      if (ch == 0) srs[r] = s;
    }
    __syncthreads();
    f32x4 oq[4][4];
    if (act && !live) {
      // dead slab: a zero share of dQ; in the storing (non-accumulating) form also zero gradient rows for its keys
      WAVE_FENCE();
      float* part = (float*)sKw;
#pragma unroll
      for (int i = 0; i < 16; ++i) *(f32x4*)(part + (i * 64 + lane) * 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (!(p.acc_kv || q0 > 0)) {
        tv4 z4;
#pragma unroll
        for (int r = 0; r < 4; ++r) z4[r] = from_f<T>(0.f);
#pragma unroll
        for (int jk = 0; jk < 4; ++jk) {
          const int key = kbase + jk * 16 + c;
          if (key < p.Nk) {
#pragma unroll
            for (int jd = 0; jd < 4; ++jd) {
              *(tv4*)((T*)p.dv + ((long long)b * p.Nk + key) * p.lddkv + h * HD + jd * 16 + 4 * g) = z4;
              *(tv4*)((T*)p.dk + ((long long)b * p.Nk + key) * p.lddkv + h * HD + jd * 16 + 4 * g) = z4;
            }
          }
        }
      }
    }
    if (act && live) {
      f32x4 acc[4][4];                        // dP^T: rows = keys 16 jk + 4 g + r of the slab, column = query 16 jq + c
      h16x8<T> vf[4][2];                      // this wave's V slab as A fragments (row = key, k = head dim), straight from global (L2): dead after this product
#pragma unroll
      for (int jk = 0; jk < 4; ++jk)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int key = kbase + jk * 16 + c;
          h16x8<T> z;
#pragma unroll
          for (int e = 0; e < 8; ++e) z[e] = (T)0.0f;
          if (key < p.Nk) z = *(const h16x8<T>*)(Vg + (long long)key * p.ldkv + ks * 32 + 8 * g);
          vf[jk][ks] = z;
        }
#pragma unroll
      for (int jk = 0; jk < 4; ++jk)
#pragma unroll
        for (int jq = 0; jq < 4; ++jq) acc[jk][jq] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        h16x8<T> bq[4];
#pragma unroll
        for (int jq = 0; jq < 4; ++jq) bq[jq] = fragKC(sdO, DS, jq * 16, ks * 32, lane);
#pragma unroll
        for (int jk = 0; jk < 4; ++jk)
#pragma unroll
          for (int jq = 0; jq < 4; ++jq) acc[jk][jq] = mma(vf[jk][ks], bq[jq], acc[jk][jq]);
      }
      tv4 dsr[4][4];
#pragma unroll
      for (int jq = 0; jq < 4; ++jq) {
        const int ql = jq * 16 + c;
        const bool qok = ql < nq;
        const float rsq = srs[ql];
#pragma unroll
        for (int jk = 0; jk < 4; ++jk) {
          const int key0 = jk * 16 + 4 * g;
          const tv4 p4 = *(const tv4*)(sSw + sw64(ql, key0));
          tv4 pm = p4;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kbase + key0 + r;
            const bool ok = qok && key < p.Nk;
            float d = acc[jk][jq][r];
            const float pp = to_f(p4[r]);
            if (ds_.on) {               // d is the gradient wrt the DROPPED probabilities: mask it, and leave P * mask in the image for dV
              const float m = ok ? drop_mul(ds_, (unsigned)((prow0 + ql) * p.Nk + key)) : 0.f;
              d *= m;
              pm[r] = from_f<T>(pp * m);
            }
            dsr[jk][jq][r] = from_f<T>(ok ? pp * (d - rsq) * p.scale : 0.f);
          }
          if (ds_.on) *(tv4*)(sSw + sw64(ql, key0)) = pm;
        }
      }
      WAVE_FENCE();
      const bool addkv = p.acc_kv || q0 > 0;
      // dV_w^T = dO^T Pd_w: formed transposed (rows = head dims, column = key), so a lane owns FOUR CONSECUTIVE HEAD DIMS of one key: 8-byte stores
#pragma unroll
      for (int jk = 0; jk < 4; ++jk) {
        f32x4 ov[4];
#pragma unroll
        for (int jd = 0; jd < 4; ++jd) ov[jd] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kq = 0; kq < 2; ++kq) {
          const auto bp = fragOC_sw(sSw, jk * 16, kq * 32, lane);
#pragma unroll
          for (int jd = 0; jd < 4; ++jd) ov[jd] = mma(fragOC(sdO, DS, jd * 16, kq * 32, lane), bp, ov[jd]);
        }
        const int key = kbase + jk * 16 + c;
        if (key < p.Nk) {
#pragma unroll
          for (int jd = 0; jd < 4; ++jd) {
            tv4* dst = (tv4*)((T*)p.dv + ((long long)b * p.Nk + key) * p.lddkv + h * HD + jd * 16 + 4 * g);
            tv4 o4;
            if (addkv) { const tv4 old = *dst; _Pragma("unroll") for (int r = 0; r < 4; ++r) o4[r] = from_f<T>(to_f(old[r]) + ov[jd][r]); }
            else { _Pragma("unroll") for (int r = 0; r < 4; ++r) o4[r] = from_f<T>(ov[jd][r]); }
            *dst = o4;
          }
        }
      }
      WAVE_FENCE();                                       // the transposed reads of the dropped P are done: the image becomes dS
#pragma unroll
      for (int jq = 0; jq < 4; ++jq)
#pragma unroll
        for (int jk = 0; jk < 4; ++jk) *(tv4*)(sSw + sw64(jq * 16 + c, jk * 16 + 4 * g)) = dsr[jk][jq];
      WAVE_FENCE();
      // dK_w^T = Q^T dS_w (transposed like dV)
#pragma unroll
      for (int jk = 0; jk < 4; ++jk) {
        f32x4 ok_[4];
#pragma unroll
        for (int jd = 0; jd < 4; ++jd) ok_[jd] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kq = 0; kq < 2; ++kq) {
          const auto bp = fragOC_sw(sSw, jk * 16, kq * 32, lane);
#pragma unroll
          for (int jd = 0; jd < 4; ++jd) ok_[jd] = mma(fragOC(sQ, DS, jd * 16, kq * 32, lane), bp, ok_[jd]);
        }
        const int key = kbase + jk * 16 + c;
        if (key < p.Nk) {
#pragma unroll
          for (int jd = 0; jd < 4; ++jd) {
            tv4* dst = (tv4*)((T*)p.dk + ((long long)b * p.Nk + key) * p.lddkv + h * HD + jd * 16 + 4 * g);
            tv4 o4;
            if (addkv) { const tv4 old = *dst; _Pragma("unroll") for (int r = 0; r < 4; ++r) o4[r] = from_f<T>(to_f(old[r]) + ok_[jd][r]); }
            else { _Pragma("unroll") for (int r = 0; r < 4; ++r) o4[r] = from_f<T>(ok_[jd][r]); }
            *dst = o4;
          }
        }
      }
      // this slab's share of dQ = dS_w K_w
#pragma unroll
      for (int jq = 0; jq < 4; ++jq)
#pragma unroll
        for (int jd = 0; jd < 4; ++jd) oq[jq][jd] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        h16x8<T> bk[4];
#pragma unroll
        for (int jd = 0; jd < 4; ++jd) bk[jd] = fragOC(sKw, DS, jd * 16, ks * 32, lane);
#pragma unroll
        for (int jq = 0; jq < 4; ++jq) {
          const auto a = fragKC_sw(sSw, jq * 16, ks * 32, lane);
#pragma unroll
          for (int jd = 0; jd < 4; ++jd) oq[jq][jd] = mma(a, bk[jd], oq[jq][jd]);
        }
      }
      WAVE_FENCE();                                       // own slabs are dead: they hold this wave's partial dQ from here on
      float* part = (float*)sKw;
#pragma unroll
      for (int jq = 0; jq < 4; ++jq)
#pragma unroll
        for (int jd = 0; jd < 4; ++jd)
#pragma unroll
          for (int r = 0; r < 4; ++r) part[(jq * 16 + 4 * g + r) * 64 + jd * 16 + c] = oq[jq][jd][r];
    }
    __syncthreads();
    {
      const int row = tid >> 3, d0 = (tid & 7) * 8;
      f32x4 s0 = (f32x4){0.f, 0.f, 0.f, 0.f}, s1 = s0;
      for (int ww = 0; ww < nact; ++ww) {
        const float* pr = (const float*)(slabs + ww * WSLAB) + row * 64 + d0;
        s0 += *(const f32x4*)pr; s1 += *(const f32x4*)(pr + 4);
      }
      if (row < nq) {
        vec ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) { ov[e] = from_f<T>(s0[e]); ov[4 + e] = from_f<T>(s1[e]); }
        *(vec*)((T*)p.dq + ((long long)b * p.Nq + q0 + row) * p.lddq + h * HD + d0) = ov;
      }
    }
  }
}

static size_t fwd_ks_lds(int Nk) { const int NKP = (Nk + 63) / 64 * 64; return (size_t)(64 + 2 * NKP) * KS_DS * 2 + 4096; }
static size_t bwd_ks_lds() { return (size_t)(2 * 64 * KS_DS + 8 * (64 * KS_DS + 64 * 64)) * 2 + 256; }
static bool ks_mode() { static int m = -1; if (m < 0) { const char* e = getenv("MAGIC_ATTN_NO_KS"); m = (e && atoi(e)) ? 0 : 1; } return m == 1; }

__host__ __device__ static inline bool bwd_alias(int NQP, int NKP, int PS, int DS) { return NQP <= 64 && NQP * PS <= NKP * DS; }

// NW waves per workgroup: 4, or 8 when a side has more than 64 rows (text self-attention, 80 x 80: six 16-row tiles per phase --
// with 4 waves two of them do two tiles in every phase and the whole workgroup waits for them)
#ifdef MAGIC_ATTN_TIMING
// stage clocks of workgroup (0, 0) of the LAST attention-backward launch (100 MHz ticks): profiles/micro/attn_bwd_probe.py
__device__ long long magic_attn_ticks[8];
extern "C" int magic_debug_attn_ticks(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(magic_attn_ticks), sizeof(long long) * 8) == hipSuccess ? 0 : -2; }
#define ATT_MARK(i) do { if (h == 0 && b == 0 && threadIdx.x == 0) magic_attn_ticks[i] = wall_clock64(); } while (0)
#else
#define ATT_MARK(i)
#endif

template <typename T, int NW>
__device__ __forceinline__ void attn_bwd_body(const AttnParams& p, const int h, const int b, unsigned char* smem_raw, float* red) {
  typedef typename AT<T>::vec vec;
  constexpr int VE = AT<T>::VE, KSTEP = AT<T>::KSTEP, DS = AT<T>::DS;
  const int NQP = (p.Nq + 31) / 32 * 32, NKP = (p.Nk + 31) / 32 * 32, PS = NKP + AT<T>::PPAD;
  T* sQ = (T*)smem_raw;             // [NQP][DS]
  T* sdO = sQ + NQP * DS;           // [NQP][DS]
  T* sK = sdO + NQP * DS;           // [NKP][DS]
  T* sV = sK + NKP * DS;            // [NKP][DS]
  T* sP = sV + NKP * DS;            // [NQP][PS]
  // dS normally has its own [NQP][PS] image.  With at most one query tile per wave (NQP <= 64) it can live in V's image, which is
  // dead once every wave has finished its dP = dO V^T products (one extra barrier): 9 KB less LDS per workgroup, which is what lets
  // a third panorama workgroup (Nq = Nk = 36) fit on a CU -- 540 workgroups then run in one round instead of two.
  const bool alias = bwd_alias(NQP, NKP, PS, DS);
  T* sdS = alias ? sV : sP + NQP * PS;           // [NQP][PS]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c16 = lane & 15;
  ATT_MARK(0);
  load_rows<T>(sQ, (const T*)p.q + (long long)b * p.Nq * p.ldq + h * HD, p.ldq, p.Nq, NQP);
  load_rows<T>(sdO, (const T*)p.dctx + (long long)b * p.Nq * p.H + h * HD, p.H, p.Nq, NQP);
  load_rows<T>(sK, (const T*)p.k + (long long)b * p.Nk * p.ldkv + h * HD, p.ldkv, p.Nk, NKP);
  load_rows<T>(sV, (const T*)p.v + (long long)b * p.Nk * p.ldkv + h * HD, p.ldkv, p.Nk, NKP);
  const long long prow0 = ((long long)b * p.nh + h) * p.Nq;
  {
    const T* Pg = (const T*)p.P + prow0 * p.ldp;
    const int cpr = PS / VE;          // PS is a multiple of VE
    for (int id = tid; id < NQP * cpr; id += NW * 64) {
      const int r = id / cpr, c = (id % cpr) * VE;
      vec z;
#pragma unroll
      for (int e = 0; e < VE; ++e) z[e] = (T)0.0f;
      if (r < p.Nq && c < p.ldp) z = *(const vec*)(Pg + (long long)r * p.ldp + c);    // ldp % VE == 0, pad columns hold zeros
      *(vec*)(sP + r * PS + c) = z;
    }
  }
  __syncthreads();
  ATT_MARK(1);
  // ---- phase 1: dP = dO V^T (+ distillation gradient), dS = P (dP - rowsum(P dP)), scaled; sprel gradients
  const int NT = NKP / 16;
  const DropState ds_ = drop_init(p.drop);
  float a0 = 0.f, a1 = 0.f;
  // The product is formed TRANSPOSED, dP^T = V dO^T: tile j then holds keys 16 j + 4 g + r (rows) of query qt 16 + c16 (column), i.e. every lane
  // owns FOUR CONSECUTIVE KEYS of ONE query per tile -- P, the dropped P and dS move through LDS as 8-byte (16-byte: fp32) accesses, the
  // distillation gradient comes in as one 16-byte load, and the row sum over the keys is 24 in-lane terms + two cross-group steps.  (The
  // untransposed form owned one key of four queries per register: 2-byte LDS reads and writes, element by element -- 5.3 us of the
  // kernel's 10.4 at 80 x 80 with dropout on, profiles/micro/attn_bwd_probe.py; now 2.x.)
  typedef T tv4 __attribute__((ext_vector_type(4)));
  auto dP_mma = [&](const int qt, f32x4 (&acc)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < HD / KSTEP; ++ks) {
      const auto bq = fragKC(sdO, DS, qt * 16, ks * KSTEP, lane);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < NT) acc[j] = mma(fragKC(sV, DS, j * 16, ks * KSTEP, lane), bq, acc[j]);
    }
  };
  auto dS_write = [&](const int qt, f32x4 (&acc)[8]) {
    const int q = qt * 16 + c16;
    const bool qok = q < p.Nq;
    float pv[8][4];
    float rs = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < NT) {
        const int key0 = j * 16 + 4 * g;
        const tv4 p4 = *(const tv4*)(sP + q * PS + key0);
        f32x4 init = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (p.dP_init && qok && key0 < p.Nk) init = *(const f32x4*)(p.dP_init + (prow0 + q) * p.ldp + key0);     // fp32 rows of pitch ldp (ldp % 4 == 0, check_common) from a 16-byte-aligned base (magic_attn_bwd checks it): 16-byte aligned; pad columns carry zeros
        tv4 pm = p4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = key0 + r;
          const bool ok = qok && (key < p.Nk);
          float d = acc[j][r];
          if (ok) d += init[r];
          const float pp = to_f(p4[r]);
          if (ds_.on) {               // d is the gradient wrt the DROPPED probabilities: mask it, and leave P*mask in sP for dV
            const float m = ok ? drop_mul(ds_, (unsigned)((prow0 + q) * p.Nk + key)) : 0.f;
            d *= m;
            pm[r] = from_f<T>(pp * m);
          }
          pv[j][r] = pp; acc[j][r] = d; rs += pp * d;
        }
        if (ds_.on) *(tv4*)(sP + q * PS + key0) = pm;
      }
    rs += __shfl_xor(rs, 16, 64);       // the four key groups of a query sit 16 lanes apart
    rs += __shfl_xor(rs, 32, 64);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < NT) {
        const int key0 = j * 16 + 4 * g;
        tv4 o4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float ds = pv[j][r] * (acc[j][r] - rs);
          o4[r] = from_f<T>(ds * p.scale);
          if (p.dist && qok && key0 + r < p.Nk) { a0 += ds * p.dist[((long long)b * p.Nq + q) * p.Nk + key0 + r]; a1 += ds; }
        }
        *(tv4*)(sdS + q * PS + key0) = o4;
      }
    };
  if (alias) {
    f32x4 acc[8];
    const bool has = w < NQP / 16;          // NQP <= 64: at most one query tile per wave
    if (has) dP_mma(w, acc);
    __syncthreads();                        // every wave is done reading V: its image becomes dS
    if (has) dS_write(w, acc);
  } else {
    for (int qt = w; qt < NQP / 16; qt += NW) {
      f32x4 acc[8];
      dP_mma(qt, acc);
      dS_write(qt, acc);
    }
  }
  if (p.dsprel_w) {
    a0 = wave_sum(a0); a1 = wave_sum(a1);
    if (lane == 0) { red[w] = a0; red[NW + w] = a1; }
  }
  __syncthreads();
  if (p.dsprel_w && tid == 0) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) { s0 += red[i]; s1 += red[NW + i]; }
    if (p.dsprel_b) { atomicAdd(p.dsprel_w, s0); atomicAdd(p.dsprel_b, s1); }
    else {                       // round 6 (dsprel_b NULL): dsprel_w is a PARTIAL buffer [B nh][2] -- this workgroup's pair, added up in order by the caller
      p.dsprel_w[2 * (b * p.nh + h)] = s0;
      p.dsprel_w[2 * (b * p.nh + h) + 1] = s1;
    }
  }
  ATT_MARK(2);
  // ---- phase 2: dQ = dS K ; dK = dS^T Q ; dV = P^T dO        (tiles of 16 rows x 64 head dims, round-robin over waves)
  for (int qt = w; qt < NQP / 16; qt += NW) {
    f32x4 o[4];
#pragma unroll
    for (int jd = 0; jd < 4; ++jd) o[jd] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < NKP / KSTEP; ++ks) {
      const auto a = fragKC(sdS, PS, qt * 16, ks * KSTEP, lane);
#pragma unroll
      for (int jd = 0; jd < 4; ++jd) o[jd] = mma(a, fragOC(sK, DS, jd * 16, ks * KSTEP, lane), o[jd]);
    }
#pragma unroll
    for (int jd = 0; jd < 4; ++jd)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = qt * 16 + 4 * g + r;
        if (q < p.Nq) ((T*)p.dq)[((long long)b * p.Nq + q) * p.lddq + h * HD + jd * 16 + c16] = from_f<T>(o[jd][r]);
      }
  }
  ATT_MARK(3);
  for (int kt = w; kt < NKP / 16; kt += NW) {
    f32x4 ok_[4], ov[4];
#pragma unroll
    for (int jd = 0; jd < 4; ++jd) { ok_[jd] = (f32x4){0.f, 0.f, 0.f, 0.f}; ov[jd] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    for (int ks = 0; ks < NQP / KSTEP; ++ks) {
      const auto as = fragOC(sdS, PS, kt * 16, ks * KSTEP, lane);
      const auto ap = fragOC(sP, PS, kt * 16, ks * KSTEP, lane);
#pragma unroll
      for (int jd = 0; jd < 4; ++jd) {
        ok_[jd] = mma(as, fragOC(sQ, DS, jd * 16, ks * KSTEP, lane), ok_[jd]);
        ov[jd] = mma(ap, fragOC(sdO, DS, jd * 16, ks * KSTEP, lane), ov[jd]);
      }
    }
#pragma unroll
    for (int jd = 0; jd < 4; ++jd)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + 4 * g + r;
        if (key < p.Nk) {
          const long long o_ = ((long long)b * p.Nk + key) * p.lddkv + h * HD + jd * 16 + c16;
          ((T*)p.dk)[o_] = from_f<T>(ok_[jd][r]);
          ((T*)p.dv)[o_] = from_f<T>(ov[jd][r]);
        }
      }
  }
  ATT_MARK(4);
#ifdef MAGIC_ATTN_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ATT_MARK(5);
#endif
}

template <typename T, int NW>
__global__ __launch_bounds__(NW * 64) void attn_bwd_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];
  __shared__ float red[2 * NW];
  attn_bwd_body<T, NW>(p, blockIdx.x, blockIdx.y, smem_dyn, red);
}
template <typename T, int NW>
__global__ __launch_bounds__(NW * 64) void attn_bwd_pair_kernel(AttnParams a, AttnParams b, int nA) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];
  __shared__ float red[2 * NW];
  const bool first = (int)blockIdx.x < nA;
  const AttnParams& p = first ? a : b;
  const int local = first ? blockIdx.x : blockIdx.x - nA;
  attn_bwd_body<T, NW>(p, local % p.nh, local / p.nh, smem_dyn, red);
}

static size_t fwd_lds(int dtype, int Nk) {
  const int NKP = (Nk + 31) / 32 * 32;
  if (dtype == DT_BF16) return (size_t)(64 * 72 + 2 * NKP * 72 + 64 * (NKP + 8)) * 2;
  else if (dtype == DT_F16) return (size_t)(64 * 72 + 2 * NKP * 72 + 64 * (NKP + 8)) * 2;
  return (size_t)(64 * 68 + 2 * NKP * 68 + 64 * (NKP + 4)) * 4;
}
static size_t fwd_tiled_lds(int dtype) {
  if (dtype == DT_BF16) return (size_t)(64 * 72 + 2 * KTILE * 72 + 64 * (KTILE + 8)) * 2;
  else if (dtype == DT_F16) return (size_t)(64 * 72 + 2 * KTILE * 72 + 64 * (KTILE + 8)) * 2;
  return (size_t)(64 * 68 + 2 * KTILE * 68 + 64 * (KTILE + 4)) * 4;
}
#define NK_TILED_MAX 512
static size_t bwd_lds(int dtype, int Nq, int Nk) {
  const int NQP = (Nq + 31) / 32 * 32, NKP = (Nk + 31) / 32 * 32;
  if (dtype == DT_BF16) return (size_t)(2 * NQP * 72 + 2 * NKP * 72 + (bwd_alias(NQP, NKP, NKP + 8, 72) ? 1 : 2) * NQP * (NKP + 8)) * 2;
  else if (dtype == DT_F16) return (size_t)(2 * NQP * 72 + 2 * NKP * 72 + (bwd_alias(NQP, NKP, NKP + 8, 72) ? 1 : 2) * NQP * (NKP + 8)) * 2;
  return (size_t)(2 * NQP * 68 + 2 * NKP * 68 + (bwd_alias(NQP, NKP, NKP + 4, 68) ? 1 : 2) * NQP * (NKP + 4)) * 4;
}
#define LDS_MAX (160 * 1024)

// returns 1 if the fused kernels support the shape (host decides fused vs GEMM+softmax path), 0 otherwise
extern "C" int magic_attn_supported(int dtype, int Nq, int Nk, int backward) {
  if (Nk <= 0 || Nq <= 0) return 0;
  if (backward == 2) return (dtype_is16(dtype) && Nk > 128 && Nk <= NK_TILED_MAX && ks_mode()) ? 1 : 0;      // key-split fused backward (magic_attn_bwd_ks)
  if (backward) return (Nk <= 128 && Nq <= 128 && bwd_lds(dtype, Nq, Nk) <= LDS_MAX) ? 1 : 0;
  if (Nk > 128) return Nk <= NK_TILED_MAX ? 1 : 0;            // K/V-tiled two-pass forward
  return fwd_lds(dtype, Nk) <= LDS_MAX ? 1 : 0;
}

static int check_common(int dtype, int B, int nh, int Nq, int Nk, int ldq, int ldkv, int ldp, int H) {
  if (!dtype_ok(dtype)) return MAGIC_ERR_ARG;
  const int ve = dtype_is16(dtype) ? 8 : 4;
  if (B <= 0 || nh <= 0 || Nq <= 0 || Nk <= 0 || H != nh * HD) return MAGIC_ERR_ARG;
  if (ldq % ve || ldkv % ve || ldp % ve || ldp < Nk || H % ve) return MAGIC_ERR_ARG;
  return MAGIC_OK;
}

extern "C" int magic_attn_fwd(int dtype, int B, int nh, int Nq, int Nk, const void* q, int ldq, const void* k, const void* v, int ldkv,
                              void* P, int ldp, void* ctx, int H, float scale, const unsigned char* kmask, const float* dist,
                              const float* sprel_w, const float* sprel_b,
                              const void* drop_seed, float drop_p, unsigned drop_site, void* Pd, void* stream) {
  int rc = check_common(dtype, B, nh, Nq, Nk, ldq, ldkv, ldp, H);
  if (rc) return rc;
  if (!drop_args_ok(drop_seed, drop_p) || (long long)B * nh * Nq * Nk > 0xFFFFFFFFll || ((uintptr_t)Pd & 15)) return MAGIC_ERR_ARG;
  if (!magic_attn_supported(dtype, Nq, Nk, 0)) return MAGIC_ERR_UNSUPPORTED;
  if (dist && (!sprel_w || !sprel_b)) return MAGIC_ERR_ARG;
  if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15) || ((uintptr_t)P & 15) || ((uintptr_t)ctx & 15)) return MAGIC_ERR_ARG;
  AttnParams p = {};
  p.q = q; p.k = k; p.v = v; p.P = P; p.ctx = ctx; p.kmask = kmask; p.dist = dist; p.sprel_w = sprel_w; p.sprel_b = sprel_b;
  p.B = B; p.nh = nh; p.Nq = Nq; p.Nk = Nk; p.ldq = ldq; p.ldkv = ldkv; p.ldp = ldp; p.H = H; p.scale = scale;
  p.drop = DropDesc{drop_p > 0.f ? (const unsigned*)drop_seed : nullptr, drop_site, drop_p}; p.Pd = drop_p > 0.f ? Pd : nullptr;
  if (group_record(KIND_ATTN_FWD, dtype, 0, &p, sizeof(p))) return MAGIC_OK;
  return launch_attn_fwd(dtype, 0, &p, nullptr, (hipStream_t)stream);
}

template <typename K> static void set_lds(K kern, size_t shm) {
  if (shm > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
}

int launch_attn_fwd(int dtype, int, const void* pa, const void* pb, hipStream_t st) {
  const AttnParams& a = *(const AttnParams*)pa;
  dim3 block(256);
  if (a.Nk > 128 || (pb && ((const AttnParams*)pb)->Nk > 128)) {          // long keys: K/V-tiled kernel, one launch per problem
    const AttnParams* ps[2] = {&a, (const AttnParams*)pb};
    for (int i = 0; i < (pb ? 2 : 1); ++i) {
      const AttnParams& q = *ps[i];
      dim3 grid((q.Nq + 63) / 64, q.nh, q.B);
      if (q.Nk > 128 && dtype_is16(dtype) && !q.dist && ks_mode()) {          // key-split: every wave owns a 64-key slab
        const size_t shm = fwd_ks_lds(q.Nk);
        dim3 blk(512);
        if (dtype == DT_BF16) { set_lds(attn_fwd_ks_kernel<bf16>, shm); hipLaunchKernelGGL(attn_fwd_ks_kernel<bf16>, grid, blk, shm, st, q); }
        else { set_lds(attn_fwd_ks_kernel<f16>, shm); hipLaunchKernelGGL(attn_fwd_ks_kernel<f16>, grid, blk, shm, st, q); }
      } else if (q.Nk > 128) {
        const size_t shm = fwd_tiled_lds(dtype);
        if (dtype == DT_BF16) { set_lds(attn_fwd_tiled_kernel<bf16>, shm); hipLaunchKernelGGL(attn_fwd_tiled_kernel<bf16>, grid, block, shm, st, q); }
        else if (dtype == DT_F16) { set_lds(attn_fwd_tiled_kernel<f16>, shm); hipLaunchKernelGGL(attn_fwd_tiled_kernel<f16>, grid, block, shm, st, q); }
        else { set_lds(attn_fwd_tiled_kernel<float>, shm); hipLaunchKernelGGL(attn_fwd_tiled_kernel<float>, grid, block, shm, st, q); }
      } else {
        const size_t shm = fwd_lds(dtype, q.Nk);
        if (dtype == DT_BF16) { set_lds(attn_fwd_kernel<bf16>, shm); hipLaunchKernelGGL(attn_fwd_kernel<bf16>, grid, block, shm, st, q); }
        else if (dtype == DT_F16) { set_lds(attn_fwd_kernel<f16>, shm); hipLaunchKernelGGL(attn_fwd_kernel<f16>, grid, block, shm, st, q); }
        else { set_lds(attn_fwd_kernel<float>, shm); hipLaunchKernelGGL(attn_fwd_kernel<float>, grid, block, shm, st, q); }
      }
    }
    return launch_status();
  }
  if (!pb) {
    dim3 grid((a.Nq + 63) / 64, a.nh, a.B);
    const size_t shm = fwd_lds(dtype, a.Nk);
    if (dtype == DT_BF16) { set_lds(attn_fwd_kernel<bf16>, shm); hipLaunchKernelGGL(attn_fwd_kernel<bf16>, grid, block, shm, st, a); }
    else if (dtype == DT_F16) { set_lds(attn_fwd_kernel<f16>, shm); hipLaunchKernelGGL(attn_fwd_kernel<f16>, grid, block, shm, st, a); }
    else { set_lds(attn_fwd_kernel<float>, shm); hipLaunchKernelGGL(attn_fwd_kernel<float>, grid, block, shm, st, a); }
    return launch_status();
  }
  const AttnParams& b = *(const AttnParams*)pb;
  const int nA = ((a.Nq + 63) / 64) * a.nh * a.B, nB = ((b.Nq + 63) / 64) * b.nh * b.B;
  const size_t sa = fwd_lds(dtype, a.Nk), sb = fwd_lds(dtype, b.Nk), shm = sa > sb ? sa : sb;
  dim3 grid(nA + nB);
  if (dtype == DT_BF16) { set_lds(attn_fwd_pair_kernel<bf16>, shm); hipLaunchKernelGGL(attn_fwd_pair_kernel<bf16>, grid, block, shm, st, a, b, nA); }
  else if (dtype == DT_F16) { set_lds(attn_fwd_pair_kernel<f16>, shm); hipLaunchKernelGGL(attn_fwd_pair_kernel<f16>, grid, block, shm, st, a, b, nA); }
  else { set_lds(attn_fwd_pair_kernel<float>, shm); hipLaunchKernelGGL(attn_fwd_pair_kernel<float>, grid, block, shm, st, a, b, nA); }
  return launch_status();
}

extern "C" int magic_attn_bwd(int dtype, int B, int nh, int Nq, int Nk, const void* q, int ldq, const void* k, const void* v, int ldkv,
                              const void* P, int ldp, const void* dctx, int H, float scale, const float* dP_init,
                              void* dq, int lddq, void* dk, void* dv, int lddkv,
                              const float* dist, float* dsprel_w, float* dsprel_b,
                              const void* drop_seed, float drop_p, unsigned drop_site, void* stream) {
  int rc = check_common(dtype, B, nh, Nq, Nk, ldq, ldkv, ldp, H);
  if (rc) return rc;
  if (!drop_args_ok(drop_seed, drop_p) || (long long)B * nh * Nq * Nk > 0xFFFFFFFFll) return MAGIC_ERR_ARG;
  if (!magic_attn_supported(dtype, Nq, Nk, 1)) return MAGIC_ERR_UNSUPPORTED;
  // dsprel_b NULL with dsprel_w set (round 6): dsprel_w is a partial buffer of 2 B nh floats, every workgroup STORES its (weight, bias) sums; the caller adds them up
  if ((dist == nullptr) != (dsprel_w == nullptr) || (dsprel_b != nullptr && dist == nullptr)) return MAGIC_ERR_ARG;
  if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15) || ((uintptr_t)P & 15) || ((uintptr_t)dctx & 15) || ((uintptr_t)dP_init & 15)) return MAGIC_ERR_ARG;
  AttnParams p = {};
  p.q = q; p.k = k; p.v = v; p.P = (void*)P; p.dist = dist; p.B = B; p.nh = nh; p.Nq = Nq; p.Nk = Nk; p.ldq = ldq; p.ldkv = ldkv;
  p.ldp = ldp; p.H = H; p.scale = scale; p.dctx = dctx; p.dP_init = dP_init; p.dq = dq; p.dk = dk; p.dv = dv; p.lddq = lddq; p.lddkv = lddkv;
  p.dsprel_w = dsprel_w; p.dsprel_b = dsprel_b;
  p.drop = DropDesc{drop_p > 0.f ? (const unsigned*)drop_seed : nullptr, drop_site, drop_p};
  if (group_record(KIND_ATTN_BWD, dtype, 0, &p, sizeof(p))) return MAGIC_OK;
  return launch_attn_bwd(dtype, 0, &p, nullptr, (hipStream_t)stream);
}


// Fused backward for long keys (key-split kernel above).  o = the forward's output [B * Nq, H] (rowsum(P dP) = dO . O); no graph-distance
// bias and no distillation seed on this path (the engine keeps the unfused chain for those); accumulate_kv: dk / dv += instead of =.
extern "C" int magic_attn_bwd_ks(int dtype, int B, int nh, int Nq, int Nk, const void* q, int ldq, const void* k, const void* v, int ldkv,
                                 const void* P, int ldp, const void* o, const void* dctx, int H, float scale,
                                 void* dq, int lddq, void* dk, void* dv, int lddkv, int accumulate_kv,
                                 const void* drop_seed, float drop_p, unsigned drop_site, void* stream) {
  int rc = check_common(dtype, B, nh, Nq, Nk, ldq, ldkv, ldp, H);
  if (rc) return rc;
  if (!drop_args_ok(drop_seed, drop_p) || (long long)B * nh * Nq * Nk > 0xFFFFFFFFll) return MAGIC_ERR_ARG;
  if (!magic_attn_supported(dtype, Nq, Nk, 2)) return MAGIC_ERR_UNSUPPORTED;
  if (!o || !dctx || !dq || !dk || !dv || lddq % 8 || lddkv % 8) return MAGIC_ERR_ARG;
  if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15) || ((uintptr_t)P & 15) || ((uintptr_t)dctx & 15) || ((uintptr_t)o & 15) ||
      ((uintptr_t)dq & 15)) return MAGIC_ERR_ARG;
  AttnParams p = {};
  p.q = q; p.k = k; p.v = v; p.P = (void*)P; p.ctx = (void*)o; p.B = B; p.nh = nh; p.Nq = Nq; p.Nk = Nk; p.ldq = ldq; p.ldkv = ldkv;
  p.ldp = ldp; p.H = H; p.scale = scale; p.dctx = dctx; p.dq = dq; p.dk = dk; p.dv = dv; p.lddq = lddq; p.lddkv = lddkv;
  p.acc_kv = accumulate_kv ? 1 : 0;
  p.drop = DropDesc{drop_p > 0.f ? (const unsigned*)drop_seed : nullptr, drop_site, drop_p};
  const size_t shm = bwd_ks_lds();
  dim3 grid(nh, B), blk(512);
  if (dtype == DT_BF16) { set_lds(attn_bwd_ks_kernel<bf16>, shm); hipLaunchKernelGGL(attn_bwd_ks_kernel<bf16>, grid, blk, shm, (hipStream_t)stream, p); }
  else { set_lds(attn_bwd_ks_kernel<f16>, shm); hipLaunchKernelGGL(attn_bwd_ks_kernel<f16>, grid, blk, shm, (hipStream_t)stream, p); }
  return launch_status();
}

static int bwd_waves8(const AttnParams& p) {
  static int mode = -1;
  if (mode < 0) { const char* e = getenv("MAGIC_ATTN_BWD_WAVES"); mode = e ? atoi(e) : 0; }      // 4 / 8 force, 0 = by shape
  if (mode == 4) return 0;
  if (mode == 8) return 1;
  return (p.Nq > 64 || p.Nk > 64) ? 1 : 0;
}

int launch_attn_bwd(int dtype, int, const void* pa, const void* pb, hipStream_t st) {
  const AttnParams& a = *(const AttnParams*)pa;
  if (!pb) {
    dim3 grid(a.nh, a.B);
    const size_t shm = bwd_lds(dtype, a.Nq, a.Nk);
#define LB(TY, NW) do { set_lds(attn_bwd_kernel<TY, NW>, shm); hipLaunchKernelGGL((attn_bwd_kernel<TY, NW>), grid, dim3(NW * 64), shm, st, a); } while (0)
    if (bwd_waves8(a)) { if (dtype == DT_BF16) LB(bf16, 8); else if (dtype == DT_F16) LB(f16, 8); else LB(float, 8); }
    else { if (dtype == DT_BF16) LB(bf16, 4); else if (dtype == DT_F16) LB(f16, 4); else LB(float, 4); }
#undef LB
    return launch_status();
  }
  const AttnParams& b = *(const AttnParams*)pb;
  const int nA = a.nh * a.B, nB = b.nh * b.B;
  const size_t sa = bwd_lds(dtype, a.Nq, a.Nk), sb = bwd_lds(dtype, b.Nq, b.Nk), shm = sa > sb ? sa : sb;
  dim3 grid(nA + nB);
#define LP(TY, NW) do { set_lds(attn_bwd_pair_kernel<TY, NW>, shm); hipLaunchKernelGGL((attn_bwd_pair_kernel<TY, NW>), grid, dim3(NW * 64), shm, st, a, b, nA); } while (0)
  if (bwd_waves8(a) || bwd_waves8(b)) { if (dtype == DT_BF16) LP(bf16, 8); else if (dtype == DT_F16) LP(f16, 8); else LP(float, 8); }
  else { if (dtype == DT_BF16) LP(bf16, 4); else if (dtype == DT_F16) LP(f16, 4); else LP(float, 4); }
#undef LP
  return launch_status();
}
