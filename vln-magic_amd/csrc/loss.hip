// Loss heads, each computing the per-row loss AND the coefficient-scaled gradient wrt the student
// logits/features in the same pass (the loss coefficients -- MKRW ability weights, kd_alpha, 1/B -- are
// known before backward, so no autograd tape is needed):
//   ce_rows : cross-entropy with -inf-masked logits and ignore_index (+ MKTD weights exp(-rate*CE))
//   kd_rows : temperature KL of pretrain_src/optim/kd_loss.py:18-41 / map_nav_src/utils/kd_loss.py:27-54
//   mse     : weighted MSE of kd_loss.py mse_loss, with (outer, inner) strides for head slicing
#include "common.hpp"
#include <cstdint>

// Dynamic loss scale (fp16 storage; train_r2r_magic.py:370-371 runs its fp16 option under amp.GradScaler): every kernel of this file -- and
// the fused SAP loss of graphops.hip -- that SEEDS a gradient multiplies its gradient coefficient (never the loss value) by the device word
// the host registered with magic_seed_scale(), so the scale can change from step to step under HIP-graph replay with no host round trip.
// NULL (the default): no scaling.  The word is read at launch RECORD time as a pointer only; its value is read by the kernels.
static const float* g_seed_scale = nullptr;          // process-wide: helper threads of a lockstep segment launch loss kernels too
const float* seed_scale_get() { return g_seed_scale; }
extern "C" int magic_seed_scale(const float* scale_dev) { g_seed_scale = scale_dev; return MAGIC_OK; }

// one block (256 threads) per row; N may be large (MLM vocab 50265).  logits dtype T (f32/bf16), ld given.
template <typename T>
__global__ __launch_bounds__(256) void ce_rows_kernel(int M, int N, const T* logits, int ld, const int* labels, int ignore_index,
                                                      float coef, const float* row_w, float* loss_row, T* dlogits, int ldd,
                                                      int accumulate, float* w_out, float w_rate, const float* ss) {
  __shared__ float red[8];
  if (ss) coef *= ss[0];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const T* x = logits + (long long)row * ld;
  const int lab = labels[row];
  const bool ignored = (lab == ignore_index) || lab < 0 || lab >= N;
  const float xl = ignored ? 0.f : to_f(x[lab]);     // read before any in-place gradient write
  float mx = -3.0e38f;
  for (int c = tid; c < N; c += 256) mx = fmaxf(mx, to_f(x[c]));
  mx = wave_max(mx);
  if (lane == 0) red[wid] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int c = tid; c < N; c += 256) s += __expf(to_f(x[c]) - mx);      // exp(-inf) = 0
  s = wave_sum(s);
  if (lane == 0) red[wid] = s;
  __syncthreads();
  s = red[0] + red[1] + red[2] + red[3];
  const float lse = mx + __logf(s);
  const float loss = ignored ? 0.f : lse - xl;
  if (tid == 0) {
    if (loss_row) loss_row[row] = loss;
    if (w_out) w_out[row] = __expf(-w_rate * loss);
  }
  if (dlogits) {
    const float cf = ignored ? 0.f : coef * (row_w ? row_w[row] : 1.f);
    T* d = dlogits + (long long)row * ldd;
    for (int c = tid; c < ldd; c += 256) {
      float g = 0.f;
      if (c < N) {
        g = cf * (__expf(to_f(x[c]) - lse) - (c == lab ? 1.f : 0.f));
        if (accumulate) g += to_f(d[c]);
      }
      d[c] = from_f<T>(g);     // columns [N, ldd) are zeroed (padding contract of the GEMM loaders)
    }
  }
}


// Wide-row variant for bf16 logits (the MLM head: ~570 masked tokens x 50 265 vocabulary entries = 57 MB per step): 16-byte vector
// loads/stores and ONE statistics pass (online max / sum-exp) instead of two, so a row is read twice and written once.  Same
// contract as ce_rows_kernel (in-place gradient allowed: a thread only rewrites vectors it has read itself, and x[label] is read
// by everybody before the first block barrier).  Needs ld, ldd multiples of 8 and 16-byte aligned bases; no accumulate.
// NT threads per row: 1024 when the launch has few rows (the bench's ~100-130 masked tokens: 256-thread blocks left the chip with four waves on
// half its CUs for a 26 MB streaming pass: 36 us -> see DESIGN section 5), 256 otherwise.
template <typename Hh, int NT>
__global__ __launch_bounds__(NT) void ce_rows_wide_kernel(int M, int N, const Hh* logits, int ld, const int* labels, int ignore_index,
                                                          float coef, const float* row_w, float* loss_row, Hh* dlogits, int ldd,
                                                          float* w_out, float w_rate, const float* ss) {
  if (ss) coef *= ss[0];
  constexpr int NWV = NT / 64;
  __shared__ float red[2 * NWV];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const Hh* x = logits + (long long)row * ld;
  const int lab = labels[row];
  const bool ignored = (lab == ignore_index) || lab < 0 || lab >= N;
  const float xl = ignored ? 0.f : to_f(x[lab]);
  const int nvec = (N + 7) >> 3;
  float mx = -3.0e38f, s = 0.f;
  for (int v = tid; v < nvec; v += NT) {
    const h16x8<Hh> q = *(const h16x8<Hh>*)(x + v * 8);
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (v * 8 + e < N) ? (float)q[e] : -3.0e38f;
    float m8 = f[0];
#pragma unroll
    for (int e = 1; e < 8; ++e) m8 = fmaxf(m8, f[e]);
    const float nm = fmaxf(mx, m8);
    float a = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) a += __expf(f[e] - nm);
    s = s * __expf(mx - nm) + a;
    mx = nm;
  }
  // merge (max, sum) pairs: wave, then block
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(mx, o, 64), os = __shfl_xor(s, o, 64);
    const float nm = fmaxf(mx, om);
    s = s * __expf(mx - nm) + os * __expf(om - nm);
    mx = nm;
  }
  if (lane == 0) { red[wid] = mx; red[NWV + wid] = s; }
  __syncthreads();
  mx = red[0];
#pragma unroll
  for (int i = 1; i < NWV; ++i) mx = fmaxf(mx, red[i]);
  s = 0.f;
#pragma unroll
  for (int i = 0; i < NWV; ++i) s += red[NWV + i] * __expf(red[i] - mx);
  const float lse = mx + __logf(s);
  const float loss = ignored ? 0.f : lse - xl;
  if (tid == 0) {
    if (loss_row) loss_row[row] = loss;
    if (w_out) w_out[row] = __expf(-w_rate * loss);
  }
  if (dlogits) {
    const float cf = ignored ? 0.f : coef * (row_w ? row_w[row] : 1.f);
    Hh* d = dlogits + (long long)row * ldd;
    const int dvec = ldd >> 3;
    for (int v = tid; v < dvec; v += NT) {
      h16x8<Hh> g;
      if (v < nvec) {
        const h16x8<Hh> q = *(const h16x8<Hh>*)(x + v * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int c = v * 8 + e;
          g[e] = (Hh)(c < N ? cf * (__expf((float)q[e] - lse) - (c == lab ? 1.f : 0.f)) : 0.f);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = (Hh)0.f;
      }
      *(h16x8<Hh>*)(d + v * 8) = g;
    }
  }
}


// Soft-target KL rows (MRC head: F.kl_div(log_softmax(logits), targets, 'none').sum(1), train_r2r_magic.py:484-485):
//   loss_row = sum_j t_j (log t_j - log p_j)   (t_j = 0 contributes 0) ;   dlogits = coef * ((sum_j t_j) p - t)
// targets fp32 [M, N] (row pitch ldt); one block per row.
template <typename T>
__global__ __launch_bounds__(256) void softkl_rows_kernel(int M, int N, const T* logits, int ld, const float* targets, int ldt,
                                                          float coef, const float* row_w, float* loss_row, T* dlogits, int ldd, const float* ss) {
  __shared__ float red[12];
  if (ss) coef *= ss[0];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const T* x = logits + (long long)row * ld;
  const float* t = targets + (long long)row * ldt;
  float mx = -3.0e38f;
  for (int c = tid; c < N; c += 256) mx = fmaxf(mx, to_f(x[c]));
  mx = wave_max(mx);
  if (lane == 0) red[wid] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f, st = 0.f, a = 0.f;            // sum exp, sum t, sum t (log t - x)
  for (int c = tid; c < N; c += 256) {
    const float xv = to_f(x[c]), tv = t[c];
    s += __expf(xv - mx); st += tv;
    if (tv > 0.f) a += tv * (__logf(tv) - xv);
  }
  s = wave_sum(s); st = wave_sum(st); a = wave_sum(a);
  if (lane == 0) { red[wid] = s; red[4 + wid] = st; red[8 + wid] = a; }
  __syncthreads();
  s = red[0] + red[1] + red[2] + red[3]; st = red[4] + red[5] + red[6] + red[7]; a = red[8] + red[9] + red[10] + red[11];
  const float lse = mx + __logf(s);
  if (tid == 0 && loss_row) loss_row[row] = a + st * lse;      // sum t (log t - x + lse)
  if (dlogits) {
    T* d = dlogits + (long long)row * ldd;
    const float cf = row_w ? coef * row_w[row] : coef;      // shape buckets: the batch's true 1 / n_rows rides in per-row weights
    for (int c = tid; c < ldd; c += 256) d[c] = from_f<T>(c < N ? cf * (st * __expf(to_f(x[c]) - lse) - t[c]) : 0.f);
  }
}

// KD rows: fp32 logits [M,N], N <= 512 (action space).  one wave per row.
// loss_row = w * sum_j p_t (log p_t - log p_s) * T^2 * norm ;  ds (+)= coef * w * T * (p_s - p_t) * norm
__global__ __launch_bounds__(256) void kd_rows_kernel(int M, int N, const float* s, const float* t, int ld, float temperature,
                                                      const float* w, float norm, float coef, const float* coef_dev, float* loss_row, float* ds, int accumulate, const float* ss) {
  if (coef_dev) coef *= coef_dev[0];
  if (ss) coef *= ss[0];
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float NEG = -__builtin_inff();
  float sv[8], tv[8];
  float ms = -3.0e38f, mt = -3.0e38f;
  const float invT = 1.0f / temperature;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int c = it * 64 + lane;
    float a = -3.0e38f, b = -3.0e38f;
    if (c < N) {
      a = s[(long long)row * ld + c]; b = t[(long long)row * ld + c];
      if (a == NEG) a = -1e6f;
      if (b == NEG) b = -1e6f;
      a *= invT; b *= invT;
      ms = fmaxf(ms, a); mt = fmaxf(mt, b);
    }
    sv[it] = a; tv[it] = b;
  }
  ms = wave_max(ms); mt = wave_max(mt);
  float zs = 0.f, zt = 0.f;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int c = it * 64 + lane;
    if (c < N) { zs += __expf(sv[it] - ms); zt += __expf(tv[it] - mt); }
  }
  zs = wave_sum(zs); zt = wave_sum(zt);
  const float ls = ms + __logf(zs), lt = mt + __logf(zt);
  const float wr = w ? w[row] : 1.f;
  float kl = 0.f;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int c = it * 64 + lane;
    if (c < N) {
      const float lpt = tv[it] - lt, lps = sv[it] - ls;
      const float pt = __expf(lpt), ps = __expf(lps);
      if (pt > 0.f) kl += pt * (lpt - lps);
      if (ds) {
        float g = coef * wr * temperature * (ps - pt) * norm;
        const long long i = (long long)row * ld + c;
        ds[i] = accumulate ? ds[i] + g : g;
      }
    }
  }
  kl = wave_sum(kl);
  if (lane == 0 && loss_row) loss_row[row] = wr * kl * temperature * temperature * norm;
}

// MSE over [outer][inner] with independent outer strides for s and t (head slicing), optional per-sample
// weight w[outer / rows_per_w].  loss (atomic, *norm) ; ds = 2 * coef * norm * w * (s - t)  (Tg dtype)
template <typename T, typename G>
__global__ __launch_bounds__(256) void mse_kernel(long long outer, long long inner, const T* s, long long s_stride, const T* t, long long t_stride,
                                                  const float* w, long long rows_per_w, float norm, float coef, const float* coef_dev, float* loss, G* ds, long long g_stride,
                                                  int accumulate, const float* ss) {
  __shared__ float red[4];
  if (coef_dev) coef *= coef_dev[0];
  if (ss) coef *= ss[0];
  const long long total = outer * inner;
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long o = i / inner, r = i % inner;
    const float d = to_f(s[o * s_stride + r]) - to_f(t[o * t_stride + r]);
    const float wv = w ? w[o / rows_per_w] : 1.f;
    acc += wv * d * d;
    if (ds) {
      float g = 2.f * coef * norm * wv * d;
      G* p = ds + o * g_stride + r;
      if (accumulate) g += to_f(*p);
      *p = from_f<G>(g);
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0 && loss) atomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) * norm);
}

// Several independent MSE problems in one launch (the ten MAKD terms of a step are computed back to back between the
// forward and the backward: one launch instead of nine).  Blocks [start[i], start[i+1]) serve problem i.
#define MSE_MAX 10
struct magic_mse_desc {
  int g_f32; long long outer, inner; const void* s; long long s_stride; const void* t; long long t_stride;
  const float* w; long long rows_per_w; float norm, coef; const float* coef_dev; float* loss; void* ds; long long g_stride; int accumulate;
  // shape-bucketed batches under graph replay (host/stream_graph.py): the launch covers the BUCKET's [outer][inner] extent; valid_dev[0..1] =
  // this batch's (outer, inner) -- elements beyond them add nothing to the loss and get a zero gradient -- and norm is multiplied by norm_dev[0]
  const int* valid_dev; const float* norm_dev; long long valid_mod;      // valid_mod > 0: element r of a block is valid iff r % valid_mod < valid_dev[1]
};
struct MseMulti { magic_mse_desc d[MSE_MAX]; int start[MSE_MAX + 1]; int vec[MSE_MAX]; int n; const float* ss; };

#define MSE_NT 1024          // threads per block of the multi-problem launch
template <typename T, typename G>
__device__ __forceinline__ void mse_body(const magic_mse_desc& p, int bid, int nblk, float* red, const float* ss) {
  float coef = p.coef;
  if (p.coef_dev) coef *= p.coef_dev[0];
  if (ss) coef *= ss[0];
  const T* s = (const T*)p.s; const T* t = (const T*)p.t; G* ds = (G*)p.ds;
  const long long total = p.outer * p.inner;
  const long long vo = p.valid_dev ? p.valid_dev[0] : p.outer, vi = p.valid_dev ? p.valid_dev[1] : p.inner;
  const float nrm = p.norm * (p.norm_dev ? p.norm_dev[0] : 1.f);
  float acc = 0.f;
  for (long long i = (long long)bid * MSE_NT + threadIdx.x; i < total; i += (long long)nblk * MSE_NT) {
    const long long o = i / p.inner, r = i % p.inner;
    const bool ok = o < vo && (p.valid_mod > 0 ? r % p.valid_mod : r) < vi;
    const float d = ok ? to_f(s[o * p.s_stride + r]) - to_f(t[o * p.t_stride + r]) : 0.f;
    const float wv = p.w ? p.w[o / p.rows_per_w] : 1.f;
    acc += wv * d * d;
    if (ds) {
      float g = 2.f * coef * nrm * wv * d;
      G* q = ds + o * p.g_stride + r;
      if (p.accumulate) g += to_f(*q);
      *q = from_f<G>(g);
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0 && p.loss) {
    float v = 0.f;
    for (int w = 0; w < MSE_NT / 64; ++w) v += red[w];
    atomicAdd(p.loss, v * nrm);
  }
}

// bf16 problems whose rows are multiples of 8 elements at 16-byte-aligned addresses: 8 elements per lane and iteration, 32-bit index
// arithmetic, two iterations' loads in flight.
template <typename Hh, typename G>
__device__ __forceinline__ void mse_body_v8(const magic_mse_desc& p, int bid, int nblk, float* red, const float* ss) {
  float coef = p.coef;
  if (p.coef_dev) coef *= p.coef_dev[0];
  if (ss) coef *= ss[0];
  const Hh* s = (const Hh*)p.s; const Hh* t = (const Hh*)p.t; G* ds = (G*)p.ds;
  const unsigned in8 = (unsigned)(p.inner >> 3), tot8 = (unsigned)p.outer * in8, rpw = (unsigned)p.rows_per_w;
  const unsigned stride = (unsigned)nblk * MSE_NT;
  const unsigned vo = p.valid_dev ? (unsigned)p.valid_dev[0] : (unsigned)p.outer, vi = p.valid_dev ? (unsigned)p.valid_dev[1] : (unsigned)p.inner;
  const float nrm = p.norm * (p.norm_dev ? p.norm_dev[0] : 1.f);
  const float c2 = 2.f * coef * nrm;
  const unsigned vm = (unsigned)p.valid_mod;
  float acc = 0.f;
  for (unsigned i0 = (unsigned)bid * MSE_NT + threadIdx.x; i0 < tot8; i0 += 2 * stride) {
    h16x8<Hh> sv[2], tv[2];
    unsigned o[2], r[2];
    bool ok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const unsigned i = i0 + u * stride;
      ok[u] = i < tot8;
      const unsigned ii = ok[u] ? i : i0;
      o[u] = ii / in8; r[u] = (ii - o[u] * in8) << 3;
      sv[u] = *(const h16x8<Hh>*)(s + (long long)o[u] * p.s_stride + r[u]);
      tv[u] = *(const h16x8<Hh>*)(t + (long long)o[u] * p.t_stride + r[u]);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (!ok[u]) continue;
      const float wv = p.w ? p.w[o[u] / rpw] : 1.f;
      float d[8];
      float a = 0.f;
      const unsigned rm = vm ? r[u] % vm : r[u];          // (valid_mod is a multiple of 8: a vector never straddles two sub-blocks)
#pragma unroll
      for (int e = 0; e < 8; ++e) { d[e] = (o[u] < vo && rm + e < vi) ? (float)sv[u][e] - (float)tv[u][e] : 0.f; a += d[e] * d[e]; }
      acc += wv * a;
      if (ds) {
        G* q = ds + (long long)o[u] * p.g_stride + r[u];
        const float cw = c2 * wv;
        if constexpr (sizeof(G) == 4) {
          f32x4 g0, g1;
#pragma unroll
          for (int e = 0; e < 4; ++e) { g0[e] = cw * d[e]; g1[e] = cw * d[4 + e]; }
          if (p.accumulate) { g0 += *(const f32x4*)q; g1 += *(const f32x4*)(q + 4); }
          *(f32x4*)q = g0; *(f32x4*)(q + 4) = g1;
        } else {
          h16x8<Hh> g;
          if (p.accumulate) {
            const h16x8<Hh> old = *(const h16x8<Hh>*)q;
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] = (Hh)(cw * d[e] + (float)old[e]);
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] = (Hh)(cw * d[e]);
          }
          *(h16x8<Hh>*)q = g;
        }
      }
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0 && p.loss) {
    float v = 0.f;
    for (int w = 0; w < MSE_NT / 64; ++w) v += red[w];
    atomicAdd(p.loss, v * nrm);
  }
}

template <typename T>
__global__ __launch_bounds__(MSE_NT) void mse_multi_kernel(MseMulti mm) {
  __shared__ float red[MSE_NT / 64];
  int i = 0;
  while (i + 1 < mm.n && (int)blockIdx.x >= mm.start[i + 1]) ++i;
  const magic_mse_desc& p = mm.d[i];
  const int bid = blockIdx.x - mm.start[i], nblk = mm.start[i + 1] - mm.start[i];
  // g_f32 bit 0: the gradient is fp32; bit 1: this problem's INPUTS are fp32 as well (the head-mean attention maps) inside a 16-bit launch
  if constexpr (sizeof(T) == 4) mse_body<float, float>(p, bid, nblk, red, mm.ss);
  else if (p.g_f32 & 2) mse_body<float, float>(p, bid, nblk, red, mm.ss);
  else if (mm.vec[i]) { if (p.g_f32 & 1) mse_body_v8<T, float>(p, bid, nblk, red, mm.ss); else mse_body_v8<T, T>(p, bid, nblk, red, mm.ss); }
  else { if (p.g_f32 & 1) mse_body<T, float>(p, bid, nblk, red, mm.ss); else mse_body<T, T>(p, bid, nblk, red, mm.ss); }
}

// Every block ends in ONE atomic on its problem's loss slot, and the slots of a step's terms share a cache line: same-line atomics
// serialise in L2 at ~15-25 ns each (measured: ~2600 blocks of 256 threads = 38 us for ~40 MB of traffic).  So the launch has about one
// 1024-thread block per CU, shared out among the problems in proportion to their sizes.
#define MSE_BLOCKS 256
extern "C" int magic_mse_multi(int dtype, int n, const magic_mse_desc* d, void* stream) {
  if (n <= 0 || n > MSE_MAX || !d) return MAGIC_ERR_ARG;
  if (!dtype_ok(dtype)) return MAGIC_ERR_ARG;
  MseMulti mm;
  mm.n = n;
  mm.ss = g_seed_scale;
  long long work[MSE_MAX], all = 0;
  for (int i = 0; i < n; ++i) {
    if (d[i].outer <= 0 || d[i].inner <= 0 || !d[i].s || !d[i].t || (d[i].w && d[i].rows_per_w <= 0)) return MAGIC_ERR_ARG;
    mm.d[i] = d[i];
    const long long tot = d[i].outer * d[i].inner;
    const magic_mse_desc& q = d[i];
    const bool vec = dtype_is16(dtype) && !(q.g_f32 & 2) && tot < 0x7FFFFFFFll && q.inner % 8 == 0 && q.s_stride % 8 == 0 && q.t_stride % 8 == 0 &&
                     !((uintptr_t)q.s & 15) && !((uintptr_t)q.t & 15) && (!q.ds || (q.g_stride % 8 == 0 && !((uintptr_t)q.ds & 15))) && q.valid_mod % 8 == 0;
    mm.vec[i] = vec ? 1 : 0;
    work[i] = vec ? (tot + 7) / 8 : tot;          // lane-iterations
    all += work[i];
  }
  int total = 0;
  for (int i = 0; i < n; ++i) {
    long long blocks = (work[i] * MSE_BLOCKS + all - 1) / all;
    const long long need = (work[i] + MSE_NT - 1) / MSE_NT;
    if (blocks > need) blocks = need;
    if (blocks < 1) blocks = 1;
    mm.start[i] = total;
    total += (int)blocks;
  }
  for (int i = n; i <= MSE_MAX; ++i) mm.start[i] = total;
  dim3 grid(total), block(MSE_NT);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_BF16) hipLaunchKernelGGL(mse_multi_kernel<bf16>, grid, block, 0, st, mm);
  else if (dtype == DT_F16) hipLaunchKernelGGL(mse_multi_kernel<f16>, grid, block, 0, st, mm);
  else hipLaunchKernelGGL(mse_multi_kernel<float>, grid, block, 0, st, mm);
  return launch_status();
}

// ---------------------------------------------------------------------------------------------
// The three in-batch contrastive terms of the CFP task in ONE launch, forward AND backward (validate_cfp, train_r2r_magic.py:548-560:
// sim = a txt^T / temp and its transpose, cross-entropy against the diagonal in both directions, for a in {map, viewpoint, fused}):
//   rows[2i][r]   = CE(sim_i[r, :], r)         rows[2i+1][c] = CE(sim_i[:, c], c)
//   G_i = coef (softmax_rows(sim_i) - I) + coef (softmax_cols(sim_i) - I)          (= d1 + d2^T of the two ce_rows calls it replaces)
//   d_a_i = G_i txt / temp                     d_txt = sum_i G_i^T a_i / temp
// B <= 64 samples, H <= 256: everything is tiny (48 x 48 x 128), so the per-op form -- 6 similarity GEMMs, 6 row losses, 6 casts and 12
// gradient GEMMs -- was ~30 launches of pure latency.  One 1024-thread workgroup per term with a_i and txt resident in LDS, fp32 FMAs; the
// three partial d_txt are summed in fixed order by whichever workgroup finishes last (no atomics on data: deterministic).
#define CFP_B 64
template <typename T>
__global__ __launch_bounds__(1024) void cfp_loss_kernel(int B, int H, const T* a0, const T* a1, const T* a2, const T* txt, float inv_temp, float coef,
                                                        float* rows, T* d0, T* d1, T* d2, T* dtxt, float* part, int* counter, const float* ss) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cfp_smem[];
  if (ss) coef *= ss[0];
  constexpr int VE = 16 / (int)sizeof(T);                        // elements per 16-byte vector
  typedef __attribute__((ext_vector_type(VE))) T vec_t;
  const int HS = H + VE;                                         // element pitch: rows stay 16-byte aligned
  T* sa = (T*)cfp_smem;
  T* st = sa + CFP_B * HS;
  float (*sg)[CFP_B + 1] = (float (*)[CFP_B + 1])(st + CFP_B * HS);          // [65][65]; row 64 and the tail hold the two lse vectors
  float* lse_r = &sg[CFP_B][0];
  float* lse_c = lse_r + CFP_B;
  __shared__ int is_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, pair = blockIdx.x;
  const T* a = pair == 0 ? a0 : pair == 1 ? a1 : a2;
  T* da = pair == 0 ? d0 : pair == 1 ? d1 : d2;
  {                                                              // both operands: one 16-byte load per lane and iteration, all issued before any is used
    const int cpr = H / VE, nchunk = CFP_B * cpr;
    for (int i0 = tid; i0 < nchunk; i0 += 2048) {
      vec_t va[2], vt[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int i = i0 + 1024 * u, r = i / cpr, c = (i - r * cpr) * VE;
#pragma unroll
        for (int e = 0; e < VE; ++e) { va[u][e] = from_f<T>(0.f); vt[u][e] = from_f<T>(0.f); }
        if (i < nchunk && r < B) { va[u] = *(const vec_t*)(a + (long long)r * H + c); vt[u] = *(const vec_t*)(txt + (long long)r * H + c); }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int i = i0 + 1024 * u, r = i / cpr, c = (i - r * cpr) * VE;
        if (i < nchunk) { *(vec_t*)(sa + r * HS + c) = va[u]; *(vec_t*)(st + r * HS + c) = vt[u]; }
      }
    }
  }
  __syncthreads();
  // ---- sim = a txt^T / temp: thread owns entries e = tid + 1024 j (r = e / 64, c = e % 64 = lane)
  // (16-byte LDS reads: the txt row of this lane's column once per 8 features -- pitch H + 8 elements = an odd number of 16-byte slots, so
  // the 16 lanes of a read phase hit 16 different slots -- and the four a rows as wave-wide broadcasts; one accumulator per entry, features in
  // order: the bits of the element-by-element loop, which spent the kernel on 2-byte reads four lanes to a bank)
  {
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < H; k0 += VE) {
      const vec_t tv = *(const vec_t*)(st + lane * HS + k0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const vec_t av = *(const vec_t*)(sa + (wave + 16 * j) * HS + k0);
#pragma unroll
        for (int e = 0; e < VE; ++e) s4[j] += to_f(av[e]) * to_f(tv[e]);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) sg[wave + 16 * j][lane] = s4[j] * inv_temp;
  }
  __syncthreads();
  // ---- log-sum-exp of every row and every column (entries outside B x B excluded); wave w owns rows / columns w, w+16, ...
  for (int r = wave; r < B; r += 16) {
    const float x = lane < B ? sg[r][lane] : -3.0e38f, y = lane < B ? sg[lane][r] : -3.0e38f;
    const float mx = wave_max(x), my = wave_max(y);
    const float sx = wave_sum(lane < B ? __expf(x - mx) : 0.f), sy = wave_sum(lane < B ? __expf(y - my) : 0.f);
    if (lane == 0) {
      const float lr = mx + __logf(sx), lc = my + __logf(sy);
      lse_r[r] = lr; lse_c[r] = lc;
      rows[(2 * pair) * B + r] = lr - sg[r][r];
      rows[(2 * pair + 1) * B + r] = lc - sg[r][r];
    }
  }
  if (!dtxt) return;
  __syncthreads();
  // ---- G = coef (P_rows - I) + coef (P_cols - I), in place (each entry read and written by its own thread)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = tid + 1024 * j, r = e >> 6, c = e & 63;
    float g = 0.f;
    if (r < B && c < B) {
      const float x = sg[r][c], d = r == c ? 1.f : 0.f;
      g = coef * ((__expf(x - lse_r[r]) - d) + (__expf(x - lse_c[c]) - d));
    }
    sg[r][c] = g;
  }
  __syncthreads();
  // ---- d_a = G txt / temp ; partial d_txt = G^T a (fp32 scratch): thread owns output pairs (row o / (H/2), features 2 (o % (H/2)), +1)
  // (one 4-byte LDS read per operand pair: the two features of a thread are neighbours)
  typedef __attribute__((ext_vector_type(2))) T vec2_t;
  const int hp = H >> 1;
  for (int o = tid; o < B * hp; o += 1024) {
    const int r = o / hp, h = (o - r * hp) * 2;
    float ga0 = 0.f, ga1 = 0.f, gt0 = 0.f, gt1 = 0.f;
#pragma unroll 8
    for (int q = 0; q < CFP_B; ++q) {
      const float g1 = sg[r][q], g2 = sg[q][r];
      const vec2_t t2 = *(const vec2_t*)(st + q * HS + h), a2v = *(const vec2_t*)(sa + q * HS + h);
      ga0 += g1 * to_f(t2[0]); ga1 += g1 * to_f(t2[1]);
      gt0 += g2 * to_f(a2v[0]); gt1 += g2 * to_f(a2v[1]);
    }
    da[(long long)r * H + h] = from_f<T>(ga0 * inv_temp); da[(long long)r * H + h + 1] = from_f<T>(ga1 * inv_temp);
    float* pp = part + (long long)pair * B * H + (long long)r * H + h;
    pp[0] = gt0; pp[1] = gt1;
  }
  __threadfence();
  __syncthreads();
  if (tid == 0) is_last = (atomicAdd(counter, 1) == 2) ? 1 : 0;
  __syncthreads();
  if (is_last) {
    __threadfence();
    for (int o = tid; o < B * H; o += 1024) {
      const float v = (__builtin_nontemporal_load(part + o) + __builtin_nontemporal_load(part + (long long)B * H + o)) + __builtin_nontemporal_load(part + 2ll * B * H + o);
      dtxt[o] = from_f<T>(v * inv_temp);
    }
    if (tid == 0) *counter = 0;
  }
}

static size_t cfp_lds_bytes(int H, int elt) { return (size_t)2 * CFP_B * (H + 16 / elt) * elt + 16 + (size_t)(CFP_B + 1) * (CFP_B + 1) * 4 + 2 * CFP_B * 4; }

extern "C" int magic_cfp_loss(int dtype, int B, int H, const void* a0, const void* a1, const void* a2, const void* txt, float temperature, float coef,
                              float* rows, void* d0, void* d1, void* d2, void* dtxt, float* part, int* counter, void* stream) {
  if (B <= 0 || B > CFP_B || H <= 0 || H > 256 || (H & 7) || !a0 || !a1 || !a2 || !txt || !rows || temperature <= 0.f) return MAGIC_ERR_ARG;
  if (((uintptr_t)a0 | (uintptr_t)a1 | (uintptr_t)a2 | (uintptr_t)txt) & 15) return MAGIC_ERR_ARG;
  if ((d0 == nullptr) != (d1 == nullptr) || (d0 == nullptr) != (d2 == nullptr) || (d0 == nullptr) != (dtxt == nullptr)) return MAGIC_ERR_ARG;
  if (dtxt && (!part || !counter)) return MAGIC_ERR_ARG;
  if (!dtype_ok(dtype)) return MAGIC_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const size_t shm = cfp_lds_bytes(H, dtype_is16(dtype) ? 2 : 4);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)cfp_loss_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cfp_lds_bytes(256, 2));
    (void)hipFuncSetAttribute((const void*)cfp_loss_kernel<f16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cfp_lds_bytes(256, 2));
    (void)hipFuncSetAttribute((const void*)cfp_loss_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cfp_lds_bytes(256, 4));
    attr_set = true;
  }
  if (dtype == DT_BF16) hipLaunchKernelGGL(cfp_loss_kernel<bf16>, dim3(3), dim3(1024), shm, st, B, H, (const bf16*)a0, (const bf16*)a1, (const bf16*)a2, (const bf16*)txt,
                                           1.f / temperature, coef, rows, (bf16*)d0, (bf16*)d1, (bf16*)d2, (bf16*)dtxt, part, counter, g_seed_scale);
  else if (dtype == DT_F16) hipLaunchKernelGGL(cfp_loss_kernel<f16>, dim3(3), dim3(1024), shm, st, B, H, (const f16*)a0, (const f16*)a1, (const f16*)a2, (const f16*)txt,
                                               1.f / temperature, coef, rows, (f16*)d0, (f16*)d1, (f16*)d2, (f16*)dtxt, part, counter, g_seed_scale);
  else hipLaunchKernelGGL(cfp_loss_kernel<float>, dim3(3), dim3(1024), shm, st, B, H, (const float*)a0, (const float*)a1, (const float*)a2, (const float*)txt,
                          1.f / temperature, coef, rows, (float*)d0, (float*)d1, (float*)d2, (float*)dtxt, part, counter, g_seed_scale);
  return launch_status();
}

extern "C" int magic_ce_rows(int dtype, int M, int N, const void* logits, int ld, const int* labels, int ignore_index,
                             float coef, const float* row_w, float* loss_row, void* dlogits, int ldd, int accumulate,
                             float* w_out, float w_rate, void* stream) {
  if (M <= 0 || N <= 0 || ld < N || (dlogits && ldd < N)) return MAGIC_ERR_ARG;
  dim3 grid(M), block(256);
  hipStream_t st = (hipStream_t)stream;
  const bool wide = dtype_is16(dtype) && N >= 2048 && !accumulate && (ld % 8) == 0 && (!dlogits || (ldd % 8) == 0) &&
                    (((uintptr_t)logits | (uintptr_t)dlogits) & 15) == 0;
  const bool big = M <= 512;            // few rows: 1024 threads per row
  if (wide && dtype == DT_BF16 && big)
    hipLaunchKernelGGL((ce_rows_wide_kernel<bf16, 1024>), grid, dim3(1024), 0, st, M, N, (const bf16*)logits, ld, labels, ignore_index, coef, row_w, loss_row, (bf16*)dlogits, ldd, w_out, w_rate, g_seed_scale);
  else if (wide && dtype == DT_BF16)
    hipLaunchKernelGGL((ce_rows_wide_kernel<bf16, 256>), grid, block, 0, st, M, N, (const bf16*)logits, ld, labels, ignore_index, coef, row_w, loss_row, (bf16*)dlogits, ldd, w_out, w_rate, g_seed_scale);
  else if (wide && big)
    hipLaunchKernelGGL((ce_rows_wide_kernel<f16, 1024>), grid, dim3(1024), 0, st, M, N, (const f16*)logits, ld, labels, ignore_index, coef, row_w, loss_row, (f16*)dlogits, ldd, w_out, w_rate, g_seed_scale);
  else if (wide)
    hipLaunchKernelGGL((ce_rows_wide_kernel<f16, 256>), grid, block, 0, st, M, N, (const f16*)logits, ld, labels, ignore_index, coef, row_w, loss_row, (f16*)dlogits, ldd, w_out, w_rate, g_seed_scale);
  else if (dtype == DT_BF16)
    hipLaunchKernelGGL(ce_rows_kernel<bf16>, grid, block, 0, st, M, N, (const bf16*)logits, ld, labels, ignore_index, coef, row_w, loss_row, (bf16*)dlogits, ldd, accumulate, w_out, w_rate, g_seed_scale);
  else if (dtype == DT_F16)
    hipLaunchKernelGGL(ce_rows_kernel<f16>, grid, block, 0, st, M, N, (const f16*)logits, ld, labels, ignore_index, coef, row_w, loss_row, (f16*)dlogits, ldd, accumulate, w_out, w_rate, g_seed_scale);
  else
    hipLaunchKernelGGL(ce_rows_kernel<float>, grid, block, 0, st, M, N, (const float*)logits, ld, labels, ignore_index, coef, row_w, loss_row, (float*)dlogits, ldd, accumulate, w_out, w_rate, g_seed_scale);
  return launch_status();
}

extern "C" int magic_softkl_rows(int dtype, int M, int N, const void* logits, int ld, const float* targets, int ldt, float coef,
                                 const float* row_w, float* loss_row, void* dlogits, int ldd, void* stream) {
  if (M <= 0 || N <= 0 || ld < N || ldt < N || !logits || !targets || (dlogits && ldd < N) || dlogits == logits) return MAGIC_ERR_ARG;
  dim3 grid(M), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(softkl_rows_kernel<bf16>, grid, block, 0, st, M, N, (const bf16*)logits, ld, targets, ldt, coef, row_w, loss_row, (bf16*)dlogits, ldd, g_seed_scale);
  else if (dtype == DT_F16)
    hipLaunchKernelGGL(softkl_rows_kernel<f16>, grid, block, 0, st, M, N, (const f16*)logits, ld, targets, ldt, coef, row_w, loss_row, (f16*)dlogits, ldd, g_seed_scale);
  else if (dtype == DT_F32)
    hipLaunchKernelGGL(softkl_rows_kernel<float>, grid, block, 0, st, M, N, (const float*)logits, ld, targets, ldt, coef, row_w, loss_row, (float*)dlogits, ldd, g_seed_scale);
  else return MAGIC_ERR_ARG;
  return launch_status();
}

extern "C" int magic_kd_rows(int M, int N, const float* s, const float* t, int ld, float temperature, const float* w, float norm,
                             float coef, const float* coef_dev, float* loss_row, float* ds, int accumulate, void* stream) {
  if (M <= 0 || N <= 0 || N > 512 || ld < N || temperature <= 0.f) return MAGIC_ERR_ARG;
  dim3 grid((M + 3) / 4), block(256);
  hipLaunchKernelGGL(kd_rows_kernel, grid, block, 0, (hipStream_t)stream, M, N, s, t, ld, temperature, w, norm, coef, coef_dev, loss_row, ds, accumulate, g_seed_scale);
  return launch_status();
}

extern "C" int magic_mse(int dtype, int g_f32, long long outer, long long inner, const void* s, long long s_stride, const void* t,
                         long long t_stride, const float* w, long long rows_per_w, float norm, float coef, const float* coef_dev, float* loss, void* ds,
                         long long g_stride, int accumulate, void* stream) {
  if (outer <= 0 || inner <= 0 || (w && rows_per_w <= 0)) return MAGIC_ERR_ARG;
  long long total = outer * inner;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 384) blocks = 384;      // every block ends in ONE atomic on the loss word: keep the fan-in small
  dim3 grid(blocks), block(256);
  hipStream_t st = (hipStream_t)stream;
#define L(TY, GY) hipLaunchKernelGGL((mse_kernel<TY, GY>), grid, block, 0, st, outer, inner, (const TY*)s, s_stride, (const TY*)t, t_stride, w, rows_per_w, norm, coef, coef_dev, loss, (GY*)ds, g_stride, accumulate, g_seed_scale)
  if (dtype == DT_BF16) { if (g_f32) L(bf16, float); else L(bf16, bf16); }
  else if (dtype == DT_F16) { if (g_f32) L(f16, float); else L(f16, f16); }
  else { L(float, float); }
#undef L
  return launch_status();
}

// ---------------------------------------------------------------------------------------------
// Step prologue (round 3): the per-step random scalars of a training step from ONE launch instead of five torch launches inside the
// captured graph (normal_, div, softmax, mul for MKRW; random_ for the dropout seed).
//   MKRW ability weights  rw = softmax(randn(5) / rw_temp) * 5          (map_nav_src/r2r/agent.py:866-871; pretrain: r2r_magic_pretrain.json:72)
//   dropout seed          two 31-bit words (the kernels' counter-based masks key on them, csrc/common.hpp)
// Randomness: murmur3-finalised counters keyed by (base_seed, step counter): the counter lives in device memory and is advanced here, so
// a replayed HIP graph draws fresh values every step; Box-Muller on two uniforms per normal.
__device__ __forceinline__ unsigned mix32(unsigned x) { x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16; return x; }
// scale_state (NULL: none) = {S, 1 / S, clean optimizer steps since the last change, pending}: the dynamic loss scale of fp16 storage with
// amp.GradScaler's rule -- `pending` is what the AdamW launch of the step that just ended left (1: it updated the weights, 2: it SKIPPED
// the update because the gradient norm was not finite, 0: no optimizer step since the last prologue, e.g. a gradient-accumulation
// micro-step): 2 -> S *= backoff, 1 -> after `interval` clean steps in a row S *= growth.  Runs before any kernel of the new step reads S.
__global__ void step_rng_kernel(unsigned base_lo, unsigned base_hi, unsigned* counter, float rw_temp, int* seed_out, float* rw_out, float* zero_me,
                                float* scale_state, float growth, float backoff, int interval) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (zero_me) zero_me[0] = 0.f;               // the gradient-norm accumulator of the step that starts here
  if (scale_state) {
    const float pend = scale_state[3];
    float S = scale_state[0], tr = scale_state[2];
    if (pend == 2.f) { S = fmaxf(S * backoff, 1.f); tr = 0.f; }
    else if (pend == 1.f) { tr += 1.f; if (tr >= (float)interval) { S = fminf(S * growth, 16777216.f); tr = 0.f; } }
    scale_state[0] = S; scale_state[1] = 1.f / S; scale_state[2] = tr; scale_state[3] = 0.f;
  }
  const unsigned c = counter[0];
  counter[0] = c + 1u;
  auto draw = [&](unsigned k) { return mix32(mix32(base_lo ^ (c * 0x9E3779B1u)) + base_hi + k * 0x85EBCA77u); };
  if (seed_out) { seed_out[0] = (int)(draw(0) & 0x7FFFFFFFu); seed_out[1] = (int)(draw(1) & 0x7FFFFFFFu); }
  if (rw_out) {
    float z[5], mx = -3.0e38f;
    for (int i = 0; i < 5; ++i) {
      const float u1 = ((float)(draw(2 + 2 * i) >> 8) + 1.0f) * (1.0f / 16777217.0f);        // (0, 1)
      const float u2 = (float)(draw(3 + 2 * i) >> 8) * (1.0f / 16777216.0f);                  // [0, 1)
      z[i] = sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958648f * u2) / rw_temp;
      mx = fmaxf(mx, z[i]);
    }
    float s = 0.f;
    for (int i = 0; i < 5; ++i) { z[i] = expf(z[i] - mx); s += z[i]; }
    for (int i = 0; i < 5; ++i) rw_out[i] = 5.0f * z[i] / s;
  }
}
extern "C" int magic_step_rng(unsigned long long base_seed, unsigned* counter, float rw_temp, int* seed_out, float* rw_out, float* zero_me,
                              float* scale_state, float growth, float backoff, int interval, void* stream) {
  if (!counter || rw_temp <= 0.f || (!seed_out && !rw_out)) return MAGIC_ERR_ARG;
  if (scale_state && (!(growth >= 1.f) || !(backoff > 0.f && backoff <= 1.f) || interval <= 0)) return MAGIC_ERR_ARG;
  hipLaunchKernelGGL(step_rng_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned)base_seed, (unsigned)(base_seed >> 32), counter, rw_temp, seed_out, rw_out, zero_me,
                     scale_state, growth, backoff, interval);
  return launch_status();
}

// Loss assembly (round 3): what `_losses` did with ~10 torch launches (sum, div, index, cat, mul, sum, mul, mul, add ...) in one:
//   sup = row_scale * sum_i rows[i] (* row_w[i]) ; slots[9] = sum kd_rows (the action-distillation rows) ;
//   terms[i] = slots[i] * rw[ability(i)] , abilities of the ten MAKD slots = {txt, txt, img, img, img, global, global, local, local, action}
//   (map_nav_src/r2r/agent.py:546-719: softmax_weights[0..4]) ; kdl = sum terms ; loss = alpha * kdl + (1 - alpha) * sup  (agent.py:1110-1123)
//   out[0] = sup, out[1..10] = terms, out[11] = kdl, out[12] = loss.  One 256-thread block.
__global__ __launch_bounds__(256) void loss_assemble_kernel(const float* rows, int n_rows, const float* row_w, float row_scale, const float* kd_rows, int n_kd,
                                                            float* slots, const float* rw, float alpha, int has_kd, float* out) {
  __shared__ float red[2][4];
  float s = 0.f, k = 0.f;
  for (int i = threadIdx.x; i < n_rows; i += 256) s += row_w ? rows[i] * row_w[i] : rows[i];
  if (kd_rows) for (int i = threadIdx.x; i < n_kd; i += 256) k += kd_rows[i];
  s = wave_sum(s); k = wave_sum(k);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = k; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float sup = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) * row_scale;
    if (kd_rows) slots[9] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const int ab[10] = {0, 0, 1, 1, 1, 2, 2, 3, 3, 4};
    float kdl = 0.f;
    for (int i = 0; i < 10; ++i) {
      const float t = has_kd ? slots[i] * (rw ? rw[ab[i]] : 1.f) : 0.f;
      out[1 + i] = t; kdl += t;
    }
    out[0] = sup; out[11] = kdl; out[12] = has_kd ? alpha * kdl + (1.f - alpha) * sup : sup;
  }
}
extern "C" int magic_loss_assemble(const float* rows, int n_rows, const float* row_w, float row_scale, const float* kd_rows, int n_kd,
                                   float* slots, const float* rw, float alpha, int has_kd, float* out, void* stream) {
  if (!rows || n_rows <= 0 || !out || (has_kd && !slots) || (kd_rows && n_kd <= 0)) return MAGIC_ERR_ARG;
  hipLaunchKernelGGL(loss_assemble_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, rows, n_rows, row_w, row_scale, kd_rows, n_kd, slots, rw, alpha, has_kd, out);
  return launch_status();
}
