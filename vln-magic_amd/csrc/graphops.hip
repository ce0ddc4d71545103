// Graph-map / panorama glue kernels: CSR row gather (map-node aggregation, candidate-view selection,
// masked-token selection; the backward is the same kernel on the transposed CSR), attention pooling of
// the 36 views (adaptive_pano_fusion), and the global/local action-logit gate + local->global fusion.
#include "common.hpp"
#include <cstring>

// out[n,:] (+)= sum_{e in [ptr[n], ptr[n+1])} w[e] * src[idx[e], :]      one wave per output row
template <typename T>
__global__ __launch_bounds__(256) void csr_gather_kernel(int n_out, int H, const T* src, const int* ptr, const int* idx, const float* w,
                                                         T* out, int accumulate) {
  const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= n_out) return;
  const int e0 = ptr[n], e1 = ptr[n + 1];
  if (accumulate && e0 == e1) return;
  for (int c = lane * 2; c < H; c += 128) {
    float a = 0.f, b = 0.f;
    for (int e = e0; e < e1; ++e) {
      const float wv = w ? w[e] : 1.f;
      const T* p = src + (long long)idx[e] * H + c;
      a += wv * to_f(p[0]); b += wv * to_f(p[1]);
    }
    T* o = out + (long long)n * H + c;
    if (accumulate) { a += to_f(o[0]); b += to_f(o[1]); }
    o[0] = from_f<T>(a); o[1] = from_f<T>(b);
  }
}

// Several independent CSR gathers in one launch (<= 4 problems; blocks [start[i], start[i+1]) serve problem i), each with an optional second
// source that is accumulated AFTER the first with the same rounding as two consecutive launches (csr_gather, then csr_gather(accumulate)).
#define CSRM_MAX 4
struct magic_csr_prob {
  int n_out, accumulate; const void* src1; const int* ptr1; const int* idx1; const float* w1;
  const void* src2; const int* ptr2; const int* idx2; const float* w2; void* out;
};
struct CsrMulti { magic_csr_prob p[CSRM_MAX]; int start[CSRM_MAX + 1]; int n, H; };
template <typename T>
__global__ __launch_bounds__(256) void csr_gather_multi_kernel(CsrMulti mm) {
  int i = 0;
  while (i + 1 < mm.n && (int)blockIdx.x >= mm.start[i + 1]) ++i;
  const magic_csr_prob& p = mm.p[i];
  const int H = mm.H, lane = threadIdx.x & 63, n = (blockIdx.x - mm.start[i]) * 4 + (threadIdx.x >> 6);
  if (n >= p.n_out) return;
  const int a0 = p.ptr1[n], a1 = p.ptr1[n + 1];
  const int b0 = p.ptr2 ? p.ptr2[n] : 0, b1 = p.ptr2 ? p.ptr2[n + 1] : 0;
  if (p.accumulate && a0 == a1 && b0 == b1) return;
  const T* s1 = (const T*)p.src1; const T* s2 = (const T*)p.src2; T* out = (T*)p.out;
  for (int c = lane * 2; c < H; c += 128) {
    T* o = out + (long long)n * H + c;
    float va = 0.f, vb = 0.f;
    if (p.accumulate) { va = to_f(o[0]); vb = to_f(o[1]); }
    if (!p.accumulate || a0 < a1) {
      float a = 0.f, b = 0.f;
      for (int e = a0; e < a1; ++e) {
        const float wv = p.w1 ? p.w1[e] : 1.f;
        const T* q = s1 + (long long)p.idx1[e] * H + c;
        a += wv * to_f(q[0]); b += wv * to_f(q[1]);
      }
      va = to_f(from_f<T>(a + va)); vb = to_f(from_f<T>(b + vb));          // (a + old value: the order csr_gather adds in)
    }
    if (b0 < b1) {
      float a = 0.f, b = 0.f;
      for (int e = b0; e < b1; ++e) {
        const float wv = p.w2 ? p.w2[e] : 1.f;
        const T* q = s2 + (long long)p.idx2[e] * H + c;
        a += wv * to_f(q[0]); b += wv * to_f(q[1]);
      }
      va = to_f(from_f<T>(a + va)); vb = to_f(from_f<T>(b + vb));
    }
    o[0] = from_f<T>(va); o[1] = from_f<T>(vb);
  }
}

// adaptive panorama fusion: p = softmax_v(x[n,v,:].wf + bf + mask), fused[n,:] = sum_v p_v x[n,v,:]
// one block (256 threads) per panorama, V <= 64 views.
template <typename T>
__global__ __launch_bounds__(256) void pano_fuse_fwd_kernel(int N, int V, int H, const T* x, const int* lens, const float* wf, const float* bf,
                                                            T* fused, float* probs, const T* P, int nh, int inner, float* pmean) {
  __shared__ float sc[64];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (P) {        // the panorama's attention map averaged over heads (the `img` attention-distillation operand): one launch less per forward
    const T* pb = P + (long long)n * nh * inner;
    for (int r = tid; r < inner; r += 256) {
      float s = 0.f;
      for (int h = 0; h < nh; ++h) s += to_f(pb[(long long)h * inner + r]);
      pmean[(long long)n * inner + r] = s / nh;
    }
  }
  const T* xb = x + (long long)n * V * H;
  for (int v = wid; v < V; v += 4) {
    float d = 0.f;
    for (int c = lane; c < H; c += 64) d += to_f(xb[(long long)v * H + c]) * wf[c];
    d = wave_sum(d);
    if (lane == 0) sc[v] = d + bf[0] + (v < lens[n] ? 0.f : -10000.0f);
  }
  __syncthreads();
  if (wid == 0) {
    float s = lane < V ? sc[lane] : -3.0e38f;
    const float mx = wave_max(s);
    float e = lane < V ? __expf(s - mx) : 0.f;
    const float z = wave_sum(e);
    if (lane < V) { sc[lane] = e / z; probs[(long long)n * V + lane] = e / z; }
  }
  __syncthreads();
  for (int c = tid; c < H; c += 256) {
    float a = 0.f;
    for (int v = 0; v < V; ++v) a += sc[v] * to_f(xb[(long long)v * H + c]);
    fused[(long long)n * H + c] = from_f<T>(a);
  }
}


// The same fusion with every view row read ONCE (round 4): wave w of the panorama's block owns views w, w + 4, ...; a lane keeps its column pairs of
// those <= 10 rows in registers across the scoring pass, the softmax and the weighted sum (the kernel above reads x twice, the second time as
// 36 dependent 2-byte loads per thread with half the block idle: 17 us for 290 panoramas, where one round trip to memory is ~3).  Same
// arithmetic order per view; the weighted sum adds the views of a wave first and the four waves after (fp32: the results differ from the
// kernel above in the last bits of the fp32 sum, not in what is rounded to T).
#define PFF_VPW 10
template <typename T, int NIT>
__global__ __launch_bounds__(256) void pano_fuse_fwd_reg_kernel(int N, int V, const T* x, const int* lens, const float* wf, const float* bf,
                                                                T* fused, float* probs, const T* P, int nh, int inner, float* pmean) {
  constexpr int H = NIT * 128;
  __shared__ float sc[64];
  __shared__ float part[4][H];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const T* xb = x + (long long)n * V * H;
  float xv[PFF_VPW][2 * NIT], wr[2 * NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) { wr[2 * it] = wf[it * 128 + lane * 2]; wr[2 * it + 1] = wf[it * 128 + lane * 2 + 1]; }
#pragma unroll
  for (int j = 0; j < PFF_VPW; ++j) {
    const int v = wid + 4 * j;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      xv[j][2 * it] = 0.f; xv[j][2 * it + 1] = 0.f;
      if (v < V) {
        if constexpr (sizeof(T) == 2) {
          const unsigned u = *(const unsigned*)(xb + (long long)v * H + it * 128 + lane * 2);
          T lo, hi;
          __builtin_memcpy(&lo, &u, 2); __builtin_memcpy(&hi, (const char*)&u + 2, 2);
          xv[j][2 * it] = to_f(lo); xv[j][2 * it + 1] = to_f(hi);
        } else {
          const float2 u = *(const float2*)(xb + (long long)v * H + it * 128 + lane * 2);
          xv[j][2 * it] = u.x; xv[j][2 * it + 1] = u.y;
        }
      }
    }
  }
  if (P) {        // the panorama's attention map averaged over heads, while the view rows are on their way
    const T* pb = P + (long long)n * nh * inner;
    for (int r = tid; r < inner; r += 256) {
      float s = 0.f;
      for (int h = 0; h < nh; ++h) s += to_f(pb[(long long)h * inner + r]);
      pmean[(long long)n * inner + r] = s / nh;
    }
  }
  const int len = lens[n];
  const float b0 = bf[0];
#pragma unroll
  for (int j = 0; j < PFF_VPW; ++j) {
    const int v = wid + 4 * j;
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < 2 * NIT; ++i) d += xv[j][i] * wr[i];
    d = wave_sum(d);
    if (lane == 0 && v < V) sc[v] = d + b0 + (v < len ? 0.f : -10000.0f);
  }
  __syncthreads();
  if (wid == 0) {
    float s = lane < V ? sc[lane] : -3.0e38f;
    const float mx = wave_max(s);
    float e = lane < V ? __expf(s - mx) : 0.f;
    const float z = wave_sum(e);
    if (lane < V) { sc[lane] = e / z; probs[(long long)n * V + lane] = e / z; }
  }
  __syncthreads();
  float a[2 * NIT];
#pragma unroll
  for (int i = 0; i < 2 * NIT; ++i) a[i] = 0.f;
#pragma unroll
  for (int j = 0; j < PFF_VPW; ++j) {
    const int v = wid + 4 * j;
    const float pv = v < V ? sc[v] : 0.f;
#pragma unroll
    for (int i = 0; i < 2 * NIT; ++i) a[i] += pv * xv[j][i];
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) { part[wid][it * 128 + lane * 2] = a[2 * it]; part[wid][it * 128 + lane * 2 + 1] = a[2 * it + 1]; }
  __syncthreads();
  for (int c = tid; c < H; c += 256) fused[(long long)n * H + c] = from_f<T>((part[0][c] + part[1][c]) + (part[2][c] + part[3][c]));
}

// backward: dx[n,v,:] += p_v * df + ds_v * wf ;  ds_v = p_v (dp_v - sum p dp), dp_v = df . x_v ; dwf, dbf atomics
// PF_PB panoramas per block, four waves each: wave w of a panorama owns views w, w+4, ... in both passes (a lane owns the column pairs
// {128 it + 2 lane}); the wf / bf gradients are reduced over the block's waves before ONE atomic per element and block (same-address
// atomics serialise in L2: one block per panorama spent 7 of its 25 us draining 290 of them per element).
#define PF_PB 2
template <typename T>
__global__ __launch_bounds__(256 * PF_PB) void pano_fuse_bwd_kernel(int N, int V, int H, const T* __restrict__ x, const float* __restrict__ probs,
                                                                    const float* __restrict__ wf, const T* __restrict__ dfused, T* __restrict__ dx,
                                                                    float* dwf, float* dbf) {
  __shared__ float dp[PF_PB][64], dsv[PF_PB][64], red[PF_PB * 4][768], rb[PF_PB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, pb = wave >> 2, wid = wave & 3;
  const int n = blockIdx.x * PF_PB + pb;
  const bool live = n < N;
  const int nn = live ? n : N - 1;
  const T* xb = x + (long long)nn * V * H;
  const T* df = dfused + (long long)nn * H;
  const float* p = probs + (long long)nn * V;
  const int nit = H / 128;
  float dfr[12], wfr[12];
  for (int it = 0; it < nit; ++it) {
    const int c = it * 128 + lane * 2;
    dfr[2 * it] = to_f(df[c]); dfr[2 * it + 1] = to_f(df[c + 1]);
    wfr[2 * it] = wf[c]; wfr[2 * it + 1] = wf[c + 1];
  }
  for (int v = wid; v < V; v += 4) {
    float d = 0.f;
    for (int it = 0; it < nit; ++it) {
      const int c = it * 128 + lane * 2;
      d += to_f(xb[(long long)v * H + c]) * dfr[2 * it] + to_f(xb[(long long)v * H + c + 1]) * dfr[2 * it + 1];
    }
    d = wave_sum(d);
    if (lane == 0) dp[pb][v] = d;
  }
  __syncthreads();
  if (wid == 0) {
    float t = lane < V ? p[lane] * dp[pb][lane] : 0.f;
    const float s = wave_sum(t);
    const float d = (lane < V && live) ? p[lane] * (dp[pb][lane] - s) : 0.f;
    if (lane < V) dsv[pb][lane] = d;
    const float tot = wave_sum(d);
    if (lane == 0) rb[pb] = tot;
  }
  __syncthreads();
  float aw[12];
  for (int i = 0; i < 2 * nit; ++i) aw[i] = 0.f;
  if (live) {
    for (int v = wid; v < V; v += 4) {
      const float pv = p[v], dv = dsv[pb][v];
      for (int it = 0; it < nit; ++it) {
        const int c = it * 128 + lane * 2;
        const long long i = ((long long)n * V + v) * H + c;
        const float x0 = to_f(xb[(long long)v * H + c]), x1 = to_f(xb[(long long)v * H + c + 1]);
        aw[2 * it] += dv * x0; aw[2 * it + 1] += dv * x1;
        dx[i] = from_f<T>(to_f(dx[i]) + pv * dfr[2 * it] + dv * wfr[2 * it]);
        dx[i + 1] = from_f<T>(to_f(dx[i + 1]) + pv * dfr[2 * it + 1] + dv * wfr[2 * it + 1]);
      }
    }
  }
  for (int it = 0; it < nit; ++it) {
    const int c = it * 128 + lane * 2;
    red[wave][c] = aw[2 * it]; red[wave][c + 1] = aw[2 * it + 1];
  }
  __syncthreads();
  // dbf NULL (round 6): dwf is a PARTIAL buffer [blocks][H + 1] -- this block's sums (weight row, then the bias) to its own row, added up in order by the caller
  float* const prow = dbf ? nullptr : dwf + (long long)blockIdx.x * (H + 1);
  for (int c = tid; c < H; c += 256 * PF_PB) {
    float v = 0.f;
    for (int w = 0; w < PF_PB * 4; ++w) v += red[w][c];
    if (prow) prow[c] = v; else atomicAdd(dwf + c, v);
  }
  if (tid == 0) {
    float v = 0.f;
    for (int q = 0; q < PF_PB; ++q) v += rb[q];
    if (prow) prow[H] = v; else atomicAdd(dbf, v);
  }
}

// SAP logits: gate fw = sigmoid(fuse_raw[b]); gl = g_raw*fw (masked -inf), ll = l_raw*(1-fw) (masked),
// fused[k] = gl[k] + (src[k] >= 0 ? ll[src[k]] : src[k] == -2 ? sum_{j in bw} ll[j] : 0)     (fp32, one block per sample)
__global__ __launch_bounds__(64) void sap_fuse_fwd_kernel(int B, int K, int Vp, const float* g_raw, const float* l_raw, const float* fuse_raw,
                                                          const unsigned char* gmask, const unsigned char* lmask, const int* fsrc,
                                                          const unsigned char* bwmask, float use_gate, float* gl, float* ll, float* fl) {
  __shared__ float sl[128];
  __shared__ float sbw;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float NEG = -__builtin_inff();
  const float fw = use_gate != 0.f ? 1.f / (1.f + __expf(-fuse_raw[b])) : 0.5f;
  float bw = 0.f;
  for (int j = tid; j < Vp; j += 64) {
    const float v = lmask[b * Vp + j] ? l_raw[b * Vp + j] * (1.f - fw) : NEG;
    ll[b * Vp + j] = v; sl[j] = v;
    if (bwmask[b * Vp + j] && lmask[b * Vp + j]) bw += v;
  }
  bw = wave_sum(bw);
  if (tid == 0) sbw = bw;
  __syncthreads();
  for (int k = tid; k < K; k += 64) {
    const float v = gmask[b * K + k] ? g_raw[b * K + k] * fw : NEG;
    gl[b * K + k] = v;
    const int s = fsrc[b * K + k];
    float add = 0.f;
    if (s >= 0) add = sl[s]; else if (s == -2) add = sbw;
    fl[b * K + k] = v + add;
  }
}

// The SAP step's logit fusion AND its row losses in one launch (round 3: every dependent launch of the replayed step costs 6-9 us whatever its
// work): sap_fuse_fwd + the three cross-entropy rows (global / local / fused logits; ce_rows arithmetic: train_r2r_magic.py:513-520) + the
// teacher-sample weights exp(-rate CE(teacher fused logits)) + the action-distillation rows (kd_rows arithmetic: optim/kd_loss.py:18-41) for one
// sample per 64-thread workgroup.  Replaces six launches.
struct SapLossParams {
  int B, K, Vp, use_gate;
  const float *g_raw, *l_raw, *fuse_raw; const unsigned char *gmask, *lmask; const int* fsrc; const unsigned char* bwmask;
  float *gl, *ll, *fl;
  const int *glab, *llab; int ignore_index; float coef;
  float *rows, *dgl, *dll, *dfl;                           // rows [3, B]; gradients may be NULL (no backward)
  const float* t_fused; float w_rate; int pad_; float* w_out;     // teacher fused logits [B, K] (or NULL); w_out [B] (or NULL: no sample weights)
  float T, kd_norm, kd_coef, pad2_; const float* kd_coef_dev; float* kd_rows;     // kd_rows NULL: no distillation term
};
__device__ __forceinline__ float sap_ce_wave(const float* x, int N, int lab, int ignore_index, float coef, float* d, int lane) {
  const bool ignored = (lab == ignore_index) || lab < 0 || lab >= N;
  const float xl = ignored ? 0.f : x[lab];
  float mx = -3.0e38f;
  for (int c = lane; c < N; c += 64) mx = fmaxf(mx, x[c]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int c = lane; c < N; c += 64) s += __expf(x[c] - mx);           // exp(-inf) = 0
  s = wave_sum(s);
  const float lse = mx + __logf(s);
  if (d) {
    const float cf = ignored ? 0.f : coef;
    for (int c = lane; c < N; c += 64) d[c] = cf * (__expf(x[c] - lse) - (c == lab ? 1.f : 0.f));
  }
  return ignored ? 0.f : lse - xl;
}
__global__ __launch_bounds__(64) void sap_fuse_loss_kernel(SapLossParams p, const float* ss) {
  __shared__ float sl[128];
  if (ss) { p.coef *= ss[0]; p.kd_coef *= ss[0]; }       // dynamic loss scale (loss.hip magic_seed_scale): gradient seeds only
  __shared__ float sbw;
  const int b = blockIdx.x, lane = threadIdx.x, B = p.B, K = p.K, Vp = p.Vp;
  const float NEG = -__builtin_inff();
  const float fw = p.use_gate ? 1.f / (1.f + __expf(-p.fuse_raw[b])) : 0.5f;
  float* gl = p.gl + (long long)b * K; float* ll = p.ll + (long long)b * Vp; float* fl = p.fl + (long long)b * K;
  float bw = 0.f;
  for (int j = lane; j < Vp; j += 64) {
    const float v = p.lmask[b * Vp + j] ? p.l_raw[b * Vp + j] * (1.f - fw) : NEG;
    ll[j] = v; sl[j] = v;
    if (p.bwmask[b * Vp + j] && p.lmask[b * Vp + j]) bw += v;
  }
  bw = wave_sum(bw);
  if (lane == 0) sbw = bw;
  __syncthreads();
  for (int k = lane; k < K; k += 64) {
    const float v = p.gmask[b * K + k] ? p.g_raw[b * K + k] * fw : NEG;
    gl[k] = v;
    const int s = p.fsrc[b * K + k];
    float add = 0.f;
    if (s >= 0) add = sl[s]; else if (s == -2) add = sbw;
    fl[k] = v + add;
  }
  __syncthreads();                                       // the wave's own global writes, read back below by other lanes
  const int gla = p.glab[b], lla = p.llab[b];
  const float l0 = sap_ce_wave(gl, K, gla, p.ignore_index, p.coef, p.dgl ? p.dgl + (long long)b * K : nullptr, lane);
  const float l1 = sap_ce_wave(ll, Vp, lla, p.ignore_index, p.coef, p.dll ? p.dll + (long long)b * Vp : nullptr, lane);
  float* dfl = p.dfl ? p.dfl + (long long)b * K : nullptr;
  const float l2 = sap_ce_wave(fl, K, gla, p.ignore_index, p.coef, dfl, lane);
  if (lane == 0) { p.rows[b] = l0; p.rows[B + b] = l1; p.rows[2 * B + b] = l2; }
  if (!p.t_fused) return;
  const float* t = p.t_fused + (long long)b * K;
  float wr = 1.f;
  if (p.w_out) {                                         // teacher_sample_hard_mining: exp(-rate * CE(teacher logits, label))
    wr = __expf(-p.w_rate * sap_ce_wave(t, K, gla, p.ignore_index, 0.f, nullptr, lane));
    if (lane == 0) p.w_out[b] = wr;
  }
  if (!p.kd_rows) return;
  __syncthreads();                                       // dfl rows written above are read-modify-written below by the same lanes
  float coef = p.kd_coef;
  if (p.kd_coef_dev) coef *= p.kd_coef_dev[0];
  const float invT = 1.0f / p.T;
  float sv[8], tv[8], ms = -3.0e38f, mt = -3.0e38f;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int c = it * 64 + lane;
    float a = -3.0e38f, q = -3.0e38f;
    if (c < K) {
      a = fl[c]; q = t[c];
      if (a == NEG) a = -1e6f;
      if (q == NEG) q = -1e6f;
      a *= invT; q *= invT;
      ms = fmaxf(ms, a); mt = fmaxf(mt, q);
    }
    sv[it] = a; tv[it] = q;
  }
  ms = wave_max(ms); mt = wave_max(mt);
  float zs = 0.f, zt = 0.f;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int c = it * 64 + lane;
    if (c < K) { zs += __expf(sv[it] - ms); zt += __expf(tv[it] - mt); }
  }
  zs = wave_sum(zs); zt = wave_sum(zt);
  const float ls = ms + __logf(zs), lt = mt + __logf(zt);
  float kl = 0.f;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int c = it * 64 + lane;
    if (c < K) {
      const float lpt = tv[it] - lt, lps = sv[it] - ls;
      const float pt = __expf(lpt), ps = __expf(lps);
      if (pt > 0.f) kl += pt * (lpt - lps);
      if (dfl) dfl[c] += coef * wr * p.T * (ps - pt) * p.kd_norm;
    }
  }
  kl = wave_sum(kl);
  if (lane == 0) p.kd_rows[b] = wr * kl * p.T * p.T * p.kd_norm;
}

// backward: given dgl, dll, dfl (any may be null) -> d g_raw, d l_raw, d fuse_raw
__global__ __launch_bounds__(64) void sap_fuse_bwd_kernel(int B, int K, int Vp, const float* g_raw, const float* l_raw, const float* fuse_raw,
                                                          const unsigned char* gmask, const unsigned char* lmask, const int* fsrc,
                                                          const unsigned char* bwmask, float use_gate, const float* dgl, const float* dll,
                                                          const float* dfl, float* dg_raw, float* dl_raw, float* dfuse_raw) {
  __shared__ float dl[128];
  __shared__ float sbw;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float fw = use_gate != 0.f ? 1.f / (1.f + __expf(-fuse_raw[b])) : 0.5f;
  for (int j = tid; j < Vp; j += 64) dl[j] = (dll && lmask[b * Vp + j]) ? dll[b * Vp + j] : 0.f;
  if (tid == 0) sbw = 0.f;
  __syncthreads();
  float dfw = 0.f;
  for (int k = tid; k < K; k += 64) {
    const bool ok = gmask[b * K + k];
    const float dg = ok ? ((dgl ? dgl[b * K + k] : 0.f) + (dfl ? dfl[b * K + k] : 0.f)) : 0.f;
    dg_raw[b * K + k] = dg * fw;
    dfw += dg * g_raw[b * K + k];
    if (dfl && ok) {   // masked global entries are -inf: their CE/KD gradient is exactly 0
      const int s = fsrc[b * K + k];
      const float d = dfl[b * K + k];
      if (s >= 0) atomicAdd(&dl[s], d); else if (s == -2) atomicAdd(&sbw, d);
    }
  }
  __syncthreads();
  for (int j = tid; j < Vp; j += 64) {
    float d = dl[j];
    if (bwmask[b * Vp + j]) d += sbw;
    if (!lmask[b * Vp + j]) d = 0.f;
    dl_raw[b * Vp + j] = d * (1.f - fw);
    dfw -= d * l_raw[b * Vp + j];
  }
  dfw = wave_sum(dfw);
  if (tid == 0) dfuse_raw[b] = use_gate != 0.f ? dfw * fw * (1.f - fw) : 0.f;
}

extern "C" int magic_csr_gather(int dtype, int n_out, int H, const void* src, const int* ptr, const int* idx, const float* w,
                                void* out, int accumulate, void* stream) {
  if (n_out <= 0 || H <= 0 || (H & 1)) return MAGIC_ERR_ARG;
  dim3 grid((n_out + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_BF16) hipLaunchKernelGGL(csr_gather_kernel<bf16>, grid, block, 0, st, n_out, H, (const bf16*)src, ptr, idx, w, (bf16*)out, accumulate);
  else if (dtype == DT_F16) hipLaunchKernelGGL(csr_gather_kernel<f16>, grid, block, 0, st, n_out, H, (const f16*)src, ptr, idx, w, (f16*)out, accumulate);
  else hipLaunchKernelGGL(csr_gather_kernel<float>, grid, block, 0, st, n_out, H, (const float*)src, ptr, idx, w, (float*)out, accumulate);
  return launch_status();
}

extern "C" int magic_csr_gather_multi(int dtype, int H, int n, const magic_csr_prob* d, void* stream) {
  if (n < 1 || n > CSRM_MAX || !d || H <= 0 || (H & 1)) return MAGIC_ERR_ARG;
  CsrMulti mm;
  mm.n = n; mm.H = H;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    if (d[i].n_out <= 0 || !d[i].src1 || !d[i].ptr1 || !d[i].idx1 || !d[i].out) return MAGIC_ERR_ARG;
    if (d[i].ptr2 && (!d[i].src2 || !d[i].idx2)) return MAGIC_ERR_ARG;
    mm.p[i] = d[i];
    mm.start[i] = total;
    total += (d[i].n_out + 3) / 4;
  }
  for (int i = n; i <= CSRM_MAX; ++i) mm.start[i] = total;
  dim3 grid(total), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_BF16) hipLaunchKernelGGL(csr_gather_multi_kernel<bf16>, grid, block, 0, st, mm);
  else if (dtype == DT_F16) hipLaunchKernelGGL(csr_gather_multi_kernel<f16>, grid, block, 0, st, mm);
  else hipLaunchKernelGGL(csr_gather_multi_kernel<float>, grid, block, 0, st, mm);
  return launch_status();
}

extern "C" int magic_pano_fuse_fwd(int dtype, int N, int V, int H, const void* x, const int* lens, const float* wf, const float* bf,
                                   void* fused, float* probs, const void* P, int nh, int inner, float* pmean, void* stream) {
  if (N <= 0 || V <= 0 || V > 64 || H <= 0 || (P && (nh <= 0 || inner <= 0 || !pmean))) return MAGIC_ERR_ARG;
  dim3 grid(N), block(256);
  hipStream_t st = (hipStream_t)stream;
  static int reg_form = -1;
  if (reg_form < 0) { const char* e = getenv("MAGIC_PANO_FUSE_REG"); reg_form = e ? atoi(e) : 1; }
  if (reg_form && V <= 4 * PFF_VPW && (H == 128 || H == 256 || H == 384 || H == 768)) {      // every view row read once, kept in registers
#define PFR(TY, NIT) hipLaunchKernelGGL((pano_fuse_fwd_reg_kernel<TY, NIT>), grid, block, 0, st, N, V, (const TY*)x, lens, wf, bf, (TY*)fused, probs, (const TY*)P, nh, inner, pmean)
#define PFR_T(TY) do { if (H == 128) PFR(TY, 1); else if (H == 256) PFR(TY, 2); else if (H == 384) PFR(TY, 3); else PFR(TY, 6); } while (0)
    if (dtype == DT_BF16) PFR_T(bf16); else if (dtype == DT_F16) PFR_T(f16); else PFR_T(float);
#undef PFR_T
#undef PFR
    return launch_status();
  }
  if (dtype == DT_BF16) hipLaunchKernelGGL(pano_fuse_fwd_kernel<bf16>, grid, block, 0, st, N, V, H, (const bf16*)x, lens, wf, bf, (bf16*)fused, probs, (const bf16*)P, nh, inner, pmean);
  else if (dtype == DT_F16) hipLaunchKernelGGL(pano_fuse_fwd_kernel<f16>, grid, block, 0, st, N, V, H, (const f16*)x, lens, wf, bf, (f16*)fused, probs, (const f16*)P, nh, inner, pmean);
  else hipLaunchKernelGGL(pano_fuse_fwd_kernel<float>, grid, block, 0, st, N, V, H, (const float*)x, lens, wf, bf, (float*)fused, probs, (const float*)P, nh, inner, pmean);
  return launch_status();
}

// workgroups (= rows of H + 1 floats of the partial buffer) magic_pano_fuse_bwd launches for N panoramas
extern "C" int magic_pano_fuse_bwd_blocks(int N) { return N > 0 ? (N + PF_PB - 1) / PF_PB : MAGIC_ERR_ARG; }
extern "C" int magic_pano_fuse_bwd(int dtype, int N, int V, int H, const void* x, const float* probs, const float* wf, const void* dfused,
                                   void* dx, float* dwf, float* dbf, void* stream) {
  if (N <= 0 || V <= 0 || V > 64 || H <= 0 || H % 128 || H > 768) return MAGIC_ERR_ARG;
  dim3 grid((N + PF_PB - 1) / PF_PB), block(256 * PF_PB);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_BF16) hipLaunchKernelGGL(pano_fuse_bwd_kernel<bf16>, grid, block, 0, st, N, V, H, (const bf16*)x, probs, wf, (const bf16*)dfused, (bf16*)dx, dwf, dbf);
  else if (dtype == DT_F16) hipLaunchKernelGGL(pano_fuse_bwd_kernel<f16>, grid, block, 0, st, N, V, H, (const f16*)x, probs, wf, (const f16*)dfused, (f16*)dx, dwf, dbf);
  else hipLaunchKernelGGL(pano_fuse_bwd_kernel<float>, grid, block, 0, st, N, V, H, (const float*)x, probs, wf, (const float*)dfused, (float*)dx, dwf, dbf);
  return launch_status();
}

extern "C" int magic_sap_fuse_fwd(int B, int K, int Vp, const float* g_raw, const float* l_raw, const float* fuse_raw,
                                  const unsigned char* gmask, const unsigned char* lmask, const int* fsrc, const unsigned char* bwmask,
                                  int use_gate, float* gl, float* ll, float* fl, void* stream) {
  if (B <= 0 || K <= 0 || Vp <= 0 || Vp > 128) return MAGIC_ERR_ARG;
  hipLaunchKernelGGL(sap_fuse_fwd_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, B, K, Vp, g_raw, l_raw, fuse_raw, gmask, lmask, fsrc, bwmask,
                     (float)use_gate, gl, ll, fl);
  return launch_status();
}

extern "C" int magic_sap_fuse_loss(const void* params, int nbytes, void* stream) {
  if (!params || nbytes != (int)sizeof(SapLossParams)) return MAGIC_ERR_ARG;
  SapLossParams p;
  memcpy(&p, params, sizeof(p));
  if (p.B <= 0 || p.K <= 0 || p.K > 512 || p.Vp <= 0 || p.Vp > 128 || !p.g_raw || !p.l_raw || !p.fuse_raw || !p.gmask || !p.lmask || !p.fsrc || !p.bwmask ||
      !p.gl || !p.ll || !p.fl || !p.glab || !p.llab || !p.rows) return MAGIC_ERR_ARG;
  if ((p.w_out || p.kd_rows) && !p.t_fused) return MAGIC_ERR_ARG;
  if (p.kd_rows && !(p.T > 0.f)) return MAGIC_ERR_ARG;
  hipLaunchKernelGGL(sap_fuse_loss_kernel, dim3(p.B), dim3(64), 0, (hipStream_t)stream, p, seed_scale_get());
  return launch_status();
}

extern "C" int magic_sap_fuse_bwd(int B, int K, int Vp, const float* g_raw, const float* l_raw, const float* fuse_raw,
                                  const unsigned char* gmask, const unsigned char* lmask, const int* fsrc, const unsigned char* bwmask,
                                  int use_gate, const float* dgl, const float* dll, const float* dfl,
                                  float* dg_raw, float* dl_raw, float* dfuse_raw, void* stream) {
  if (B <= 0 || K <= 0 || Vp <= 0 || Vp > 128) return MAGIC_ERR_ARG;
  hipLaunchKernelGGL(sap_fuse_bwd_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, B, K, Vp, g_raw, l_raw, fuse_raw, gmask, lmask, fsrc, bwmask,
                     (float)use_gate, dgl, dll, dfl, dg_raw, dl_raw, dfuse_raw);
  return launch_status();
}

// ---------------------------------------------------------------------------------------------------------------
// Feature ingest: the panorama features live ONCE in HBM as a packed table [n_viewpoints, 36, D]
// (dataset.py:246-254 keeps the same rows in a host dict keyed "{scan}_{vp}"); a batch is described by indices only.
// out[p, j, :] = table[vp_row[p], order[p, j], :]   (order < 0 -> zeros: padded view slots, pad_tensors common.py:9)
// `order` carries the reference's token order: candidate views first, then the remaining views (dataset.py:742-756).
// Pure HBM stream over 16-byte vectors; algorithmic bytes = 2 * D * sizeof(T) per view.
template <typename T>
__global__ __launch_bounds__(256) void view_gather_kernel(long long rows, int V, int D, const T* table, int n_vp, const int* vp_row, const int* order, T* out) {
  constexpr int VE = 16 / sizeof(T), UN = 4;
  typedef __attribute__((ext_vector_type(4))) unsigned int u4;
  const int nvec = D / VE;                                    // 16-byte vectors per view row (96 for 768 bf16)
  const long long total = rows * nvec, stride = (long long)gridDim.x * 256;
  u4* dst = (u4*)out;
  // flat index space over 16-byte vectors: every lane busy whatever D is; UN independent loads in flight per lane
  for (long long v0 = (long long)blockIdx.x * 256 + threadIdx.x; v0 < total; v0 += stride * UN) {
    u4 val[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long long v = v0 + u * stride;
      val[u] = (u4){0u, 0u, 0u, 0u};
      if (v < total) {
        const long long r = v / nvec;
        const int c = (int)(v - r * nvec), sv = order[r];
        const int vr = vp_row[r / V];
        if (sv >= 0 && sv < 36 && (unsigned)vr < (unsigned)n_vp)        // indices come from the data side: never read outside the table
          val[u] = __builtin_nontemporal_load((const u4*)(table + ((long long)vr * 36 + sv) * D) + c);   // read once
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long long v = v0 + u * stride;
      if (v < total) dst[v] = val[u];
    }
  }
}

extern "C" int magic_view_gather(int dtype, int Np, int V, int D, const void* table, int n_viewpoints, const int* vp_row,
                                 const int* order, void* out, void* stream) {
  if (Np <= 0 || V <= 0 || D <= 0 || n_viewpoints <= 0 || !table || !vp_row || !order || !out) return MAGIC_ERR_ARG;
  if (!dtype_ok(dtype)) return MAGIC_ERR_ARG;
  const int ve = dtype_is16(dtype) ? 8 : 4;
  if (D % ve || ((uintptr_t)table & 15) || ((uintptr_t)out & 15)) return MAGIC_ERR_ARG;
  const long long rows = (long long)Np * V;
  const long long chunks = (rows * (D / ve) + 256 * 4 - 1) / (256 * 4);
  dim3 grid((unsigned)(chunks < 1 ? 1 : chunks)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_BF16) hipLaunchKernelGGL(view_gather_kernel<bf16>, grid, block, 0, st, rows, V, D, (const bf16*)table, n_viewpoints, vp_row, order, (bf16*)out);
  else if (dtype == DT_F16) hipLaunchKernelGGL(view_gather_kernel<f16>, grid, block, 0, st, rows, V, D, (const f16*)table, n_viewpoints, vp_row, order, (f16*)out);
  else hipLaunchKernelGGL(view_gather_kernel<float>, grid, block, 0, st, rows, V, D, (const float*)table, n_viewpoints, vp_row, order, (float*)out);
  return launch_status();
}
