// Row-wise kernels (HBM-bound): fused [dense + gathered-embedding] sum -> LayerNorm (fwd/bwd), tiny-K
// linear + LayerNorm (position features), masked softmax with graph-distance bias (fwd/bwd),
// LayerNorm + dot head (ClsPrediction tail).  One 64-lane wave owns one row; row statistics are
// wave-shuffle reductions; parameter gradients are reduced per block (registers -> LDS) before a
// single fp32 atomic per element per block.
#include "common.hpp"
#include "group.hpp"
#include <cstdlib>

#define MAXIT 6   // H = 128*NIT, NIT in {1,2,3,6}: each lane owns elements {it*128 + lane*2, +1}

template <typename T> __device__ __forceinline__ void ld2(const T* p, float& a, float& b);
template <> __device__ __forceinline__ void ld2<float>(const float* p, float& a, float& b) {
  float2 v = *(const float2*)p; a = v.x; b = v.y;
}
template <> __device__ __forceinline__ void ld2<bf16>(const bf16* p, float& a, float& b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  bf16x2 v = *(const bf16x2*)p; a = (float)v[0]; b = (float)v[1];
}
template <> __device__ __forceinline__ void ld2<f16>(const f16* p, float& a, float& b) {
  typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
  f16x2 v = *(const f16x2*)p; a = (float)v[0]; b = (float)v[1];
}
template <typename T> __device__ __forceinline__ void st2(T* p, float a, float b);
template <> __device__ __forceinline__ void st2<float>(float* p, float a, float b) { *(float2*)p = make_float2(a, b); }
template <> __device__ __forceinline__ void st2<bf16>(bf16* p, float a, float b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  bf16x2 v; v[0] = (bf16)a; v[1] = (bf16)b; *(bf16x2*)p = v;
}

template <> __device__ __forceinline__ void st2<f16>(f16* p, float a, float b) {
  typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
  f16x2 v; v[0] = (f16)a; v[1] = (f16)b; *(f16x2*)p = v;
}

struct TabRef {          // embedding-table term: row = idx ? idx[r] : (mod ? r % mod + off : off)
  const void* tab; const int* idx; int mod; int off;
};
__device__ __forceinline__ int tab_row(const TabRef& t, int r) {
  return t.idx ? t.idx[r] : (t.mod ? (r % t.mod) + t.off : t.off);
}

// ---------------------------------------------------------------------------------------------
// out = LN(in0 + in1 + tab0[..] + tab1[..] + tab2[..]) * gamma + beta      (do_ln)   | plain sum
// NIT = H / 128 is a template parameter so the per-lane row fragment lives in exactly 2*NIT registers.
struct LnfParams {
  int M; const void* in0; const void* in1; TabRef t0, t1, t2; const float* gamma; const float* beta; float eps; void* out; float* rstd_out; int do_ln;
  DropDesc din, dout; void* out_drop;
};
template <typename T, int NIT>
__device__ __forceinline__ void ln_fwd_body(const LnfParams& pp, const int bid) {
  const int M = pp.M, do_ln = pp.do_ln;
  const T* in0 = (const T*)pp.in0; const T* in1 = (const T*)pp.in1;
  const TabRef t0 = pp.t0, t1 = pp.t1, t2 = pp.t2;
  const float* gamma = pp.gamma; const float* beta = pp.beta; const float eps = pp.eps;
  T* out = (T*)pp.out; float* rstd_out = pp.rstd_out; T* out_drop = (T*)pp.out_drop;
  const DropDesc din = pp.din, dout = pp.dout;
  constexpr int H = NIT * 128;
  const int lane = threadIdx.x & 63, row = bid * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const DropState sin = drop_init(din), sout = drop_init(dout);     // dropout on in0 (dense -> dropout -> +residual) / on the output
  float x[2 * NIT];
  float s = 0.f;
  const int r0 = t0.tab ? tab_row(t0, row) : 0, r1 = t1.tab ? tab_row(t1, row) : 0, r2 = t2.tab ? tab_row(t2, row) : 0;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = it * 128 + lane * 2;
    float a = 0.f, b = 0.f, u, v;
    if (in0) {
      ld2<T>(in0 + (long long)row * H + c, u, v);
      if (sin.on) { u *= drop_mul(sin, (unsigned)(row * H + c)); v *= drop_mul(sin, (unsigned)(row * H + c + 1)); }
      a += u; b += v;
    }
    if (in1) { ld2<T>(in1 + (long long)row * H + c, u, v); a += u; b += v; }
    if (t0.tab) { ld2<T>((const T*)t0.tab + (long long)r0 * H + c, u, v); a += u; b += v; }
    if (t1.tab) { ld2<T>((const T*)t1.tab + (long long)r1 * H + c, u, v); a += u; b += v; }
    if (t2.tab) { ld2<T>((const T*)t2.tab + (long long)r2 * H + c, u, v); a += u; b += v; }
    x[2 * it] = a; x[2 * it + 1] = b; s += a + b;
  }
  if (!do_ln) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) st2<T>(out + (long long)row * H + it * 128 + lane * 2, x[2 * it], x[2 * it + 1]);
    return;
  }
  // gamma / beta requested BEFORE the two wave reductions: after them the loads' latency was the tail of the launch (a step's LayerNorms run on a
  // few hundred rows: one row per wave, nothing else to hide it behind)
  float2 gmv[NIT], btv[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) { gmv[it] = *(const float2*)(gamma + it * 128 + lane * 2); btv[it] = *(const float2*)(beta + it * 128 + lane * 2); }
  const float mean = wave_sum(s) / H;
  float q = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) { float a = x[2 * it] - mean, b = x[2 * it + 1] - mean; q += a * a + b * b; }
  const float rstd = rsqrtf(wave_sum(q) / H + eps);
  if (rstd_out && lane == 0) rstd_out[row] = rstd;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = it * 128 + lane * 2;
    const float2 g = gmv[it], b = btv[it];
    const float y0 = (x[2 * it] - mean) * rstd * g.x + b.x, y1 = (x[2 * it + 1] - mean) * rstd * g.y + b.y;
    st2<T>(out + (long long)row * H + c, y0, y1);          // pre-dropout y: the backward recovers xhat from it
    if (sout.on) st2<T>(out_drop + (long long)row * H + c, y0 * drop_mul(sout, (unsigned)(row * H + c)), y1 * drop_mul(sout, (unsigned)(row * H + c + 1)));
  }
}
template <typename T, int NIT>
__global__ __launch_bounds__(256) void ln_fwd_kernel(LnfParams a) { ln_fwd_body<T, NIT>(a, blockIdx.x); }
// two independent problems of one width in one launch (lockstep segments / L.group: the text and the panorama embedding norms)
template <typename T, int NIT>
__global__ __launch_bounds__(256) void ln_fwd_pair_kernel(LnfParams a, LnfParams b, int split) {
  if ((int)blockIdx.x < split) ln_fwd_body<T, NIT>(a, blockIdx.x);
  else ln_fwd_body<T, NIT>(b, blockIdx.x - split);
}

// Backward of the above.  dx (optional) = gradient wrt the pre-LN sum (shared by in0/in1).
// xhat is recomputed from y: xhat = (y - beta) / gamma.  Parameter/table gradients are fp32 atomics:
//   dgamma, dbeta: per-lane registers -> LDS block reduce -> one atomic per element per block;
//   const-row / tiny (<=3 rows, `small`) tables: LDS atomics into per-block slots -> one global atomic per element per block;
//   indexed tables (word / position / step embeddings): one global atomic row per input row.
#define LNB_ROWS 4   // rows per wave
struct LnbParams {
  int M; const void* dy; const void* y; const float* gamma; const float* beta; const float* rstd; void* dx; float* dgamma; float* dbeta;
  TabRef t0; float* d0; int small0; TabRef t1; float* d1; int small1; TabRef t2; float* d2; int small2; int do_ln;
  DropDesc ddy;               // the forward dropped its OUTPUT: dy is masked on load
  void* dxm; DropDesc ddx;    // the forward dropped its in0: second output dxm = dx * mask (gradient of the dense branch)
  int hot0;                   // >= 0: row of indexed table 0 that very many input rows hit (the padding token): summed per block in LDS
  int pg_partial;             // != 0: dgamma / dbeta are PARTIAL buffers [blocks][H]: every block stores its sums in its own row (no atomics)
  // TAIL instantiation only (magic_ln_bwd_tail: the MLM head's transform): dy arrives as fp32 (the split-K accumulator of the vocabulary
  // projection's input gradient) and dx leaves multiplied by the derivative of the activation in FRONT of the LayerNorm
  const float* dy32; const void* act_pre; int act;
  int nslab;                  // dy32 = the sum, in slab order, of nslab slabs of M x H (a deterministic split-K accumulator: magic_gemm with splitk < 0)
};

// NW = waves per block.  Every block ends in one same-address atomic per parameter / const-table element, and those serialise in L2
// (measured, M = 10.7 k rows, H = 128: 16.8 us with the gamma / beta gradients, 5.6 without): 16 waves per block = 4x fewer blocks.
// TAB = false: no table gradients in this launch (the LayerNorms inside the transformer blocks: 25 of a navigator step's 26 launches).  The
// table machinery costs 9 x 2 NIT registers per lane even when unused: at H = 768 the general form needs 256 VGPRs (one wave per SIMD:
// 17.5 us for 608 rows, pure latency), the plain form half of that.
template <typename T, int NIT, int NW, bool TAB = true, int RPI_ = 0, bool TAIL = false>
__device__ __forceinline__ void ln_bwd_body(const LnbParams& pp, const int bid, const int nblk, float* red) {
  const int M = pp.M, do_ln = pp.do_ln, small0 = pp.small0, small1 = pp.small1, small2 = pp.small2;
  const T* dy = (const T*)pp.dy; const T* y = (const T*)pp.y; T* dx = (T*)pp.dx;
  const float* gamma = pp.gamma; const float* beta = pp.beta; const float* rstd = pp.rstd;
  float* dgamma = pp.dgamma; float* dbeta = pp.dbeta; float* d0 = TAB ? pp.d0 : nullptr; float* d1 = TAB ? pp.d1 : nullptr; float* d2 = TAB ? pp.d2 : nullptr;
  const TabRef t0 = pp.t0, t1 = pp.t1, t2 = pp.t2;
  const DropState sdy = drop_init(pp.ddy), sdx = drop_init(pp.ddx);
  T* dxm = (T*)pp.dxm;
  constexpr int H = NIT * 128;     // red: [2][NW waves][H] gamma/beta partials | [9][H] table slots
  constexpr int NT = NW * 64;
  float* tacc = red + 2 * NW * H;
  const int hot0 = (d0 && t0.idx && !small0) ? pp.hot0 : -1;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const bool c0 = d0 && !t0.idx && !t0.mod, c1 = d1 && !t1.idx && !t1.mod, c2 = d2 && !t2.idx && !t2.mod;
  const bool l0 = d0 && (c0 || small0), l1 = d1 && (c1 || small1), l2 = d2 && (c2 || small2);   // LDS-slot tables
  const bool any_lds = l0 || l1 || l2 || hot0 >= 0;
  if (any_lds) {
    for (int i = threadIdx.x; i < 9 * H; i += NT) tacc[i] = 0.f;
    __syncthreads();
  }
  float ag[2 * NIT], ab[2 * NIT];
#pragma unroll
  for (int i = 0; i < 2 * NIT; ++i) { ag[i] = 0.f; ab[i] = 0.f; }
  // gradients of the const-row / tiny tables: per-lane registers over all of the wave's rows (an LDS atomic per row and element -- 16 waves
  // on the same 128 addresses -- cost 20 of the panorama embedding backward's 30 us), one LDS add per wave at the end
  float ts[3][3][2 * NIT];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int i = 0; i < 2 * NIT; ++i) ts[k][j][i] = 0.f;
  // Position-major traversal when a table is indexed by (row % mod) (the position embeddings): logical row i is physical row
  // (i % Bs) * mod + i / Bs, so the RPI rows a wave handles together share ONE table row and their gradients are merged before the
  // atomics (measured on the text-embedding backward: every position row took B = 48 same-address atomics per element from 48 different
  // waves; any bijection of rows is valid -- the rows are independent).
  const int pm_mod = (d0 && !t0.idx && t0.mod) ? t0.mod : (d1 && !t1.idx && t1.mod) ? t1.mod : (d2 && !t2.idx && t2.mod) ? t2.mod : 0;
  const int pm_Bs = (pm_mod > 0 && M % pm_mod == 0) ? M / pm_mod : 0;
  auto phys = [&](int i) { return pm_Bs ? (i % pm_Bs) * pm_mod + i / pm_Bs : i; };

  // grid-stride over groups of RPI rows per wave: the RPI rows' loads are issued together and their wave reductions
  // interleave (ILP), instead of one latency-bound row after another; the grid is capped so the per-block
  // parameter-gradient atomics stay few.
  constexpr int RPI = RPI_ ? RPI_ : (NIT <= 2 ? 4 : 2);
  float gmr[2 * NIT], btr[2 * NIT], igm[2 * NIT];
  if (do_ln) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = it * 128 + lane * 2;
      const float2 gm = *(const float2*)(gamma + c), bt = *(const float2*)(beta + c);
      gmr[2 * it] = gm.x; gmr[2 * it + 1] = gm.y; btr[2 * it] = bt.x; btr[2 * it + 1] = bt.y;
      igm[2 * it] = gm.x != 0.f ? 1.f / gm.x : 0.f; igm[2 * it + 1] = gm.y != 0.f ? 1.f / gm.y : 0.f;
    }
  }
  for (int base = (bid * NW + wid) * RPI; base < M; base += nblk * NW * RPI) {
    float g[RPI][2 * NIT], xh[RPI][2 * NIT];
    float s1[RPI], s2[RPI], rs[RPI];
#pragma unroll
    for (int u = 0; u < RPI; ++u) {
      const int row = phys(min(base + u, M - 1));    // clamped; rows >= M are masked below
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int c = it * 128 + lane * 2;
        if constexpr (TAIL) {
          float2 v = *(const float2*)(pp.dy32 + (long long)row * H + c);
          // the other slabs eight at a time: eight loads in flight, added in slab order (a fixed order: the sum is reproducible)
          for (int s0 = 1; s0 < pp.nslab; s0 += 8) {
            float2 w8[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const int sl = min(s0 + q, pp.nslab - 1);
              w8[q] = *(const float2*)(pp.dy32 + ((long long)sl * M + row) * H + c);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
              if (s0 + q < pp.nslab) { v.x += w8[q].x; v.y += w8[q].y; }
          }
          g[u][2 * it] = v.x; g[u][2 * it + 1] = v.y;
        } else {
          ld2<T>(dy + (long long)row * H + c, g[u][2 * it], g[u][2 * it + 1]);
        }
        if (sdy.on) { g[u][2 * it] *= drop_mul(sdy, (unsigned)(row * H + c)); g[u][2 * it + 1] *= drop_mul(sdy, (unsigned)(row * H + c + 1)); }
        if (do_ln) ld2<T>(y + (long long)row * H + c, xh[u][2 * it], xh[u][2 * it + 1]);
      }
      rs[u] = do_ln ? rstd[row] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < RPI; ++u) {
      const float live = (base + u < M) ? 1.f : 0.f;
      s1[u] = 0.f; s2[u] = 0.f;
      if (do_ln) {
#pragma unroll
        for (int i = 0; i < 2 * NIT; ++i) {
          const float da = g[u][i] * live;
          const float xa = (xh[u][i] - btr[i]) * igm[i];
          ag[i] += da * xa; ab[i] += da;
          const float ga = da * gmr[i];
          g[u][i] = ga; xh[u][i] = xa;
          s1[u] += ga; s2[u] += ga * xa;
        }
      }
    }
    if (do_ln) {
#pragma unroll
      for (int u = 0; u < RPI; ++u) { s1[u] = wave_sum(s1[u]); s2[u] = wave_sum(s2[u]); }
#pragma unroll
      for (int u = 0; u < RPI; ++u) {
        const float m1 = s1[u] / H, m2 = s2[u] / H;
#pragma unroll
        for (int i = 0; i < 2 * NIT; ++i) g[u][i] = rs[u] * (g[u][i] - m1 - xh[u][i] * m2);
      }
    }
#pragma unroll
    for (int u = 0; u < RPI; ++u) {
      if (base + u < M) {
        const int row = phys(base + u);
        if constexpr (TAIL) {
          const T* pre = (const T*)pp.act_pre;
#pragma unroll
          for (int it = 0; it < NIT; ++it) {
            const int c = it * 128 + lane * 2;
            float za, zb;
            ld2<T>(pre + (long long)row * H + c, za, zb);
            const float da = pp.act == 1 ? dgelu_f(za) : (za > 0.f ? 1.f : 0.f), db = pp.act == 1 ? dgelu_f(zb) : (zb > 0.f ? 1.f : 0.f);
            st2<T>(dx + (long long)row * H + c, g[u][2 * it] * da, g[u][2 * it + 1] * db);
          }
        } else if (dx) {
#pragma unroll
          for (int it = 0; it < NIT; ++it) st2<T>(dx + (long long)row * H + it * 128 + lane * 2, g[u][2 * it], g[u][2 * it + 1]);
        }
        if (dxm) {
#pragma unroll
          for (int it = 0; it < NIT; ++it) {
            const int c = it * 128 + lane * 2;
            st2<T>(dxm + (long long)row * H + c, g[u][2 * it] * drop_mul(sdx, (unsigned)(row * H + c)), g[u][2 * it + 1] * drop_mul(sdx, (unsigned)(row * H + c + 1)));
          }
        }
      }
    }
    // table gradients.  LDS-slot tables: LDS atomics per row.  Indexed tables: consecutive rows of the wave's group that hit the SAME
    // table row (position-major traversal above; runs of step id 0 in the map-token table) are summed in registers first.
#define TAB_GRAD(K, TK, DK, LK, CK)                                                          \
    if (DK) {                                                                                \
      if (LK) {                                                                              \
        _Pragma("unroll") for (int u = 0; u < RPI; ++u)                                      \
          if (base + u < M) {                                                                \
            const int sid = CK ? 0 : TK.idx[phys(base + u)];                                 \
            _Pragma("unroll") for (int j = 0; j < 3; ++j)                                    \
              _Pragma("unroll") for (int i = 0; i < 2 * NIT; ++i) ts[K][j][i] += (sid == j) ? g[u][i] : 0.f;   \
          }                                                                                  \
      } else {                                                                               \
        float run[2 * NIT];                                                                  \
        _Pragma("unroll") for (int i = 0; i < 2 * NIT; ++i) run[i] = 0.f;                    \
        _Pragma("unroll") for (int u = 0; u < RPI; ++u)                                      \
          if (base + u < M) {                                                                \
            const int tr = tab_row(TK, phys(base + u));                                      \
            const bool more = (u + 1 < RPI) && (base + u + 1 < M) && tab_row(TK, phys(base + u + 1)) == tr;   \
            _Pragma("unroll") for (int i = 0; i < 2 * NIT; ++i) run[i] += g[u][i];           \
            if (!more) {                                                                     \
              float* dst = DK + (long long)tr * H;                                           \
              const bool hot = (K == 0) && tr == hot0;                                       \
              _Pragma("unroll") for (int it = 0; it < NIT; ++it) {                           \
                const int c = it * 128 + lane * 2;                                           \
                if (hot) { atomicAdd(tacc + c, run[2 * it]); atomicAdd(tacc + c + 1, run[2 * it + 1]); }   \
                else { atomicAdd(dst + c, run[2 * it]); atomicAdd(dst + c + 1, run[2 * it + 1]); }         \
                run[2 * it] = 0.f; run[2 * it + 1] = 0.f;                                    \
              }                                                                              \
            }                                                                                \
          }                                                                                  \
      }                                                                                      \
    }
    TAB_GRAD(0, t0, d0, l0, c0)
    TAB_GRAD(1, t1, d1, l1, c1)
    TAB_GRAD(2, t2, d2, l2, c2)
#undef TAB_GRAD
  }
#define TAB_WAVE(K, LK, CK)                                                                 \
  if (LK) {                                                                                  \
    _Pragma("unroll") for (int j = 0; j < 3; ++j)                                            \
      if (j == 0 || !CK) {                                                                   \
        _Pragma("unroll") for (int it = 0; it < NIT; ++it) {                                 \
          const int c = it * 128 + lane * 2;                                                 \
          atomicAdd(tacc + (K * 3 + j) * H + c, ts[K][j][2 * it]); atomicAdd(tacc + (K * 3 + j) * H + c + 1, ts[K][j][2 * it + 1]);   \
        }                                                                                    \
      }                                                                                      \
  }
  TAB_WAVE(0, l0, c0)
  TAB_WAVE(1, l1, c1)
  TAB_WAVE(2, l2, c2)
#undef TAB_WAVE
  const bool pg = do_ln && dgamma != nullptr;      // gamma/beta grads here, or by ln_pgrad_kernel (dgamma == nullptr)
  if (pg) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = it * 128 + lane * 2;
      red[(0 * NW + wid) * H + c] = ag[2 * it]; red[(0 * NW + wid) * H + c + 1] = ag[2 * it + 1];
      red[(1 * NW + wid) * H + c] = ab[2 * it]; red[(1 * NW + wid) * H + c + 1] = ab[2 * it + 1];
    }
  }
  __syncthreads();
  if (pg) {
    for (int c = threadIdx.x; c < 2 * H; c += NT) {       // [gamma | beta] x H columns
      const int q = c / H, cc = c % H;
      float v = 0.f;
#pragma unroll
      for (int ww = 0; ww < NW; ++ww) v += red[(q * NW + ww) * H + cc];
      if (pp.pg_partial) (q ? dbeta : dgamma)[(long long)bid * H + cc] = v;
      else atomicAdd((q ? dbeta : dgamma) + cc, v);
    }
  }
  if (any_lds) {
#define TAB_FLUSH(K, TK, DK, LK, CK)                                                        \
    if (LK) {                                                                                \
      const int nrow = CK ? 1 : 3;                                                           \
      for (int i = threadIdx.x; i < nrow * H; i += NT) {                                     \
        const float v = tacc[K * 3 * H + i];                                                 \
        if (v != 0.f) atomicAdd(DK + (long long)(CK ? TK.off : 0) * H + i, v);               \
      }                                                                                      \
    }
    TAB_FLUSH(0, t0, d0, l0, c0)
    TAB_FLUSH(1, t1, d1, l1, c1)
    TAB_FLUSH(2, t2, d2, l2, c2)
#undef TAB_FLUSH
    if (hot0 >= 0)
      for (int i = threadIdx.x; i < H; i += NT) {
        const float v = tacc[i];
        if (v != 0.f) atomicAdd(d0 + (long long)hot0 * H + i, v);
      }
  }
}

template <typename T, int NIT, int NW, bool TAB = true, int RPI_ = 0>
__global__ __launch_bounds__(NW * 64) void ln_bwd_kernel(LnbParams p) {
  extern __shared__ __attribute__((aligned(16))) float red_dyn[];
  ln_bwd_body<T, NIT, NW, TAB, RPI_>(p, blockIdx.x, gridDim.x, red_dyn);
}
// TAIL: fp32 dy in, dx x act'(pre) out (see LnbParams); no table gradients
template <typename T, int NIT, int NW>
__global__ __launch_bounds__(NW * 64) void ln_bwd_tail_kernel(LnbParams p) {
  extern __shared__ __attribute__((aligned(16))) float red_dyn[];
  ln_bwd_body<T, NIT, NW, false, 0, true>(p, blockIdx.x, gridDim.x, red_dyn);
}
template <typename T, int NIT, int NW, bool TAB = true, int RPI_ = 0>
__global__ __launch_bounds__(NW * 64) void ln_bwd_pair_kernel(LnbParams a, LnbParams b, int nA) {
  extern __shared__ __attribute__((aligned(16))) float red_dyn[];
  if ((int)blockIdx.x < nA) ln_bwd_body<T, NIT, NW, TAB, RPI_>(a, blockIdx.x, nA, red_dyn);
  else ln_bwd_body<T, NIT, NW, TAB, RPI_>(b, blockIdx.x - nA, gridDim.x - nA, red_dyn);
}

// gamma/beta gradients of a LayerNorm as a separate column reduction: dgamma[c] += sum_m dy*xhat, dbeta[c] += sum_m dy
// (xhat from y).  A block owns PG_ROWS rows: its 4 waves stride over them with independent loads (no per-row wave
// reductions on the chain), reduce through LDS, and issue one atomic per column -- M/PG_ROWS-way contention only.
#define PG_ROWS 128
template <typename T, int NIT>
__global__ __launch_bounds__(256) void ln_pgrad_kernel(int M, const T* dy, const T* y, const float* gamma, const float* beta,
                                                       float* dgamma, float* dbeta) {
  constexpr int H = NIT * 128;
  extern __shared__ __attribute__((aligned(16))) float red[];   // [2][4][H]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float ag[2 * NIT], ab[2 * NIT], gm[2 * NIT], bt[2 * NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = it * 128 + lane * 2;
    const float2 g2 = *(const float2*)(gamma + c), b2 = *(const float2*)(beta + c);
    gm[2 * it] = g2.x != 0.f ? 1.f / g2.x : 0.f; gm[2 * it + 1] = g2.y != 0.f ? 1.f / g2.y : 0.f;
    bt[2 * it] = b2.x; bt[2 * it + 1] = b2.y;
    ag[2 * it] = ag[2 * it + 1] = ab[2 * it] = ab[2 * it + 1] = 0.f;
  }
  const int r0 = blockIdx.x * PG_ROWS, r1 = min(M, r0 + PG_ROWS);
#pragma unroll 4
  for (int row = r0 + wid; row < r1; row += 4) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = it * 128 + lane * 2;
      float da, db, ya, yb;
      ld2<T>(dy + (long long)row * H + c, da, db);
      ld2<T>(y + (long long)row * H + c, ya, yb);
      ag[2 * it] += da * (ya - bt[2 * it]) * gm[2 * it]; ag[2 * it + 1] += db * (yb - bt[2 * it + 1]) * gm[2 * it + 1];
      ab[2 * it] += da; ab[2 * it + 1] += db;
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = it * 128 + lane * 2;
    red[(0 * 4 + wid) * H + c] = ag[2 * it]; red[(0 * 4 + wid) * H + c + 1] = ag[2 * it + 1];
    red[(1 * 4 + wid) * H + c] = ab[2 * it]; red[(1 * 4 + wid) * H + c + 1] = ab[2 * it + 1];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < H; c += 256) {
    atomicAdd(dgamma + c, red[c] + red[H + c] + red[2 * H + c] + red[3 * H + c]);
    atomicAdd(dbeta + c, red[4 * H + c] + red[5 * H + c] + red[6 * H + c] + red[7 * H + c]);
  }
}

// ---------------------------------------------------------------------------------------------
// y = LN(x[M,Kin] @ W[H,Kin]^T + b) * gamma + beta, Kin <= 16 (position features; fp32 inputs/params)
// A wave owns SKF_ROWS consecutive rows; for H <= 256 its lanes keep their W rows, bias and gamma / beta in registers across them (one
// wave per row re-read the 3.5 KB of W per row through a non-unrolled k loop: 17 us for the 10.7 k panorama rows).
#define SKF_ROWS 4
template <typename T, int NIT>
__global__ __launch_bounds__(256) void smallk_ln_fwd_kernel(int M, int H, int Kin, const float* __restrict__ x, const float* __restrict__ W,
                                                            const float* __restrict__ b, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, T* __restrict__ out, float* __restrict__ rstd_out) {
  const int lane = threadIdx.x & 63, row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * SKF_ROWS;
  if (row0 >= M) return;
  constexpr bool WREG = NIT <= 2;
  float wr[WREG ? 2 * NIT : 1][16], br[2 * NIT], gr[2 * NIT], er[2 * NIT];
#pragma unroll
  for (int i = 0; i < 2 * NIT; ++i) {
    const int c = (i >> 1) * 128 + lane * 2 + (i & 1);
    br[i] = b[c]; gr[i] = gamma[c]; er[i] = beta[c];
    if constexpr (WREG) {
#pragma unroll
      for (int k = 0; k < 16; ++k) wr[i][k] = k < Kin ? W[c * Kin + k] : 0.f;
    }
  }
  for (int rr = 0; rr < SKF_ROWS; ++rr) {
    const int row = row0 + rr;
    if (row >= M) break;
    float xr[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) xr[k] = k < Kin ? x[(long long)row * Kin + k] : 0.f;
    float z[2 * NIT];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2 * NIT; ++i) {
      float a = br[i];
      if constexpr (WREG) {
#pragma unroll
        for (int k = 0; k < 16; ++k) a += xr[k] * wr[i][k];
      } else {
        const int c = (i >> 1) * 128 + lane * 2 + (i & 1);
#pragma unroll
        for (int k = 0; k < 16; ++k) a += (k < Kin) ? xr[k] * W[c * Kin + k] : 0.f;
      }
      z[i] = a; s += a;
    }
    const float mean = wave_sum(s) / H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 2 * NIT; ++i) { const float a = z[i] - mean; q += a * a; }
    const float rstd = rsqrtf(wave_sum(q) / H + eps);
    if (rstd_out && lane == 0) rstd_out[row] = rstd;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = it * 128 + lane * 2;
      st2<T>(out + (long long)row * H + c, (z[2 * it] - mean) * rstd * gr[2 * it] + er[2 * it],
             (z[2 * it + 1] - mean) * rstd * gr[2 * it + 1] + er[2 * it + 1]);
    }
  }
}

// Input stage of a cross-modal encoder in ONE launch (1 or 2 encoders): out = [gathered panorama embeddings] + LN(pos W^T + b) + [step
// embedding], i.e. csr_gather (+ csr_gather accumulate) -> smallk_ln_fwd -> ln_fwd(do_ln = 0) of the per-op path with the SAME rounding points
// (the gathered sum is rounded to T after each source, the position embedding is rounded to T before the sum), so both paths agree bit for bit.
struct magic_node_in {
  int M, Kin; const float* x; const float* W; const float* b; const float* gamma; const float* beta; float eps; int pad_;
  void* A; float* rstd; void* out;                                   // A = LN(pos W^T + b) (saved for the backward), out = the encoder's input
  const void* add0;                                                  // plain [M,H] addend instead of the gathers (node embeddings kept by the caller)
  const void* src1; const int* ptr1; const int* idx1; const float* w1;      // CSR row gathers out of src1 / src2 ([*,H]); ptr == NULL: none
  const void* src2; const int* ptr2; const int* idx2; const float* w2;
  const void* tab; const int* tab_idx;                               // + tab[tab_idx[row], :]
};
template <typename T, int NIT>
__device__ __forceinline__ void node_in_body(const magic_node_in& p, const int bid) {
  constexpr int H = NIT * 128;
  const int lane = threadIdx.x & 63, row0 = (bid * 4 + (threadIdx.x >> 6)) * SKF_ROWS, M = p.M, Kin = p.Kin;
  if (row0 >= M) return;
  constexpr bool WREG = NIT <= 2;
  float wr[WREG ? 2 * NIT : 1][16], br[2 * NIT], gr[2 * NIT], er[2 * NIT];
#pragma unroll
  for (int i = 0; i < 2 * NIT; ++i) {
    const int c = (i >> 1) * 128 + lane * 2 + (i & 1);
    br[i] = p.b[c]; gr[i] = p.gamma[c]; er[i] = p.beta[c];
    if constexpr (WREG) {
#pragma unroll
      for (int k = 0; k < 16; ++k) wr[i][k] = k < Kin ? p.W[c * Kin + k] : 0.f;
    }
  }
  T* A = (T*)p.A; T* out = (T*)p.out;
  for (int rr = 0; rr < SKF_ROWS; ++rr) {
    const int row = row0 + rr;
    if (row >= M) break;
    float xr[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) xr[k] = k < Kin ? p.x[(long long)row * Kin + k] : 0.f;
    float z[2 * NIT];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2 * NIT; ++i) {
      float a = br[i];
      if constexpr (WREG) {
#pragma unroll
        for (int k = 0; k < 16; ++k) a += xr[k] * wr[i][k];
      } else {
        const int c = (i >> 1) * 128 + lane * 2 + (i & 1);
#pragma unroll
        for (int k = 0; k < 16; ++k) a += (k < Kin) ? xr[k] * p.W[c * Kin + k] : 0.f;
      }
      z[i] = a; s += a;
    }
    const float mean = wave_sum(s) / H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 2 * NIT; ++i) { const float a = z[i] - mean; q += a * a; }
    const float rstd = rsqrtf(wave_sum(q) / H + p.eps);
    if (p.rstd && lane == 0) p.rstd[row] = rstd;
    // gathered embeddings (rounded to T after each source, as csr_gather / csr_gather(accumulate) store them)
    float g[2 * NIT];
#pragma unroll
    for (int i = 0; i < 2 * NIT; ++i) g[i] = 0.f;
    if (p.add0) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) ld2<T>((const T*)p.add0 + (long long)row * H + it * 128 + lane * 2, g[2 * it], g[2 * it + 1]);
    } else {
      if (p.ptr1) {
        const int e0 = p.ptr1[row], e1 = p.ptr1[row + 1];
        for (int e = e0; e < e1; ++e) {
          const float wv = p.w1 ? p.w1[e] : 1.f;
          const T* sp = (const T*)p.src1 + (long long)p.idx1[e] * H;
#pragma unroll
          for (int it = 0; it < NIT; ++it) { float u, v; ld2<T>(sp + it * 128 + lane * 2, u, v); g[2 * it] += wv * u; g[2 * it + 1] += wv * v; }
        }
#pragma unroll
        for (int i = 0; i < 2 * NIT; ++i) g[i] = to_f(from_f<T>(g[i]));
      }
      if (p.ptr2) {
        const int e0 = p.ptr2[row], e1 = p.ptr2[row + 1];
        if (e0 < e1) {
          float h[2 * NIT];
#pragma unroll
          for (int i = 0; i < 2 * NIT; ++i) h[i] = 0.f;
          for (int e = e0; e < e1; ++e) {
            const float wv = p.w2 ? p.w2[e] : 1.f;
            const T* sp = (const T*)p.src2 + (long long)p.idx2[e] * H;
#pragma unroll
            for (int it = 0; it < NIT; ++it) { float u, v; ld2<T>(sp + it * 128 + lane * 2, u, v); h[2 * it] += wv * u; h[2 * it + 1] += wv * v; }
          }
#pragma unroll
          for (int i = 0; i < 2 * NIT; ++i) g[i] = to_f(from_f<T>(h[i] + g[i]));
        }
      }
    }
    const int tr = p.tab ? p.tab_idx[row] : 0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = it * 128 + lane * 2;
      const float y0 = (z[2 * it] - mean) * rstd * gr[2 * it] + er[2 * it], y1 = (z[2 * it + 1] - mean) * rstd * gr[2 * it + 1] + er[2 * it + 1];
      st2<T>(A + (long long)row * H + c, y0, y1);
      float o0 = g[2 * it] + to_f(from_f<T>(y0)), o1 = g[2 * it + 1] + to_f(from_f<T>(y1));
      if (p.tab) { float u, v; ld2<T>((const T*)p.tab + (long long)tr * H + c, u, v); o0 += u; o1 += v; }
      st2<T>(out + (long long)row * H + c, o0, o1);
    }
  }
}
template <typename T, int NIT>
__global__ __launch_bounds__(256) void node_in_fwd_kernel(magic_node_in a, magic_node_in b, int nA) {
  if ((int)blockIdx.x < nA) node_in_body<T, NIT>(a, blockIdx.x);
  else node_in_body<T, NIT>(b, blockIdx.x - nA);
}


// Input stage of the panorama encoder in ONE launch (round 4), together with the text embedding rows:
//   A1  = LN_img(P0)                       P0 = img_linear(view features), the GEMM in front of this launch
//   A2  = LN_loc(loc W^T + b)              loc [M, Kin <= 16] fp32
//   X0  = LN(A1 + A2 + nav_tab[nav_type] + tok_tab[0])   (+ its dropped copy X0d)
// i.e. ln_fwd -> smallk_ln_fwd -> ln_fwd of the per-op path (three launches on the critical path in front of the whole-encoder launch) with
// the SAME rounding points and summation order -- A1 / A2 are rounded to T before the sum, the sum runs in0 + in1 + t0 + t1 as ln_fwd_body's
// -- so every saved tensor (A1, rstd_a1, A2, rstd_a2, X0, rstd_x0, X0d) is bit-identical and the per-op backward kernels read them unchanged.
// Blocks >= nA run ln_fwd_body on a second, independent problem (the text embedding: three table gathers + LayerNorm + dropout).
struct magic_pano_in {
  int M, Kin; float eps; int pad_;
  const void* P0; const float* g1; const float* b1; void* A1; float* rstd1;
  const float* loc; const float* W; const float* b; const float* g2; const float* b2; void* A2; float* rstd2;
  const void* nav_tab; const int* nav_idx; const void* tok_tab;
  const float* g3; const float* b3; void* X0; float* rstd3; void* X0d; DropDesc dout;
};
template <typename T, int NIT>
__device__ __forceinline__ void pano_in_body(const magic_pano_in& p, const int bid) {
  constexpr int H = NIT * 128;
  const int lane = threadIdx.x & 63, row0 = (bid * 4 + (threadIdx.x >> 6)) * SKF_ROWS, M = p.M, Kin = p.Kin;
  if (row0 >= M) return;
  constexpr bool WREG = NIT <= 2;
  float wr[WREG ? 2 * NIT : 1][16], br[2 * NIT];
#pragma unroll
  for (int i = 0; i < 2 * NIT; ++i) {
    const int c = (i >> 1) * 128 + lane * 2 + (i & 1);
    br[i] = p.b[c];
    if constexpr (WREG) {
#pragma unroll
      for (int k = 0; k < 16; ++k) wr[i][k] = k < Kin ? p.W[c * Kin + k] : 0.f;
    }
  }
  const DropState sout = drop_init(p.dout);
  for (int rr = 0; rr < SKF_ROWS; ++rr) {
    const int row = row0 + rr;
    if (row >= M) break;
    // ---- A1 = LN_img(P0 row)  (ln_fwd_body with in0 only)
    float x[2 * NIT], s = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      float u, v;
      ld2<T>((const T*)p.P0 + (long long)row * H + it * 128 + lane * 2, u, v);
      x[2 * it] = 0.f + u; x[2 * it + 1] = 0.f + v; s += x[2 * it] + x[2 * it + 1];
    }
    float mean = wave_sum(s) / H, q = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) { const float a = x[2 * it] - mean, b = x[2 * it + 1] - mean; q += a * a + b * b; }
    float rstd = rsqrtf(wave_sum(q) / H + p.eps);
    if (lane == 0) p.rstd1[row] = rstd;
    float a1[2 * NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = it * 128 + lane * 2;
      const float2 g = *(const float2*)(p.g1 + c), b = *(const float2*)(p.b1 + c);
      const float y0 = (x[2 * it] - mean) * rstd * g.x + b.x, y1 = (x[2 * it + 1] - mean) * rstd * g.y + b.y;
      st2<T>((T*)p.A1 + (long long)row * H + c, y0, y1);
      a1[2 * it] = to_f(from_f<T>(y0)); a1[2 * it + 1] = to_f(from_f<T>(y1));
    }
    // ---- A2 = LN_loc(loc W^T + b)  (smallk_ln_fwd_kernel)
    float xr[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) xr[k] = k < Kin ? p.loc[(long long)row * Kin + k] : 0.f;
    float z[2 * NIT];
    s = 0.f;
#pragma unroll
    for (int i = 0; i < 2 * NIT; ++i) {
      float a = br[i];
      if constexpr (WREG) {
#pragma unroll
        for (int k = 0; k < 16; ++k) a += xr[k] * wr[i][k];
      } else {
        const int c = (i >> 1) * 128 + lane * 2 + (i & 1);
#pragma unroll
        for (int k = 0; k < 16; ++k) a += (k < Kin) ? xr[k] * p.W[c * Kin + k] : 0.f;
      }
      z[i] = a; s += a;
    }
    mean = wave_sum(s) / H;
    q = 0.f;
#pragma unroll
    for (int i = 0; i < 2 * NIT; ++i) { const float a = z[i] - mean; q += a * a; }
    rstd = rsqrtf(wave_sum(q) / H + p.eps);
    if (lane == 0) p.rstd2[row] = rstd;
    float a2[2 * NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = it * 128 + lane * 2;
      const float2 g = *(const float2*)(p.g2 + c), b = *(const float2*)(p.b2 + c);
      const float y0 = (z[2 * it] - mean) * rstd * g.x + b.x, y1 = (z[2 * it + 1] - mean) * rstd * g.y + b.y;
      st2<T>((T*)p.A2 + (long long)row * H + c, y0, y1);
      a2[2 * it] = to_f(from_f<T>(y0)); a2[2 * it + 1] = to_f(from_f<T>(y1));
    }
    // ---- X0 = LN(A1 + A2 + nav_tab[nav_type] + tok_tab[0]), X0d = dropout(X0)   (ln_fwd_body with in0, in1, t0, t1)
    const int nr = p.nav_idx[row];
    s = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = it * 128 + lane * 2;
      float a = 0.f, b = 0.f, u, v;
      a += a1[2 * it]; b += a1[2 * it + 1];
      a += a2[2 * it]; b += a2[2 * it + 1];
      ld2<T>((const T*)p.nav_tab + (long long)nr * H + c, u, v); a += u; b += v;
      ld2<T>((const T*)p.tok_tab + c, u, v); a += u; b += v;
      x[2 * it] = a; x[2 * it + 1] = b; s += a + b;
    }
    mean = wave_sum(s) / H;
    q = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) { const float a = x[2 * it] - mean, b = x[2 * it + 1] - mean; q += a * a + b * b; }
    rstd = rsqrtf(wave_sum(q) / H + p.eps);
    if (lane == 0) p.rstd3[row] = rstd;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = it * 128 + lane * 2;
      const float2 g = *(const float2*)(p.g3 + c), b = *(const float2*)(p.b3 + c);
      const float y0 = (x[2 * it] - mean) * rstd * g.x + b.x, y1 = (x[2 * it + 1] - mean) * rstd * g.y + b.y;
      st2<T>((T*)p.X0 + (long long)row * H + c, y0, y1);
      if (sout.on) st2<T>((T*)p.X0d + (long long)row * H + c, y0 * drop_mul(sout, (unsigned)(row * H + c)), y1 * drop_mul(sout, (unsigned)(row * H + c + 1)));
    }
  }
}
template <typename T, int NIT>
__global__ __launch_bounds__(256) void embed_in_fwd_kernel(magic_pano_in a, LnfParams b, int nA) {
  if ((int)blockIdx.x < nA) pano_in_body<T, NIT>(a, blockIdx.x);
  else ln_fwd_body<T, NIT>(b, blockIdx.x - nA);
}


// Backward of the panorama encoder's input stage in ONE launch (round 4), together with the text embedding's LayerNorm backward:
//   dy (gradient of the stage's output, masked by the output dropout) -> sum-LayerNorm backward -> dsum (rounded to T: the per-op path stores it)
//     -> nav-type / token-type row gradients, gamma3 / beta3
//   dsum -> image-LayerNorm backward -> dP0 (operand of the image projection's deferred weight gradient), gamma1 / beta1
//   dsum -> location-LayerNorm backward -> dz -> dW_loc[H, Kin] += dz^T loc, db_loc += colsum(dz), gamma2 / beta2
// i.e. magic_ln_bwd -> magic_ln_bwd -> magic_smallk_ln_bwd of the per-op path (three launches at the very end of the backward chain, each
// ending in its own round of same-address parameter-gradient atomics) with the same formulas and rounding points: dP0 is bit-identical, the
// parameter gradients agree to fp32 summation order.  Every parameter gradient of the stage -- 11 + Kin vectors of H -- is accumulated in
// registers over the wave's rows and reduced through LDS four vectors at a time: one round of atomics per block instead of three.
// Blocks >= nA run ln_bwd_body on a second, independent problem (the text embedding: three table scatters).
#define PIB_KMAX 8
struct magic_pano_in_bwd {
  int M, Kin, pad0_, pad1_;
  const void* dy; DropDesc ddy;
  const void* X0; const float* rstd3; const float* g3; const float* b3; float* dg3; float* db3;
  const int* nav_idx; float* d_nav; float* d_tok;
  const void* A1; const float* rstd1; const float* g1; const float* b1; float* dg1; float* db1; void* dP0;
  const void* A2; const float* rstd2; const float* g2; const float* b2; float* dg2; float* db2;
  const float* loc; float* dW; float* dbl;
  // round 6: != NULL -> every block STORES its (11 + Kin) H sums in row `block` of this buffer (pad0_ rows of pad1_ floats: dg3 | db3 | d_nav[3 H] | d_tok | dg1 | db1 |
  // dg2 | db2 | dbl | dW[H Kin], each laid out as its destination) instead of ending in as many same-address atomics; the caller adds the rows up in block order
  // (magic_colsum_add_v): the atomics WERE the launch's time, and the sums become reproducible
  float* part;
};
#ifdef PIB_VARIANT
__device__ long long magic_pib_ticks[8];
extern "C" int magic_debug_pib_ticks(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(magic_pib_ticks), sizeof(long long) * 8) == hipSuccess ? 0 : -2; }
#define PIB_MARK(i) do { if (bid == 0 && threadIdx.x == 0) magic_pib_ticks[i] = wall_clock64(); } while (0)
#else
#define PIB_MARK(i)
#endif
template <typename T, int NIT, int NW>
__device__ __forceinline__ void pano_in_bwd_body(const magic_pano_in_bwd& p, const int bid, const int nblk, float* red) {
  constexpr int H = NIT * 128, NT = NW * 64, NS = 11 + PIB_KMAX, E = 2 * NIT;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, M = p.M, Kin = p.Kin;
  PIB_MARK(0);
  const DropState sdy = drop_init(p.ddy);
  // accumulator sets: 0 dg3 | 1 db3 | 2-4 nav rows | 5 token-type row | 6 dg1 | 7 db1 | 8 dg2 | 9 db2 | 10 db_loc | 11.. dW_loc[:, k]
  float acc[NS][E];
#pragma unroll
  for (int q = 0; q < NS; ++q)
#pragma unroll
    for (int i = 0; i < E; ++i) acc[q][i] = 0.f;
  float g3r[E], b3r[E], i3r[E], g1r[E], b1r[E], i1r[E], g2r[E], b2r[E];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = it * 128 + lane * 2;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      g3r[2 * it + e] = p.g3[c + e]; b3r[2 * it + e] = p.b3[c + e]; i3r[2 * it + e] = g3r[2 * it + e] != 0.f ? 1.f / g3r[2 * it + e] : 0.f;
      g1r[2 * it + e] = p.g1[c + e]; b1r[2 * it + e] = p.b1[c + e]; i1r[2 * it + e] = g1r[2 * it + e] != 0.f ? 1.f / g1r[2 * it + e] : 0.f;
      g2r[2 * it + e] = p.g2[c + e]; b2r[2 * it + e] = p.b2[c + e];
    }
  }
  // PIB_RPI rows per wave and iteration: their loads (four row vectors, three rstd, the nav id, the location features -- all cold: written by the
  // forward a millisecond earlier) are issued together, so a wave pays one memory round trip per PIB_RPI rows
  constexpr int PIB_RPI = 2;
  PIB_MARK(1);
  int it_ = 0;
  for (int base = (bid * NW + wid) * PIB_RPI; base < M; base += nblk * NW * PIB_RPI) {
    if (it_ < 2) { PIB_MARK(2 + 2 * it_); }
    float dv[PIB_RPI][E], y3v[PIB_RPI][E], y1v[PIB_RPI][E], y2v[PIB_RPI][E], lxv[PIB_RPI][PIB_KMAX];
    float rs3v[PIB_RPI], rs1v[PIB_RPI], rs2v[PIB_RPI];
    int sidv[PIB_RPI];
#pragma unroll
    for (int u = 0; u < PIB_RPI; ++u) {
      const int row = min(base + u, M - 1);          // clamped; rows >= M are skipped below
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int c = it * 128 + lane * 2;
        ld2<T>((const T*)p.dy + (long long)row * H + c, dv[u][2 * it], dv[u][2 * it + 1]);
        ld2<T>((const T*)p.X0 + (long long)row * H + c, y3v[u][2 * it], y3v[u][2 * it + 1]);
        ld2<T>((const T*)p.A1 + (long long)row * H + c, y1v[u][2 * it], y1v[u][2 * it + 1]);
        ld2<T>((const T*)p.A2 + (long long)row * H + c, y2v[u][2 * it], y2v[u][2 * it + 1]);
      }
      rs3v[u] = p.rstd3[row]; rs1v[u] = p.rstd1[row]; rs2v[u] = p.rstd2[row];
      sidv[u] = p.nav_idx[row];
#pragma unroll
      for (int k = 0; k < PIB_KMAX; ++k) lxv[u][k] = k < Kin ? p.loc[(long long)row * Kin + k] : 0.f;
    }
#ifdef PIB_VARIANT
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (it_ < 2) { PIB_MARK(3 + 2 * it_); }
    ++it_;
#endif
#pragma unroll
    for (int u = 0; u < PIB_RPI; ++u) {
    if (base + u >= M) break;
    const int row = base + u;
    float d[E], y3[E], y1[E], y2[E], lx[PIB_KMAX];
#pragma unroll
    for (int i = 0; i < E; ++i) { d[i] = dv[u][i]; y3[i] = y3v[u][i]; y1[i] = y1v[u][i]; y2[i] = y2v[u][i]; }
#pragma unroll
    for (int k = 0; k < PIB_KMAX; ++k) lx[k] = lxv[u][k];
    if (sdy.on) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int c = it * 128 + lane * 2;
        d[2 * it] *= drop_mul(sdy, (unsigned)(row * H + c)); d[2 * it + 1] *= drop_mul(sdy, (unsigned)(row * H + c + 1));
      }
    }
    const float rs3 = rs3v[u], rs1 = rs1v[u], rs2 = rs2v[u];
    const int sid = sidv[u];
    // ---- sum LayerNorm (ln_bwd_body's arithmetic)
    float s1 = 0.f, s2 = 0.f, ga[E], xa[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
      xa[i] = (y3[i] - b3r[i]) * i3r[i];
      acc[0][i] += d[i] * xa[i]; acc[1][i] += d[i];
      ga[i] = d[i] * g3r[i];
      s1 += ga[i]; s2 += ga[i] * xa[i];
    }
    float m1 = wave_sum(s1) / H, m2 = wave_sum(s2) / H;
    float ds[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const float v = rs3 * (ga[i] - m1 - xa[i] * m2);
      acc[2][i] += sid == 0 ? v : 0.f; acc[3][i] += sid == 1 ? v : 0.f; acc[4][i] += sid == 2 ? v : 0.f;
      acc[5][i] += v;
      ds[i] = to_f(from_f<T>(v));                     // the per-op path hands dsum on as a T tensor
    }
    // ---- image LayerNorm -> dP0
    s1 = 0.f; s2 = 0.f;
#pragma unroll
    for (int i = 0; i < E; ++i) {
      xa[i] = (y1[i] - b1r[i]) * i1r[i];
      acc[6][i] += ds[i] * xa[i]; acc[7][i] += ds[i];
      ga[i] = ds[i] * g1r[i];
      s1 += ga[i]; s2 += ga[i] * xa[i];
    }
    m1 = wave_sum(s1) / H; m2 = wave_sum(s2) / H;
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      st2<T>((T*)p.dP0 + (long long)row * H + it * 128 + lane * 2, rs1 * (ga[2 * it] - m1 - xa[2 * it] * m2), rs1 * (ga[2 * it + 1] - m1 - xa[2 * it + 1] * m2));
    // ---- location LayerNorm (smallk_ln_bwd_body's arithmetic) -> dz -> loc_linear gradients
    s1 = 0.f; s2 = 0.f;
#pragma unroll
    for (int i = 0; i < E; ++i) {
      xa[i] = g2r[i] != 0.f ? (y2[i] - b2r[i]) / g2r[i] : 0.f;
      acc[8][i] += ds[i] * xa[i]; acc[9][i] += ds[i];
      ga[i] = ds[i] * g2r[i];
      s1 += ga[i]; s2 += ga[i] * xa[i];
    }
    m1 = wave_sum(s1) / H; m2 = wave_sum(s2) / H;
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const float dz = rs2 * (ga[i] - m1 - xa[i] * m2);
      acc[10][i] += dz;
#pragma unroll
      for (int k = 0; k < PIB_KMAX; ++k) acc[11 + k][i] += dz * lx[k];
    }
    }
  }
  PIB_MARK(6);
#if defined(PIB_VARIANT) && PIB_VARIANT == 1
  if (acc[0][0] == 123.456f) p.dg3[0] = acc[3][1] + acc[12][0] + acc[18][1];      // (timing variant: no tail)
  return;
#endif
  // ---- one round of atomics per block: four accumulator sets at a time through LDS ([NW][4][H])
  for (int q0 = 0; q0 < 11 + Kin; q0 += 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int c = it * 128 + lane * 2;
        float v0 = 0.f, v1 = 0.f;
#pragma unroll
        for (int q = 0; q < NS; ++q)
          if (q == q0 + j) { v0 = acc[q][2 * it]; v1 = acc[q][2 * it + 1]; }
        red[(wid * 4 + j) * H + c] = v0; red[(wid * 4 + j) * H + c + 1] = v1;
      }
    __syncthreads();
    for (int t = threadIdx.x; t < 4 * H; t += NT) {
      const int j = t / H, c = t % H, q = q0 + j;
      if (q < 11 + Kin) {
        float v = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) v += red[(ww * 4 + j) * H + c];
        if (p.part) {                // partial row of this block: vectors 0..10 at q H + c, the location weight at 11 H + c Kin + k (its destination's layout)
          p.part[(long long)bid * p.pad1_ + (q < 11 ? q * H + c : 11 * H + c * Kin + (q - 11))] = v;
          continue;
        }
        float* dst = q == 0 ? p.dg3 + c : q == 1 ? p.db3 + c : q <= 4 ? p.d_nav + (long long)(q - 2) * H + c : q == 5 ? p.d_tok + c
                   : q == 6 ? p.dg1 + c : q == 7 ? p.db1 + c : q == 8 ? p.dg2 + c : q == 9 ? p.db2 + c : q == 10 ? p.dbl + c
                   : p.dW + (long long)c * Kin + (q - 11);
#if defined(PIB_VARIANT) && PIB_VARIANT == 2
        if (v == 123.456f) atomicAdd(dst, v);                                         // (timing variant: no atomics)
#else
        if (v != 0.f) atomicAdd(dst, v);
#endif
      }
    }
    __syncthreads();
  }
}
template <typename T, int NIT, int NW>
__global__ __launch_bounds__(NW * 64) void embed_in_bwd_kernel(magic_pano_in_bwd a, LnbParams b, int nB, ColsumJobs cs) {
  extern __shared__ __attribute__((aligned(16))) float red_dyn[];
  // the LAST cs.n workgroups of the grid: the column sums of the row-block backward's partial LayerNorm gradients (csrc/encbwd.hip) -- every
  // row-block launch of the backward precedes this one, and the launch has CUs to spare
  if ((int)blockIdx.x >= (int)gridDim.x - cs.n) { colsum_body(cs, blockIdx.x - (gridDim.x - cs.n), NIT * 128, red_dyn); return; }
  // the text problem's blocks come FIRST in the grid: they are the long pole (the word-embedding scatter's atomics) and a block of this kernel
  // fills a CU (16 waves), so blocks past the 256th wait for a free CU -- behind the panorama blocks they started a whole round late (68 us
  // for the launch instead of ~35)
  if ((int)blockIdx.x < nB) ln_bwd_body<T, NIT, NW>(b, blockIdx.x, nB, red_dyn);
  else pano_in_bwd_body<T, NIT, NW>(a, blockIdx.x - nB, gridDim.x - nB - cs.n, red_dyn);
}

// backward: dW[H,Kin], db[H], dgamma, dbeta (fp32 atomics, block-reduced).  SK_ROWS rows per block (NW waves);
// dz rows are parked in LDS so the dW outer product is a cooperative (c,k) loop.  Every block ends in H (Kin + 3) same-address atomics,
// so at H = 128 and M >= 4096 a block takes 64 rows with 16 waves (half as many blocks as the 32-row / 4-wave shape, 4 rows per wave).
struct SkbParams { int M, Kin; const float* x; const void* dy; const void* y; const float* gamma; const float* beta; const float* rstd;
                   float* dW; float* db; float* dgamma; float* dbeta;
                   float* part; };       // != NULL: every block STORES its sums in its own row [H Kin | H | H | H] (dW, db, dgamma, dbeta) instead of the atomics
template <typename T, int NIT, int NW, int SK_ROWS>
__device__ __forceinline__ void smallk_ln_bwd_body(const SkbParams& pp, const int bid) {
  constexpr int H = NIT * 128;
  const int M = pp.M, Kin = pp.Kin;
  const float* x = pp.x; const T* dy = (const T*)pp.dy; const T* y = (const T*)pp.y;
  const float* gamma = pp.gamma; const float* beta = pp.beta; const float* rstd = pp.rstd;
  float* dW = pp.dW; float* db = pp.db; float* dgamma = pp.dgamma; float* dbeta = pp.dbeta;
  float* prow = pp.part ? pp.part + (long long)bid * H * (Kin + 3) : nullptr;
  extern __shared__ __attribute__((aligned(16))) float sm[];   // dz[SK_ROWS][H] | xs[SK_ROWS][16] | red[2][NW][H]
  float* dz = sm;
  float* xs = sm + SK_ROWS * H;
  float* red = xs + SK_ROWS * 16;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  constexpr int nit = NIT;
  float ag[2 * NIT], ab[2 * NIT];
#pragma unroll
  for (int i = 0; i < 2 * NIT; ++i) { ag[i] = 0.f; ab[i] = 0.f; }
  const int base = bid * SK_ROWS;
  for (int rr = wid; rr < SK_ROWS; rr += NW) {
    const int row = base + rr;
    const bool ok = row < M;
    float g[2 * NIT], xh[2 * NIT];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      if (it < nit) {
        const int c = it * 128 + lane * 2;
        float da = 0.f, dbv = 0.f, ya = 0.f, yb = 0.f;
        if (ok) { ld2<T>(dy + (long long)row * H + c, da, dbv); ld2<T>(y + (long long)row * H + c, ya, yb); }
        const float g0 = gamma[c], g1 = gamma[c + 1];
        const float xa = (ok && g0 != 0.f) ? (ya - beta[c]) / g0 : 0.f, xb = (ok && g1 != 0.f) ? (yb - beta[c + 1]) / g1 : 0.f;
        ag[2 * it] += da * xa; ag[2 * it + 1] += dbv * xb; ab[2 * it] += da; ab[2 * it + 1] += dbv;
        g[2 * it] = da * g0; g[2 * it + 1] = dbv * g1; xh[2 * it] = xa; xh[2 * it + 1] = xb;
        s1 += g[2 * it] + g[2 * it + 1]; s2 += g[2 * it] * xa + g[2 * it + 1] * xb;
      }
    const float m1 = wave_sum(s1) / H, m2 = wave_sum(s2) / H, rs = ok ? rstd[row] : 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      if (it < nit) {
        const int c = it * 128 + lane * 2;
        dz[rr * H + c] = rs * (g[2 * it] - m1 - xh[2 * it] * m2);
        dz[rr * H + c + 1] = rs * (g[2 * it + 1] - m1 - xh[2 * it + 1] * m2);
      }
    if (lane < 16) xs[rr * 16 + lane] = (ok && lane < Kin) ? x[(long long)row * Kin + lane] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < H * Kin; i += NW * 64) {
    const int c = i / Kin, k = i % Kin;
    float s = 0.f;
#pragma unroll 8
    for (int r = 0; r < SK_ROWS; ++r) s += dz[r * H + c] * xs[r * 16 + k];
    if (prow) prow[i] = s; else atomicAdd(dW + i, s);
  }
  for (int c = threadIdx.x; c < H; c += NW * 64) {
    float s = 0.f;
#pragma unroll 8
    for (int r = 0; r < SK_ROWS; ++r) s += dz[r * H + c];
    if (prow) prow[H * Kin + c] = s; else atomicAdd(db + c, s);
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it)
    if (it < nit) {
      const int c = it * 128 + lane * 2;
      red[(0 * NW + wid) * H + c] = ag[2 * it]; red[(0 * NW + wid) * H + c + 1] = ag[2 * it + 1];
      red[(1 * NW + wid) * H + c] = ab[2 * it]; red[(1 * NW + wid) * H + c + 1] = ab[2 * it + 1];
    }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * H; c += NW * 64) {
    const int q = c / H, cc = c % H;
    float v = 0.f;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) v += red[(q * NW + ww) * H + cc];
    if (prow) prow[H * (Kin + 1 + q) + cc] = v; else atomicAdd((q ? dbeta : dgamma) + cc, v);
  }
}

template <typename T, int NIT, int NW, int SK_ROWS>
__global__ __launch_bounds__(NW * 64) void smallk_ln_bwd_kernel(SkbParams a, SkbParams b, int nA) {
  if ((int)blockIdx.x < nA) smallk_ln_bwd_body<T, NIT, NW, SK_ROWS>(a, blockIdx.x);
  else smallk_ln_bwd_body<T, NIT, NW, SK_ROWS>(b, blockIdx.x - nA);
}

// ---------------------------------------------------------------------------------------------
// P[b,h,q,:] = softmax(scale * S + keymask + sprel(dist) ), rows of length Nk <= 512, ld = ldp (zero pad)
#define SM_IT 8
template <typename T>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(int B, int nh, int Nq, int Nk, int ldp, const float* S, T* P, float scale,
                                                          const unsigned char* kmask, const float* dist, const float* sprel_w, const float* sprel_b) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long rows = (long long)B * nh * Nq;
  if (row >= rows) return;
  const int q = row % Nq, b = row / ((long long)nh * Nq);
  const float sw = dist ? sprel_w[0] : 0.f, sb = dist ? sprel_b[0] : 0.f;
  float v[SM_IT];
  float mx = -3.0e38f;
#pragma unroll
  for (int it = 0; it < SM_IT; ++it) {
    const int k = it * 64 + lane;
    float x = -3.0e38f;
    if (k < Nk) {
      x = S[row * ldp + k] * scale;
      if (kmask && !kmask[(long long)b * Nk + k]) x += -10000.0f;
      if (dist) x += sw * dist[((long long)b * Nq + q) * Nk + k] + sb;
    }
    v[it] = x; mx = fmaxf(mx, x);
  }
  mx = wave_max(mx);
  float s = 0.f;
#pragma unroll
  for (int it = 0; it < SM_IT; ++it) {
    const int k = it * 64 + lane;
    const float e = k < Nk ? __expf(v[it] - mx) : 0.f;
    v[it] = e; s += e;
  }
  const float inv = 1.0f / wave_sum(s);
#pragma unroll
  for (int it = 0; it < SM_IT; ++it) {
    const int k = it * 64 + lane;
    if (k < ldp) P[row * ldp + k] = from_f<T>(k < Nk ? v[it] * inv : 0.f);
  }
}

// dS = P * (dP - sum(dP*P)) ; writes scale*dS (T, zero pad); optional sprel grads: d[0] += sum dS*dist, d[1] += sum dS
template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(int B, int nh, int Nq, int Nk, int ldp, const T* P, const float* dP, T* dS, float scale,
                                                          const float* dist, float* dsprel_w, float* dsprel_b) {
  __shared__ float red[8];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const long long row = (long long)blockIdx.x * 4 + wid;
  const long long rows = (long long)B * nh * Nq;
  float a0 = 0.f, a1 = 0.f;
  if (row < rows) {
    const int q = row % Nq, b = row / ((long long)nh * Nq);
    float p[SM_IT], g[SM_IT];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < SM_IT; ++it) {
      const int k = it * 64 + lane;
      p[it] = 0.f; g[it] = 0.f;
      if (k < Nk) { p[it] = to_f(P[row * ldp + k]); g[it] = dP[row * ldp + k]; s += p[it] * g[it]; }
    }
    s = wave_sum(s);
#pragma unroll
    for (int it = 0; it < SM_IT; ++it) {
      const int k = it * 64 + lane;
      const float d = p[it] * (g[it] - s);
      if (k < ldp) dS[row * ldp + k] = from_f<T>(k < Nk ? d * scale : 0.f);
      if (dist && k < Nk) { a0 += d * dist[((long long)b * Nq + q) * Nk + k]; a1 += d; }
    }
  }
  if (dsprel_w) {
    a0 = wave_sum(a0); a1 = wave_sum(a1);
    if (lane == 0) { red[wid] = a0; red[4 + wid] = a1; }
    __syncthreads();
    if (threadIdx.x == 0) {
      atomicAdd(dsprel_w, red[0] + red[1] + red[2] + red[3]);
      atomicAdd(dsprel_b, red[4] + red[5] + red[6] + red[7]);
    }
  }
}

// mean over heads: out[b,q,k] = (1/nh) sum_h P[b,h,q,k]   (fp32 out, ld = ldp)
template <typename T>
__global__ void head_mean_fwd_kernel(int B, int nh, long long inner, const T* P, float* out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * inner) return;
  const long long b = i / inner, r = i % inner;
  float s = 0.f;
  for (int h = 0; h < nh; ++h) s += to_f(P[(b * nh + h) * inner + r]);
  out[i] = s / nh;
}
// dP[b,h,q,k] (+)= g[b,q,k] / nh   (fp32 dP)
__global__ void head_mean_bwd_kernel(int B, int nh, long long inner, const float* g, float* dP, int accumulate) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * nh * inner) return;
  const long long b = i / (nh * inner), r = i % inner;
  const float v = g[b * inner + r] / nh;
  dP[i] = accumulate ? dP[i] + v : v;
}

// ---------------------------------------------------------------------------------------------
// ClsPrediction tail: logit[m] = dot(LN(Y[m]) * gamma + beta, w2) + b2    (Y = relu(linear) from the GEMM)
template <typename T, int NIT>
__global__ __launch_bounds__(256) void lndot_fwd_kernel(int M, int H, const T* Y, const float* gamma, const float* beta, float eps,
                                                        const float* w2, const float* b2, float* logit) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  constexpr int nit = NIT;
  float x[2 * NIT];
  float s = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it)
    if (it < nit) { ld2<T>(Y + (long long)row * H + it * 128 + lane * 2, x[2 * it], x[2 * it + 1]); s += x[2 * it] + x[2 * it + 1]; }
  const float mean = wave_sum(s) / H;
  float q = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it)
    if (it < nit) { float a = x[2 * it] - mean, b = x[2 * it + 1] - mean; q += a * a + b * b; }
  const float rstd = rsqrtf(wave_sum(q) / H + eps);
  float d = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it)
    if (it < nit) {
      const int c = it * 128 + lane * 2;
      d += ((x[2 * it] - mean) * rstd * gamma[c] + beta[c]) * w2[c] + ((x[2 * it + 1] - mean) * rstd * gamma[c + 1] + beta[c + 1]) * w2[c + 1];
    }
  d = wave_sum(d);
  if (lane == 0) logit[row] = d + b2[0];
}

// backward: dZ[m,:] = relu'(Y) * LNbwd(dlogit[m] * w2); dgamma, dbeta, dw2, db2 block-reduced atomics
template <typename T, int NIT>
__global__ __launch_bounds__(256) void lndot_bwd_kernel(int M, int H, const T* Y, const float* gamma, const float* beta, float eps,
                                                        const float* w2, const float* dlogit, T* dZ,
                                                        float* dgamma, float* dbeta, float* dw2, float* db2, float* part) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [3][4][H] + [4]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  constexpr int nit = NIT;
  float ag[2 * NIT], ab[2 * NIT], aw[2 * NIT];
#pragma unroll
  for (int i = 0; i < 2 * NIT; ++i) { ag[i] = 0.f; ab[i] = 0.f; aw[i] = 0.f; }
  float adb = 0.f;
  const int row0 = (blockIdx.x * 4 + wid) * LNB_ROWS;
  for (int rr = 0; rr < LNB_ROWS; ++rr) {
    const int row = row0 + rr;
    if (row >= M) break;
    float x[2 * NIT];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      if (it < nit) { ld2<T>(Y + (long long)row * H + it * 128 + lane * 2, x[2 * it], x[2 * it + 1]); s += x[2 * it] + x[2 * it + 1]; }
    const float mean = wave_sum(s) / H;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      if (it < nit) { float a = x[2 * it] - mean, b = x[2 * it + 1] - mean; q += a * a + b * b; }
    const float rstd = rsqrtf(wave_sum(q) / H + eps);
    const float dl = dlogit[row];
    adb += dl;
    float g[2 * NIT], xh[2 * NIT];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      if (it < nit) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int c = it * 128 + lane * 2 + e, i = 2 * it + e;
          xh[i] = (x[i] - mean) * rstd;
          const float dln = dl * w2[c];                 // grad wrt LN output
          aw[i] += dl * (xh[i] * gamma[c] + beta[c]);
          ag[i] += dln * xh[i]; ab[i] += dln;
          g[i] = dln * gamma[c];
          s1 += g[i]; s2 += g[i] * xh[i];
        }
      }
    const float m1 = wave_sum(s1) / H, m2 = wave_sum(s2) / H;
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      if (it < nit) {
        const float da = x[2 * it] > 0.f ? rstd * (g[2 * it] - m1 - xh[2 * it] * m2) : 0.f;
        const float dbv = x[2 * it + 1] > 0.f ? rstd * (g[2 * it + 1] - m1 - xh[2 * it + 1] * m2) : 0.f;
        st2<T>(dZ + (long long)row * H + it * 128 + lane * 2, da, dbv);
      }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it)
    if (it < nit) {
      const int c = it * 128 + lane * 2;
      red[(0 * 4 + wid) * H + c] = ag[2 * it]; red[(0 * 4 + wid) * H + c + 1] = ag[2 * it + 1];
      red[(1 * 4 + wid) * H + c] = ab[2 * it]; red[(1 * 4 + wid) * H + c + 1] = ab[2 * it + 1];
      red[(2 * 4 + wid) * H + c] = aw[2 * it]; red[(2 * 4 + wid) * H + c + 1] = aw[2 * it + 1];
    }
  if (lane == 0) red[12 * H + wid] = adb;     // every lane of the wave holds the same adb
  __syncthreads();
  if (part) {          // round 6: this block's sums to its own row [dgamma | dbeta | dw2 | db2] (3 H + 1 floats); the caller adds the rows up in block order
    float* row = part + (long long)blockIdx.x * (3 * H + 1);
    for (int c = threadIdx.x; c < H; c += 256) {
      row[c] = red[c] + red[H + c] + red[2 * H + c] + red[3 * H + c];
      row[H + c] = red[4 * H + c] + red[5 * H + c] + red[6 * H + c] + red[7 * H + c];
      row[2 * H + c] = red[8 * H + c] + red[9 * H + c] + red[10 * H + c] + red[11 * H + c];
    }
    if (threadIdx.x == 0) row[3 * H] = red[12 * H] + red[12 * H + 1] + red[12 * H + 2] + red[12 * H + 3];
    return;
  }
  for (int c = threadIdx.x; c < H; c += 256) {
    atomicAdd(dgamma + c, red[c] + red[H + c] + red[2 * H + c] + red[3 * H + c]);
    atomicAdd(dbeta + c, red[4 * H + c] + red[5 * H + c] + red[6 * H + c] + red[7 * H + c]);
    atomicAdd(dw2 + c, red[8 * H + c] + red[9 * H + c] + red[10 * H + c] + red[11 * H + c]);
  }
  if (threadIdx.x == 0) atomicAdd(db2, red[12 * H] + red[12 * H + 1] + red[12 * H + 2] + red[12 * H + 3]);
}

// ============================================================================================
static inline bool okH(int H) { return H >= 128 && H <= 128 * MAXIT && (H % 128) == 0; }
// CALL is written once over the type name TY
#define DISPATCH_T(dtype, ...)                                                        \
  do {                                                                                \
    if ((dtype) == DT_BF16) { typedef bf16 TY; __VA_ARGS__; }                          \
    else if ((dtype) == DT_F16) { typedef f16 TY; __VA_ARGS__; }                       \
    else { typedef float TY; __VA_ARGS__; }                                           \
  } while (0)
// H in {128, 256, 384, 768} = the S / M / B / L family (run_r2r_kdl_valid.sh:85-94)
#define DISPATCH_NIT(dtype, H, F)                                                                     \
  do {                                                                                                \
    if ((dtype) == DT_BF16) {                                                                         \
      if ((H) == 128) F(bf16, 1); else if ((H) == 256) F(bf16, 2); else if ((H) == 384) F(bf16, 3);   \
      else if ((H) == 768) F(bf16, 6); else return MAGIC_ERR_UNSUPPORTED;                             \
    } else if ((dtype) == DT_F16) {                                                                   \
      if ((H) == 128) F(f16, 1); else if ((H) == 256) F(f16, 2); else if ((H) == 384) F(f16, 3);      \
      else if ((H) == 768) F(f16, 6); else return MAGIC_ERR_UNSUPPORTED;                              \
    } else {                                                                                          \
      if ((H) == 128) F(float, 1); else if ((H) == 256) F(float, 2); else if ((H) == 384) F(float, 3); \
      else if ((H) == 768) F(float, 6); else return MAGIC_ERR_UNSUPPORTED;                            \
    }                                                                                                 \
  } while (0)

extern "C" int magic_ln_fwd(int dtype, int M, int H, const void* in0, const void* in1,
                            const void* tab0, const int* idx0, int mod0, int off0,
                            const void* tab1, const int* idx1, int mod1, int off1,
                            const void* tab2, const int* idx2, int mod2, int off2,
                            const float* gamma, const float* beta, float eps, void* out, float* rstd, int do_ln,
                            const void* drop_seed, float drop_p, unsigned site_in0, unsigned site_out, void* out_drop, void* stream) {
  if (M <= 0 || !okH(H) || !out) return MAGIC_ERR_ARG;
  if (do_ln && (!gamma || !beta)) return MAGIC_ERR_ARG;
  if (!drop_args_ok(drop_seed, drop_p) || (long long)M * H > 0xFFFFFFFFll) return MAGIC_ERR_ARG;
  const bool don = drop_p > 0.f;
  if (don && site_out && (!out_drop || !do_ln)) return MAGIC_ERR_ARG;
  if (don && site_in0 && !in0) return MAGIC_ERR_ARG;
  DropDesc din{(don && site_in0) ? (const unsigned*)drop_seed : nullptr, site_in0, drop_p};
  DropDesc dout{(don && site_out) ? (const unsigned*)drop_seed : nullptr, site_out, drop_p};
  TabRef t0{tab0, idx0, mod0, off0}, t1{tab1, idx1, mod1, off1}, t2{tab2, idx2, mod2, off2};
  LnfParams p{M, in0, in1, t0, t1, t2, gamma, beta, eps, out, rstd, do_ln, din, dout, out_drop};
  if (group_record(KIND_LNF, dtype, H, &p, sizeof(p))) return MAGIC_OK;
  return launch_lnf(dtype, H, &p, nullptr, (hipStream_t)stream);
}

int launch_lnf(int dtype, int H, const void* pa, const void* pb, hipStream_t st) {
  const LnfParams& a = *(const LnfParams*)pa;
  const int na = (a.M + 3) / 4;
  dim3 block(256);
  if (!pb) {
    dim3 grid(na);
#define LNF(TY, NIT) hipLaunchKernelGGL((ln_fwd_kernel<TY, NIT>), grid, block, 0, st, a)
    DISPATCH_NIT(dtype, H, LNF);
#undef LNF
  } else {
    const LnfParams& b = *(const LnfParams*)pb;
    dim3 grid(na + (b.M + 3) / 4);
#define LNF2(TY, NIT) hipLaunchKernelGGL((ln_fwd_pair_kernel<TY, NIT>), grid, block, 0, st, a, b, na)
    DISPATCH_NIT(dtype, H, LNF2);
#undef LNF2
  }
  return launch_status();
}

static inline int lnb_waves(int nit) { return nit == 1 ? 16 : nit == 2 ? 8 : 4; }      // waves per block (LDS: (2 NW + 9) H floats; registers)
// H = 768 without table gradients on the few hundred rows of a navigator step (the LayerNorms inside the transformer blocks): `lean` form --
// MAGIC_LNB_LEAN = "<waves per block><rows per wave>" picks the launch shape of that form.  Measured in a dependent chain over cold operands
// (profiles/micro/nav_row_probe.py, M = 624): 4 waves x 2 rows 11.0 us, 4 x 1 7.6, 8 x 1 7.7 (half the partial rows of 4 x 1: the default),
// 8 x 2 12.1 -- a wave's second row is a second memory round trip on the launch's critical path
static int lnb_lean_cfg() {
  static int cfg = -1;
  if (cfg < 0) { const char* e = getenv("MAGIC_LNB_LEAN"); cfg = e ? atoi(e) : 81; if (cfg != 42 && cfg != 41 && cfg != 81 && cfg != 82) cfg = 81; }
  return cfg;
}
static inline bool lnb_lean(const LnbParams& p, int nit) { return nit == 6 && !p.d0 && !p.d1 && !p.d2; }
static inline int lnb_blocks(const LnbParams& p, int nit, bool single = false) {
  int rpi = nit <= 2 ? 4 : 2;                 // rows per wave per iteration (ln_bwd_body::RPI)
  int nw = lnb_waves(nit);
  if (single && lnb_lean(p, nit)) { nw = lnb_lean_cfg() / 10; rpi = lnb_lean_cfg() % 10; }
  const int nb = (p.M + nw * rpi - 1) / (nw * rpi);
  // with in-kernel gamma/beta grads every block ends in 2H same-address atomics -> cap the grid; without them one row group per wave
  const int cap = p.dgamma ? 512 : 4096;
  return nb > cap ? cap : nb;
}

extern "C" int magic_ln_bwd(int dtype, int M, int H, const void* dy, const void* y, const float* gamma, const float* beta,
                            const float* rstd, void* dx, float* dgamma, float* dbeta,
                            const int* idx0, int mod0, int off0, float* d0, int small0,
                            const int* idx1, int mod1, int off1, float* d1, int small1,
                            const int* idx2, int mod2, int off2, float* d2, int small2,
                            int do_ln, const void* drop_seed, float drop_p, unsigned site_dy, unsigned site_dx, void* dxm, int hot0, int pg_partial,
                            void* stream) {
  if (M <= 0 || !okH(H) || !dy || (H != 128 && H != 256 && H != 384 && H != 768)) return MAGIC_ERR_ARG;
  if (pg_partial && (!do_ln || !dgamma)) return MAGIC_ERR_ARG;
  if (hot0 >= 0 && (!idx0 || !d0)) return MAGIC_ERR_ARG;
  if (!drop_args_ok(drop_seed, drop_p) || (long long)M * H > 0xFFFFFFFFll) return MAGIC_ERR_ARG;
  const bool don = drop_p > 0.f;
  if (don && site_dx && !dxm) return MAGIC_ERR_ARG;
  if (do_ln && (!y || !gamma || !beta || !rstd)) return MAGIC_ERR_ARG;
  if ((dgamma == nullptr) != (dbeta == nullptr)) return MAGIC_ERR_ARG;
  if ((small0 && !idx0) || (small1 && !idx1) || (small2 && !idx2)) return MAGIC_ERR_ARG;
  LnbParams p{M, dy, y, gamma, beta, rstd, dx, dgamma, dbeta, TabRef{d0, idx0, mod0, off0}, d0, small0, TabRef{d1, idx1, mod1, off1}, d1, small1,
              TabRef{d2, idx2, mod2, off2}, d2, small2, do_ln,
              DropDesc{(don && site_dy) ? (const unsigned*)drop_seed : nullptr, site_dy, drop_p},
              (don && site_dx) ? dxm : nullptr, DropDesc{(don && site_dx) ? (const unsigned*)drop_seed : nullptr, site_dx, drop_p}, hot0, pg_partial ? 1 : 0};
  const int nit = H / 128;
  // (partial rows are sized for a single launch: a paired launch takes the 4-wave x 2-row shape, 8 rows per workgroup -- fine when the single form has 8 too)
  if (pg_partial && group_state().active && lnb_lean(p, nit) && (lnb_lean_cfg() / 10) * (lnb_lean_cfg() % 10) != 8) return MAGIC_ERR_ARG;
  if (group_record(KIND_LNB, dtype, nit, &p, sizeof(p))) return MAGIC_OK;
  return launch_lnb(dtype, nit, &p, nullptr, (hipStream_t)stream);
}

// dst_j[c] += sum over b < nblk_j of part_j[b * stride_j + c], c < len_j: the finisher of every "each workgroup stores its partial sums in its own
// row" epilogue (LayerNorm / position-embedding parameter gradients at the wide model sizes: the same-address fp32 atomics they replace were the
// launches' whole cost -- ln_bwd 17.6 us for 608 x 768, 4 us without).  Sums run in block order: reproducible.  One launch holds a destination once.
#define CSV_MAX 96
struct ColsumVJobs { const float* part[CSV_MAX]; float* dst[CSV_MAX]; int nblk[CSV_MAX]; int len[CSV_MAX]; int stride[CSV_MAX]; int n; };
// (round 6: a workgroup owns 64 columns and its four waves a quarter of the rows each, eight loads in flight per lane, combined through LDS in wave order -- one
// thread per column walking all rows four at a time was a chain of nblk / 4 dependent round trips: 10.9 us for 80 rows x 2.3 k columns, 23.7 us per launch at H = 768)
__global__ __launch_bounds__(256) void colsum_v_kernel(ColsumVJobs js) {
  __shared__ float red[4][64];
  const int j = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = blockIdx.x * 64 + lane;
  const int len = js.len[j];
  if ((int)blockIdx.x * 64 >= len) return;                 // (block-uniform)
  const int nb = js.nblk[j], st = js.stride[j];
  const int per = (nb + 3) / 4, b0 = w * per, b1 = min(nb, b0 + per);
  float s[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) s[q] = 0.f;
  if (c < len) {
    const float* pt = js.part[j] + c;
    int b = b0;
    for (; b + 7 < b1; b += 8) {
#pragma unroll
      for (int q = 0; q < 8; ++q) s[q] += pt[(long long)(b + q) * st];
    }
    for (; b < b1; ++b) s[0] += pt[(long long)b * st];
  }
  red[w][lane] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (w == 0 && c < len) js.dst[j][c] += (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}
extern "C" int magic_colsum_add_v(int n, const float* const* parts, float* const* dsts, const int* nblks, const int* lens, const int* strides, void* stream) {
  if (n <= 0 || n > CSV_MAX || !parts || !dsts || !nblks || !lens || !strides) return MAGIC_ERR_ARG;
  ColsumVJobs js;
  js.n = n;
  int mx = 0;
  for (int i = 0; i < n; ++i) {
    if (!parts[i] || !dsts[i] || nblks[i] <= 0 || lens[i] <= 0 || strides[i] < lens[i]) return MAGIC_ERR_ARG;
    js.part[i] = parts[i]; js.dst[i] = dsts[i]; js.nblk[i] = nblks[i]; js.len[i] = lens[i]; js.stride[i] = strides[i];
    mx = lens[i] > mx ? lens[i] : mx;
  }
  hipLaunchKernelGGL(colsum_v_kernel, dim3((mx + 63) / 64, n), dim3(256), 0, (hipStream_t)stream, js);
  return launch_status();
}

// workgroups magic_ln_bwd launches for M rows with in-kernel gamma / beta gradients: the row count of its PARTIAL buffers (pg_partial).  has_tables: the
// launch also carries table gradients (d0 / d1 / d2) -- those never take the lean shape (MAGIC_LNB_LEAN), whatever it is set to
extern "C" int magic_ln_bwd_blocks(int M, int H, int has_tables) {
  if (M <= 0 || (H != 128 && H != 256 && H != 384 && H != 768)) return MAGIC_ERR_ARG;
  LnbParams p{};
  p.M = M; p.dgamma = (float*)1;
  if (has_tables) p.d0 = (float*)1;
  return lnb_blocks(p, H / 128, true);
}

int launch_lnb(int dtype, int nit, const void* pa, const void* pb, hipStream_t st) {
  const LnbParams& a = *(const LnbParams*)pa;
  const int H = nit * 128;
  const int nw = lnb_waves(nit);
  dim3 block(nw * 64);
  const int nA = lnb_blocks(a, nit, !pb);
  // launches without table gradients (the LayerNorms inside the blocks) take the plain instantiation at the wide sizes
  const bool tab = a.d0 || a.d1 || a.d2 || (pb && (((const LnbParams*)pb)->d0 || ((const LnbParams*)pb)->d1 || ((const LnbParams*)pb)->d2));
  if (!pb && lnb_lean(a, nit) && lnb_lean_cfg() != 42) {         // single lean launch in another shape
    const int cfg = lnb_lean_cfg();
#define LNBL(TY)                                                                                                                                      \
    do {                                                                                                                                              \
      if (cfg == 41) hipLaunchKernelGGL((ln_bwd_kernel<TY, 6, 4, false, 1>), dim3(nA), dim3(256), (size_t)8 * H * sizeof(float), st, a);             \
      else if (cfg == 81) hipLaunchKernelGGL((ln_bwd_kernel<TY, 6, 8, false, 1>), dim3(nA), dim3(512), (size_t)16 * H * sizeof(float), st, a);       \
      else hipLaunchKernelGGL((ln_bwd_kernel<TY, 6, 8, false, 2>), dim3(nA), dim3(512), (size_t)16 * H * sizeof(float), st, a);                      \
    } while (0)
    if (dtype == DT_BF16) LNBL(bf16); else if (dtype == DT_F16) LNBL(f16); else LNBL(float);
#undef LNBL
    return launch_status();
  }
  if (pb && lnb_lean(a, nit) && lnb_lean(*(const LnbParams*)pb, nit) && lnb_lean_cfg() == 81) {
    // a PAIR of lean launches (the twin LayerNorms of a navigator step's two cross-modal encoders) in the lean shape too: 8 waves x 1 row, the
    // same 8 rows per workgroup as the 4 x 2 form, so the partial-row buffers keep their size
    const LnbParams& b = *(const LnbParams*)pb;
    const int nLA = lnb_blocks(a, nit, true), nLB = lnb_blocks(b, nit, true);
#define LNBP(TY) hipLaunchKernelGGL((ln_bwd_pair_kernel<TY, 6, 8, false, 1>), dim3(nLA + nLB), dim3(512), (size_t)16 * H * sizeof(float), st, a, b, nLA)
    if (dtype == DT_BF16) LNBP(bf16); else if (dtype == DT_F16) LNBP(f16); else LNBP(float);
#undef LNBP
    return launch_status();
  }
  const size_t shm = (size_t)(2 * nw + ((nit >= 3 && !tab) ? 0 : 9)) * H * sizeof(float);
#define LNB2(TY, NIT, NW, TAB)                                                                          \
  do {                                                                                                  \
    if (!pb) hipLaunchKernelGGL((ln_bwd_kernel<TY, NIT, NW, TAB>), dim3(nA), block, shm, st, a);        \
    else {                                                                                              \
      const LnbParams& b = *(const LnbParams*)pb;                                                       \
      hipLaunchKernelGGL((ln_bwd_pair_kernel<TY, NIT, NW, TAB>), dim3(nA + lnb_blocks(b, nit)), block, shm, st, a, b, nA); \
    }                                                                                                   \
  } while (0)
#define LNB1(TY, NIT, NW) do { if (NIT >= 3 && !tab) LNB2(TY, NIT, NW, false); else LNB2(TY, NIT, NW, true); } while (0)
  if (dtype == DT_BF16) { if (nit == 1) LNB1(bf16, 1, 16); else if (nit == 2) LNB1(bf16, 2, 8); else if (nit == 3) LNB1(bf16, 3, 4); else LNB1(bf16, 6, 4); }
  else if (dtype == DT_F16) { if (nit == 1) LNB1(f16, 1, 16); else if (nit == 2) LNB1(f16, 2, 8); else if (nit == 3) LNB1(f16, 3, 4); else LNB1(f16, 6, 4); }
  else { if (nit == 1) LNB1(float, 1, 16); else if (nit == 2) LNB1(float, 2, 8); else if (nit == 3) LNB1(float, 3, 4); else LNB1(float, 6, 4); }
#undef LNB2
#undef LNB1
  return launch_status();
}

// LayerNorm backward of a head's transform in one launch with its neighbours: dy in fp32 (no cast launch), dx = LN'(dy) x act'(pre) (no activation-
// derivative launch); gamma / beta gradients as magic_ln_bwd's (atomics).  act: 1 gelu (erf), 2 relu.
extern "C" int magic_ln_bwd_tail(int dtype, int M, int H, const float* dy32, const void* y, const float* gamma, const float* beta, const float* rstd,
                                 const void* act_pre, int act, void* dx, float* dgamma, float* dbeta, void* stream) {
  if (M <= 0 || (H != 128 && H != 256 && H != 384 && H != 768) || !dy32 || !y || !gamma || !beta || !rstd || !act_pre || !dx) return MAGIC_ERR_ARG;
  // act: bits 0..7 the activation (1 gelu, 2 relu); bits 8.. = S > 0: dy32 is S slabs of M x H to be added in slab order (magic_gemm with splitk = -S)
  const int nslab = (act >> 8) > 0 ? (act >> 8) : 1;
  const int partial = (act & 0x40) ? 1 : 0;          // bit 6: dgamma / dbeta are PARTIAL buffers [magic_ln_bwd_blocks(M, H, 0)][H]
  act &= 0x3F;
  if (nslab > 256) return MAGIC_ERR_ARG;
  if ((act != 1 && act != 2) || (dgamma == nullptr) != (dbeta == nullptr) || (long long)M * H > 0xFFFFFFFFll) return MAGIC_ERR_ARG;
  if (((uintptr_t)dy32 & 7) || !dtype_ok(dtype)) return MAGIC_ERR_ARG;
  LnbParams p{};
  p.M = M; p.y = y; p.gamma = gamma; p.beta = beta; p.rstd = rstd; p.dx = dx; p.dgamma = dgamma; p.dbeta = dbeta; p.do_ln = 1; p.hot0 = -1;
  p.dy32 = dy32; p.act_pre = act_pre; p.act = act; p.nslab = nslab; p.pg_partial = (partial && dgamma) ? 1 : 0;
  const int nit = H / 128, nw = lnb_waves(nit);
  const int nb = lnb_blocks(p, nit);
  const size_t shm = (size_t)(2 * nw) * H * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
#define LNT(TY, NIT, NW) hipLaunchKernelGGL((ln_bwd_tail_kernel<TY, NIT, NW>), dim3(nb), dim3(NW * 64), shm, st, p)
#define LNTD(TY) do { if (nit == 1) LNT(TY, 1, 16); else if (nit == 2) LNT(TY, 2, 8); else if (nit == 3) LNT(TY, 3, 4); else LNT(TY, 6, 4); } while (0)
  if (dtype == DT_BF16) LNTD(bf16); else if (dtype == DT_F16) LNTD(f16); else LNTD(float);
#undef LNTD
#undef LNT
  return launch_status();
}

// public mirror of magic_ln_bwd's arguments for the second problem of magic_embed_in_bwd (the text embedding rows)
struct magic_ln_bwd_in {
  int M, do_ln; const void* dy; const void* y; const float* gamma; const float* beta; const float* rstd; void* dx; float* dgamma; float* dbeta;
  const int* idx[3]; int mod[3]; int off[3]; float* d[3]; int small[3];
  const unsigned* drop_seed; float drop_p; unsigned site_dy, site_dx; int hot0; void* dxm;
  int partial;               // round 6: != 0 -> dgamma / dbeta are PARTIAL buffers [magic_ln_bwd_blocks(M, H, tables)][H] (magic_ln_bwd's `partial`)
};
extern "C" int magic_embed_in_bwd_supported(int H, int Kin) { return (H == 128 || H == 256) && Kin >= 1 && Kin <= PIB_KMAX; }
// workgroups of the panorama half of magic_embed_in_bwd for M rows beside nb_text workgroups of the text half: one row per wave and iteration; with the atomic
// epilogue every block ends in (11 + Kin) H same-address-class atomics and those, not the rows, are the launch's time (256 blocks 51 us, 96 blocks 30 us, 32
// blocks 40 us: profiles/micro/embed_bwd_probe.py) -> 80 blocks (MAGIC_PIB_BLOCKS); with partial rows (round 6) 128 blocks (MAGIC_PIB_BLOCKS_PART; 0: bounded by the chip alone -- 24.7-26.9 us
// for the launch at 80 / 128 / ~200 blocks, the column-sum launch behind it 7.0 / 6.8 / 7.9 us)
static int pib_blocks(int M, int H, int nb_text, bool partial) {
  const int nit = H / 128, nw = lnb_waves(nit);
  int na = (M + 2 * nw - 1) / (2 * nw);               // (PIB_RPI = 2 rows per wave and iteration)
  static int ncu = 0;
  if (!ncu) { int dev = 0; hipDeviceProp_t pr; ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256; }
  int room = ncu - nb_text > 64 ? ncu - nb_text : 64;
  static int cap = -1, cap_part = -1;
  if (cap < 0) { const char* e = getenv("MAGIC_PIB_BLOCKS"); cap = e ? atoi(e) : 80; }
  if (cap_part < 0) { const char* e = getenv("MAGIC_PIB_BLOCKS_PART"); cap_part = e ? atoi(e) : 128; }
  const int c = partial ? cap_part : cap;
  if (c > 0 && room > c) room = c;
  return na > room ? room : na;
}
// the row count of the partial buffer a caller of magic_embed_in_bwd must provide (and the number of rows magic_colsum_add_v then adds up)
extern "C" int magic_embed_in_bwd_blocks(int M, int H, int nb_text, int partial) {
  if (M <= 0 || (H != 128 && H != 256) || nb_text < 0) return MAGIC_ERR_ARG;
  return pib_blocks(M, H, nb_text, partial != 0);
}
extern "C" int magic_embed_in_bwd(int dtype, int H, const magic_pano_in_bwd* pa, const magic_ln_bwd_in* tx,
                                  int n_cs, const float* const* cs_parts, float* const* cs_dsts, const int* cs_nblks, void* stream) {
  if (!pa || !dtype_ok(dtype)) return MAGIC_ERR_ARG;
  if (n_cs < 0 || n_cs > CSJ_MAX || (n_cs > 0 && (!cs_parts || !cs_dsts || !cs_nblks))) return MAGIC_ERR_ARG;
  ColsumJobs cs;
  cs.n = n_cs;
  for (int i = 0; i < n_cs; ++i) {
    if (!cs_parts[i] || !cs_dsts[i] || cs_nblks[i] <= 0) return MAGIC_ERR_ARG;
    cs.part[i] = cs_parts[i]; cs.dst[i] = cs_dsts[i]; cs.nblk[i] = cs_nblks[i];
  }
  const magic_pano_in_bwd& a = *pa;
  if (!magic_embed_in_bwd_supported(H, a.Kin)) return MAGIC_ERR_UNSUPPORTED;
  if (a.M <= 0 || (long long)a.M * H > 0xFFFFFFFFll || !drop_args_ok(a.ddy.seed, a.ddy.p)) return MAGIC_ERR_ARG;
  const void* req[] = {a.dy, a.X0, a.rstd3, a.g3, a.b3, a.dg3, a.db3, a.nav_idx, a.d_nav, a.d_tok, a.A1, a.rstd1, a.g1, a.b1, a.dg1, a.db1, a.dP0,
                       a.A2, a.rstd2, a.g2, a.b2, a.dg2, a.db2, a.loc, a.dW, a.dbl};
  for (const void* q : req)
    if (!q) return MAGIC_ERR_ARG;
  const int nit = H / 128, nw = lnb_waves(nit);
  LnbParams b{};
  int nb = 0;
  if (tx) {
    const magic_ln_bwd_in& t = *tx;
    if (t.M <= 0 || !t.dy || (long long)t.M * H > 0xFFFFFFFFll || !drop_args_ok(t.drop_seed, t.drop_p)) return MAGIC_ERR_ARG;
    if (t.hot0 >= 0 && (!t.idx[0] || !t.d[0])) return MAGIC_ERR_ARG;
    const bool don = t.drop_p > 0.f;
    if (don && t.site_dx && !t.dxm) return MAGIC_ERR_ARG;
    if (t.do_ln && (!t.y || !t.gamma || !t.beta || !t.rstd)) return MAGIC_ERR_ARG;
    if ((t.dgamma == nullptr) != (t.dbeta == nullptr)) return MAGIC_ERR_ARG;
    for (int k = 0; k < 3; ++k)
      if (t.small[k] && !t.idx[k]) return MAGIC_ERR_ARG;
    b = LnbParams{t.M, t.dy, t.y, t.gamma, t.beta, t.rstd, t.dx, t.dgamma, t.dbeta,
                  TabRef{t.d[0], t.idx[0], t.mod[0], t.off[0]}, t.d[0], t.small[0], TabRef{t.d[1], t.idx[1], t.mod[1], t.off[1]}, t.d[1], t.small[1],
                  TabRef{t.d[2], t.idx[2], t.mod[2], t.off[2]}, t.d[2], t.small[2], t.do_ln,
                  DropDesc{(don && t.site_dy) ? t.drop_seed : nullptr, t.site_dy, t.drop_p},
                  (don && t.site_dx) ? t.dxm : nullptr, DropDesc{(don && t.site_dx) ? t.drop_seed : nullptr, t.site_dx, t.drop_p}, t.hot0};
    b.pg_partial = (t.partial && t.dgamma) ? 1 : 0;
    nb = lnb_blocks(b, nit);
  }
  const int na = pib_blocks(a.M, H, nb, a.part != nullptr);
  if (a.part && (a.pad0_ < na || a.pad1_ < (11 + a.Kin) * H || ((uintptr_t)a.part & 3))) return MAGIC_ERR_ARG;      // (rows, stride of the partial buffer)
  const size_t sa = (size_t)nw * 4 * H * sizeof(float), sb = (size_t)(2 * nw + 9) * H * sizeof(float), shm = sa > sb ? sa : sb;
  dim3 grid(na + nb + n_cs), block(nw * 64);
  hipStream_t st = (hipStream_t)stream;
#define EIB(TY, NIT, NW) hipLaunchKernelGGL((embed_in_bwd_kernel<TY, NIT, NW>), grid, block, shm, st, a, b, nb, cs)
  if (dtype == DT_BF16) { if (nit == 1) EIB(bf16, 1, 16); else EIB(bf16, 2, 8); }
  else if (dtype == DT_F16) { if (nit == 1) EIB(f16, 1, 16); else EIB(f16, 2, 8); }
  else { if (nit == 1) EIB(float, 1, 16); else EIB(float, 2, 8); }
#undef EIB
  return launch_status();
}

extern "C" int magic_ln_pgrad(int dtype, int M, int H, const void* dy, const void* y, const float* gamma, const float* beta,
                              float* dgamma, float* dbeta, void* stream) {
  if (M <= 0 || !okH(H) || !dy || !y || !gamma || !beta || !dgamma || !dbeta) return MAGIC_ERR_ARG;
  dim3 grid((M + PG_ROWS - 1) / PG_ROWS), block(256);
  size_t shm = (size_t)8 * H * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
#define LPG(TY, NIT) hipLaunchKernelGGL((ln_pgrad_kernel<TY, NIT>), grid, block, shm, st, M, (const TY*)dy, (const TY*)y, gamma, beta, dgamma, dbeta)
  DISPATCH_NIT(dtype, H, LPG);
#undef LPG
  return launch_status();
}

extern "C" int magic_smallk_ln_fwd(int dtype, int M, int H, int Kin, const float* x, const float* W, const float* b,
                                   const float* gamma, const float* beta, float eps, void* out, float* rstd, void* stream) {
  if (M <= 0 || !okH(H) || Kin <= 0 || Kin > 16) return MAGIC_ERR_ARG;
  dim3 grid((M + 4 * SKF_ROWS - 1) / (4 * SKF_ROWS)), block(256);
  hipStream_t st = (hipStream_t)stream;
#define SKF(TY, NIT) hipLaunchKernelGGL((smallk_ln_fwd_kernel<TY, NIT>), grid, block, 0, st, M, H, Kin, x, W, b, gamma, beta, eps, (TY*)out, rstd)
  DISPATCH_NIT(dtype, H, SKF);
#undef SKF
  return launch_status();
}


// public mirror of LnfParams for the second problem of magic_embed_in_fwd (the text embedding rows): same meaning as magic_ln_fwd's arguments
struct magic_ln_in {
  int M, do_ln; const void* in0; const void* in1;
  const void* tab[3]; const int* idx[3]; int mod[3]; int off[3];
  const float* gamma; const float* beta; float eps; int pad_; void* out; float* rstd;
  const unsigned* drop_seed; float drop_p; unsigned site_in0, site_out, pad2_; void* out_drop;
};
extern "C" int magic_embed_in_fwd(int dtype, int H, const magic_pano_in* pa, const magic_ln_in* tx, void* stream) {
  if (!okH(H) || !pa) return MAGIC_ERR_ARG;
  const magic_pano_in& a = *pa;
  if (a.M <= 0 || a.Kin <= 0 || a.Kin > 16 || (long long)a.M * H > 0xFFFFFFFFll) return MAGIC_ERR_ARG;
  const void* req[] = {a.P0, a.g1, a.b1, a.A1, a.rstd1, a.loc, a.W, a.b, a.g2, a.b2, a.A2, a.rstd2, a.nav_tab, a.nav_idx, a.tok_tab, a.g3, a.b3, a.X0, a.rstd3};
  for (const void* q : req)
    if (!q) return MAGIC_ERR_ARG;
  if (!drop_args_ok(a.dout.seed, a.dout.p) || (a.dout.seed && a.dout.p > 0.f && !a.X0d)) return MAGIC_ERR_ARG;
  LnfParams b{};
  int nb = 0;
  if (tx) {
    const magic_ln_in& t = *tx;
    if (t.M <= 0 || !t.out || (t.do_ln && (!t.gamma || !t.beta))) return MAGIC_ERR_ARG;
    if (!drop_args_ok(t.drop_seed, t.drop_p) || (long long)t.M * H > 0xFFFFFFFFll) return MAGIC_ERR_ARG;
    const bool don = t.drop_p > 0.f;
    if (don && t.site_out && (!t.out_drop || !t.do_ln)) return MAGIC_ERR_ARG;
    if (don && t.site_in0 && !t.in0) return MAGIC_ERR_ARG;
    DropDesc din{(don && t.site_in0) ? t.drop_seed : nullptr, t.site_in0, t.drop_p};
    DropDesc dout{(don && t.site_out) ? t.drop_seed : nullptr, t.site_out, t.drop_p};
    TabRef t0{t.tab[0], t.idx[0], t.mod[0], t.off[0]}, t1{t.tab[1], t.idx[1], t.mod[1], t.off[1]}, t2{t.tab[2], t.idx[2], t.mod[2], t.off[2]};
    b = LnfParams{t.M, t.in0, t.in1, t0, t1, t2, t.gamma, t.beta, t.eps, t.out, t.rstd, t.do_ln, din, dout, t.out_drop};
    nb = (t.M + 3) / 4;
  }
  const int na = (a.M + 4 * SKF_ROWS - 1) / (4 * SKF_ROWS);
  dim3 grid(na + nb), block(256);
  hipStream_t st = (hipStream_t)stream;
#define EIF(TY, NIT) hipLaunchKernelGGL((embed_in_fwd_kernel<TY, NIT>), grid, block, 0, st, a, b, na)
  DISPATCH_NIT(dtype, H, EIF);
#undef EIF
  return launch_status();
}

extern "C" int magic_node_in_fwd(int dtype, int H, int n, const magic_node_in* d, void* stream) {
  if (!okH(H) || n < 1 || n > 2 || !d) return MAGIC_ERR_ARG;
  int nb[2] = {0, 0};
  for (int i = 0; i < n; ++i) {
    const magic_node_in& p = d[i];
    if (p.M <= 0 || p.Kin <= 0 || p.Kin > 16 || !p.x || !p.W || !p.b || !p.gamma || !p.beta || !p.A || !p.out) return MAGIC_ERR_ARG;
    if (p.ptr1 && (!p.src1 || !p.idx1)) return MAGIC_ERR_ARG;
    if (p.ptr2 && (!p.src2 || !p.idx2 || !p.ptr1)) return MAGIC_ERR_ARG;
    if (p.tab && !p.tab_idx) return MAGIC_ERR_ARG;
    nb[i] = (p.M + 4 * SKF_ROWS - 1) / (4 * SKF_ROWS);
  }
  const magic_node_in& a = d[0];
  const magic_node_in& b = d[n - 1];
  dim3 grid(nb[0] + nb[1]), block(256);
  hipStream_t st = (hipStream_t)stream;
#define NIF(TY, NIT) hipLaunchKernelGGL((node_in_fwd_kernel<TY, NIT>), grid, block, 0, st, a, b, nb[0])
  DISPATCH_NIT(dtype, H, NIF);
#undef NIF
  return launch_status();
}

// rows per workgroup: 64 rows x 16 waves at H = 128 with many rows; 8 rows x 4 waves for the wide models on few rows (the navigator's
// M ~ 600 steps at H = 768 ran on 19 workgroups: 58 us per launch); 32 rows x 4 waves otherwise
static inline int skb_rows(int H, int Mmax) { return (H == 128 && Mmax >= 4096) ? 64 : (H >= 384 && Mmax <= 4096) ? 8 : 32; }
// workgroups of a problem of M rows in a launch whose largest problem has Mmax rows (the pair entry picks ONE tile shape): rows of `part`
extern "C" int magic_smallk_ln_bwd_blocks(int M, int H, int Mmax) {
  return (M <= 0 || Mmax < M || !okH(H)) ? MAGIC_ERR_ARG : (M + skb_rows(H, Mmax) - 1) / skb_rows(H, Mmax);
}

static int launch_skb(int dtype, int H, const SkbParams& a, const SkbParams* b, hipStream_t st) {
  const int Mmax = b ? (a.M > b->M ? a.M : b->M) : a.M;
  const int rows = skb_rows(H, Mmax), nw = rows == 64 ? 16 : 4;
  const bool wide = rows == 64, narrow = rows == 8;
  const int nA = (a.M + rows - 1) / rows, nB = b ? (b->M + rows - 1) / rows : 0;
  dim3 grid(nA + nB), block(nw * 64);
  size_t shm = (size_t)(rows * H + rows * 16 + 2 * nw * H) * sizeof(float);
  const SkbParams& bb = b ? *b : a;
#define SKB1(TY, NIT, NW, ROWS)                                                                                              \
  do {                                                                                                                       \
    if (shm > 64 * 1024) hipFuncSetAttribute((const void*)smallk_ln_bwd_kernel<TY, NIT, NW, ROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
    hipLaunchKernelGGL((smallk_ln_bwd_kernel<TY, NIT, NW, ROWS>), grid, block, shm, st, a, bb, nA);                            \
  } while (0)
#define SKB(TY, NIT)                                                                                                         \
  do { if (NIT == 1 && wide) SKB1(TY, 1, 16, 64); else if (NIT >= 3 && narrow) SKB1(TY, NIT, 4, 8); else SKB1(TY, NIT, 4, 32); } while (0)
  DISPATCH_NIT(dtype, H, SKB);
#undef SKB
#undef SKB1
  return launch_status();
}

extern "C" int magic_smallk_ln_bwd(int dtype, int M, int H, int Kin, const float* x, const void* dy, const void* y,
                                   const float* gamma, const float* beta, const float* rstd,
                                   float* dW, float* db, float* dgamma, float* dbeta, float* part, void* stream) {
  if (M <= 0 || !okH(H) || Kin <= 0 || Kin > 16) return MAGIC_ERR_ARG;
  SkbParams a{M, Kin, x, dy, y, gamma, beta, rstd, dW, db, dgamma, dbeta, part};
  return launch_skb(dtype, H, a, nullptr, (hipStream_t)stream);
}

// two position-embedding backwards (map tokens || viewpoint tokens) in one launch; arguments as magic_smallk_ln_bwd, per problem
struct magic_skb_prob { int M, Kin; const float* x; const void* dy; const void* y; const float* gamma; const float* beta; const float* rstd;
                        float* dW; float* db; float* dgamma; float* dbeta; float* part; };
extern "C" int magic_smallk_ln_bwd_pair(int dtype, int H, const magic_skb_prob* d, void* stream) {
  if (!d || !okH(H)) return MAGIC_ERR_ARG;
  SkbParams p[2];
  for (int i = 0; i < 2; ++i) {
    if (d[i].M <= 0 || d[i].Kin <= 0 || d[i].Kin > 16 || !d[i].x || !d[i].dy || !d[i].y || !d[i].dW) return MAGIC_ERR_ARG;
    p[i] = SkbParams{d[i].M, d[i].Kin, d[i].x, d[i].dy, d[i].y, d[i].gamma, d[i].beta, d[i].rstd, d[i].dW, d[i].db, d[i].dgamma, d[i].dbeta, d[i].part};
  }
  return launch_skb(dtype, H, p[0], &p[1], (hipStream_t)stream);
}

extern "C" int magic_softmax_fwd(int dtype, int B, int nh, int Nq, int Nk, int ldp, const float* S, void* P, float scale,
                                 const unsigned char* kmask, const float* dist, const float* sprel_w, const float* sprel_b, void* stream) {
  if (B <= 0 || nh <= 0 || Nq <= 0 || Nk <= 0 || Nk > 64 * SM_IT || ldp < Nk || ldp > 64 * SM_IT) return MAGIC_ERR_ARG;
  if (dist && (!sprel_w || !sprel_b)) return MAGIC_ERR_ARG;
  long long rows = (long long)B * nh * Nq;
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((softmax_fwd_kernel<TY>), grid, block, 0, st, B, nh, Nq, Nk, ldp, S, (TY*)P, scale, kmask, dist, sprel_w, sprel_b));
  return launch_status();
}

extern "C" int magic_softmax_bwd(int dtype, int B, int nh, int Nq, int Nk, int ldp, const void* P, const float* dP, void* dS, float scale,
                                 const float* dist, float* dsprel_w, float* dsprel_b, void* stream) {
  if (B <= 0 || nh <= 0 || Nq <= 0 || Nk <= 0 || Nk > 64 * SM_IT || ldp < Nk || ldp > 64 * SM_IT) return MAGIC_ERR_ARG;
  if ((dist == nullptr) != (dsprel_w == nullptr) || (dist == nullptr) != (dsprel_b == nullptr)) return MAGIC_ERR_ARG;
  long long rows = (long long)B * nh * Nq;
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((softmax_bwd_kernel<TY>), grid, block, 0, st, B, nh, Nq, Nk, ldp, (const TY*)P, dP, (TY*)dS, scale, dist, dsprel_w, dsprel_b));
  return launch_status();
}

extern "C" int magic_head_mean_fwd(int dtype, int B, int nh, long long inner, const void* P, float* out, void* stream) {
  if (B <= 0 || nh <= 0 || inner <= 0) return MAGIC_ERR_ARG;
  long long n = (long long)B * inner;
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((head_mean_fwd_kernel<TY>), grid, block, 0, st, B, nh, inner, (const TY*)P, out));
  return launch_status();
}

extern "C" int magic_head_mean_bwd(int B, int nh, long long inner, const float* g, float* dP, int accumulate, void* stream) {
  if (B <= 0 || nh <= 0 || inner <= 0) return MAGIC_ERR_ARG;
  long long n = (long long)B * nh * inner;
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  hipLaunchKernelGGL(head_mean_bwd_kernel, grid, block, 0, (hipStream_t)stream, B, nh, inner, g, dP, accumulate);
  return launch_status();
}

extern "C" int magic_lndot_fwd(int dtype, int M, int H, const void* Y, const float* gamma, const float* beta, float eps,
                               const float* w2, const float* b2, float* logit, void* stream) {
  if (M <= 0 || !okH(H)) return MAGIC_ERR_ARG;
  dim3 grid((M + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
#define LDF(TY, NIT) hipLaunchKernelGGL((lndot_fwd_kernel<TY, NIT>), grid, block, 0, st, M, H, (const TY*)Y, gamma, beta, eps, w2, b2, logit)
  DISPATCH_NIT(dtype, H, LDF);
#undef LDF
  return launch_status();
}

// workgroups (= rows of 3 H + 1 floats of the `part` buffer) magic_lndot_bwd launches for M rows
extern "C" int magic_lndot_bwd_blocks(int M) { return M > 0 ? (M + 4 * LNB_ROWS - 1) / (4 * LNB_ROWS) : MAGIC_ERR_ARG; }
extern "C" int magic_lndot_bwd(int dtype, int M, int H, const void* Y, const float* gamma, const float* beta, float eps,
                               const float* w2, const float* dlogit, void* dZ, float* dgamma, float* dbeta, float* dw2, float* db2,
                               float* part, void* stream) {
  if (M <= 0 || !okH(H)) return MAGIC_ERR_ARG;
  if (!part && (!dgamma || !dbeta || !dw2 || !db2)) return MAGIC_ERR_ARG;
  dim3 grid((M + 4 * LNB_ROWS - 1) / (4 * LNB_ROWS)), block(256);
  size_t shm = (size_t)(12 * H + 4) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
#define LDB(TY, NIT) hipLaunchKernelGGL((lndot_bwd_kernel<TY, NIT>), grid, block, shm, st, M, H, (const TY*)Y, gamma, beta, eps, w2, dlogit, (TY*)dZ, dgamma, dbeta, dw2, db2, part)
  DISPATCH_NIT(dtype, H, LDB);
#undef LDB
  return launch_status();
}

// ---------------------------------------------------------------------------------------------
// Row dot / row gate (round 5): s[m] = x[m] . wx (+ e[m] . we) + b0 (+ b1), one wave per row, H any multiple of 128 up to 1024.
//   mode 0 (dot):  out_s[m] = s                           -- the value head's 512 -> 1 Linear (`Critic.state2value[3]`, DUET lineage)
//   mode 1 (gate): g = sigmoid(s), out[m, :] = e[m, :] g  -- the 'door' gate of the causal-intervention blocks (parser.py:129-142 do_add_method)
// backward: mode 0  dx = dy wx ; mode 1  ds = (dout . e) g (1 - g), de = dout g + ds we, dx = ds wx; parameter gradients per block through LDS, one
// atomic per element and block (dormant paths: no shipped script calls the value head, no shipped config switches the gate on).
template <typename T>
__global__ __launch_bounds__(256) void rowgate_fwd_kernel(int M, int H, int mode, const T* x, const T* e, const float* wx, const float* we,
                                                          const float* b0, const float* b1, float* out_s, T* out, float* gsave) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float s = 0.f;
  for (int c = lane * 2; c < H; c += 128) {
    float a, b;
    ld2<T>(x + (long long)row * H + c, a, b);
    s += a * wx[c] + b * wx[c + 1];
    if (e) { ld2<T>(e + (long long)row * H + c, a, b); s += a * we[c] + b * we[c + 1]; }
  }
  s = wave_sum(s) + (b0 ? b0[0] : 0.f) + (b1 ? b1[0] : 0.f);
  if (mode == 0) { if (lane == 0) out_s[row] = s; return; }
  const float g = 1.f / (1.f + __expf(-s));
  if (lane == 0 && gsave) gsave[row] = g;
  for (int c = lane * 2; c < H; c += 128) {
    float a, b;
    ld2<T>(e + (long long)row * H + c, a, b);
    st2<T>(out + (long long)row * H + c, a * g, b * g);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void rowgate_bwd_kernel(int M, int H, int mode, const T* x, const T* e, const float* wx, const float* we, const float* gsave,
                                                          const float* dy, const T* dout, T* dx, T* de, float* dwx, float* dwe, float* db0, float* db1) {
  extern __shared__ __attribute__((aligned(16))) float red[];      // [2][H] + [1]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 2 * H + 1; i += 256) red[i] = 0.f;
  __syncthreads();
  float bsum = 0.f;
  for (int row = blockIdx.x * 4 + wid; row < M; row += gridDim.x * 4) {
    float ds;
    float g = 0.f;
    if (mode == 0) ds = dy[row];
    else {
      g = gsave[row];
      float t = 0.f;
      for (int c = lane * 2; c < H; c += 128) {
        float a, b, u, v;
        ld2<T>(dout + (long long)row * H + c, a, b);
        ld2<T>(e + (long long)row * H + c, u, v);
        t += a * u + b * v;
      }
      ds = wave_sum(t) * g * (1.f - g);
    }
    bsum += ds;
    for (int c = lane * 2; c < H; c += 128) {
      float a, b;
      ld2<T>(x + (long long)row * H + c, a, b);
      atomicAdd(red + c, ds * a); atomicAdd(red + c + 1, ds * b);
      if (dx) st2<T>(dx + (long long)row * H + c, ds * wx[c], ds * wx[c + 1]);
      if (mode == 1) {
        float u, v, p, q;
        ld2<T>(e + (long long)row * H + c, u, v);
        ld2<T>(dout + (long long)row * H + c, p, q);
        atomicAdd(red + H + c, ds * u); atomicAdd(red + H + c + 1, ds * v);
        if (de) st2<T>(de + (long long)row * H + c, p * g + ds * we[c], q * g + ds * we[c + 1]);
      }
    }
  }
  if (lane == 0) atomicAdd(red + 2 * H, bsum);
  __syncthreads();
  for (int i = threadIdx.x; i < H; i += 256) {
    if (dwx && red[i] != 0.f) atomicAdd(dwx + i, red[i]);
    if (mode == 1 && dwe && red[H + i] != 0.f) atomicAdd(dwe + i, red[H + i]);
  }
  if (threadIdx.x == 0) {
    if (db0) atomicAdd(db0, red[2 * H]);
    if (db1) atomicAdd(db1, red[2 * H]);
  }
}
extern "C" int magic_rowgate_fwd(int dtype, int M, int H, int mode, const void* x, const void* e, const float* wx, const float* we,
                                 const float* b0, const float* b1, float* out_s, void* out, float* gsave, void* stream) {
  if (M <= 0 || H < 128 || H > 1024 || (H % 128) || !x || !wx || (mode != 0 && mode != 1) || !dtype_ok(dtype)) return MAGIC_ERR_ARG;
  if ((e != nullptr) != (we != nullptr) || (mode == 0 && !out_s) || (mode == 1 && (!e || !out))) return MAGIC_ERR_ARG;
  dim3 grid((M + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_BF16) hipLaunchKernelGGL(rowgate_fwd_kernel<bf16>, grid, block, 0, st, M, H, mode, (const bf16*)x, (const bf16*)e, wx, we, b0, b1, out_s, (bf16*)out, gsave);
  else if (dtype == DT_F16) hipLaunchKernelGGL(rowgate_fwd_kernel<f16>, grid, block, 0, st, M, H, mode, (const f16*)x, (const f16*)e, wx, we, b0, b1, out_s, (f16*)out, gsave);
  else hipLaunchKernelGGL(rowgate_fwd_kernel<float>, grid, block, 0, st, M, H, mode, (const float*)x, (const float*)e, wx, we, b0, b1, out_s, (float*)out, gsave);
  return launch_status();
}
extern "C" int magic_rowgate_bwd(int dtype, int M, int H, int mode, const void* x, const void* e, const float* wx, const float* we, const float* gsave,
                                 const float* dy, const void* dout, void* dx, void* de, float* dwx, float* dwe, float* db0, float* db1, void* stream) {
  if (M <= 0 || H < 128 || H > 1024 || (H % 128) || !x || !wx || (mode != 0 && mode != 1) || !dtype_ok(dtype)) return MAGIC_ERR_ARG;
  if ((mode == 0 && !dy) || (mode == 1 && (!e || !we || !gsave || !dout))) return MAGIC_ERR_ARG;
  int nb = (M + 15) / 16;
  if (nb > 256) nb = 256;
  dim3 grid(nb), block(256);
  const size_t shm = (size_t)(2 * H + 1) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_BF16) hipLaunchKernelGGL(rowgate_bwd_kernel<bf16>, grid, block, shm, st, M, H, mode, (const bf16*)x, (const bf16*)e, wx, we, gsave, dy, (const bf16*)dout, (bf16*)dx, (bf16*)de, dwx, dwe, db0, db1);
  else if (dtype == DT_F16) hipLaunchKernelGGL(rowgate_bwd_kernel<f16>, grid, block, shm, st, M, H, mode, (const f16*)x, (const f16*)e, wx, we, gsave, dy, (const f16*)dout, (f16*)dx, (f16*)de, dwx, dwe, db0, db1);
  else hipLaunchKernelGGL(rowgate_bwd_kernel<float>, grid, block, shm, st, M, H, mode, (const float*)x, (const float*)e, wx, we, gsave, dy, (const float*)dout, (float*)dx, (float*)de, dwx, dwe, db0, db1);
  return launch_status();
}

// ---------------------------------------------------------------------------------------------
// out[r, c] = in[r, c] * mask(site, r*cols + c) / (1-p)   (rows x cols logical, row pitch ld; in may alias out).
// The unfused attention path (key length > 128) drops its probabilities / masks their gradient with it, and the tests
// export the masks the fused kernels regenerate (in = ones).
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(long long rows, int cols, int ld, const T* in, T* out, DropDesc d) {
  const DropState s = drop_init(d);
  const long long n = rows * cols;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long r = i / cols; const int c = (int)(i - r * cols);
    const float m = s.on ? drop_mul(s, (unsigned)i) : 1.f;
    out[r * ld + c] = from_f<T>(to_f(in[r * ld + c]) * m);
  }
}
extern "C" int magic_dropout(int dtype, long long rows, int cols, int ld, const void* in, void* out,
                             const void* drop_seed, float drop_p, unsigned site, void* stream) {
  if (rows <= 0 || cols <= 0 || ld < cols || !in || !out || rows * cols > 0xFFFFFFFFll) return MAGIC_ERR_ARG;
  if (!drop_args_ok(drop_seed, drop_p)) return MAGIC_ERR_ARG;
  DropDesc d{drop_p > 0.f ? (const unsigned*)drop_seed : nullptr, site, drop_p};
  const long long n = rows * cols;
  dim3 grid((unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_BF16) hipLaunchKernelGGL(dropout_kernel<bf16>, grid, block, 0, st, rows, cols, ld, (const bf16*)in, (bf16*)out, d);
  else if (dtype == DT_F16) hipLaunchKernelGGL(dropout_kernel<f16>, grid, block, 0, st, rows, cols, ld, (const f16*)in, (f16*)out, d);
  else if (dtype == DT_F32) hipLaunchKernelGGL(dropout_kernel<float>, grid, block, 0, st, rows, cols, ld, (const float*)in, (float*)out, d);
  else return MAGIC_ERR_ARG;
  return launch_status();
}
