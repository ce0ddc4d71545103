// Backward of the per-token half of a post-LN BERT block in ONE launch (gfx950, bf16, H = 128, FFN 512), on 32-row blocks over the
// whole chip.  Between two attention backwards the explicit backward of the text / panorama encoders is a chain of four per-token
// products with two LayerNorm backwards in it:
//
//     [tail of block j+1]   dx   = dQKV_{j+1} Wqkv_{j+1} + d_ao_{j+1}            -> LayerNorm-backward through block j's OUTPUT norm
//                                  -> d_fo (residual branch), d_fod = d_fo * dropout mask (dense branch; operand of dW2)
//     [FFN of block j]      d_z  = (d_fod W2) * gelu'(z_j)                        (operand of dW1)
//                           d_a  = d_z W1 + d_fo                                  -> LayerNorm-backward through block j's ATTENTION-OUTPUT norm
//                                  -> d_ao (residual branch, feeds the next tail), d_aod (dense branch; operand of dWo)
//     [output projection]   d_ctx = d_aod Wo                                      (input of block j's attention backward)
//
// The per-op path spends four launches on it (GEMM+LN-backward, GEMM, GEMM+LN-backward, GEMM); here a 512-thread workgroup owns 32
// rows for the whole chain: intermediate gradients stay in LDS / registers, and -- as in the forward encoder kernels -- every weight
// element is used once per workgroup and is loaded straight from L2 into MFMA B-fragments.  `dy W` contracts over W's ROWS, so the
// fragments (8 consecutive k per lane) need the TRANSPOSED weights: a bf16 transposed shadow of the four matrices per block is kept
// next to the ordinary shadow (magic_transpose_spans, refreshed with it after every optimizer step).
// The weight gradients stay with the engine's deferred grouped GEMM: this kernel writes their dY operands (d_fod, d_z, d_aod).
// Same rounding points (bf16 tensors between the products), LayerNorm-backward arithmetic, dropout masks and gamma / beta gradient
// atomics as magic_linear_lnbwd / magic_gemm, so the two paths agree to bf16 rounding.
#include "enc_common.hpp"
#include <cstdlib>
#include <cstring>


template <typename Hh> struct RbwSegT {
  int M, kt;                         // kt: k-steps of the tail product (12: dQKV [M,3H]; 4: a dQ [M,H]; 0: no product, dx = dao_n: top of a stack)
  const Hh* dqkv_n; const Hh* WqkvT_n; const Hh* dao_n;      // tail of the next block (dqkv_n == null: d_fo / d_fod are given)
  const Hh* dfo_in; const Hh* dfod_in;
  const Hh* y2; const float* rstd2; const float* g2; const float* b2; float* dg2; float* db2;     // this block's output LayerNorm
  const Hh* z; const Hh* W2T; const Hh* W1T;                                                  // [M, I]; [I, H]; [H, I]
  const Hh* y1; const float* rstd1; const float* g1; const float* b1; float* dg1; float* db1;     // attention-output LayerNorm
  const Hh* WoT;                                                                                  // [H, H]
  Hh *dfo, *dfod, *dz, *daod, *dao, *dctx;                        // outputs (dfo / dfod only when the tail runs here)
  unsigned site_out, site_ao;
  // ---- round 6: the ATTENTION BACKWARD of the block above inside this launch (mode != 0; see attn_tile_stage) ----
  // mode 1: [attention backward of block j+1 for this workgroup's 16 rows of one sample] -> the chain above with the tail's dQKV taken from LDS;
  // mode 2: [attention backward of block 0] -> dx0 = dQKV Wqkv + d_ao, written to dfo (the gradient wrt the stack's input), nothing else
  int mode, N, ntile, ldp;                                         // rows per sample (uniform), 16-row tiles per sample, row pitch of P
  const Hh* qkv_a; const Hh* P_a; const Hh* o_a; const Hh* dctx_a; const float* dP_init;     // block j+1: [M, 3H], [nsamp, 2, N, ldp], [M, H], [M, H]; optional fp32 seed
  Hh* dqkv_out;                                                    // [M, 3H]: the dY operand of dWqkv
  unsigned site_attn, pad_;
  // graph-distance bias of the map encoder's self-attention (softmax(... + w dist + b), r2r_magic_model_config.json:28): dist [nsamp, N, N] fp32 or NULL;
  // its two gradients are added here (one atomic pair per workgroup), as magic_attn_bwd does
  const float* dist; float* dsprel_w; float* dsprel_b;
};
template <typename Hh> struct RbwParamsT { RbwSegT<Hh> seg[2]; int nseg, blocks0; float p_hidden; int pad1; const unsigned* seed; float p_attn, scale; };
typedef RbwParamsT<bf16> RbwParams; typedef RbwSegT<bf16> RbwSeg;      // host side: pointers only, one layout for both 16-bit types

// exact-enough gelu'(x) = Phi(x) + x phi(x) with the same rational erf as gelu_fast (one exp shared by both terms)
__device__ __forceinline__ float dgelu_fast(float x) {
  const float ax = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float e = __expf(-ax * ax);                         // exp(-x^2 / 2)
  const float erf_abs = 1.0f - poly * e;
  const float cdf = 0.5f * (1.0f + (x < 0.f ? -erf_abs : erf_abs));
  return cdf + x * 0.39894228040143268f * e;
}

// rows x cols bf16 from global rows (row < nvalid, else zeros) into an LDS image
template <typename Hh> __device__ __forceinline__ void load_rows_img(Hh* s, int pitch, const Hh* g, long long ldg, int rows, int cols, int nvalid, int tid) {
  const int cpr = cols / 8;
  for (int id = tid; id < rows * cpr; id += NWAVE * 64) {
    const int r = id / cpr, c = (id % cpr) * 8;
    h16x8<Hh> v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (Hh)0.0f;
    if (r < nvalid) v = *(const h16x8<Hh>*)(g + (long long)r * ldg + c);
    *(h16x8<Hh>*)(s + r * pitch + c) = v;
  }
}

// LayerNorm backward over full rows, wave w owning columns [16w, 16w+16): v = acc (+ residual already added) is dL/dy; writes
// dx (residual branch) and dx * dropout mask (dense branch) as bf16 into two LDS images; gamma / beta gradients by one atomic per column
template <int NRT, typename Hh>
__device__ __forceinline__ void ln_bwd_rows(f32x4 (&acc)[NRT], const Hh* sY, const float* rstd_g, const float gm, const float bt, float* dgamma,
                                            float* dbeta, float* red, Hh* sSum, Hh* sDense, const int m0, const int M, const DropState& ds,
                                            const int w, const int lane, const int part_blk = -1) {
  constexpr int RB_ROWS = NRT * 16;
  const int g = lane >> 4, c16 = lane & 15, col = w * 16 + c16;
  const float ig = gm != 0.f ? 1.f / gm : 0.f;
  float xh[NRT][4], s1[NRT][4], s2[NRT][4];
  float pg = 0.f, pb = 0.f;
#pragma unroll
  for (int i = 0; i < NRT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = i * 16 + 4 * g + r;
      const bool live = m0 + rr < M;
      const float v = live ? acc[i][r] : 0.f;
      const float x = live ? (to_f(sY[rr * XS + col]) - bt) * ig : 0.f;
      pg += v * x; pb += v;
      const float ga = v * gm;
      acc[i][r] = ga; xh[i][r] = x;
      s1[i][r] = row16_sum(ga); s2[i][r] = row16_sum(ga * x);
    }
  if (dgamma) {      // fold the four row groups of the wave (lanes 16 apart), one atomic per column and workgroup
    pg += __shfl_xor(pg, 16, 64); pg += __shfl_xor(pg, 32, 64);
    pb += __shfl_xor(pb, 16, 64); pb += __shfl_xor(pb, 32, 64);
    if (g == 0) {
      // part_blk >= 0 (round 4): dgamma / dbeta are PARTIAL buffers [blocks][H] -- this workgroup's sums are stored in its own row and a
      // column-sum launch (magic_colsum_add) adds them up later in block order: no atomics (240-330 workgroups x 512 same-address-class
      // atomics per launch were 3 % of the training step), and the LayerNorm gradients of these blocks become reproducible
      if (part_blk >= 0) { dgamma[(long long)part_blk * EH + col] = pg; dbeta[(long long)part_blk * EH + col] = pb; }
      else { atomicAdd(dgamma + col, pg); atomicAdd(dbeta + col, pb); }
    }
  }
  if (c16 == 0) {
#pragma unroll
    for (int i = 0; i < NRT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) { red[w * RB_ROWS + i * 16 + 4 * g + r] = s1[i][r]; red[(NWAVE + w) * RB_ROWS + i * 16 + 4 * g + r] = s2[i][r]; }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NRT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = i * 16 + 4 * g + r;
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int ww = 0; ww < NWAVE; ++ww) { t1 += red[ww * RB_ROWS + rr]; t2 += red[(NWAVE + ww) * RB_ROWS + rr]; }
      const float m1 = t1 * (1.0f / EH), m2 = t2 * (1.0f / EH);
      const float rs = (m0 + rr < M) ? rstd_g[m0 + rr] : 0.f;
      const float d = rs * (acc[i][r] - m1 - xh[i][r] * m2);
      sSum[rr * XS + col] = from_f<Hh>(d);
      sDense[rr * XS + col] = from_f<Hh>(ds.on ? d * drop_mul(ds, (unsigned)((m0 + rr) * EH + col)) : d);
    }
}

#ifdef RBW_TIMING
__device__ long long rbw_ticks[32];            // wall_clock64 (100 MHz) marks of workgroup 0 (profiles/micro/rowbwd_timing.hip)
#define RBW_MARK(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) rbw_ticks[i] = wall_clock64(); } while (0)
#else
#define RBW_MARK(i)
#endif
// ---- round 6: attention backward per 16-row tile --------------------------------------------------------------------------------------------
// The per-token chain of block j and the attention backward of block j+1 used to alternate as two launches per block (magic_rowbwd, magic_attn_bwd:
// one workgroup per (sample, head) -- 96 workgroups for the text stack) because the attention backward sums over queries.  A 16-row tile of ONE
// sample can do its share alone, flash-attention style, once d_ctx of the WHOLE sample is in memory (it is: the previous launch wrote it):
//   as QUERIES (rows R of the tile, all keys):   dP = dO_R V^T (+ seed), mask, dS_R = P_R o (dP - rs) scale, dQ_R = dS_R K
//   as KEYS    (rows R as keys, all queries):    dS[:, R] the same way from dO (all rows) V_R^T, dK_R = dS[:, R]^T Q, dV_R = (P o mask)[:, R]^T dO
// with rs_q = sum_k P[q,k] dP[q,k] = dO_q . O_q (+ sum_k (P o mask)[q,k] seed[q,k] when a distillation gradient seeds dP) -- no pass over the keys.
// dP of the tile's own rows is computed twice (once per role: 1/5 of the products of an 80-token sample); everything else is the tile's own share.
// Operands: V rows are MFMA A fragments straight from global memory (k = head dim, contiguous); Q, K, dO of the sample sit in LDS images because
// dQ / dK / dV contract over rows (transposing reads); one head at a time through the same images.  Output: the tile's [16, 3H] dQKV rows in the
// chain's z image (QS pitch) -- the tail product's A operand -- and in global memory for the deferred dWqkv.
#define AT_DS 72          // [rows][64] images of one head
#define AT_PS 104         // [16][<= 96] dS / dS^T / (P o mask)^T tiles
#define AT_ROWS 96
static inline size_t attn_stage_lds() { return (size_t)(3 * AT_ROWS * AT_DS + 3 * 16 * AT_PS) * 2 + AT_ROWS * sizeof(float); }
template <typename Hh, bool DIST>          // DIST = the FULL form: graph-distance bias gradients + a gradient seeded into the attention map (dP_init).  An instantiation of its own:
                                           // their live registers through stage A (two sums, two f32x4 seeds) spill in the lean kernel and cost the form its gain
__device__ __forceinline__ void attn_tile_stage(const RbwSegT<Hh>& sg, const RbwParamsT<Hh>& p, const int b, const int ti, const int nv,
                                                Hh* sDq, unsigned char* scratch, const int tid) {
  typedef h16x4<Hh> v4;
  const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, c16 = lane & 15;
  const int N = sg.N, NT = sg.ntile, NK32 = (N + 31) / 32 * 32, ldp = sg.ldp;
  Hh* sQ = (Hh*)scratch;
  Hh* sK = sQ + AT_ROWS * AT_DS;
  Hh* sdO = sK + AT_ROWS * AT_DS;
  Hh* sdS = sdO + AT_ROWS * AT_DS;          // [16 queries of the tile][keys]
  Hh* sdST = sdS + 16 * AT_PS;              // [16 keys of the tile][queries]
  Hh* sPT = sdST + 16 * AT_PS;              // (P o mask)^T, same shape
  float* sRs = (float*)(sPT + 16 * AT_PS);  // [AT_ROWS]
  const long long row0 = (long long)b * N;
  DropDesc dd;
  dd.seed = p.seed; dd.site = sg.site_attn; dd.p = p.p_attn;
  const DropState ds = drop_init(dd);
  h16x8<Hh> zero8;
#pragma unroll
  for (int e = 0; e < 8; ++e) zero8[e] = (Hh)0.0f;
  // the three small tiles: columns [16 NT, NK32) are contraction padding that no stage writes -- zero everything once
  for (int id = tid; id < 3 * 16 * AT_PS / 8; id += NWAVE * 64) *(h16x8<Hh>*)(sdS + id * 8) = zero8;
  // Every global operand of a head is REQUESTED before anything waits (a first version loaded, waited, stored, loaded ... through ~7 dependent L2
  // round trips per head and took 40-70 us per launch where the separate attention backward took 12): the image rows and O rows of head h + 1 are in
  // flight under head h's stages, the V fragments / P / seed values of a wave's stage-A jobs under the image stores and the first barrier.
  constexpr int RIT = (AT_ROWS * 8 + NWAVE * 64 - 1) / (NWAVE * 64);       // 16-byte row chunks per thread and image (id = tid + it * 512: row id >> 3, chunk id & 7; rows >= NK32 skipped)
  static_assert(RIT == 2, "image-row load mapping");
  h16x8<Hh> rq[RIT], rk[RIT], ro[RIT], rO[RIT];         // Q, K, dO rows of the images; O rows (same (row, chunk) as dO: rs_q = dO_q . O_q needs no second dO load)
  auto img_issue = [&](const int h) {
#pragma unroll
    for (int it = 0; it < RIT; ++it) {
      const int id = tid + it * NWAVE * 64, r = id >> 3, c = (id & 7) * 8;
      rq[it] = zero8; rk[it] = zero8; ro[it] = zero8; rO[it] = zero8;
      if (r < N) {
        const Hh* base = sg.qkv_a + (row0 + r) * (3 * EH) + h * EHD + c;
        rq[it] = *(const h16x8<Hh>*)base;
        rk[it] = *(const h16x8<Hh>*)(base + EH);
        ro[it] = *(const h16x8<Hh>*)(sg.dctx_a + (row0 + r) * EH + h * EHD + c);
        rO[it] = *(const h16x8<Hh>*)(sg.o_a + (row0 + r) * EH + h * EHD + c);
      }
    }
  };
  auto img_store = [&](const int h) {                    // images + rs_q (8 lanes per row: the lanes of one row are an aligned group of 8)
#pragma unroll
    for (int it = 0; it < RIT; ++it) {
      const int id = tid + it * NWAVE * 64, r = id >> 3, c = (id & 7) * 8;
      if (r < NK32) {
        *(h16x8<Hh>*)(sQ + r * AT_DS + c) = rq[it];
        *(h16x8<Hh>*)(sK + r * AT_DS + c) = rk[it];
        *(h16x8<Hh>*)(sdO + r * AT_DS + c) = ro[it];
      }
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) s += to_f(ro[it][e]) * to_f(rO[it][e]);
      if (DIST && sg.dP_init && r < N) {    // + sum_k (P o mask)[q, k] seed[q, k]: the distillation gradient enters dP before the mask (top block only)
        const long long prow = ((long long)b * ENH + h) * N + r;
        for (int c8 = (id & 7) * 8; c8 < ldp; c8 += 64) {
          const h16x8<Hh> pv = *(const h16x8<Hh>*)(sg.P_a + prow * ldp + c8);
          const f32x4 i0 = *(const f32x4*)(sg.dP_init + prow * ldp + c8), i1 = *(const f32x4*)(sg.dP_init + prow * ldp + c8 + 4);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int key = c8 + e;
            if (key < N) {
              const float m = ds.on ? drop_mul(ds, (unsigned)(prow * N + key)) : 1.f;
              s += to_f(pv[e]) * m * (e < 4 ? i0[e & 3] : i1[e & 3]);
            }
          }
        }
      }
      s += dpp_mov<DPP_QUAD_XOR1>(s);
      s += dpp_mov<DPP_QUAD_XOR2>(s);
      s += dpp_mov<DPP_ROW_HALF_MIRROR>(s);
      if ((id & 7) == 0 && r < NK32) sRs[r] = s;
    }
  };
  // stage-A jobs of this wave: job = w + 8 jj < 2 NT; jobs [0, NT): the tile's rows as QUERIES against key tile job; [NT, 2 NT): as KEYS against query tile job - NT
  constexpr int AJ = (2 * (AT_ROWS / 16) + NWAVE - 1) / NWAVE;
  h16x8<Hh> av[AJ][EHD / 32];
  v4 ap[AJ];
  f32x4 ai[DIST ? AJ : 1];
  auto vp_issue = [&](const int h) {
#pragma unroll
    for (int jj = 0; jj < AJ; ++jj) {
      const int job = w + NWAVE * jj;
      const bool asq = job < NT;
      const int t = asq ? job : job - NT;
      const int ktile = asq ? t : ti, qtile = asq ? ti : t;
      const int vkey = ktile * 16 + c16, q = qtile * 16 + c16, key0 = ktile * 16 + 4 * g;
      const long long prow = ((long long)b * ENH + h) * N + q;
#pragma unroll
      for (int ks = 0; ks < EHD / 32; ++ks) {
        av[jj][ks] = zero8;
        if (job < 2 * NT && vkey < N) av[jj][ks] = *(const h16x8<Hh>*)(sg.qkv_a + (row0 + vkey) * (3 * EH) + 2 * EH + h * EHD + ks * 32 + 8 * g);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) ap[jj][r] = (Hh)0.0f;
      if constexpr (DIST) ai[jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (job < 2 * NT && q < N) {
        if (key0 < ldp) ap[jj] = *(const v4*)(sg.P_a + prow * ldp + key0);
        if constexpr (DIST) {
          if (sg.dP_init && key0 < N) ai[jj] = *(const f32x4*)(sg.dP_init + prow * ldp + key0);
        }
      }
    }
  };
  float a0 = 0.f, a1 = 0.f;            // sprel_linear gradients (as-queries role only: every (query, key) pair of the sample exactly once over its tiles)
  RBW_MARK(16);
  img_issue(0);
  vp_issue(0);
  for (int h = 0; h < ENH; ++h) {
    img_store(h);
    __syncthreads();
    RBW_MARK(17 + 3 * h);
    if (h + 1 < ENH) img_issue(h + 1);                   // in flight under stages A and B of this head
    // ---- stage A: dP^T tiles (rows = keys, columns = queries: a lane owns four consecutive keys of one query), dS into the role's image
#pragma unroll
    for (int jj = 0; jj < AJ; ++jj) {
      const int job = w + NWAVE * jj;
      if (job < 2 * NT) {
        const bool asq = job < NT;
        const int t = asq ? job : job - NT;
        const int ktile = asq ? t : ti, qtile = asq ? ti : t;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < EHD / 32; ++ks) acc = emma(av[jj][ks], lfrag(sdO, AT_DS, qtile * 16, ks * 32, lane), acc);
        const int q = qtile * 16 + c16, key0 = ktile * 16 + 4 * g;
        const bool qok = q < N;
        const long long prow = ((long long)b * ENH + h) * N + q;
        const float rsq = sRs[q];
        v4 o4, pm4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = key0 + r;
          const bool ok = qok && key < N;
          float d = acc[r];
          if constexpr (DIST) d += ok ? ai[jj][r] : 0.f;
          const float pp = to_f(ap[jj][r]);
          float pm = pp;
          if (ds.on) {
            const float m = ok ? drop_mul(ds, (unsigned)(prow * N + key)) : 0.f;
            d *= m; pm = pp * m;
          }
          const float dsu = pp * (d - rsq);
          o4[r] = from_f<Hh>(dsu * p.scale);
          pm4[r] = from_f<Hh>(pm);
          if constexpr (DIST) {
            if (sg.dist && asq && ok) { a0 += dsu * sg.dist[((long long)b * N + q) * N + key]; a1 += dsu; }
          }
        }
        if (asq) *(v4*)(sdS + c16 * AT_PS + key0) = o4;
        else {
#pragma unroll
          for (int r = 0; r < 4; ++r) { sdST[(4 * g + r) * AT_PS + q] = o4[r]; sPT[(4 * g + r) * AT_PS + q] = pm4[r]; }
        }
      }
    }
    if (h + 1 < ENH) vp_issue(h + 1);                    // in flight under stage B
    __syncthreads();
    RBW_MARK(18 + 3 * h);
    // ---- stage B: dQ_R = dS_R K, dK_R = dS[:, R]^T Q, dV_R = (P o mask)[:, R]^T dO: twelve 16 x 16 output tiles over the waves
    for (int job = w; job < 12; job += NWAVE) {
      const int prod = job >> 2, jd = job & 3;
      const Hh* A = prod == 0 ? sdS : (prod == 1 ? sdST : sPT);
      const Hh* Bm = prod == 0 ? sK : (prod == 1 ? sQ : sdO);
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int ks = 0; ks < NK32 / 32; ++ks) acc = emma(lfrag(A, AT_PS, 0, ks * 32, lane), tfrag(Bm, AT_DS, jd * 16, ks * 32, lane), acc);
      const int col = prod * EH + h * EHD + jd * 16 + c16;
#pragma unroll
      for (int r = 0; r < 4; ++r) sDq[(4 * g + r) * QS + col] = from_f<Hh>(acc[r]);
    }
    __syncthreads();
    RBW_MARK(19 + 3 * h);
  }
  copy_out(sDq, QS, sg.dqkv_out + (row0 + ti * 16) * (3 * EH), 3 * EH, nv, 3 * EH, tid);
  if (DIST && sg.dist) {                // (sRs is dead: the head loop ended on a barrier)
    a0 = wave_sum(a0); a1 = wave_sum(a1);
    if (lane == 0) { sRs[w] = a0; sRs[NWAVE + w] = a1; }
    __syncthreads();
    if (tid == 0) {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int i = 0; i < NWAVE; ++i) { s0 += sRs[i]; s1 += sRs[NWAVE + i]; }
      atomicAdd(sg.dsprel_w, s0);
      atomicAdd(sg.dsprel_b, s1);
    }
    __syncthreads();
  }
  RBW_MARK(23);
}

// NRT = row tiles of 16 per workgroup.  2 (32 rows): two workgroups per CU, 128 registers per lane, weight fragments one chunk of four
// k-steps ahead of their use.  4 (64 rows): one workgroup per CU, 256 registers, a whole product's fragments ahead -- and half the
// weight bytes streamed from L2 per row (every workgroup streams all 393 KB of the block's matrices).
template <int NRT, typename Hh, int ATT = 0>          // ATT: 0 no attention stage; 1 the stage; 2 the stage with the graph-distance bias gradients
__device__ __forceinline__ void rowbwd_body(const RbwParamsT<Hh>& p, unsigned char* rb_smem) {
  constexpr int RB_ROWS = NRT * 16;
  constexpr bool DEEP = NRT >= 4;
  static_assert(!ATT || NRT == 1, "the attention stage works on 16-row tiles of one sample");
  int blk = blockIdx.x, sidx = 0;
  if (ATT) {
    // XCD-aware order (an experiment, off): the dispatcher deals consecutive workgroup ids round-robin over the 8 XCDs, and the 5 tiles of a sample each
    // read the sample's whole Q / K / V / dO / O -- with consecutive LOGICAL ids on one XCD they would meet in one L2.  Measured SLOWER in the step.
    if (p.seg[0].pad_ & 1u) {               // bit 0 of seg[0].pad_: MAGIC_RBW_XCD=1.  OFF by default: measured 1.507 vs 1.458 ms/step on one box (profiles/micro/r06_ab_rbw_xcd.txt)
      const int G = gridDim.x, x = blk & 7, i = blk >> 3, q8 = G >> 3, r8 = G & 7;
      blk = x * q8 + min(x, r8) + i;
    }
  }
  if (blk >= p.blocks0) { blk -= p.blocks0; sidx = 1; }
  const RbwSegT<Hh>& sg = p.seg[sidx];
  const int mode = ATT ? sg.mode : 0;
  Hh* sZ = (Hh*)rb_smem;                   // [32][GS]  z, overwritten in place by d_z
  Hh* sY2 = sZ + RB_ROWS * GS;               // [32][XS]  this block's output (its LayerNorm's y); later the d_ctx staging image
  Hh* sR = sY2 + RB_ROWS * XS;               // [32][XS]  d_ao of the next block (residual of the tail)
  Hh* sFo = sR + RB_ROWS * XS;               // [32][XS]  d_fo
  Hh* sD = sFo + RB_ROWS * XS;               // [32][XS]  d_fod, later d_aod
  Hh* sY1 = sD + RB_ROWS * XS;               // [32][XS]  a (the attention-output LayerNorm's y)
  Hh* sAo = sR;                              // [..][XS]  d_ao (the tail's residual image is dead by then)
  float* red = (float*)(sY1 + RB_ROWS * XS);   // [2][8][rows]
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, c16 = lane & 15;
  const int M = sg.M;
  // rows of this workgroup: a flat 16 / 32 / 64-row block of the [M, H] buffers, or (mode != 0) 16-row tile blk % ntile of sample blk / ntile
  int m0 = blk * RB_ROWS, nv = min(RB_ROWS, M - m0);
  if (mode) { const int b = blk / sg.ntile, ti = blk - b * sg.ntile; m0 = b * sg.N + ti * 16; nv = min(16, sg.N - ti * 16); }
  const int colw = w * 16 + c16;
  const bool tail = mode ? true : sg.dqkv_n != nullptr;
  const bool shortm = sg.z == nullptr;           // short chain (cross-attention query side): tail -> LayerNorm backward -> output projection
  const int kt = mode ? 12 : sg.kt, ldq = kt * 32;
  DropDesc dd;
  dd.seed = p.seed; dd.p = p.p_hidden;
  RBW_MARK(0);
  if (mode) {          // dQKV rows of the block above: computed here, left in the z image's space (QS pitch) where the tail expects them
    const int b = blk / sg.ntile;
    attn_tile_stage<Hh, ATT == 2>(sg, p, b, blk - b * sg.ntile, nv, sZ, rb_smem + (size_t)RB_ROWS * GS * sizeof(Hh), tid);
    // (the stage ends on a barrier: its scratch behind the z image is dead and becomes the chain's images below)
  }
  // ---- weights of the first two products + small parameters, issued before anything else
  // (two workgroups per CU = 128 registers per lane: weight fragments arrive in chunks of four k-steps, one chunk ahead of their use)
  h16x8<Hh> wq[12];
  if (tail) {
#pragma unroll
    for (int ks = 0; ks < (DEEP ? 12 : 4); ++ks)
      if (ks < kt) wq[ks] = gfrag(sg.WqkvT_n, ldq, w * 16, ks * 32, lane);
  }
  h16x8<Hh> w2[4][4];
  const float gm2 = mode == 2 ? 0.f : sg.g2[colw], bt2 = mode == 2 ? 0.f : sg.b2[colw];
  const float gm1 = shortm ? 0.f : sg.g1[colw], bt1 = shortm ? 0.f : sg.b1[colw];     // (the short chain has no second LayerNorm)
  // ---- stage the block's rows: z, a, and either (out, d_ao of the next block) for the tail or the given (d_fo, d_fod)
  // tail: the dQKV rows of the block above pass through the z image's space first (z itself is fetched during the tail's epilogue)
  if (mode) ;                                    // (the attention stage left the dQKV rows in the image)
  else if (tail) load_rows_img(sZ, QS, sg.dqkv_n + (long long)m0 * ldq, ldq, RB_ROWS, ldq, nv, tid);
  else load_rows_img(sZ, GS, sg.z + (long long)m0 * EI, EI, RB_ROWS, EI, nv, tid);
  if (!shortm) load_rows_img(sY1, XS, sg.y1 + (long long)m0 * EH, EH, RB_ROWS, EH, nv, tid);
  if (tail) {
    if (mode != 2) load_rows_img(sY2, XS, sg.y2 + (long long)m0 * EH, EH, RB_ROWS, EH, nv, tid);
    load_rows_img(sR, XS, sg.dao_n + (long long)m0 * EH, EH, RB_ROWS, EH, nv, tid);
  } else {
    load_rows_img(sFo, XS, sg.dfo_in + (long long)m0 * EH, EH, RB_ROWS, EH, nv, tid);
    load_rows_img(sD, XS, sg.dfod_in + (long long)m0 * EH, EH, RB_ROWS, EH, nv, tid);
  }
  __syncthreads();
  RBW_MARK(1);
  // ================= tail: dx = dQKV W_qkv + d_ao -> LayerNorm backward (output norm) =================
  if (tail) {
    f32x4 acc[NRT];
#pragma unroll
    for (int i = 0; i < NRT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      if (4 * ch >= kt) break;
      if (!DEEP && ch < 2 && 4 * ch + 4 < kt) {
#pragma unroll
        for (int ks = 4 * ch + 4; ks < 4 * ch + 8; ++ks) wq[ks] = gfrag(sg.WqkvT_n, ldq, w * 16, ks * 32, lane);
      }
#pragma unroll
      for (int ks = 4 * ch; ks < 4 * ch + 4; ++ks) {
#pragma unroll
        for (int i = 0; i < NRT; ++i) acc[i] = emma(lfrag(sZ, QS, i * 16, ks * 32, lane), wq[ks], acc[i]);
      }
      KSTEP_FENCE();
    }
    RBW_MARK(2);
    if (mode == 2) {      // bottom of the stack: dx0 = dQKV Wqkv + d_ao is the gradient wrt the stack's input -- no LayerNorm, nothing below
#pragma unroll
      for (int i = 0; i < NRT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rr = i * 16 + 4 * g + r;
          sFo[rr * XS + colw] = from_f<Hh>(acc[i][r] + to_f(sR[rr * XS + colw]));
        }
      __syncthreads();
      copy_out(sFo, XS, sg.dfo + (long long)m0 * EH, EH, nv, EH, tid);
      return;
    }
    // z rows -> registers now (their round trip hides under the LayerNorm backward); they go into the image once every wave is past
    // the barrier inside ln_bwd_rows, i.e. done reading the dQKV rows
    constexpr int ZIT = RB_ROWS * (EI / 8) / (NWAVE * 64);
    h16x8<Hh> zr[ZIT];
#pragma unroll
    for (int it = 0; it < ZIT; ++it) {
      const int id = tid + it * NWAVE * 64, r = id / (EI / 8), c = (id % (EI / 8)) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) zr[it][e] = (Hh)0.0f;
      if (r < nv && !shortm) zr[it] = *(const h16x8<Hh>*)(sg.z + (long long)(m0 + r) * EI + c);
    }
#pragma unroll
    for (int i = 0; i < NRT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][r] += to_f(sR[(i * 16 + 4 * g + r) * XS + colw]);
    dd.site = sg.site_out;
    const DropState ds = drop_init(dd);
    ln_bwd_rows<NRT>(acc, sY2, sg.rstd2, gm2, bt2, sg.dg2, sg.db2, red, sFo, sD, m0, m0 + nv, ds, w, lane, p.pad1 ? blk : -1);
#pragma unroll
    for (int it = 0; it < ZIT; ++it) {
      const int id = tid + it * NWAVE * 64, r = id / (EI / 8), c = (id % (EI / 8)) * 8;
      *(h16x8<Hh>*)(sZ + r * GS + c) = zr[it];
    }
    __syncthreads();                             // d_fo / d_fod images and the z image complete
    RBW_MARK(3);
    copy_out(sFo, XS, sg.dfo + (long long)m0 * EH, EH, nv, EH, tid);
    copy_out(sD, XS, sg.dfod + (long long)m0 * EH, EH, nv, EH, tid);
  }
  RBW_MARK(4);
  h16x8<Hh> wo[4];
  if (shortm) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) wo[ks] = gfrag(sg.WoT, EH, w * 16, ks * 32, lane);
  } else {
  // ================= FFN: d_z = (d_fod W2) * gelu'(z) : 32 column tiles, 4 per wave =================
  h16x8<Hh> w1[16];
  {
    f32x4 acc[NRT][4];
#pragma unroll
    for (int i = 0; i < NRT; ++i)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[i][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) w2[ct][0] = gfrag(sg.W2T, EH, (4 * w + ct) * 16, 0, lane);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < 3) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) w2[ct][ks + 1] = gfrag(sg.W2T, EH, (4 * w + ct) * 16, (ks + 1) * 32, lane);
      }
#pragma unroll
      for (int i = 0; i < NRT; ++i) {
        const h16x8<Hh> a = lfrag(sD, XS, i * 16, ks * 32, lane);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[i][ct] = emma(a, w2[ct][ks], acc[i][ct]);
      }
      KSTEP_FENCE();
    }
#pragma unroll
    for (int ks = 0; ks < (DEEP ? 8 : 4); ++ks) w1[ks] = gfrag(sg.W1T, EI, w * 16, ks * 32, lane);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const int col = (4 * w + ct) * 16 + c16;
#pragma unroll
      for (int i = 0; i < NRT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rr = i * 16 + 4 * g + r;
          Hh* e = sZ + rr * GS + col;                      // each element of the image is read and rewritten by exactly one lane
          *e = from_f<Hh>(acc[i][ct][r] * dgelu_fast(to_f(*e)));
        }
    }
  }
  __syncthreads();                               // d_z image complete
  RBW_MARK(5);
  copy_out(sZ, GS, sg.dz + (long long)m0 * EI, EI, nv, EI, tid);
  RBW_MARK(6);
  // ================= d_a = d_z W1 + d_fo -> LayerNorm backward (attention-output norm) =================
  {
    f32x4 acc[NRT];
#pragma unroll
    for (int i = 0; i < NRT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int CHK = DEEP ? 8 : 4;              // k-steps per weight chunk
#pragma unroll
    for (int ch = 0; ch < 16 / CHK; ++ch) {
      if (ch + 1 < 16 / CHK) {
#pragma unroll
        for (int ks = CHK * (ch + 1); ks < CHK * (ch + 2); ++ks) w1[ks] = gfrag(sg.W1T, EI, w * 16, ks * 32, lane);
      } else {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wo[ks] = gfrag(sg.WoT, EH, w * 16, ks * 32, lane);
      }
#pragma unroll
      for (int ks = CHK * ch; ks < CHK * (ch + 1); ++ks) {
#pragma unroll
        for (int i = 0; i < NRT; ++i) acc[i] = emma(lfrag(sZ, GS, i * 16, ks * 32, lane), w1[ks], acc[i]);
        if ((ks & 3) == 3) KSTEP_FENCE();
      }
    }
#pragma unroll
    for (int i = 0; i < NRT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][r] += to_f(sFo[(i * 16 + 4 * g + r) * XS + colw]);
    RBW_MARK(7);
    dd.site = sg.site_ao;
    const DropState ds = drop_init(dd);
    // d_aod goes where d_fod was: every wave is past the FFN product (the barrier above) and reads sD no more
    ln_bwd_rows<NRT>(acc, sY1, sg.rstd1, gm1, bt1, sg.dg1, sg.db1, red, sAo, sD, m0, m0 + nv, ds, w, lane, p.pad1 ? blk : -1);
  }
  __syncthreads();                               // d_ao / d_aod images complete
  RBW_MARK(8);
  copy_out(sAo, XS, sg.dao + (long long)m0 * EH, EH, nv, EH, tid);
  copy_out(sD, XS, sg.daod + (long long)m0 * EH, EH, nv, EH, tid);
  }
  RBW_MARK(9);
  // ================= d_ctx = d_aod Wo (short chain: the tail's dense-branch gradient) =================
  {
    f32x4 acc[NRT];
#pragma unroll
    for (int i = 0; i < NRT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < NRT; ++i) acc[i] = emma(lfrag(sD, XS, i * 16, ks * 32, lane), wo[ks], acc[i]);
#pragma unroll
    for (int i = 0; i < NRT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) sY2[(i * 16 + 4 * g + r) * XS + colw] = from_f<Hh>(acc[i][r]);
  }
  __syncthreads();
  RBW_MARK(10);
  copy_out(sY2, XS, sg.dctx + (long long)m0 * EH, EH, nv, EH, tid);
  RBW_MARK(11);
}

template <typename Hh> __global__ __launch_bounds__(512, 4) void rowbwd16_kernel(RbwParamsT<Hh> p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rb_smem[];
  rowbwd_body<1>(p, rb_smem);
}
template <typename Hh> __global__ __launch_bounds__(512, 4) void rowbwd32_kernel(RbwParamsT<Hh> p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rb_smem[];
  rowbwd_body<2>(p, rb_smem);
}
template <typename Hh> __global__ __launch_bounds__(512, 2) void rowbwd64_kernel(RbwParamsT<Hh> p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rb_smem[];
  rowbwd_body<4>(p, rb_smem);
}
// round 6: the 16-row chain with the attention backward of the block above in front (segments with mode != 0)
template <typename Hh> __global__ __launch_bounds__(512, 4) void rowbwd16a_kernel(RbwParamsT<Hh> p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rb_smem[];
  rowbwd_body<1, Hh, 1>(p, rb_smem);
}
template <typename Hh> __global__ __launch_bounds__(512, 4) void rowbwd16ad_kernel(RbwParamsT<Hh> p) {      // ... with a graph-distance bias (map encoder)
  extern __shared__ __attribute__((aligned(16))) unsigned char rb_smem[];
  rowbwd_body<1, Hh, 2>(p, rb_smem);
}
static size_t rbw_lds_bytes(int rows) { return (size_t)(rows * GS + 5 * rows * XS) * 2 + (size_t)2 * NWAVE * rows * sizeof(float); }
// rows per workgroup: MAGIC_RBW_ROWS = 16 / 32 / 64 forces one shape; default (0) = 16 rows when that still leaves the launch with no more
// workgroups than CUs (the text stack alone: 3840 rows = 240 workgroups instead of 120 on 256 CUs), else 32 (measured: 64 rows 51.6 us vs 42.7)
static int rbw_rows_env() {
  static int r = -1;
  if (r < 0) { const char* e = getenv("MAGIC_RBW_ROWS"); const int v = e ? atoi(e) : 0; r = (v == 16 || v == 32 || v == 64) ? v : 0; }
  return r;
}
static int rbw_ncu() {
  static int n = 0;
  if (!n) { hipDeviceProp_t pr; int d = 0; (void)hipGetDevice(&d); n = (hipGetDeviceProperties(&pr, d) == hipSuccess) ? pr.multiProcessorCount : 256; }
  return n;
}

// rows per workgroup magic_rowbwd will choose for a launch of `total_rows` rows (all segments): a caller that passes PARTIAL gamma / beta
// buffers (params.pad1 != 0) sizes them with it -- ceil(M_seg / rows) x H floats per vector
extern "C" int magic_rowbwd_rows(long long total_rows) { return rbw_rows_env() ? rbw_rows_env() : (total_rows <= 16ll * rbw_ncu() ? 16 : 32); }

// dst_j[c] += sum_b part_j[b][c], b < nblk_j, c < H, in block order (fixed: reproducible), for n <= 96 jobs in one launch: the finisher of the
// partial-buffer mode of magic_rowbwd.  One 1024-thread workgroup per job: 1024 / H row groups take every (1024 / H)-th block, LDS fold.
__global__ __launch_bounds__(1024) void colsum_add_kernel(ColsumJobs js, int H) {
  __shared__ float red[1024];
  colsum_body(js, blockIdx.x, H, red);
}
extern "C" int magic_colsum_add(int H, int n, const float* const* parts, float* const* dsts, const int* nblks, void* stream) {
  if (n <= 0 || n > CSJ_MAX || !parts || !dsts || !nblks || H <= 0 || H > 1024 || (1024 % H)) return MAGIC_ERR_ARG;
  ColsumJobs js;
  js.n = n;
  for (int i = 0; i < n; ++i) {
    if (!parts[i] || !dsts[i] || nblks[i] <= 0) return MAGIC_ERR_ARG;
    js.part[i] = parts[i]; js.dst[i] = dsts[i]; js.nblk[i] = nblks[i];
  }
  hipLaunchKernelGGL(colsum_add_kernel, dim3(n), dim3(1024), 0, (hipStream_t)stream, js, H);
  return launch_status();
}

// workgroups a segment of `M` rows takes in a launch with an attention-stage segment (mode != 0: nsamp x ntile; else flat 16-row blocks)
extern "C" int magic_rowbwd_attn_supported(int dtype, int H, int I, int nh, int N) { return dtype_is16(dtype) && H == EH && I == EI && nh == ENH && N > 0 && N <= AT_ROWS; }
extern "C" int magic_rowbwd_supported(int dtype, int H, int I) { return dtype_is16(dtype) && H == EH && I == EI; }
extern "C" int magic_rowbwd_params_bytes() { return (int)sizeof(RbwParams); }

extern "C" int magic_rowbwd(int dtype, const void* params, int nbytes, void* stream) {
  if (!params || nbytes != (int)sizeof(RbwParams) || !dtype_is16(dtype)) return MAGIC_ERR_ARG;
  RbwParams p;
  memcpy(&p, params, sizeof(p));
  if (p.nseg < 1 || p.nseg > 2 || !drop_args_ok(p.seed, p.p_hidden)) return MAGIC_ERR_ARG;
  int blocks = 0;
  long long total_rows = 0;
  bool att = false;
  for (int s = 0; s < p.nseg; ++s) { total_rows += p.seg[s].M > 0 ? p.seg[s].M : 0; att = att || p.seg[s].mode != 0; }
  const int rows = att ? 16 : (rbw_rows_env() ? rbw_rows_env() : (total_rows <= 16ll * rbw_ncu() ? 16 : 32));
  if (att && !drop_args_ok(p.seed, p.p_attn)) return MAGIC_ERR_ARG;
  for (int s = 0; s < 2; ++s) {
    RbwSeg& sg = p.seg[s];
    if (s >= p.nseg) { sg.M = 0; sg.mode = 0; continue; }
    if (sg.M <= 0 || (long long)sg.M * EI > 0x7FFFFFFFll) return MAGIC_ERR_ARG;
    if (sg.mode) {                    // attention backward of the block above inside the launch: per-sample 16-row tiles
      if (sg.mode != 1 && sg.mode != 2) return MAGIC_ERR_ARG;
      if (sg.N <= 0 || sg.N > AT_ROWS || sg.M % sg.N || sg.ldp < sg.N || sg.ldp % 8 || sg.ntile != (sg.N + 15) / 16) return MAGIC_ERR_ARG;
      const void* need[] = {sg.qkv_a, sg.P_a, sg.o_a, sg.dctx_a, sg.dqkv_out, sg.WqkvT_n, sg.dao_n, sg.dfo};
      for (const void* q : need)
        if (!q || ((uintptr_t)q & 15)) return MAGIC_ERR_ARG;
      if (((uintptr_t)sg.dP_init & 15) || (long long)(sg.M / sg.N) * ENH * sg.N * sg.N > 0xFFFFFFFFll) return MAGIC_ERR_ARG;
      if ((sg.dist == nullptr) != (sg.dsprel_w == nullptr) || (sg.dist == nullptr) != (sg.dsprel_b == nullptr)) return MAGIC_ERR_ARG;
      if (sg.mode == 2) {             // bottom of a stack: attention backward + dx0 only
        const int nb2 = (sg.M / sg.N) * sg.ntile;
        if (s == 0) p.blocks0 = nb2;
        blocks += nb2;
        continue;
      }
      if (!sg.z || !sg.dfod) return MAGIC_ERR_ARG;       // (mode 1 runs the full chain; the short chain keeps its own launches)
    }
    const void* req[] = {sg.y2, sg.rstd2, sg.g2, sg.b2, sg.WoT, sg.dctx};
    for (const void* q : req)
      if (!q) return MAGIC_ERR_ARG;
    if (sg.z) {                       // full chain
      const void* full[] = {sg.W2T, sg.W1T, sg.y1, sg.rstd1, sg.g1, sg.b1, sg.dz, sg.daod, sg.dao};
      for (const void* q : full)
        if (!q) return MAGIC_ERR_ARG;
    } else if (!sg.dqkv_n) return MAGIC_ERR_ARG;      // the short chain is a tail by definition
    if (sg.mode) ;
    else if (sg.dqkv_n) { if (!sg.WqkvT_n || !sg.dao_n || !sg.dfo || !sg.dfod || (sg.kt != 0 && sg.kt != 4 && sg.kt != 12)) return MAGIC_ERR_ARG; }
    else if (!sg.dfo_in || !sg.dfod_in) return MAGIC_ERR_ARG;
    if ((sg.dg2 == nullptr) != (sg.db2 == nullptr) || (sg.dg1 == nullptr) != (sg.db1 == nullptr)) return MAGIC_ERR_ARG;
    const void* al[] = {sg.dqkv_n, sg.WqkvT_n, sg.dao_n, sg.dfo_in, sg.dfod_in, sg.y2, sg.z, sg.W2T, sg.W1T, sg.y1, sg.WoT, sg.dfo, sg.dfod, sg.dz, sg.daod, sg.dao, sg.dctx};
    for (const void* q : al)
      if ((uintptr_t)q & 15) return MAGIC_ERR_ARG;
    const int nb = sg.mode ? (sg.M / sg.N) * sg.ntile : (sg.M + rows - 1) / rows;
    if (s == 0) p.blocks0 = nb;
    blocks += nb;
  }
  if (att) {
    static int xcd_off = -1;
    if (xcd_off < 0) { const char* e = getenv("MAGIC_RBW_XCD"); xcd_off = (e && atoi(e) == 1) ? 1 : 0; }      // (1 = the XCD-aware order ON)
    p.seg[0].pad_ = xcd_off ? 1u : 0u;
    const size_t rest = rbw_lds_bytes(16) - (size_t)16 * GS * 2;
    const size_t shm = (size_t)16 * GS * 2 + (attn_stage_lds() > rest ? attn_stage_lds() : rest);
    static bool attr_a = false;
    if (!attr_a) {
      (void)hipFuncSetAttribute((const void*)rowbwd16a_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
      (void)hipFuncSetAttribute((const void*)rowbwd16a_kernel<f16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
      (void)hipFuncSetAttribute((const void*)rowbwd16ad_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
      (void)hipFuncSetAttribute((const void*)rowbwd16ad_kernel<f16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
      attr_a = true;
    }
    const bool with_dist = (p.seg[0].mode && (p.seg[0].dist || p.seg[0].dP_init)) || (p.nseg > 1 && p.seg[1].mode && (p.seg[1].dist || p.seg[1].dP_init));      // the full form
    RbwParamsT<f16> pf;
    static_assert(sizeof(pf) == sizeof(p), "layout");
    memcpy(&pf, &p, sizeof(pf));
    if (with_dist) {
      if (dtype == DT_BF16) hipLaunchKernelGGL(rowbwd16ad_kernel<bf16>, dim3(blocks), dim3(512), shm, (hipStream_t)stream, p);
      else hipLaunchKernelGGL(rowbwd16ad_kernel<f16>, dim3(blocks), dim3(512), shm, (hipStream_t)stream, pf);
    } else {
      if (dtype == DT_BF16) hipLaunchKernelGGL(rowbwd16a_kernel<bf16>, dim3(blocks), dim3(512), shm, (hipStream_t)stream, p);
      else hipLaunchKernelGGL(rowbwd16a_kernel<f16>, dim3(blocks), dim3(512), shm, (hipStream_t)stream, pf);
    }
    return launch_status();
  }
  const size_t shm = rbw_lds_bytes(rows);
  static bool attr_set = false;
  if (!attr_set) {
#define RBW_ATTR(TY)                                                                                                              \
    (void)hipFuncSetAttribute((const void*)rowbwd16_kernel<TY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rbw_lds_bytes(16)); \
    (void)hipFuncSetAttribute((const void*)rowbwd32_kernel<TY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rbw_lds_bytes(32)); \
    (void)hipFuncSetAttribute((const void*)rowbwd64_kernel<TY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rbw_lds_bytes(64))
    RBW_ATTR(bf16); RBW_ATTR(f16);
#undef RBW_ATTR
    attr_set = true;
  }
#define RBW_LAUNCH(TY, PP)                                                                                          \
  do {                                                                                                              \
    if (rows == 16) hipLaunchKernelGGL(rowbwd16_kernel<TY>, dim3(blocks), dim3(512), shm, (hipStream_t)stream, PP); \
    else if (rows == 32) hipLaunchKernelGGL(rowbwd32_kernel<TY>, dim3(blocks), dim3(512), shm, (hipStream_t)stream, PP); \
    else hipLaunchKernelGGL(rowbwd64_kernel<TY>, dim3(blocks), dim3(512), shm, (hipStream_t)stream, PP);            \
  } while (0)
  if (dtype == DT_BF16) RBW_LAUNCH(bf16, p);
  else { RbwParamsT<f16> pf; static_assert(sizeof(pf) == sizeof(p), "layout"); memcpy(&pf, &p, sizeof(pf)); RBW_LAUNCH(f16, pf); }
#undef RBW_LAUNCH
  return launch_status();
}

// ---- transposed bf16 shadow: dst[off .. off + rows*cols) = transpose of the [rows, cols] matrix at src[off ..) -----------------------
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;      // 2-byte moves: the same kernel serves the bf16 and the f16 shadow
#define TSP_MAX 160
struct TSpans { long long off[TSP_MAX]; int rows[TSP_MAX]; int cols[TSP_MAX]; int tile0[TSP_MAX + 1]; int n; };
__global__ __launch_bounds__(256) void transpose_spans_kernel(const unsigned short* __restrict__ src, unsigned short* __restrict__ dst, TSpans t) {
  __shared__ unsigned short tile[32][34];
  const int id = blockIdx.x;
  int s = 0;
  for (int i = 1; i < t.n; ++i) s += (id >= t.tile0[i]) ? 1 : 0;
  const int local = id - t.tile0[s], R = t.rows[s], C = t.cols[s];
  const int tc = (C + 31) / 32, r0 = (local / tc) * 32, c0 = (local % tc) * 32;
  const unsigned short* a = src + t.off[s];
  unsigned short* b = dst + t.off[s];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8)
    if (r0 + j < R && c0 + tx < C) tile[j][tx] = a[(long long)(r0 + j) * C + c0 + tx];
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
    if (c0 + j < C && r0 + tx < R) b[(long long)(c0 + j) * R + r0 + tx] = tile[tx][j];
}
// every span a multiple of 64 x 64 at 16-byte-aligned offsets (all weight matrices of the H = 128 / FFN 512 blocks): 64 x 64 tiles, 16-byte
// global loads and stores on both sides (the 32 x 32 kernel above moves 2 bytes per lane and access: 20 us for the 3 M elements of a step)
__global__ __launch_bounds__(256) void transpose_spans64_kernel(const unsigned short* __restrict__ src, unsigned short* __restrict__ dst, TSpans t) {
  __shared__ unsigned short tile[64][72];
  const int id = blockIdx.x;
  int s = 0;
  for (int i = 1; i < t.n; ++i) s += (id >= t.tile0[i]) ? 1 : 0;
  const int local = id - t.tile0[s], R = t.rows[s], C = t.cols[s];
  const int tc = C / 64, r0 = (local / tc) * 64, c0 = (local % tc) * 64;
  const unsigned short* a = src + t.off[s];
  unsigned short* b = dst + t.off[s];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int ch = threadIdx.x + it * 256, r = ch >> 3, c = (ch & 7) * 8;
    *(u16x8*)&tile[r][c] = *(const u16x8*)(a + (long long)(r0 + r) * C + c0 + c);
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int ch = threadIdx.x + it * 256, cc = ch >> 3, rr = (ch & 7) * 8;        // output row c0 + cc, 8 consecutive source rows
    u16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[rr + e][cc];
    *(u16x8*)(b + (long long)(c0 + cc) * R + r0 + rr) = v;
  }
}
// offs / rows / cols: host arrays of n spans (element offsets into the flat bf16 buffers)
extern "C" int magic_transpose_spans(const void* src, void* dst, int n, const long long* offs, const int* rows, const int* cols, void* stream) {
  if (!src || !dst || n < 0 || (n && (!offs || !rows || !cols))) return MAGIC_ERR_ARG;
  bool all64 = !((uintptr_t)src & 15) && !((uintptr_t)dst & 15);
  for (int i = 0; i < n && all64; ++i) all64 = rows[i] > 0 && cols[i] > 0 && rows[i] % 64 == 0 && cols[i] % 64 == 0 && offs[i] % 8 == 0;
  const int ts = all64 ? 64 : 32;
  for (int base = 0; base < n; base += TSP_MAX) {
    TSpans t;
    t.n = n - base < TSP_MAX ? n - base : TSP_MAX;
    int tiles = 0;
    for (int i = 0; i < t.n; ++i) {
      if (rows[base + i] <= 0 || cols[base + i] <= 0) return MAGIC_ERR_ARG;
      t.off[i] = offs[base + i]; t.rows[i] = rows[base + i]; t.cols[i] = cols[base + i];
      t.tile0[i] = tiles;
      tiles += ((rows[base + i] + ts - 1) / ts) * ((cols[base + i] + ts - 1) / ts);
    }
    t.tile0[t.n] = tiles;
    if (all64) hipLaunchKernelGGL(transpose_spans64_kernel, dim3(tiles), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)src, (unsigned short*)dst, t);
    else hipLaunchKernelGGL(transpose_spans_kernel, dim3(tiles), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)src, (unsigned short*)dst, t);
  }
  return launch_status();
}
