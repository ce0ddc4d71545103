// Whole self-attention encoders in ONE launch (gfx950, bf16, H = 128): the text encoder (6 post-LN BERT blocks over <= 80 tokens)
// and the panorama encoder (2 blocks over 36-38 view tokens) of MAGIC-S.
//
// One 512-thread workgroup owns ONE sample for ALL layers of its encoder: the sample's activations (<= 80 x 128 bf16 = 20 KB) never
// leave the CU between the QKV projection, the 2-head attention, the output projection + add&norm and the FFN + add&norm, nor
// between layers.  Weights are not staged through LDS at all: in y = x W^T every weight element is used exactly once per workgroup
// (all <= 80 rows multiply it at once), so each lane loads its MFMA B-fragment (8 consecutive k of one weight row = 16 bytes)
// straight from L2 into registers, a whole stage's fragments ahead of their use.  The workgroup writes exactly the tensors the
// engine's backward kernels read (qkv, P [+ dropped P], ctx, attention-output LN result + rstd, FFN pre-activation, GELU output,
// block output + rstd), with the same rounding points and the same counter-based dropout masks as the per-op kernels
// (gemm.hip / attention.hip / linear_ln), so the unfused backward runs unchanged on what this kernel saved.
//
// Replaces 5 launches per block (QKV GEMM, fused attention, dense+LN, FFN1 GEMM, dense+LN) x (6 | 2) blocks by one launch for both
// encoders: the text workgroups (long pole: 6 layers) come first in the grid, the panorama workgroups fill the remaining CUs.
// LDS per workgroup ~145 KB (one workgroup per CU); MFMA v_mfma_f32_16x16x32_bf16, fp32 accumulation, fp32 LayerNorm / softmax.
#include "enc_common.hpp"
#include <cstdlib>
#include <cstring>

template <typename Hh> struct EncLayerT {
  const Hh* Wqkv; const float* bqkv;                                   // [3H, H] (q | k | v rows), [3H]
  const Hh* Wo; const float* bo; const float* g1; const float* be1;    // attention.output.dense / LayerNorm
  const Hh* W1; const float* bi;                                       // intermediate.dense [I, H]
  const Hh* W2; const float* bo2; const float* g2; const float* be2;   // output.dense [H, I] / LayerNorm
  Hh *qkv, *P, *Pd, *ctx, *a, *z, *g, *out;                            // saved for the backward ([M,3H], [B,nh,N,ldp] x2, [M,H], [M,H], [M,I], [M,I], [M,H])
  float *rstd_a, *rstd_o;
  unsigned site_attn, site_ao, site_out, pad_;
};
template <typename Hh> struct EncSegT { const Hh* x; const unsigned char* kmask; int nsamp, N, ldp, nlayers; EncLayerT<Hh> L[6]; };
template <typename Hh> struct EncParamsT { EncSegT<Hh> seg[2]; int nseg; float p_attn, p_hidden, eps, scale; const unsigned* seed; };
typedef EncParamsT<bf16> EncParams; typedef EncSegT<bf16> EncSeg; typedef EncLayerT<bf16> EncLayer;      // host side: the layout holds pointers only, the same for both 16-bit types

// out[row][w*16 + c16] = LayerNorm_row(acc + bias (dropped) + residual) for the workgroup's NRT*16 rows; every wave owns 16 of the
// 128 columns, row statistics go through LDS (two passes: mean, then centred variance -- as linear_ln_kernel).
template <int NRT, typename Hh>
__device__ __forceinline__ void add_norm(f32x4 (&acc)[NRT], const float bv, const float gv, const float btv, const Hh* sRes,
                                         float* red, Hh* sOut, float* gRstd, int N, long long row_base, float eps,
                                         const DropState& ds, int w, int lane) {
  const int g = lane >> 4, c16 = lane & 15, col = w * 16 + c16;
  float s[NRT][4];
#pragma unroll
  for (int i = 0; i < NRT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = i * 16 + 4 * g + r;
      float v = acc[i][r] + bv;
      if (ds.on) v *= drop_mul(ds, (unsigned)((row_base + row) * EH + col));
      v += to_f(sRes[row * XS + col]);
      acc[i][r] = v;
      s[i][r] = g16_sum(v);
    }
  if (c16 == 0) {
#pragma unroll
    for (int i = 0; i < NRT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[w * MAXROWS + i * 16 + 4 * g + r] = s[i][r];
  }
  __syncthreads();
  float mean[NRT][4];
#pragma unroll
  for (int i = 0; i < NRT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = i * 16 + 4 * g + r;
      float t = 0.f;
#pragma unroll
      for (int ww = 0; ww < NWAVE; ++ww) t += red[ww * MAXROWS + rr];
      mean[i][r] = t * (1.0f / EH);
    }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NRT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float d = acc[i][r] - mean[i][r];
      s[i][r] = g16_sum(d * d);
    }
  if (c16 == 0) {
#pragma unroll
    for (int i = 0; i < NRT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[w * MAXROWS + i * 16 + 4 * g + r] = s[i][r];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NRT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = i * 16 + 4 * g + r;
      float t = 0.f;
#pragma unroll
      for (int ww = 0; ww < NWAVE; ++ww) t += red[ww * MAXROWS + rr];
      const float rstd = rsqrtf(t * (1.0f / EH) + eps);
      const Hh y = from_f<Hh>((acc[i][r] - mean[i][r]) * rstd * gv + btv);
      sOut[rr * XS + col] = (rr < N) ? y : (Hh)0.0f;        // rows past the sample stay zero (they feed the next GEMM as padding)
      if (rr < N && w == 0 && c16 == 0) gRstd[row_base + rr] = rstd;
    }
}

#ifdef ENC_TIMING
__device__ long long enc_ticks[2][16];        // [segment][stage mark] of block 0 of each segment, last layer
#define ENC_MARK(i) do { if (samp == 0 && tid == 0) enc_ticks[&sg == &p.seg[0] ? 0 : 1][i] = wall_clock64(); } while (0)
#else
#define ENC_MARK(i)
#endif

template <int NRT, typename Hh>
__device__ __forceinline__ void enc_body(const EncParamsT<Hh>& p, const EncSegT<Hh>& sg, const int samp, unsigned char* smem) {
  Hh* sX = (Hh*)smem;                        // [80][XS]   layer input / residual of the attention block
  Hh* sA = sX + MAXROWS * XS;                  // [80][XS]   attention context, then (after the norm) the FFN's input / residual
  Hh* sQKV = sA + MAXROWS * XS;                // [96][QS]   Q | K | V
  Hh* sP = sQKV + KROWS * QS;                  // [8][16][PSW] per-wave probability tiles
  Hh* sG = sQKV;                               // [80][GS]   GELU output (aliases Q|K|V and the probability tiles, dead by then)
  float* red = (float*)(sP + NWAVE * 16 * PSW);  // [8][80]    LayerNorm partial sums
  const int tid = threadIdx.x, lane0 = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N = sg.N, ldp = sg.ldp;
  const int NKP = (N + 31) / 32 * 32;
  const long long row_base = (long long)samp * N;
  constexpr int ROWS = NRT * 16;

  // ---- layer-0 input -> sX (rows past the sample zero)
  {
    const Hh* x = sg.x + row_base * EH;
    for (int id = tid; id < ROWS * (EH / 8); id += NWAVE * 64) {
      const int r = id / (EH / 8), c = (id % (EH / 8)) * 8;
      h16x8<Hh> v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (Hh)0.0f;
      if (r < N) v = *(const h16x8<Hh>*)(x + (long long)r * EH + c);
      *(h16x8<Hh>*)(sX + r * XS + c) = v;
    }
  }
  DropDesc dd;
  dd.seed = p.seed;
  // additive key-mask bias of this lane's key columns (layer-invariant): -10000 on padded keys, as HF's extended attention mask
  float kbias[NRT];
#pragma unroll
  for (int j = 0; j < NRT; ++j) {
    const int key = j * 16 + (lane0 & 15);
    kbias[j] = (key < N && sg.kmask && !sg.kmask[(long long)samp * N + key]) ? -10000.0f : 0.f;
  }
  for (int l = 0; l < sg.nlayers; ++l) {
    const EncLayerT<Hh>& L = sg.L[l];
    // every per-lane index below derives from `lane`; laundering it per layer keeps the compiler from hoisting the layer-invariant
    // row / column / predicate values of all five stages out of this loop (hundreds of registers, all spilled)
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int g = lane >> 4, c16 = lane & 15;
    ENC_MARK(0);
    // every small parameter this lane needs in the layer's five epilogues, fetched now: a load at its point of use would sit in
    // the in-order memory queue behind the stage's stores and expose a full round trip per stage
    float pb_qkv[3], pb_ffn[4];
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) pb_qkv[ct] = L.bqkv[(3 * w + ct) * 16 + c16];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) pb_ffn[ct] = L.bi[(4 * w + ct) * 16 + c16];
    const float pb_o = L.bo[w * 16 + c16], pg_1 = L.g1[w * 16 + c16], pe_1 = L.be1[w * 16 + c16];
    const float pb_2 = L.bo2[w * 16 + c16], pg_2 = L.g2[w * 16 + c16], pe_2 = L.be2[w * 16 + c16];
    // ================= A: Q|K|V = x Wqkv^T + b : 24 column tiles, 3 per wave =================
    h16x8<Hh> bw[3][4];
#pragma unroll
    for (int ct = 0; ct < 3; ++ct)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bw[ct][ks] = gfrag(L.Wqkv, EH, (3 * w + ct) * 16, ks * 32, lane);
    // key rows past the computed ones must be finite zeros (PV multiplies them by zero probabilities); the GELU image of the
    // previous layer overlapped them
    for (int id = tid; id < (KROWS - ROWS) * (384 / 8); id += NWAVE * 64) {
      const int r = ROWS + id / 48, c = (id % 48) * 8;
      h16x8<Hh> zv;
#pragma unroll
      for (int e = 0; e < 8; ++e) zv[e] = (Hh)0.0f;
      *(h16x8<Hh>*)(sQKV + r * QS + c) = zv;
    }
    __syncthreads();                              // sX complete (layer input), zero rows in place
    ENC_MARK(1);
    {
      f32x4 acc[NRT][3];
#pragma unroll
      for (int i = 0; i < NRT; ++i)
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) acc[i][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int i = 0; i < NRT; ++i) {
          const h16x8<Hh> a = lfrag(sX, XS, i * 16, ks * 32, lane);
#pragma unroll
          for (int ct = 0; ct < 3; ++ct) acc[i][ct] = emma(a, bw[ct][ks], acc[i][ct]);
        }
        KSTEP_FENCE();
      }
#pragma unroll
      for (int ct = 0; ct < 3; ++ct) {
        const int col = (3 * w + ct) * 16 + c16;
        const float bv = pb_qkv[ct];
#pragma unroll
        for (int i = 0; i < NRT; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) sQKV[(i * 16 + 4 * g + r) * QS + col] = from_f<Hh>(acc[i][ct][r] + bv);
      }
    }
    // prefetch the output projection's fragments (used after the attention)
    h16x8<Hh> wo[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) wo[ks] = gfrag(L.Wo, EH, w * 16, ks * 32, lane);
    __syncthreads();                              // Q|K|V image complete
    copy_out(sQKV, QS, L.qkv + row_base * 3 * EH, 3 * EH, N, 3 * EH, tid);
    ENC_MARK(2);
    // ================= B: attention, unit = (head, 16-query tile) =================
    // first FFN matrix (16 fragments: this wave's 4 column tiles x 4 k-steps): issued ahead of the attention's stores
    h16x8<Hh> w1[4][4];
#pragma unroll
    for (int ct = 0; ct < (NRT > 4 ? 2 : 4); ++ct)      // 80-row samples: half now, half after the attention (register budget)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) w1[ct][ks] = gfrag(L.W1, EH, (4 * w + ct) * 16, ks * 32, lane);
    dd.site = L.site_attn; dd.p = p.p_attn;
    const DropState dsa = drop_init(dd);
    Hh* sPw = sP + w * 16 * PSW;
    for (int u = w; u < ENH * NRT; u += NWAVE) {
      const int h = u / NRT, rt = u % NRT;
      f32x4 sc[NRT];
#pragma unroll
      for (int j = 0; j < NRT; ++j) sc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const h16x8<Hh> a = lfrag(sQKV, QS, rt * 16, h * EHD + ks * 32, lane);
#pragma unroll
        for (int j = 0; j < NRT; ++j) sc[j] = emma(a, lfrag(sQKV, QS, j * 16, EH + h * EHD + ks * 32, lane), sc[j]);
      }
      float mx[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
      for (int j = 0; j < NRT; ++j) {
        const int key = j * 16 + c16;
        const bool kv = key < N;
        const float mb = kbias[j];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = sc[j][r] * p.scale + mb;
          x = kv ? x : -3.0e38f;
          sc[j][r] = x; mx[r] = fmaxf(mx[r], x);
        }
      }
      float sum[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { mx[r] = g16_max(mx[r]); sum[r] = 0.f; }
#pragma unroll
      for (int j = 0; j < NRT; ++j) {
        const bool kv = (j * 16 + c16) < N;
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float e = kv ? __expf(sc[j][r] - mx[r]) : 0.f; sc[j][r] = e; sum[r] += e; }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) sum[r] = 1.0f / g16_sum(sum[r]);
      // clean probabilities -> the wave's tile (columns up to NKP: zeros past the sample's keys)
#pragma unroll
      for (int j = 0; j < NRT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) sPw[(4 * g + r) * PSW + j * 16 + c16] = from_f<Hh>(sc[j][r] * sum[r]);
      if (NRT * 16 < NKP) {
        for (int id = lane; id < 16 * (NKP - NRT * 16); id += 64) {
          const int r = id / (NKP - NRT * 16), c = NRT * 16 + id % (NKP - NRT * 16);
          sPw[r * PSW + c] = (Hh)0.0f;
        }
      }
      wave_lds_sync();                             // the tile is wave-private: no workgroup barrier needed
      const int nq = min(16, N - rt * 16);
      {
        Hh* Pg = L.P + (((long long)samp * ENH + h) * N + rt * 16) * ldp;
        const int cpr = ldp / 8;
        for (int id = lane; id < nq * cpr; id += 64) {
          const int r = id / cpr, c = (id % cpr) * 8;
          *(h16x8<Hh>*)(Pg + (long long)r * ldp + c) = *(const h16x8<Hh>*)(sPw + r * PSW + c);
        }
      }
      if (dsa.on) {
#pragma unroll
        for (int j = 0; j < NRT; ++j) {
          const int key = j * 16 + c16;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int ql = 4 * g + r;
            const unsigned idx = (unsigned)(((((long long)samp * ENH + h) * N + rt * 16 + ql) * N) + key);
            const float m = (ql < nq && key < N) ? drop_mul(dsa, idx) : 0.f;
            sPw[ql * PSW + key] = from_f<Hh>(sc[j][r] * sum[r] * m);
          }
        }
        wave_lds_sync();
        if (L.Pd) {
          Hh* Pg = L.Pd + (((long long)samp * ENH + h) * N + rt * 16) * ldp;
          const int cpr = ldp / 8;
          for (int id = lane; id < nq * cpr; id += 64) {
            const int r = id / cpr, c = (id % cpr) * 8;
            *(h16x8<Hh>*)(Pg + (long long)r * ldp + c) = *(const h16x8<Hh>*)(sPw + r * PSW + c);
          }
        }
      }
      f32x4 o[4];
#pragma unroll
      for (int jd = 0; jd < 4; ++jd) o[jd] = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int ks = 0; ks < NKP / 32; ++ks) {
        const h16x8<Hh> a = lfrag(sPw, PSW, 0, ks * 32, lane);
#pragma unroll
        for (int jd = 0; jd < 4; ++jd) o[jd] = emma(a, tfrag(sQKV + 2 * EH + h * EHD, QS, jd * 16, ks * 32, lane), o[jd]);
      }
#pragma unroll
      for (int jd = 0; jd < 4; ++jd)
#pragma unroll
        for (int r = 0; r < 4; ++r) sA[(rt * 16 + 4 * g + r) * XS + h * EHD + jd * 16 + c16] = from_f<Hh>(o[jd][r]);
    }
    ENC_MARK(3);
    if (NRT > 4) {
#pragma unroll
      for (int ct = 2; ct < 4; ++ct)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) w1[ct][ks] = gfrag(L.W1, EH, (4 * w + ct) * 16, ks * 32, lane);
    }
    __syncthreads();                              // context image complete (sA)
    copy_out(sA, XS, L.ctx + row_base * EH, EH, N, EH, tid);
    ENC_MARK(4);
    // ================= C: a = LayerNorm(x + dropout(ctx Wo^T + bo)) : 8 column tiles, 1 per wave =================
    {
      f32x4 acc[NRT];
#pragma unroll
      for (int i = 0; i < NRT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int i = 0; i < NRT; ++i) acc[i] = emma(lfrag(sA, XS, i * 16, ks * 32, lane), wo[ks], acc[i]);
        KSTEP_FENCE();
      }
      dd.site = L.site_ao; dd.p = p.p_hidden;
      const DropState dsh = drop_init(dd);
      // the first statistics barrier inside add_norm also orders "every wave has read the context image" before it is overwritten
      add_norm<NRT>(acc, pb_o, pg_1, pe_1, sX, red, sA, L.rstd_a, N, row_base, p.eps, dsh, w, lane);
    }
    __syncthreads();                              // sA = attention-block output, complete
    copy_out(sA, XS, L.a + row_base * EH, EH, N, EH, tid);
    ENC_MARK(5);
    // ================= D: z = a W1^T + bi ; g = gelu(z) : 32 column tiles, 4 per wave =================
    h16x8<Hh> w2[16];
    {
      f32x4 acc[NRT][4];
#pragma unroll
      for (int i = 0; i < NRT; ++i)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[i][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int i = 0; i < NRT; ++i) {
          const h16x8<Hh> a = lfrag(sA, XS, i * 16, ks * 32, lane);
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) acc[i][ct] = emma(a, w1[ct][ks], acc[i][ct]);
        }
        KSTEP_FENCE();
      }
      // second FFN matrix (this wave's 16 output columns x 512 k = 16 fragments): issued before this stage's stores
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) w2[ks] = gfrag(L.W2, EI, w * 16, ks * 32, lane);
      // pre-activation z (kept for the backward) -> image -> global; then the GELU output g takes the image's place
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const int col = (4 * w + ct) * 16 + c16;
#pragma unroll
        for (int i = 0; i < NRT; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            acc[i][ct][r] += pb_ffn[ct];
            sG[(i * 16 + 4 * g + r) * GS + col] = from_f<Hh>(acc[i][ct][r]);
          }
      }
      __syncthreads();
      copy_out(sG, GS, L.z + row_base * EI, EI, N, EI, tid);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int i = 0; i < NRT; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][ct][r] = gelu_fast(acc[i][ct][r]);
      __syncthreads();                            // every thread has read its share of the z image
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const int col = (4 * w + ct) * 16 + c16;
#pragma unroll
        for (int i = 0; i < NRT; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) sG[(i * 16 + 4 * g + r) * GS + col] = from_f<Hh>(acc[i][ct][r]);
      }
    }
    ENC_MARK(6);
    __syncthreads();                              // GELU image complete
    copy_out(sG, GS, L.g + row_base * EI, EI, N, EI, tid);
    ENC_MARK(7);
    // ================= E: out = LayerNorm(a + dropout(g W2^T + bo2)) =================
    {
      f32x4 acc[NRT];
#pragma unroll
      for (int i = 0; i < NRT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
        for (int i = 0; i < NRT; ++i) acc[i] = emma(lfrag(sG, GS, i * 16, ks * 32, lane), w2[ks], acc[i]);
        if ((ks & 1) == 1) KSTEP_FENCE();
      }
      dd.site = L.site_out; dd.p = p.p_hidden;
      const DropState dsh = drop_init(dd);
      add_norm<NRT>(acc, pb_2, pg_2, pe_2, sA, red, sX, L.rstd_o, N, row_base, p.eps, dsh, w, lane);
    }
    __syncthreads();                              // sX = block output, complete; every wave is done with the GELU image
    copy_out(sX, XS, L.out + row_base * EH, EH, N, EH, tid);
    ENC_MARK(8);
  }
}

template <typename Hh>
__global__ __launch_bounds__(512) void encoder_fwd_kernel(EncParamsT<Hh> p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char enc_smem[];
  int b = blockIdx.x, s = 0;
  if (b >= p.seg[0].nsamp) { b -= p.seg[0].nsamp; s = 1; }
  const EncSegT<Hh>& sg = p.seg[s];
  const int nrt = max(2, (sg.N + 15) / 16);
  switch (nrt) {
    case 2: enc_body<2>(p, sg, b, enc_smem); break;
    case 3: enc_body<3>(p, sg, b, enc_smem); break;
    case 4: enc_body<4>(p, sg, b, enc_smem); break;
    default: enc_body<5>(p, sg, b, enc_smem); break;
  }
}

static size_t enc_lds_bytes() {
  return (size_t)(2 * MAXROWS * XS + KROWS * QS + NWAVE * 16 * PSW) * 2 + (size_t)NWAVE * MAXROWS * sizeof(float);
}

extern "C" int magic_encoder_supported(int dtype, int H, int I, int nh, int N, int nlayers) {
  return dtype_is16(dtype) && H == EH && I == EI && nh == ENH && N >= 1 && N <= MAXROWS && nlayers >= 1 && nlayers <= 6;
}
extern "C" int magic_encoder_params_bytes() { return (int)sizeof(EncParams); }

// params: a host copy of EncParams (mirrored field by field by host/lib.py); nothing is read from it after this call returns
extern "C" int magic_encoder_fwd(int dtype, const void* params, int nbytes, void* stream) {
  if (!params || nbytes != (int)sizeof(EncParams) || !dtype_is16(dtype)) return MAGIC_ERR_ARG;
  EncParams p;
  memcpy(&p, params, sizeof(p));
  if (p.nseg < 1 || p.nseg > 2) return MAGIC_ERR_ARG;
  if (!drop_args_ok(p.seed, p.p_attn) || !drop_args_ok(p.seed, p.p_hidden)) return MAGIC_ERR_ARG;
  int blocks = 0;
  for (int s = 0; s < 2; ++s) {
    EncSeg& sg = p.seg[s];
    if (s >= p.nseg) { sg.nsamp = 0; continue; }
    if (sg.nsamp <= 0 || sg.N < 1 || sg.N > MAXROWS || sg.nlayers < 1 || sg.nlayers > 6 || !sg.x) return MAGIC_ERR_ARG;
    if (sg.ldp < sg.N || (sg.ldp & 7) || sg.ldp > KROWS) return MAGIC_ERR_ARG;
    if ((long long)sg.nsamp * ENH * sg.N * sg.N > 0xFFFFFFFFll || (long long)sg.nsamp * sg.N * EI > 0x7FFFFFFFll) return MAGIC_ERR_ARG;
    if ((uintptr_t)sg.x & 15) return MAGIC_ERR_ARG;
    for (int l = 0; l < sg.nlayers; ++l) {
      const EncLayer& L = sg.L[l];
      const void* req[] = {L.Wqkv, L.bqkv, L.Wo, L.bo, L.g1, L.be1, L.W1, L.bi, L.W2, L.bo2, L.g2, L.be2, L.qkv, L.P, L.ctx, L.a, L.z, L.g, L.out, L.rstd_a, L.rstd_o};
      for (const void* q : req)
        if (!q) return MAGIC_ERR_ARG;
      const void* al[] = {L.Wqkv, L.Wo, L.W1, L.W2, L.qkv, L.P, L.Pd, L.ctx, L.a, L.g, L.out};
      for (const void* q : al)
        if ((uintptr_t)q & 15) return MAGIC_ERR_ARG;
      if (p.p_attn > 0.f && !L.Pd) return MAGIC_ERR_ARG;
    }
    blocks += sg.nsamp;
  }
  const size_t shm = enc_lds_bytes();
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)encoder_fwd_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    (void)hipFuncSetAttribute((const void*)encoder_fwd_kernel<f16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    attr_set = true;
  }
  if (dtype == DT_BF16) hipLaunchKernelGGL(encoder_fwd_kernel<bf16>, dim3(blocks), dim3(512), shm, (hipStream_t)stream, p);
  else { EncParamsT<f16> pf; static_assert(sizeof(pf) == sizeof(p), "layout"); memcpy(&pf, &p, sizeof(pf)); hipLaunchKernelGGL(encoder_fwd_kernel<f16>, dim3(blocks), dim3(512), shm, (hipStream_t)stream, pf); }
  return launch_status();
}

// =====================================================================================================================================
// Cross-modal encoders in ONE launch: the global (map) and local (viewpoint) co-attention encoders of MAGIC-S, 3 METER BertCrossLayer
// blocks each (self-attention [+ graph-distance bias] -> cross-attention to the context (the instruction; or, for the MLM path, the
// text attending to the map) -> FFN), one 512-thread workgroup per sample for all layers of its encoder.  Same construction as
// encoder_fwd_kernel above: activations resident in LDS, weights streamed from L2 as MFMA B-fragments, and the workgroup writes what
// the per-op backward kernels read (self: qkv, P [+Pd], ctx, a, rstd_a; cross: q, kv, P [+Pd], ctx, c, rstd_c; FFN: z, g, out, rstd).
// Queries <= 80 rows (NRT tiles), context <= 80 rows; the context's key/value projection reads the context rows straight from
// global memory as MFMA A-fragments (they are layer-invariant and used by one GEMM per layer, not worth 21 KB of LDS).
// =====================================================================================================================================
template <typename Hh> struct XLayerT {
  const Hh* Wqkv; const float* bqkv; const Hh* Wo; const float* bo; const float* g1; const float* be1;          // attention.*
  const Hh* Wq; const float* bq; const Hh* Wkv; const float* bkv;                                               // crossattention.self (q | k,v adjacent)
  const Hh* Woc; const float* boc; const float* gc; const float* bec;                                             // crossattention.output
  const Hh* W1; const float* bi; const Hh* W2; const float* bo2; const float* g2; const float* be2;             // intermediate / output
  Hh *qkv, *P, *Pd, *ctx, *a; float* rstd_a;                      // self-attention block
  Hh *q, *kv, *Pc, *Pdc, *cctx, *c; float* rstd_c;                // cross-attention block
  Hh *z, *g, *out; float* rstd_o;                                 // FFN
  unsigned site_attn, site_ao, site_cattn, site_co, site_out, pad_;
};
template <typename Hh> struct XSegT {
  const Hh* x; const Hh* cx;                                    // queries [nsamp*Nq, H], context [nsamp*Nk, H]
  const unsigned char* qmask; const unsigned char* cmask;           // [nsamp, Nq], [nsamp, Nk]  (1 = valid)
  const float* dist; const float* sprel_w; const float* sprel_b;    // graph-distance bias of the self-attention (global encoder) or null
  int nsamp, Nq, Nk, ldps, ldpc, nlayers;
  XLayerT<Hh> L[3];
};
template <typename Hh> struct XParamsT { XSegT<Hh> seg[2]; int nseg; float p_attn, p_hidden, eps, scale; const unsigned* seed; };
typedef XParamsT<bf16> XParams; typedef XSegT<bf16> XSeg; typedef XLayerT<bf16> XLayer;

// one attention unit: this wave's 16 query rows (tile rt) of head h against NKT key tiles of the Q|K|V image; writes the clean (and,
// under dropout, the dropped) probabilities and the 16 x 64 context tile into sCtx
template <int NKT, typename Hh>
__device__ __forceinline__ void attn_unit(const Hh* sQKV, Hh* sPw, Hh* sCtx, Hh* Pg0, Hh* Pdg0, const int h, const int rt, const int Nq,
                                          const int Nk, const int ldp, const float (&kbias)[NKT], const float* dist, const float sw, const float sb,
                                          const long long samp, const float scale, const DropState& dsa, const int lane) {
  const int g = lane >> 4, c16 = lane & 15;
  const int NKP = (Nk + 31) / 32 * 32;
  f32x4 sc[NKT];
#pragma unroll
  for (int j = 0; j < NKT; ++j) sc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const h16x8<Hh> a = lfrag(sQKV, QS, rt * 16, h * EHD + ks * 32, lane);
#pragma unroll
    for (int j = 0; j < NKT; ++j) sc[j] = emma(a, lfrag(sQKV, QS, j * 16, EH + h * EHD + ks * 32, lane), sc[j]);
  }
  float mx[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
  for (int j = 0; j < NKT; ++j) {
    const int key = j * 16 + c16;
    const bool kv = key < Nk;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float x = sc[j][r] * scale + kbias[j];
      if (dist && kv) { const int qrow = min(rt * 16 + 4 * g + r, Nq - 1); x += sw * dist[((long long)samp * Nq + qrow) * Nk + key] + sb; }
      x = kv ? x : -3.0e38f;
      sc[j][r] = x; mx[r] = fmaxf(mx[r], x);
    }
  }
  float sum[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { mx[r] = g16_max(mx[r]); sum[r] = 0.f; }
#pragma unroll
  for (int j = 0; j < NKT; ++j) {
    const bool kv = (j * 16 + c16) < Nk;
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float e = kv ? __expf(sc[j][r] - mx[r]) : 0.f; sc[j][r] = e; sum[r] += e; }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) sum[r] = 1.0f / g16_sum(sum[r]);
#pragma unroll
  for (int j = 0; j < NKT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) sPw[(4 * g + r) * PSW + j * 16 + c16] = from_f<Hh>(sc[j][r] * sum[r]);
  if (NKT * 16 < NKP) {
    for (int id = lane; id < 16 * (NKP - NKT * 16); id += 64) {
      const int r = id / (NKP - NKT * 16), c = NKT * 16 + id % (NKP - NKT * 16);
      sPw[r * PSW + c] = (Hh)0.0f;
    }
  }
  wave_lds_sync();
  const int nq = min(16, Nq - rt * 16);
  {
    Hh* Pg = Pg0 + (((long long)samp * ENH + h) * Nq + rt * 16) * ldp;
    const int cpr = ldp / 8;
    for (int id = lane; id < nq * cpr; id += 64) {
      const int r = id / cpr, c = (id % cpr) * 8;
      *(h16x8<Hh>*)(Pg + (long long)r * ldp + c) = *(const h16x8<Hh>*)(sPw + r * PSW + c);
    }
  }
  if (dsa.on) {
#pragma unroll
    for (int j = 0; j < NKT; ++j) {
      const int key = j * 16 + c16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ql = 4 * g + r;
        const unsigned idx = (unsigned)(((((long long)samp * ENH + h) * Nq + rt * 16 + ql) * Nk) + key);
        const float m = (ql < nq && key < Nk) ? drop_mul(dsa, idx) : 0.f;
        sPw[ql * PSW + key] = from_f<Hh>(sc[j][r] * sum[r] * m);
      }
    }
    wave_lds_sync();
    if (Pdg0) {
      Hh* Pg = Pdg0 + (((long long)samp * ENH + h) * Nq + rt * 16) * ldp;
      const int cpr = ldp / 8;
      for (int id = lane; id < nq * cpr; id += 64) {
        const int r = id / cpr, c = (id % cpr) * 8;
        *(h16x8<Hh>*)(Pg + (long long)r * ldp + c) = *(const h16x8<Hh>*)(sPw + r * PSW + c);
      }
    }
  }
  f32x4 o[4];
#pragma unroll
  for (int jd = 0; jd < 4; ++jd) o[jd] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int ks = 0; ks < NKP / 32; ++ks) {
    const h16x8<Hh> a = lfrag(sPw, PSW, 0, ks * 32, lane);
#pragma unroll
    for (int jd = 0; jd < 4; ++jd) o[jd] = emma(a, tfrag(sQKV + 2 * EH + h * EHD, QS, jd * 16, ks * 32, lane), o[jd]);
  }
#pragma unroll
  for (int jd = 0; jd < 4; ++jd)
#pragma unroll
    for (int r = 0; r < 4; ++r) sCtx[(rt * 16 + 4 * g + r) * XS + h * EHD + jd * 16 + c16] = from_f<Hh>(o[jd][r]);
}

// one 16-column output tile per wave: acc[i] = sIn[rows of tile i] . W[w*16 .. +16]^T over K = 128
template <int NRT, typename Hh>
__device__ __forceinline__ void proj16(f32x4 (&acc)[NRT], const Hh* sIn, const h16x8<Hh> (&wf)[4], const int lane) {
#pragma unroll
  for (int i = 0; i < NRT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
    for (int i = 0; i < NRT; ++i) acc[i] = emma(lfrag(sIn, XS, i * 16, ks * 32, lane), wf[ks], acc[i]);
    KSTEP_FENCE();
  }
}

template <int NRT, typename Hh>
__device__ __forceinline__ void xenc_body(const XParamsT<Hh>& p, const XSegT<Hh>& sg, const int samp, unsigned char* smem) {
  constexpr int NKT = 5;                         // context key tiles (<= 80 rows); unused tiles cost a few MFMAs on zeros
  Hh* sX = (Hh*)smem;                        // [80][XS]   layer input x; later the cross context image / c / the block output
  Hh* sA = sX + MAXROWS * XS;                  // [80][XS]   self-attention context, then a (attention-block output)
  Hh* sQKV = sA + MAXROWS * XS;                // [96][QS]   Q | K | V of the self-attention, then Q | K,V(context) of the cross-attention
  Hh* sP = sQKV + KROWS * QS;
  Hh* sG = sQKV;
  float* red = (float*)(sP + NWAVE * 16 * PSW);
  const int tid = threadIdx.x, lane0 = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Nq = sg.Nq, Nk = sg.Nk;
  const long long qbase = (long long)samp * Nq, kbase = (long long)samp * Nk;
  constexpr int ROWS = NRT * 16;
  {
    const Hh* x = sg.x + qbase * EH;
    for (int id = tid; id < ROWS * (EH / 8); id += NWAVE * 64) {
      const int r = id / (EH / 8), c = (id % (EH / 8)) * 8;
      h16x8<Hh> v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (Hh)0.0f;
      if (r < Nq) v = *(const h16x8<Hh>*)(x + (long long)r * EH + c);
      *(h16x8<Hh>*)(sX + r * XS + c) = v;
    }
  }
  DropDesc dd;
  dd.seed = p.seed;
  float qbias[NRT], cbias[NKT];
#pragma unroll
  for (int j = 0; j < NRT; ++j) {
    const int key = j * 16 + (lane0 & 15);
    qbias[j] = (key < Nq && sg.qmask && !sg.qmask[qbase + key]) ? -10000.0f : 0.f;
  }
#pragma unroll
  for (int j = 0; j < NKT; ++j) {
    const int key = j * 16 + (lane0 & 15);
    cbias[j] = (key < Nk && sg.cmask && !sg.cmask[kbase + key]) ? -10000.0f : 0.f;
  }
  const float sw = sg.dist ? sg.sprel_w[0] : 0.f, sb = sg.dist ? sg.sprel_b[0] : 0.f;
  for (int l = 0; l < sg.nlayers; ++l) {
    const XLayerT<Hh>& L = sg.L[l];
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int g = lane >> 4, c16 = lane & 15;
    const int colw = w * 16 + c16;
    float pb_qkv[3], pb_ffn[4], pb_kv[2];
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) pb_qkv[ct] = L.bqkv[(3 * w + ct) * 16 + c16];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) pb_ffn[ct] = L.bi[(4 * w + ct) * 16 + c16];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) pb_kv[ct] = L.bkv[(2 * w + ct) * 16 + c16];
    const float pb_o = L.bo[colw], pg_1 = L.g1[colw], pe_1 = L.be1[colw], pb_q = L.bq[colw];
    const float pb_oc = L.boc[colw], pg_c = L.gc[colw], pe_c = L.bec[colw];
    const float pb_2 = L.bo2[colw], pg_2 = L.g2[colw], pe_2 = L.be2[colw];
    // ================= self-attention: Q|K|V =================
    h16x8<Hh> bw[3][4];
#pragma unroll
    for (int ct = 0; ct < 3; ++ct)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bw[ct][ks] = gfrag(L.Wqkv, EH, (3 * w + ct) * 16, ks * 32, lane);
    for (int id = tid; id < (KROWS - ROWS) * (384 / 8); id += NWAVE * 64) {
      const int r = ROWS + id / 48, c = (id % 48) * 8;
      h16x8<Hh> zv;
#pragma unroll
      for (int e = 0; e < 8; ++e) zv[e] = (Hh)0.0f;
      *(h16x8<Hh>*)(sQKV + r * QS + c) = zv;
    }
    __syncthreads();
    {
      f32x4 acc[NRT][3];
#pragma unroll
      for (int i = 0; i < NRT; ++i)
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) acc[i][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int i = 0; i < NRT; ++i) {
          const h16x8<Hh> a = lfrag(sX, XS, i * 16, ks * 32, lane);
#pragma unroll
          for (int ct = 0; ct < 3; ++ct) acc[i][ct] = emma(a, bw[ct][ks], acc[i][ct]);
        }
        KSTEP_FENCE();
      }
#pragma unroll
      for (int ct = 0; ct < 3; ++ct) {
        const int col = (3 * w + ct) * 16 + c16;
#pragma unroll
        for (int i = 0; i < NRT; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) sQKV[(i * 16 + 4 * g + r) * QS + col] = from_f<Hh>(acc[i][ct][r] + pb_qkv[ct]);
      }
    }
    h16x8<Hh> wo[4], wq[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { wo[ks] = gfrag(L.Wo, EH, w * 16, ks * 32, lane); wq[ks] = gfrag(L.Wq, EH, w * 16, ks * 32, lane); }
    __syncthreads();
    copy_out(sQKV, QS, L.qkv + qbase * 3 * EH, 3 * EH, Nq, 3 * EH, tid);
    dd.site = L.site_attn; dd.p = p.p_attn;
    const DropState dsa = drop_init(dd);
    Hh* sPw = sP + w * 16 * PSW;
    for (int u = w; u < ENH * NRT; u += NWAVE)
      attn_unit<NRT>(sQKV, sPw, sA, L.P, L.Pd, u / NRT, u % NRT, Nq, Nq, sg.ldps, qbias, sg.dist, sw, sb, samp, p.scale, dsa, lane);
    __syncthreads();                              // self-attention context complete (sA)
    copy_out(sA, XS, L.ctx + qbase * EH, EH, Nq, EH, tid);
    {
      f32x4 acc[NRT];
      proj16<NRT>(acc, sA, wo, lane);
      dd.site = L.site_ao; dd.p = p.p_hidden;
      const DropState dsh = drop_init(dd);
      add_norm<NRT>(acc, pb_o, pg_1, pe_1, sX, red, sA, L.rstd_a, Nq, qbase, p.eps, dsh, w, lane);
    }
    // context key / value projection weights: 2 column tiles per wave (K|V = 256 columns)
    h16x8<Hh> wkv[2][4];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) wkv[ct][ks] = gfrag(L.Wkv, EH, (2 * w + ct) * 16, ks * 32, lane);
    __syncthreads();                              // sA = a (self-attention block output); every wave is done with the Q|K|V image
    copy_out(sA, XS, L.a + qbase * EH, EH, Nq, EH, tid);
    // ================= cross-attention: Q from a, K|V from the context rows (global) =================
    {
      f32x4 acc[NRT];
      proj16<NRT>(acc, sA, wq, lane);
#pragma unroll
      for (int i = 0; i < NRT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) sQKV[(i * 16 + 4 * g + r) * QS + colw] = from_f<Hh>(acc[i][r] + pb_q);
    }
    {
      f32x4 acc[NKT][2];
#pragma unroll
      for (int i = 0; i < NKT; ++i)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[i][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const Hh* cx = sg.cx + kbase * EH;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int i = 0; i < NKT; ++i) {
          const int row = min(i * 16 + c16, Nk - 1);                 // rows past the context: any valid row (their keys are masked out below)
          const h16x8<Hh> a = *(const h16x8<Hh>*)(cx + (long long)row * EH + ks * 32 + 8 * g);
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) acc[i][ct] = emma(a, wkv[ct][ks], acc[i][ct]);
        }
        KSTEP_FENCE();
      }
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const int col = EH + (2 * w + ct) * 16 + c16;
#pragma unroll
        for (int i = 0; i < NKT; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = i * 16 + 4 * g + r;
            sQKV[row * QS + col] = (row < Nk) ? from_f<Hh>(acc[i][ct][r] + pb_kv[ct]) : (Hh)0.0f;     // rows >= Nk: zero (PV pads)
          }
      }
    }
    // rows NKT*16 .. 95 of the K|V columns: zero (the self-attention's zero rows may have been overwritten only below ROWS)
    for (int id = tid; id < (KROWS - NKT * 16) * (256 / 8); id += NWAVE * 64) {
      const int r = NKT * 16 + id / 32, c = EH + (id % 32) * 8;
      h16x8<Hh> zv;
#pragma unroll
      for (int e = 0; e < 8; ++e) zv[e] = (Hh)0.0f;
      *(h16x8<Hh>*)(sQKV + r * QS + c) = zv;
    }
    h16x8<Hh> woc[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) woc[ks] = gfrag(L.Woc, EH, w * 16, ks * 32, lane);
    h16x8<Hh> w1[4][4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) w1[ct][ks] = gfrag(L.W1, EH, (4 * w + ct) * 16, ks * 32, lane);
    __syncthreads();                              // cross Q | K,V image complete
    // q [Nq, H] and kv [Nk, 2H] for the backward
    copy_out(sQKV, QS, L.q + qbase * EH, EH, Nq, EH, tid);
    copy_out(sQKV + EH, QS, L.kv + kbase * 2 * EH, 2 * EH, Nk, 2 * EH, tid);
    dd.site = L.site_cattn; dd.p = p.p_attn;
    const DropState dsc = drop_init(dd);
    for (int u = w; u < ENH * NRT; u += NWAVE)
      attn_unit<NKT>(sQKV, sPw, sX, L.Pc, L.Pdc, u / NRT, u % NRT, Nq, Nk, sg.ldpc, cbias, nullptr, 0.f, 0.f, samp, p.scale, dsc, lane);
    __syncthreads();                              // cross context complete (sX; the layer input there is dead since the first add&norm)
    copy_out(sX, XS, L.cctx + qbase * EH, EH, Nq, EH, tid);
    {
      f32x4 acc[NRT];
      proj16<NRT>(acc, sX, woc, lane);
      dd.site = L.site_co; dd.p = p.p_hidden;
      const DropState dsh = drop_init(dd);
      add_norm<NRT>(acc, pb_oc, pg_c, pe_c, sA, red, sX, L.rstd_c, Nq, qbase, p.eps, dsh, w, lane);
    }
    __syncthreads();                              // sX = c (cross-attention block output)
    copy_out(sX, XS, L.c + qbase * EH, EH, Nq, EH, tid);
    // ================= FFN =================
    h16x8<Hh> w2[16];
    {
      f32x4 acc[NRT][4];
#pragma unroll
      for (int i = 0; i < NRT; ++i)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[i][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int i = 0; i < NRT; ++i) {
          const h16x8<Hh> a = lfrag(sX, XS, i * 16, ks * 32, lane);
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) acc[i][ct] = emma(a, w1[ct][ks], acc[i][ct]);
        }
        KSTEP_FENCE();
      }
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) w2[ks] = gfrag(L.W2, EI, w * 16, ks * 32, lane);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const int col = (4 * w + ct) * 16 + c16;
#pragma unroll
        for (int i = 0; i < NRT; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            acc[i][ct][r] += pb_ffn[ct];
            sG[(i * 16 + 4 * g + r) * GS + col] = from_f<Hh>(acc[i][ct][r]);
          }
      }
      __syncthreads();
      copy_out(sG, GS, L.z + qbase * EI, EI, Nq, EI, tid);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int i = 0; i < NRT; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][ct][r] = gelu_fast(acc[i][ct][r]);
      __syncthreads();
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const int col = (4 * w + ct) * 16 + c16;
#pragma unroll
        for (int i = 0; i < NRT; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) sG[(i * 16 + 4 * g + r) * GS + col] = from_f<Hh>(acc[i][ct][r]);
      }
    }
    __syncthreads();
    copy_out(sG, GS, L.g + qbase * EI, EI, Nq, EI, tid);
    {
      f32x4 acc[NRT];
#pragma unroll
      for (int i = 0; i < NRT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
        for (int i = 0; i < NRT; ++i) acc[i] = emma(lfrag(sG, GS, i * 16, ks * 32, lane), w2[ks], acc[i]);
        if ((ks & 1) == 1) KSTEP_FENCE();
      }
      dd.site = L.site_out; dd.p = p.p_hidden;
      const DropState dsh = drop_init(dd);
      // residual = c (sX), output -> sX: every wave has added its residual values before add_norm's first barrier
      add_norm<NRT>(acc, pb_2, pg_2, pe_2, sX, red, sX, L.rstd_o, Nq, qbase, p.eps, dsh, w, lane);
    }
    __syncthreads();
    copy_out(sX, XS, L.out + qbase * EH, EH, Nq, EH, tid);
  }
}

template <typename Hh>
__global__ __launch_bounds__(512) void xencoder_fwd_kernel(XParamsT<Hh> p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char enc_smem[];
  int b = blockIdx.x, s = 0;
  if (b >= p.seg[0].nsamp) { b -= p.seg[0].nsamp; s = 1; }
  const XSegT<Hh>& sg = p.seg[s];
  const int nrt = max(2, (sg.Nq + 15) / 16);
  switch (nrt) {
    case 2: xenc_body<2>(p, sg, b, enc_smem); break;
    case 3: xenc_body<3>(p, sg, b, enc_smem); break;
    case 4: xenc_body<4>(p, sg, b, enc_smem); break;
    default: xenc_body<5>(p, sg, b, enc_smem); break;
  }
}

extern "C" int magic_xencoder_supported(int dtype, int H, int I, int nh, int Nq, int Nk, int nlayers) {
  return dtype_is16(dtype) && H == EH && I == EI && nh == ENH && Nq >= 1 && Nq <= MAXROWS && Nk >= 1 && Nk <= MAXROWS && nlayers >= 1 && nlayers <= 3;
}
extern "C" int magic_xencoder_params_bytes() { return (int)sizeof(XParams); }

extern "C" int magic_xencoder_fwd(int dtype, const void* params, int nbytes, void* stream) {
  if (!params || nbytes != (int)sizeof(XParams) || !dtype_is16(dtype)) return MAGIC_ERR_ARG;
  XParams p;
  memcpy(&p, params, sizeof(p));
  if (p.nseg < 1 || p.nseg > 2) return MAGIC_ERR_ARG;
  if (!drop_args_ok(p.seed, p.p_attn) || !drop_args_ok(p.seed, p.p_hidden)) return MAGIC_ERR_ARG;
  int blocks = 0;
  for (int s = 0; s < 2; ++s) {
    XSeg& sg = p.seg[s];
    if (s >= p.nseg) { sg.nsamp = 0; continue; }
    if (sg.nsamp <= 0 || sg.Nq < 1 || sg.Nq > MAXROWS || sg.Nk < 1 || sg.Nk > MAXROWS || sg.nlayers < 1 || sg.nlayers > 3 || !sg.x || !sg.cx) return MAGIC_ERR_ARG;
    if (sg.ldps < sg.Nq || (sg.ldps & 7) || sg.ldps > KROWS || sg.ldpc < sg.Nk || (sg.ldpc & 7) || sg.ldpc > KROWS) return MAGIC_ERR_ARG;
    if ((long long)sg.nsamp * ENH * sg.Nq * MAXROWS > 0xFFFFFFFFll) return MAGIC_ERR_ARG;
    if (((uintptr_t)sg.x & 15) || ((uintptr_t)sg.cx & 15)) return MAGIC_ERR_ARG;
    if (sg.dist && (!sg.sprel_w || !sg.sprel_b)) return MAGIC_ERR_ARG;
    for (int l = 0; l < sg.nlayers; ++l) {
      const XLayer& L = sg.L[l];
      const void* req[] = {L.Wqkv, L.bqkv, L.Wo, L.bo, L.g1, L.be1, L.Wq, L.bq, L.Wkv, L.bkv, L.Woc, L.boc, L.gc, L.bec, L.W1, L.bi, L.W2, L.bo2, L.g2, L.be2,
                           L.qkv, L.P, L.ctx, L.a, L.rstd_a, L.q, L.kv, L.Pc, L.cctx, L.c, L.rstd_c, L.z, L.g, L.out, L.rstd_o};
      for (const void* q : req)
        if (!q) return MAGIC_ERR_ARG;
      const void* al[] = {L.Wqkv, L.Wo, L.Wq, L.Wkv, L.Woc, L.W1, L.W2, L.qkv, L.P, L.Pd, L.ctx, L.a, L.q, L.kv, L.Pc, L.Pdc, L.cctx, L.c, L.z, L.g, L.out};
      for (const void* q : al)
        if ((uintptr_t)q & 15) return MAGIC_ERR_ARG;
      if (p.p_attn > 0.f && (!L.Pd || !L.Pdc)) return MAGIC_ERR_ARG;
    }
    blocks += sg.nsamp;
  }
  const size_t shm = enc_lds_bytes();
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)xencoder_fwd_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    (void)hipFuncSetAttribute((const void*)xencoder_fwd_kernel<f16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    attr_set = true;
  }
  if (dtype == DT_BF16) hipLaunchKernelGGL(xencoder_fwd_kernel<bf16>, dim3(blocks), dim3(512), shm, (hipStream_t)stream, p);
  else { XParamsT<f16> pf; static_assert(sizeof(pf) == sizeof(p), "layout"); memcpy(&pf, &p, sizeof(pf)); hipLaunchKernelGGL(xencoder_fwd_kernel<f16>, dim3(blocks), dim3(512), shm, (hipStream_t)stream, pf); }
  return launch_status();
}
