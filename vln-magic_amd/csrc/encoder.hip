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
template <typename Hh> struct EncParamsT {
  EncSegT<Hh> seg[2]; int nseg; float p_attn, p_hidden, eps, scale; const unsigned* seed;
  unsigned* sync; int sync_words, pad2_;        // row-split form: >= 6 * (samples of all segments) + 4 words, zeroed by the launch function
};
typedef EncParamsT<bf16> EncParams; typedef EncSegT<bf16> EncSeg; typedef EncLayerT<bf16> EncLayer;      // host side: the layout holds pointers only, the same for both 16-bit types

// out[row][w*16 + c16] = LayerNorm_row(acc + bias (dropped) + residual) for the workgroup's NRT*16 rows; every wave owns 16 of the
// 128 columns, row statistics go through LDS (two passes: mean, then centred variance -- as linear_ln_kernel).
template <int NRT, int RSTR, typename Hh>
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
      for (int r = 0; r < 4; ++r) red[w * RSTR + i * 16 + 4 * g + r] = s[i][r];
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
      for (int ww = 0; ww < NWAVE; ++ww) t += red[ww * RSTR + rr];
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
      for (int r = 0; r < 4; ++r) red[w * RSTR + i * 16 + 4 * g + r] = s[i][r];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NRT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = i * 16 + 4 * g + r;
      float t = 0.f;
#pragma unroll
      for (int ww = 0; ww < NWAVE; ++ww) t += red[ww * RSTR + rr];
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

// Launches of the encoder kernels that have got their LAST workgroup onto a CU (monotonic; never reset) and the 100 MHz tick of the latest
// one.  magic_encoder_start_gate parks another stream until the count moves: the frozen teacher's forward, which runs next to the student's
// step on a side stream, then starts once the student's whole-encoder launch has its workgroups resident instead of taking CUs and LDS
// from under it (bench: 1.615 -> 1.57 ms).
__device__ unsigned magic_enc_starts;
__device__ long long magic_enc_start_tick;
__device__ __forceinline__ void enc_mark_start() {
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    __hip_atomic_store(&magic_enc_start_tick, (long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&magic_enc_starts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// LDS layout of the per-sample body.  Samples of <= 48 rows (panoramas: 36-38 views) take the COMPACT layout, 76.5 KB: images sized for 48
// rows, ONE zero row behind the V image (the PV product's key steps past row 47 are clamped onto it) instead of 48 zeroed rows, six
// probability tiles (2 heads x 3 query tiles: waves 6 and 7 have no attention unit) of pitch 72, the LayerNorm partial sums behind the
// GELU image.  With the row-split text tiles at 79.6 KB the mixed launch then fits TWO workgroups per CU.  (Measured: the launch takes the
// same 180-230 us either way -- at 145 KB a text tile has its CU to itself, 14 us per layer, and the panoramas run after the tiles; at
// 79.6 KB everything is resident at once and a text tile takes 22 us per layer next to its neighbour.  Kept for the launches that are all
// small samples: twice the workgroups per CU.)
template <int NRT> struct EncLay {
  static constexpr bool COMPACT = NRT <= 3;
  static constexpr int RA = COMPACT ? 48 : MAXROWS;        // rows of the [rows][XS] images and row stride of the partial sums
  static constexpr int KR = COMPACT ? 49 : KROWS;          // rows of the Q|K|V image
  static constexpr int PW = COMPACT ? 72 : PSW;            // probability tile pitch
};
// B fragment from a [k][n] image whose rows >= kmax do not exist: they read row `zrow` (zeros)
template <typename Hh> __device__ __forceinline__ h16x8<Hh> tfrag_z(const Hh* s, int pitch, int n0, int k0, int lane, int kmax, int zrow) {
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  int r0 = k0 + 8 * g + q, r1 = r0 + 4;
  r0 = r0 < kmax ? r0 : zrow; r1 = r1 < kmax ? r1 : zrow;
  const h16x4<Hh> lo = lds_tr4(s + r0 * pitch + n0 + 4 * pp), hi = lds_tr4(s + r1 * pitch + n0 + 4 * pp);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int NRT, typename Hh>
__device__ __forceinline__ void enc_body(const EncParamsT<Hh>& p, const EncSegT<Hh>& sg, const int samp, unsigned char* smem) {
  typedef EncLay<NRT> LY;
  constexpr int RA = LY::RA, KR = LY::KR, PW = LY::PW;
  Hh* sX = (Hh*)smem;                        // [RA][XS]   layer input / residual of the attention block
  Hh* sA = sX + RA * XS;                       // [RA][XS]   attention context, then (after the norm) the FFN's input / residual
  Hh* sQKV = sA + RA * XS;                     // [KR][QS]   Q | K | V
  Hh* sP = sQKV + KR * QS;                     // [8 | 6][16][PW] per-wave probability tiles
  Hh* sG = sQKV;                               // [RA][GS]   GELU output (aliases Q|K|V and the probability tiles, dead by then)
  float* red = LY::COMPACT ? (float*)(sQKV + RA * GS) : (float*)(sP + NWAVE * 16 * PSW);   // [8][RA] LayerNorm partial sums (compact: behind the GELU image)
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
    for (int id = tid; id < (KR - (LY::COMPACT ? 48 : ROWS)) * (384 / 8); id += NWAVE * 64) {      // (compact: the one zero row, 48)
      const int r = (LY::COMPACT ? 48 : ROWS) + id / 48, c = (id % 48) * 8;
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
    Hh* sPw = sP + w * 16 * PW;
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
        for (int r = 0; r < 4; ++r) sPw[(4 * g + r) * PW + j * 16 + c16] = from_f<Hh>(sc[j][r] * sum[r]);
      if (NRT * 16 < NKP) {
        for (int id = lane; id < 16 * (NKP - NRT * 16); id += 64) {
          const int r = id / (NKP - NRT * 16), c = NRT * 16 + id % (NKP - NRT * 16);
          sPw[r * PW + c] = (Hh)0.0f;
        }
      }
      wave_lds_sync();                             // the tile is wave-private: no workgroup barrier needed
      const int nq = min(16, N - rt * 16);
      {
        Hh* Pg = L.P + (((long long)samp * ENH + h) * N + rt * 16) * ldp;
        const int cpr = ldp / 8;
        for (int id = lane; id < nq * cpr; id += 64) {
          const int r = id / cpr, c = (id % cpr) * 8;
          *(h16x8<Hh>*)(Pg + (long long)r * ldp + c) = *(const h16x8<Hh>*)(sPw + r * PW + c);
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
            sPw[ql * PW + key] = from_f<Hh>(sc[j][r] * sum[r] * m);
          }
        }
        wave_lds_sync();
        if (L.Pd) {
          Hh* Pg = L.Pd + (((long long)samp * ENH + h) * N + rt * 16) * ldp;
          const int cpr = ldp / 8;
          for (int id = lane; id < nq * cpr; id += 64) {
            const int r = id / cpr, c = (id % cpr) * 8;
            *(h16x8<Hh>*)(Pg + (long long)r * ldp + c) = *(const h16x8<Hh>*)(sPw + r * PW + c);
          }
        }
      }
      f32x4 o[4];
#pragma unroll
      for (int jd = 0; jd < 4; ++jd) o[jd] = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int ks = 0; ks < NKP / 32; ++ks) {
        const h16x8<Hh> a = lfrag(sPw, PW, 0, ks * 32, lane);
#pragma unroll
        for (int jd = 0; jd < 4; ++jd) o[jd] = emma(a, LY::COMPACT ? tfrag_z(sQKV + 2 * EH + h * EHD, QS, jd * 16, ks * 32, lane, 48, 48)
                                    : tfrag(sQKV + 2 * EH + h * EHD, QS, jd * 16, ks * 32, lane), o[jd]);
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
      add_norm<NRT, RA>(acc, pb_o, pg_1, pe_1, sX, red, sA, L.rstd_a, N, row_base, p.eps, dsh, w, lane);
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
      add_norm<NRT, RA>(acc, pb_2, pg_2, pe_2, sA, red, sX, L.rstd_o, N, row_base, p.eps, dsh, w, lane);
    }
    __syncthreads();                              // sX = block output, complete; every wave is done with the GELU image
    copy_out(sX, XS, L.out + row_base * EH, EH, N, EH, tid);
    ENC_MARK(8);
  }
}

template <typename Hh>
__global__ __launch_bounds__(512) void encoder_fwd_kernel(EncParamsT<Hh> p) {
  enc_mark_start();
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

// =====================================================================================================================================
// Row-split form of the same encoders (round 3).  One 512-thread workgroup owns ONE 16-row tile of ONE sample for all layers, so an
// 80-token instruction is five workgroups instead of one (48 instructions: 240 workgroups instead of 48 on 256 CUs) and a panorama
// three.  A layer's output rows are the only thing the tiles of a sample exchange: every tile needs ALL rows of the layer input for
// its key / value projection (recomputed per tile: 320 of the ~700 MFMAs a tile issues per layer -- the matrix pipe is idle anyway),
// everything else (Q, attention of its 16 queries, both add&norms, the FFN) is per row.  The hand-off stays inside the launch
// (cdna_hip_programming.md section 6, Guideline 16, form R1): the block output is stored write-through (`sc1`), every storing wave
// drains its stores, the workgroup's barrier, then ONE lane adds to the (sample, layer) arrival counter with an agent-scope atomic;
// the next layer starts with one lane polling that counter (relaxed `sc1` loads + s_sleep, bounded) and EVERY load of the handed-off
// rows is an `sc1` buffer load, so no acquire fence is needed.  The counters are zeroed by a memset node in front of the launch.
// Residency: all tiles of a sample must make progress together; the grid puts the text tiles first and is sized by the launch
// function so that every workgroup of the launch is resident at once or the text tiles alone are (2 workgroups per CU: 79.6 KB LDS).
// Same saved tensors, rounding points and dropout masks as encoder_fwd_kernel (tests/test_encoder_gpu.py compares the two).
// =====================================================================================================================================
#define RS_ZROW 80                 // zero row of the V image: the PV product's key steps past the sample's last tile read it
#define RS_SS 84                   // fp32 score row pitch (<= 80 keys); the clean probabilities later reuse the row as 168 16-bit slots
#define RS_PP 104                  // dropped-probability row pitch (<= 96 keys)
#define RS_SPIN_MAX (1u << 21)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(1))) unsigned gu32_t;
// A bounded hand-off wait that ran out (a tile of the sample never arrived: the launch was not fully resident -- a grid larger than the chip, a
// second process holding CUs): the launch's own word (sync[0], zeroed per launch) and a STICKY process-wide count the host reads at a point
// that may synchronise (magic_encoder_health): the rows that were not handed off make the activations wrong, so a trainer must raise.
__device__ unsigned magic_enc_gave_up;
__device__ __forceinline__ void enc_give_up(gu32_t* err) {
  __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_fetch_add(&magic_enc_gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename Hh> __device__ __forceinline__ h16x8<Hh> tfrag_clamp(const Hh* s, int pitch, int n0, int k0, int lane, int kmax) {
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  int r0 = k0 + 8 * g + q, r1 = r0 + 4;
  r0 = r0 < kmax ? r0 : RS_ZROW; r1 = r1 < kmax ? r1 : RS_ZROW;
  const h16x4<Hh> lo = lds_tr4(s + r0 * pitch + n0 + 4 * pp), hi = lds_tr4(s + r1 * pitch + n0 + 4 * pp);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// LayerNorm(acc + bias (dropped) + residual) over the tile's 16 rows; wave w owns columns [16w, 16w + 16).  red: [2][8][16] floats.
template <typename Hh>
__device__ __forceinline__ void add_norm16(f32x4& acc, const float bv, const float gv, const float btv, const Hh* sRes, float* red, Hh* sOut,
                                           float* gRstd, const int nq, const long long row0, const float eps, const DropState& ds, const int w, const int lane) {
  const int g = lane >> 4, c16 = lane & 15, col = w * 16 + c16;
  float s[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * g + r;
    float v = acc[r] + bv;
    if (ds.on) v *= drop_mul(ds, (unsigned)((row0 + row) * EH + col));
    v += to_f(sRes[row * XS + col]);
    acc[r] = v;
    s[r] = g16_sum(v);
  }
  if (c16 == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) red[w * 16 + 4 * g + r] = s[r];
  }
  __syncthreads();
  float mean[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float t = 0.f;
#pragma unroll
    for (int ww = 0; ww < NWAVE; ++ww) t += red[ww * 16 + 4 * g + r];
    mean[r] = t * (1.0f / EH);
    const float d = acc[r] - mean[r];
    s[r] = g16_sum(d * d);
  }
  if (c16 == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) red[128 + w * 16 + 4 * g + r] = s[r];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int rr = 4 * g + r;
    float t = 0.f;
#pragma unroll
    for (int ww = 0; ww < NWAVE; ++ww) t += red[128 + ww * 16 + rr];
    const float rstd = rsqrtf(t * (1.0f / EH) + eps);
    const Hh y = from_f<Hh>((acc[r] - mean[r]) * rstd * gv + btv);
    sOut[rr * XS + col] = (rr < nq) ? y : (Hh)0.0f;
    if (rr < nq && w == 0 && c16 == 0) gRstd[row0 + rr] = rstd;
  }
}

// 16 rows x `cols` from an LDS image to global rows, 16-byte chunks; SC1: write-through stores (the layer hand-off's payload)
template <bool SC1, typename Hh>
__device__ __forceinline__ void copy_tile(const Hh* s, int pitch, Hh* g, int ldg, int rows, int cols, int tid) {
  const int cpr = cols / 8;
  if constexpr (SC1) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, rows * ldg * 2, 0x00020000);
    for (int id = tid; id < rows * cpr; id += NWAVE * 64) {
      const int r = id / cpr, c = (id % cpr) * 8;
      __builtin_amdgcn_raw_buffer_store_b128(*(const u32x4*)(s + r * pitch + c), rsrc, (r * ldg + c) * 2, 0, 16);
    }
  } else {
    for (int id = tid; id < rows * cpr; id += NWAVE * 64) {
      const int r = id / cpr, c = (id % cpr) * 8;
      *(h16x8<Hh>*)(g + (long long)r * ldg + c) = *(const h16x8<Hh>*)(s + r * pitch + c);
    }
  }
}

#ifdef ENC_TIMING
__device__ long long rs_ticks[8][16];         // [layer][mark] of tile 0 of sample 0 of segment 0
#define RS_MARK(i) do { if (samp == 0 && tile == 0 && tid == 0 && &sg == &p.seg[0]) rs_ticks[l][i] = wall_clock64(); } while (0)
#else
#define RS_MARK(i)
#endif
template <int NRT, typename Hh>
__device__ __forceinline__ void enc_rs_body(const EncParamsT<Hh>& p, const EncSegT<Hh>& sg, const int samp, const int tile, unsigned* cnt,
                                            unsigned* err, unsigned char* smem) {
  Hh* sK = (Hh*)smem;                          // [80][XS]  keys of all tiles
  Hh* sV = sK + 80 * XS;                       // [81][XS]  values of all tiles; row 80 stays zero
  Hh* sU = sV + 81 * XS;                       // union, 80 * XS elements:
  Hh* sX = sU;                                 //   [16 NRT][XS]  layer input, all rows            (stage A)
  float* sS = (float*)sU;                      //   [2][16][RS_SS] fp32 scores, then the clean probabilities in place    (stage B)
  Hh* sPd = sU + 2 * 16 * RS_SS * 2;           //   [2][16][RS_PP] dropped probabilities                                  (stage B)
  Hh* sG = sU;                                 //   [16][GS]      GELU output                       (stages D, E)
  Hh* sZ = sK;                                 // [16][GS] FFN pre-activation (the K image is dead by then)
  Hh* sQ = sU + 80 * XS;                       // [16][XS]  Q, then the attention context
  Hh* sO = sQ + 16 * XS;                       // [16][XS]  own rows of the layer input (residual); then the layer output
  Hh* sA = sO + 16 * XS;                       // [16][XS]  attention-block output
  float* red = (float*)(sA + 16 * XS);         // [2][8][16]
  const int tid = threadIdx.x, lane0 = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N = sg.N, ldp = sg.ldp;
  const int NKP = (N + 31) / 32 * 32;
  const long long row_base = (long long)samp * N;
  const int r0 = tile * 16, nq = min(16, N - r0);
  const long long row0 = row_base + r0;
  // zero row of the V image + the V rows the last tile does not cover are written once per layer below (bias rows are finite)
  if (tid < XS / 8) {
    h16x8<Hh> zv;
#pragma unroll
    for (int e = 0; e < 8; ++e) zv[e] = (Hh)0.0f;
    *(h16x8<Hh>*)(sV + RS_ZROW * XS + tid * 8) = zv;
  }
  DropDesc dd;
  dd.seed = p.seed;
  float kbias[NRT];
#pragma unroll
  for (int j = 0; j < NRT; ++j) {
    const int key = j * 16 + (lane0 & 15);
    kbias[j] = (key < N && sg.kmask && !sg.kmask[(long long)samp * N + key]) ? -10000.0f : 0.f;
  }
  for (int l = 0; l < sg.nlayers; ++l) {
    const EncLayerT<Hh>& L = sg.L[l];
    int lane = lane0;
    asm volatile("" : "+v"(lane));             // (see enc_body: keeps the layer-invariant per-lane indices from being hoisted and spilled)
    const int g = lane >> 4, c16 = lane & 15;
    RS_MARK(0);
    // ---- small parameters of the layer's epilogues, fetched now ----
    const float pb_q = L.bqkv[w * 16 + c16], pb_kv0 = L.bqkv[EH + (2 * w) * 16 + c16], pb_kv1 = L.bqkv[EH + (2 * w + 1) * 16 + c16];
    float pb_ffn[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) pb_ffn[ct] = L.bi[(4 * w + ct) * 16 + c16];
    const float pb_o = L.bo[w * 16 + c16], pg_1 = L.g1[w * 16 + c16], pe_1 = L.be1[w * 16 + c16];
    const float pb_2 = L.bo2[w * 16 + c16], pg_2 = L.g2[w * 16 + c16], pe_2 = L.be2[w * 16 + c16];
    // weight fragments of stage A: this wave's Q column tile and its two K|V column tiles
    h16x8<Hh> wq[4], wkv[2][4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      wq[ks] = gfrag(L.Wqkv, EH, w * 16, ks * 32, lane);
      wkv[0][ks] = gfrag(L.Wqkv, EH, EH + (2 * w) * 16, ks * 32, lane);
      wkv[1][ks] = gfrag(L.Wqkv, EH, EH + (2 * w + 1) * 16, ks * 32, lane);
    }
    // ---- layer input: all rows of the sample -> sX.  Layer 0 reads what an earlier launch wrote; later layers read the rows the sample's
    //      other tiles handed off (sc1 loads, after the arrival counter of the previous layer has reached the tile count)
    if (l > 0) {
      if (tid == 0) {
        gu32_t* c = (gu32_t*)(cnt + (long long)samp * 6 + (l - 1));
        unsigned spins = 0;
        while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)NRT) {
          __builtin_amdgcn_s_sleep(4);
          if (++spins > RS_SPIN_MAX) { enc_give_up((gu32_t*)err); break; }
        }
      }
      __syncthreads();
    }
    RS_MARK(1);
    {
      const Hh* x = (l == 0 ? sg.x : sg.L[l - 1].out) + row_base * EH;
      const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, N * EH * 2, 0x00020000);     // rows >= N: out of range -> zeros
      for (int id = tid; id < NRT * 16 * (EH / 8); id += NWAVE * 64) {
        const int r = id / (EH / 8), c = (id % (EH / 8)) * 8;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (r * EH + c) * 2, 0, 16);
        *(u32x4*)(sX + r * XS + c) = v;
        if (r >= r0 && r < r0 + 16) *(u32x4*)(sO + (r - r0) * XS + c) = v;
      }
    }
    __syncthreads();
    RS_MARK(2);
    // ================= A: K|V of all rows (2 of 16 column tiles per wave), Q of the own tile (1 of 8) =================
    {
      f32x4 acc[NRT][2], aq = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < NRT; ++i) { acc[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int i = 0; i < NRT; ++i) {
          const h16x8<Hh> a = lfrag(sX, XS, i * 16, ks * 32, lane);
          acc[i][0] = emma(a, wkv[0][ks], acc[i][0]);
          acc[i][1] = emma(a, wkv[1][ks], acc[i][1]);
        }
        aq = emma(lfrag(sO, XS, 0, ks * 32, lane), wq[ks], aq);
        KSTEP_FENCE();
      }
      // K column tiles 0..7 come from waves 0..3, V from waves 4..7
      Hh* dst = (w < 4 ? sK : sV) + ((2 * w) & 7) * 16 + c16;
#pragma unroll
      for (int i = 0; i < NRT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dst[(i * 16 + 4 * g + r) * XS] = from_f<Hh>(acc[i][0][r] + pb_kv0);
          dst[(i * 16 + 4 * g + r) * XS + 16] = from_f<Hh>(acc[i][1][r] + pb_kv1);
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) sQ[(4 * g + r) * XS + w * 16 + c16] = from_f<Hh>(aq[r] + pb_q);
    }
    // fragments of the output projection and the first FFN matrix: in flight under the attention
    h16x8<Hh> wo[4], w1[4][4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) wo[ks] = gfrag(L.Wo, EH, w * 16, ks * 32, lane);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) w1[ct][ks] = gfrag(L.W1, EH, (4 * w + ct) * 16, ks * 32, lane);
    __syncthreads();                              // K, V, Q images complete; sX is dead
    RS_MARK(3);
    // own rows of Q|K|V -> global (the backward reads qkv [M, 3H])
    {
      Hh* qg = L.qkv + row0 * 3 * EH;
      for (int id = tid; id < nq * 48; id += NWAVE * 64) {
        const int r = id / 48, c = (id % 48) * 8;
        const Hh* src = c < EH ? sQ + r * XS + c : (c < 2 * EH ? sK + (r0 + r) * XS + (c - EH) : sV + (r0 + r) * XS + (c - 2 * EH));
        *(h16x8<Hh>*)(qg + (long long)r * 3 * EH + c) = *(const h16x8<Hh>*)src;
      }
    }
    // ================= B1: scores of the tile's 16 queries, unit = (head, key tile) =================
    for (int u = w; u < ENH * NRT; u += NWAVE) {
      const int h = u / NRT, j = u % NRT;
      f32x4 sc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) sc = emma(lfrag(sQ, XS, 0, h * EHD + ks * 32, lane), lfrag(sK, XS, j * 16, h * EHD + ks * 32, lane), sc);
      const int key = j * 16 + c16;
      float mb = 0.f;
#pragma unroll
      for (int jj = 0; jj < NRT; ++jj) mb = (jj == j) ? kbias[jj] : mb;
#pragma unroll
      for (int r = 0; r < 4; ++r) sS[(h * 16 + 4 * g + r) * RS_SS + key] = key < N ? sc[r] * p.scale + mb : -3.0e38f;
    }
    __syncthreads();
    RS_MARK(4);
    // ================= B2: softmax, 4 of the 32 (head, query) rows per wave, 16 lanes per row =================
    dd.site = L.site_attn; dd.p = p.p_attn;
    const DropState dsa = drop_init(dd);
    {
      const int rr = 4 * w + g, h = rr >> 4, ql = rr & 15;
      float e[NRT], mx = -3.0e38f;
#pragma unroll
      for (int j = 0; j < NRT; ++j) { e[j] = sS[rr * RS_SS + j * 16 + c16]; mx = fmaxf(mx, e[j]); }
      mx = g16_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < NRT; ++j) { e[j] = (j * 16 + c16) < N ? __expf(e[j] - mx) : 0.f; sum += e[j]; }
      sum = 1.0f / g16_sum(sum);
      Hh* pc = (Hh*)(sS + rr * RS_SS);          // the row's own bytes: only this wave reads or writes them in this phase
      Hh* pd = sPd + rr * RS_PP;
#pragma unroll
      for (int j = 0; j < NRT; ++j) {
        const int key = j * 16 + c16;
        const float pv = e[j] * sum;
        pc[key] = from_f<Hh>(pv);
        if (dsa.on) {
          const unsigned idx = (unsigned)(((((long long)samp * ENH + h) * N + r0 + ql) * N) + key);
          pd[key] = from_f<Hh>((ql < nq && key < N) ? pv * drop_mul(dsa, idx) : 0.f);
        }
      }
      if (NRT * 16 < NKP && c16 < NKP - NRT * 16) { pc[NRT * 16 + c16] = (Hh)0.0f; if (dsa.on) pd[NRT * 16 + c16] = (Hh)0.0f; }
    }
    __syncthreads();
    RS_MARK(5);
    // probabilities -> global (clean: the backward's P; dropped: the exposed attention map)
    {
      const int cpr = ldp / 8;
      for (int id = tid; id < ENH * nq * cpr; id += NWAVE * 64) {
        const int h = id / (nq * cpr), rem = id % (nq * cpr), r = rem / cpr, c = (rem % cpr) * 8;
        const long long go = (((long long)samp * ENH + h) * N + r0 + r) * ldp + c;
        *(h16x8<Hh>*)(L.P + go) = *(const h16x8<Hh>*)((const Hh*)(sS + (h * 16 + r) * RS_SS) + c);
        if (dsa.on && L.Pd) *(h16x8<Hh>*)(L.Pd + go) = *(const h16x8<Hh>*)(sPd + (h * 16 + r) * RS_PP + c);
      }
    }
    // ================= B3: context = P V, unit = (head, 16 of the head's 64 columns): one per wave =================
    {
      const int h = w >> 2, jd = w & 3;
      const Hh* pa = dsa.on ? sPd + h * 16 * RS_PP : (const Hh*)(sS + h * 16 * RS_SS);
      const int pp = dsa.on ? RS_PP : 2 * RS_SS;
      f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int ks = 0; ks < NKP / 32; ++ks)
        o = emma(lfrag(pa, pp, 0, ks * 32, lane), tfrag_clamp(sV + h * EHD, XS, jd * 16, ks * 32, lane, NRT * 16), o);
#pragma unroll
      for (int r = 0; r < 4; ++r) sQ[(4 * g + r) * XS + h * EHD + jd * 16 + c16] = from_f<Hh>(o[r]);      // Q is dead: the context takes its place
    }
    __syncthreads();
    RS_MARK(6);
    copy_tile<false>(sQ, XS, L.ctx + row0 * EH, EH, nq, EH, tid);
    // ================= C: a = LayerNorm(x + dropout(ctx Wo^T + bo)) =================
    {
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) acc = emma(lfrag(sQ, XS, 0, ks * 32, lane), wo[ks], acc);
      dd.site = L.site_ao; dd.p = p.p_hidden;
      const DropState dsh = drop_init(dd);
      add_norm16(acc, pb_o, pg_1, pe_1, sO, red, sA, L.rstd_a, nq, row0, p.eps, dsh, w, lane);
    }
    __syncthreads();
    RS_MARK(7);
    copy_tile<false>(sA, XS, L.a + row0 * EH, EH, nq, EH, tid);
    // ================= D: z = a W1^T + bi ; g = gelu(z) =================
    h16x8<Hh> w2[16];
    {
      f32x4 acc[4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const h16x8<Hh> a = lfrag(sA, XS, 0, ks * 32, lane);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = emma(a, w1[ct][ks], acc[ct]);
      }
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) w2[ks] = gfrag(L.W2, EI, w * 16, ks * 32, lane);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const int col = (4 * w + ct) * 16 + c16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float zv = acc[ct][r] + pb_ffn[ct];
          sZ[(4 * g + r) * GS + col] = from_f<Hh>(zv);
          sG[(4 * g + r) * GS + col] = from_f<Hh>(gelu_fast(zv));
        }
      }
    }
    __syncthreads();
    RS_MARK(8);
    copy_tile<false>(sZ, GS, L.z + row0 * EI, EI, nq, EI, tid);
    copy_tile<false>(sG, GS, L.g + row0 * EI, EI, nq, EI, tid);
    // ================= E: out = LayerNorm(a + dropout(g W2^T + bo2)) =================
    {
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) acc = emma(lfrag(sG, GS, 0, ks * 32, lane), w2[ks], acc);
      dd.site = L.site_out; dd.p = p.p_hidden;
      const DropState dsh = drop_init(dd);
      add_norm16(acc, pb_2, pg_2, pe_2, sA, red, sO, L.rstd_o, nq, row0, p.eps, dsh, w, lane);
    }
    __syncthreads();
    RS_MARK(9);
    // ---- hand-off: the tile's output rows, write-through; every wave drains its stores; one arrival per workgroup ----
    copy_tile<true>(sO, XS, L.out + row0 * EH, EH, nq, EH, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                               // (also: every wave is done with sG / sZ / sK before the next layer overwrites them)
    RS_MARK(10);
    if (tid == 0 && l + 1 < sg.nlayers) __hip_atomic_fetch_add((gu32_t*)(cnt + (long long)samp * 6 + l), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// every segment row-split: nt0 / nt1 = 16-row tiles per sample
template <typename Hh>
__global__ __launch_bounds__(512) void encoder_rs_kernel(EncParamsT<Hh> p, int nt0, int nt1) {
  enc_mark_start();
  extern __shared__ __attribute__((aligned(16))) unsigned char enc_smem[];
  int b = blockIdx.x, s = 0, nt = nt0;
  unsigned* cnt = p.sync + 4;
  if (b >= p.seg[0].nsamp * nt0) { b -= p.seg[0].nsamp * nt0; s = 1; nt = nt1; cnt += 6ll * p.seg[0].nsamp; }
  const EncSegT<Hh>& sg = p.seg[s];
  const int samp = b / nt, tile = b - samp * nt;
  switch (nt) {
    case 1: enc_rs_body<1>(p, sg, samp, tile, cnt, p.sync, enc_smem); break;
    case 2: enc_rs_body<2>(p, sg, samp, tile, cnt, p.sync, enc_smem); break;
    case 3: enc_rs_body<3>(p, sg, samp, tile, cnt, p.sync, enc_smem); break;
    case 4: enc_rs_body<4>(p, sg, samp, tile, cnt, p.sync, enc_smem); break;
    default: enc_rs_body<5>(p, sg, samp, tile, cnt, p.sync, enc_smem); break;
  }
}
// The headline launch: segment 0 (the text encoder, 4-5 tiles per instruction) row-split, its tiles first in the grid and all resident
// together; segment 1 (the panorama encoder, <= 48 rows) one workgroup per sample (enc_body), which never waits.  k0 = 4 | 5 tiles,
// k1 = -2 | -3 row tiles.  ONE flat switch over FOUR instantiations: with more bodies inlined into one kernel (or a nested
// `if (row_split) switch ... else switch ...`) hipcc keeps the 2.5 KB parameter block in scratch memory and the launch runs 1.7x slower.
template <typename Hh>
__global__ __launch_bounds__(512) void encoder_mix_kernel(EncParamsT<Hh> p, int k0, int k1) {
  enc_mark_start();
  extern __shared__ __attribute__((aligned(16))) unsigned char enc_smem[];
  int b = blockIdx.x, s = 0, nt = k0;
  unsigned* cnt = p.sync + 4;
  const int blocks0 = p.seg[0].nsamp * k0;
  // (measured and not kept: panoramas first in the grid, 1.567 -> 1.582 ms/step; text tiles and panoramas alternating, 1.57 -> 1.64)
  if (b >= blocks0) { b -= blocks0; s = 1; nt = k1; }
  const EncSegT<Hh>& sg = p.seg[s];
  const int na = nt > 0 ? nt : 1;
  const int samp = b / na, tile = b - samp * na;
  switch (nt) {
    case 4: enc_rs_body<4>(p, sg, samp, tile, cnt, p.sync, enc_smem); break;
    case 5: enc_rs_body<5>(p, sg, samp, tile, cnt, p.sync, enc_smem); break;
    case -3: enc_body<3>(p, sg, samp, enc_smem); break;
    default: enc_body<2>(p, sg, samp, enc_smem); break;
  }
}
static size_t enc_rs_lds_bytes() { return (size_t)(80 * XS + 81 * XS + 80 * XS + 3 * 16 * XS) * 2 + 256 * sizeof(float); }

static size_t enc_lds_bytes() {
  return (size_t)(2 * MAXROWS * XS + KROWS * QS + NWAVE * 16 * PSW) * 2 + (size_t)NWAVE * MAXROWS * sizeof(float);
}
static size_t enc_lds_bytes_compact() { return (size_t)(2 * 48 * XS + 49 * QS + 6 * 16 * 72) * 2; }      // EncLay<NRT <= 3>

// Nothing in HIP promises that two streams run side by side (they may share a hardware queue; a profiler may serialise them), and a gate
// that starts AFTER the launch it waits for would sleep for nothing.  So the gate (i) passes at once when a whole-encoder launch became
// resident within the last `recent_ticks` (that IS the launch it was meant to follow), (ii) counts what it does in `st`, and (iii) switches
// ITSELF off after GATE_MAX_CONSEC consecutive timeouts: from then on it is an empty launch -- and re-arms itself after a backoff of skipped calls
// (32, doubling to 1024 each time it has to switch off again, back to 32 once it opens): a TRANSIENT loss of overlap -- RCCL building its
// communicator in the first data-parallel steps, a profiler attached for a while -- no longer leaves the teacher ungated for the rest of the run
// (round 5: the data-parallel structure ran at 2.15 ms/step for that reason alone), a PERSISTENT one costs three timeouts per 1024 steps.
// st: 0 calls | 1 opened by a launch while waiting | 2 launch already resident at entry | 3 timeouts | 4 consecutive timeouts | 5 disabled
//     | 6 calls skipped while disabled | 7 (backoff period << 16) | skipped calls of the current backoff
#define GATE_MAX_CONSEC 3
__global__ void enc_start_gate_kernel(long long timeout_ticks, long long recent_ticks, unsigned* st) {
  if (threadIdx.x) return;
  st[0] += 1;
  if (st[5]) {
    st[6] += 1;
    unsigned period = st[7] >> 16, n = (st[7] & 0xFFFFu) + 1;
    if (period == 0) period = 32;
    if (n >= period) { st[5] = 0; st[4] = 0; st[7] = (period < 1024 ? period * 2 : 1024) << 16; }      // re-arm for the next call; the NEXT switch-off waits twice as long
    else st[7] = (period << 16) | n;
    return;
  }
  const unsigned c0 = __hip_atomic_load(&magic_enc_starts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const long long t0 = wall_clock64();                       // 100 MHz
  const long long tm = __hip_atomic_load(&magic_enc_start_tick, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (c0 != 0 && t0 - tm >= 0 && t0 - tm < recent_ticks) { st[2] += 1; st[4] = 0; st[7] = 0; return; }
  while (__hip_atomic_load(&magic_enc_starts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == c0) {
    if (wall_clock64() - t0 >= timeout_ticks) {
      st[3] += 1;
      if (++st[4] >= GATE_MAX_CONSEC) st[5] = 1;
      return;
    }
    __builtin_amdgcn_s_sleep(32);
  }
  st[1] += 1;
  st[4] = 0;
  st[7] = 0;                         // overlap is back: the next switch-off starts from the shortest backoff again
}
// park `stream` (one sleeping wave) until the next whole-encoder launch of this process has all its workgroups on CUs, at most timeout_us
extern "C" int magic_encoder_start_gate(int timeout_us, int recent_us, unsigned* stats, void* stream) {
  if (timeout_us < 0 || timeout_us > 100000 || recent_us < 0 || recent_us > 100000 || !stats) return MAGIC_ERR_ARG;
  hipLaunchKernelGGL(enc_start_gate_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)timeout_us * 100, (long long)recent_us * 100, stats);
  return launch_status();
}

// Do two streams run SIDE BY SIDE?  The runtime deals its streams onto a handful of hardware queues (GPU_MAX_HW_QUEUES, 4 by default) and
// two streams that land on one queue execute in order.  w[0]: the flag, w[1]: 1 = the waiter saw the flag, 2 = it ran out of time.
__global__ void stream_probe_wait_kernel(long long timeout_ticks, unsigned* w) {
  if (threadIdx.x) return;
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(&w[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
    if (wall_clock64() - t0 >= timeout_ticks) { w[1] = 2; return; }
    __builtin_amdgcn_s_sleep(32);
  }
  w[1] = 1;
}
__global__ void stream_probe_set_kernel(unsigned* w) {
  if (threadIdx.x == 0) __hip_atomic_store(&w[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// One waiter (a sleeping wave, at most timeout_us) on `stream_wait`, then the launch that releases it on `stream_set`: on streams that
// share a hardware queue the release sits behind the waiter and the waiter times out.  word2: 2 x uint32 of ZEROED device memory; the caller
// synchronises both streams and reads word2[1] (1 = side by side, 2 = in order).  Both streams must be idle when this is called.
extern "C" int magic_stream_probe(unsigned* word2, int timeout_us, void* stream_wait, void* stream_set) {
  if (!word2 || timeout_us < 1 || timeout_us > 100000 || stream_wait == stream_set) return MAGIC_ERR_ARG;
  hipLaunchKernelGGL(stream_probe_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_wait, (long long)timeout_us * 100, word2);
  if (int rc = launch_status()) return rc;
  hipLaunchKernelGGL(stream_probe_set_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_set, word2);
  return launch_status();
}

__global__ void enc_health_kernel(unsigned* out) {
  if (threadIdx.x == 0) {
    out[0] = __hip_atomic_load(&magic_enc_gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    out[1] = __hip_atomic_load(&magic_enc_starts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// out[0] = bounded hand-off waits of the row-split encoder kernels that gave up since the process started (must stay 0), out[1] = whole-encoder
// launches that became resident.  One tiny launch on `stream`; the caller reads `out` (2 x uint32, device) where it may synchronise.
extern "C" int magic_encoder_health(unsigned* out, void* stream) {
  if (!out) return MAGIC_ERR_ARG;
  hipLaunchKernelGGL(enc_health_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
  return launch_status();
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
  if (p.sync) {           // row-split form: one workgroup per (sample, 16-row tile)
    const int nt0 = (p.seg[0].N + 15) / 16, nt1 = p.nseg > 1 ? (p.seg[1].N + 15) / 16 : 1;
    const int ns1 = p.nseg > 1 ? p.seg[1].nsamp : 0;
    const long long words = 4 + 6ll * (p.seg[0].nsamp + ns1);
    if (p.sync_words < words || ((uintptr_t)p.sync & 15)) return MAGIC_ERR_ARG;
    // a segment runs row-split when all its tiles are resident together with one workgroup per CU, and only segments that come out ahead:
    // >= 4 tiles per sample (a 36-view panorama is 3 tiles, the third almost empty, and its per-sample workgroups already fill the chip)
    static int ncu = 0;
    if (!ncu) { hipDeviceProp_t pr; int d = 0; (void)hipGetDevice(&d); ncu = (hipGetDeviceProperties(&pr, d) == hipSuccess) ? pr.multiProcessorCount : 256; }
    static int force = -2;
    if (force == -2) { const char* e = getenv("MAGIC_ENC_RS_ALL"); force = e ? atoi(e) : -1; }      // 1: every segment row-split (tests, measurements)
    const bool rs0 = nt0 >= 4 && p.seg[0].nsamp * nt0 <= ncu, rs1 = p.nseg > 1 && nt1 >= 4 && p.seg[0].nsamp * nt0 + ns1 * nt1 <= ncu;
    const int form = (force == 1 || (rs0 && (rs1 || p.nseg == 1))) ? 2 : (rs0 && p.nseg == 2 && nt1 <= 3) ? 1 : 0;      // 2: all row-split, 1: mixed, 0: per sample
    static bool rs_attr = false;
    if (!rs_attr) {
      (void)hipFuncSetAttribute((const void*)encoder_rs_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)enc_rs_lds_bytes());
      (void)hipFuncSetAttribute((const void*)encoder_rs_kernel<f16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)enc_rs_lds_bytes());
      (void)hipFuncSetAttribute((const void*)encoder_mix_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)enc_lds_bytes());
      (void)hipFuncSetAttribute((const void*)encoder_mix_kernel<f16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)enc_lds_bytes());
      rs_attr = true;
    }
    if (form) {
      // every polled word is zeroed in front of EVERY launch (a memset node under graph capture); a multiple of 16 bytes from the allocation's start
      if (hipMemsetAsync(p.sync, 0, (size_t)((words + 3) / 4 * 4) * sizeof(unsigned), (hipStream_t)stream) != hipSuccess) return MAGIC_ERR_LAUNCH;
      EncParamsT<f16> pf;
      static_assert(sizeof(pf) == sizeof(p), "layout");
      memcpy(&pf, &p, sizeof(pf));
      if (form == 2) {
        const int grid = p.seg[0].nsamp * nt0 + ns1 * nt1;
        if (dtype == DT_BF16) hipLaunchKernelGGL(encoder_rs_kernel<bf16>, dim3(grid), dim3(512), enc_rs_lds_bytes(), (hipStream_t)stream, p, nt0, nt1);
        else hipLaunchKernelGGL(encoder_rs_kernel<f16>, dim3(grid), dim3(512), enc_rs_lds_bytes(), (hipStream_t)stream, pf, nt0, nt1);
      } else {
        const int grid = p.seg[0].nsamp * nt0 + ns1, k1 = -(nt1 < 2 ? 2 : nt1);
        // the panoramas' per-sample body takes its compact layout (nt1 <= 3): both bodies fit twice into a CU's LDS
        const size_t shm_mix = enc_rs_lds_bytes() > enc_lds_bytes_compact() ? enc_rs_lds_bytes() : enc_lds_bytes_compact();
        if (dtype == DT_BF16) hipLaunchKernelGGL(encoder_mix_kernel<bf16>, dim3(grid), dim3(512), shm_mix, (hipStream_t)stream, p, nt0, k1);
        else hipLaunchKernelGGL(encoder_mix_kernel<f16>, dim3(grid), dim3(512), shm_mix, (hipStream_t)stream, pf, nt0, k1);
      }
      return launch_status();
    }
    if (hipMemsetAsync(p.sync, 0, 16, (hipStream_t)stream) != hipSuccess) return MAGIC_ERR_LAUNCH;      // per-sample form: the give-up word still reads 0
  }
  bool all_small = true;        // every sample <= 48 rows: the compact layout, two workgroups per CU
  for (int sgi = 0; sgi < p.nseg; ++sgi) all_small = all_small && p.seg[sgi].N <= 48;
  const size_t shm = all_small ? enc_lds_bytes_compact() : enc_lds_bytes();
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)encoder_fwd_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)enc_lds_bytes());
    (void)hipFuncSetAttribute((const void*)encoder_fwd_kernel<f16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)enc_lds_bytes());
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
template <typename Hh> struct XParamsT {
  XSegT<Hh> seg[2]; int nseg; float p_attn, p_hidden, eps, scale; const unsigned* seed;
  unsigned* sync; int sync_words, pad2_;        // row-split form (as EncParamsT): >= 4 + 6 * (samples of all segments) words
};
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
      add_norm<NRT, MAXROWS>(acc, pb_o, pg_1, pe_1, sX, red, sA, L.rstd_a, Nq, qbase, p.eps, dsh, w, lane);
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
      add_norm<NRT, MAXROWS>(acc, pb_oc, pg_c, pe_c, sA, red, sX, L.rstd_c, Nq, qbase, p.eps, dsh, w, lane);
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
      add_norm<NRT, MAXROWS>(acc, pb_2, pg_2, pe_2, sX, red, sX, L.rstd_o, Nq, qbase, p.eps, dsh, w, lane);
    }
    __syncthreads();
    copy_out(sX, XS, L.out + qbase * EH, EH, Nq, EH, tid);
  }
}

// =====================================================================================================================================
// Row-split form of the cross-modal encoders (round 3; construction and hand-off protocol: encoder_rs_kernel above).  One workgroup per
// (sample, 16-row tile of the queries): a 20-node map is two workgroups, a 37-token viewpoint three, an 80-token instruction (the MLM
// text-attends-map path) five -- 240 workgroups for B = 48 instead of 96 (or 48).  Per layer a tile recomputes the key / value projections of
// ALL its encoder's rows (self-attention) and of the CONTEXT's <= 80 rows (cross-attention), and does everything else for its own 16 rows.
// =====================================================================================================================================
// attention of the tile's 16 queries (sQ: [16][XS], both heads) against NKT key tiles of the K / V images, on all 8 waves:
//   phase 1 scores per (head, key tile) -> sS ; phase 2 softmax, 4 (head, query) rows per wave -> clean probabilities in place of the
//   scores (+ dropped probabilities in sPd) ; phase 3 P V per (head, 16 context columns) -> sQ (Q is dead by then).
// Writes P (and Pd) rows of the tile to global.  dist: optional graph-distance bias rows [Nq_total, Nk] of this sample (global encoder).
template <int NKT, typename Hh>
__device__ __forceinline__ void rs_attn(Hh* sQ, const Hh* sK, const Hh* sV, float* sS, Hh* sPd, const float (&kbias)[NKT], const int Nk, const int kmax,
                                        const float* dist, const float sw, const float sb, const int q0, const int nq, const int Nq,
                                        Hh* Pg, Hh* Pdg, const int ldp, const long long samp, const float scale, const DropState& dsa,
                                        const int tid, const int w, const int lane) {
  const int g = lane >> 4, c16 = lane & 15;
  const int NKP = (Nk + 31) / 32 * 32;
  for (int u = w; u < ENH * NKT; u += NWAVE) {
    const int h = u / NKT, j = u % NKT;
    f32x4 sc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) sc = emma(lfrag(sQ, XS, 0, h * EHD + ks * 32, lane), lfrag(sK, XS, j * 16, h * EHD + ks * 32, lane), sc);
    const int key = j * 16 + c16;
    float mb = 0.f;
#pragma unroll
    for (int jj = 0; jj < NKT; ++jj) mb = (jj == j) ? kbias[jj] : mb;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float x = sc[r] * scale + mb;
      if (dist && key < Nk) { const int qrow = min(q0 + 4 * g + r, Nq - 1); x += sw * dist[((long long)samp * Nq + qrow) * Nk + key] + sb; }
      sS[(h * 16 + 4 * g + r) * RS_SS + key] = key < Nk ? x : -3.0e38f;
    }
  }
  __syncthreads();
  {
    const int rr = 4 * w + g, h = rr >> 4, ql = rr & 15;
    float e[NKT], mx = -3.0e38f;
#pragma unroll
    for (int j = 0; j < NKT; ++j) { e[j] = sS[rr * RS_SS + j * 16 + c16]; mx = fmaxf(mx, e[j]); }
    mx = g16_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NKT; ++j) { e[j] = (j * 16 + c16) < Nk ? __expf(e[j] - mx) : 0.f; sum += e[j]; }
    sum = 1.0f / g16_sum(sum);
    Hh* pc = (Hh*)(sS + rr * RS_SS);
    Hh* pd = sPd + rr * RS_PP;
#pragma unroll
    for (int j = 0; j < NKT; ++j) {
      const int key = j * 16 + c16;
      const float pv = e[j] * sum;
      pc[key] = from_f<Hh>(pv);
      if (dsa.on) {
        const unsigned idx = (unsigned)(((((long long)samp * ENH + h) * Nq + q0 + ql) * Nk) + key);
        pd[key] = from_f<Hh>((ql < nq && key < Nk) ? pv * drop_mul(dsa, idx) : 0.f);
      }
    }
    if (NKT * 16 < NKP && c16 < NKP - NKT * 16) { pc[NKT * 16 + c16] = (Hh)0.0f; if (dsa.on) pd[NKT * 16 + c16] = (Hh)0.0f; }
  }
  __syncthreads();
  {
    const int cpr = ldp / 8;
    for (int id = tid; id < ENH * nq * cpr; id += NWAVE * 64) {
      const int h = id / (nq * cpr), rem = id % (nq * cpr), r = rem / cpr, c = (rem % cpr) * 8;
      const long long go = (((long long)samp * ENH + h) * Nq + q0 + r) * ldp + c;
      *(h16x8<Hh>*)(Pg + go) = *(const h16x8<Hh>*)((const Hh*)(sS + (h * 16 + r) * RS_SS) + c);
      if (dsa.on && Pdg) *(h16x8<Hh>*)(Pdg + go) = *(const h16x8<Hh>*)(sPd + (h * 16 + r) * RS_PP + c);
    }
  }
  {
    const int h = w >> 2, jd = w & 3;
    const Hh* pa = dsa.on ? sPd + h * 16 * RS_PP : (const Hh*)(sS + h * 16 * RS_SS);
    const int pp = dsa.on ? RS_PP : 2 * RS_SS;
    f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < NKP / 32; ++ks)
      o = emma(lfrag(pa, pp, 0, ks * 32, lane), tfrag_clamp(sV + h * EHD, XS, jd * 16, ks * 32, lane, kmax), o);
#pragma unroll
    for (int r = 0; r < 4; ++r) sQ[(4 * g + r) * XS + h * EHD + jd * 16 + c16] = from_f<Hh>(o[r]);
  }
  __syncthreads();
}

template <int NRT, typename Hh>
__device__ __forceinline__ void xenc_rs_body(const XParamsT<Hh>& p, const XSegT<Hh>& sg, const int samp, const int tile, const int ntile,
                                             unsigned* cnt, unsigned* err, unsigned char* smem) {
  // NRT: compile-time bound on the encoder's row tiles (3 or 5; tiles past the sample are zero rows), ntile: the real count (the hand-off waits for it)
  constexpr int NKT = 5;                       // context key tiles (<= 80 rows)
  Hh* sK = (Hh*)smem;                          // [80][XS]  self keys, then the context's keys
  Hh* sV = sK + 80 * XS;                       // [81][XS]  self values, then the context's; row 80 stays zero
  Hh* sU = sV + 81 * XS;                       // union (80 * XS elements): layer input (all rows) | scores + probabilities | GELU output
  Hh* sX = sU;
  float* sS = (float*)sU;
  Hh* sPd = sU + 2 * 16 * RS_SS * 2;
  Hh* sG = sU;
  Hh* sZ = sK;                                 // [16][GS] FFN pre-activation (K image dead by then)
  Hh* sQ = sU + 80 * XS;                       // [16][XS] Q -> context (self), then Q -> context (cross)
  Hh* sO = sQ + 16 * XS;                       // [16][XS] own rows of the layer input; then c (cross-attention block output)
  Hh* sA = sO + 16 * XS;                       // [16][XS] a (self-attention block output); then the layer output
  float* red = (float*)(sA + 16 * XS);
  const int tid = threadIdx.x, lane0 = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Nq = sg.Nq, Nk = sg.Nk;
  const long long qbase = (long long)samp * Nq, kbase = (long long)samp * Nk;
  const int r0 = tile * 16, nq = min(16, Nq - r0);
  const long long row0 = qbase + r0;
  if (tid < XS / 8) {
    h16x8<Hh> zv;
#pragma unroll
    for (int e = 0; e < 8; ++e) zv[e] = (Hh)0.0f;
    *(h16x8<Hh>*)(sV + RS_ZROW * XS + tid * 8) = zv;
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
    const float pb_q0 = L.bqkv[colw], pb_kv0 = L.bqkv[EH + (2 * w) * 16 + c16], pb_kv1 = L.bqkv[EH + (2 * w + 1) * 16 + c16];
    const float pb_ck0 = L.bkv[(2 * w) * 16 + c16], pb_ck1 = L.bkv[(2 * w + 1) * 16 + c16];
    float pb_ffn[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) pb_ffn[ct] = L.bi[(4 * w + ct) * 16 + c16];
    const float pb_o = L.bo[colw], pg_1 = L.g1[colw], pe_1 = L.be1[colw], pb_q = L.bq[colw];
    const float pb_oc = L.boc[colw], pg_c = L.gc[colw], pe_c = L.bec[colw];
    const float pb_2 = L.bo2[colw], pg_2 = L.g2[colw], pe_2 = L.be2[colw];
    h16x8<Hh> wq[4], wkv[2][4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      wq[ks] = gfrag(L.Wqkv, EH, w * 16, ks * 32, lane);
      wkv[0][ks] = gfrag(L.Wqkv, EH, EH + (2 * w) * 16, ks * 32, lane);
      wkv[1][ks] = gfrag(L.Wqkv, EH, EH + (2 * w + 1) * 16, ks * 32, lane);
    }
    if (l > 0) {
      if (tid == 0) {
        gu32_t* c = (gu32_t*)(cnt + (long long)samp * 6 + (l - 1));
        unsigned spins = 0;
        while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)ntile) {
          __builtin_amdgcn_s_sleep(4);
          if (++spins > RS_SPIN_MAX) { enc_give_up((gu32_t*)err); break; }
        }
      }
      __syncthreads();
    }
    {
      const Hh* x = (l == 0 ? sg.x : sg.L[l - 1].out) + qbase * EH;
      const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, Nq * EH * 2, 0x00020000);
      for (int id = tid; id < NRT * 16 * (EH / 8); id += NWAVE * 64) {
        const int r = id / (EH / 8), c = (id % (EH / 8)) * 8;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (r * EH + c) * 2, 0, 16);
        *(u32x4*)(sX + r * XS + c) = v;
        if (r >= r0 && r < r0 + 16) *(u32x4*)(sO + (r - r0) * XS + c) = v;
      }
    }
    __syncthreads();
    // ================= self-attention: K|V of all the encoder's rows, Q of the own tile =================
    {
      f32x4 acc[NRT][2], aq = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < NRT; ++i) { acc[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int i = 0; i < NRT; ++i) {
          const h16x8<Hh> a = lfrag(sX, XS, i * 16, ks * 32, lane);
          acc[i][0] = emma(a, wkv[0][ks], acc[i][0]);
          acc[i][1] = emma(a, wkv[1][ks], acc[i][1]);
        }
        aq = emma(lfrag(sO, XS, 0, ks * 32, lane), wq[ks], aq);
        KSTEP_FENCE();
      }
      Hh* dst = (w < 4 ? sK : sV) + ((2 * w) & 7) * 16 + c16;
#pragma unroll
      for (int i = 0; i < NRT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dst[(i * 16 + 4 * g + r) * XS] = from_f<Hh>(acc[i][0][r] + pb_kv0);
          dst[(i * 16 + 4 * g + r) * XS + 16] = from_f<Hh>(acc[i][1][r] + pb_kv1);
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) sQ[(4 * g + r) * XS + colw] = from_f<Hh>(aq[r] + pb_q0);
    }
    h16x8<Hh> wo[4], wcq[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { wo[ks] = gfrag(L.Wo, EH, w * 16, ks * 32, lane); wcq[ks] = gfrag(L.Wq, EH, w * 16, ks * 32, lane); }
    __syncthreads();
    {
      Hh* qg = L.qkv + row0 * 3 * EH;
      for (int id = tid; id < nq * 48; id += NWAVE * 64) {
        const int r = id / 48, c = (id % 48) * 8;
        const Hh* src = c < EH ? sQ + r * XS + c : (c < 2 * EH ? sK + (r0 + r) * XS + (c - EH) : sV + (r0 + r) * XS + (c - 2 * EH));
        *(h16x8<Hh>*)(qg + (long long)r * 3 * EH + c) = *(const h16x8<Hh>*)src;
      }
    }
    dd.site = L.site_attn; dd.p = p.p_attn;
    const DropState dsa = drop_init(dd);
    rs_attn<NRT>(sQ, sK, sV, sS, sPd, qbias, Nq, NRT * 16, sg.dist, sw, sb, r0, nq, Nq, L.P, L.Pd, sg.ldps, samp, p.scale, dsa, tid, w, lane);
    copy_tile<false>(sQ, XS, L.ctx + row0 * EH, EH, nq, EH, tid);
    {
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) acc = emma(lfrag(sQ, XS, 0, ks * 32, lane), wo[ks], acc);
      dd.site = L.site_ao; dd.p = p.p_hidden;
      const DropState dsh = drop_init(dd);
      add_norm16(acc, pb_o, pg_1, pe_1, sO, red, sA, L.rstd_a, nq, row0, p.eps, dsh, w, lane);
    }
    // weights of the context's key / value projection (2 of its 16 column tiles per wave): in flight under the barrier
    h16x8<Hh> wck[2][4];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) wck[ct][ks] = gfrag(L.Wkv, EH, (2 * w + ct) * 16, ks * 32, lane);
    __syncthreads();                              // sA = a ; the self K / V images are dead
    copy_tile<false>(sA, XS, L.a + row0 * EH, EH, nq, EH, tid);
    // ================= cross-attention: Q of the own tile from a, K|V of ALL context rows (read from global as A-fragments) =================
    {
      f32x4 aq = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) aq = emma(lfrag(sA, XS, 0, ks * 32, lane), wcq[ks], aq);
#pragma unroll
      for (int r = 0; r < 4; ++r) sQ[(4 * g + r) * XS + colw] = from_f<Hh>(aq[r] + pb_q);
      f32x4 acc[NKT][2];
#pragma unroll
      for (int i = 0; i < NKT; ++i) { acc[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
      const Hh* cx = sg.cx + kbase * EH;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int i = 0; i < NKT; ++i) {
          const int row = min(i * 16 + c16, Nk - 1);
          const h16x8<Hh> a = *(const h16x8<Hh>*)(cx + (long long)row * EH + ks * 32 + 8 * g);
          acc[i][0] = emma(a, wck[0][ks], acc[i][0]);
          acc[i][1] = emma(a, wck[1][ks], acc[i][1]);
        }
        KSTEP_FENCE();
      }
      Hh* dst = (w < 4 ? sK : sV) + ((2 * w) & 7) * 16 + c16;
#pragma unroll
      for (int i = 0; i < NKT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = i * 16 + 4 * g + r;
          dst[row * XS] = row < Nk ? from_f<Hh>(acc[i][0][r] + pb_ck0) : (Hh)0.0f;
          dst[row * XS + 16] = row < Nk ? from_f<Hh>(acc[i][1][r] + pb_ck1) : (Hh)0.0f;
        }
    }
    h16x8<Hh> woc[4], w1[4][4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) woc[ks] = gfrag(L.Woc, EH, w * 16, ks * 32, lane);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) w1[ct][ks] = gfrag(L.W1, EH, (4 * w + ct) * 16, ks * 32, lane);
    __syncthreads();                              // cross Q, K, V images complete
    copy_tile<false>(sQ, XS, L.q + row0 * EH, EH, nq, EH, tid);
    // kv [Nk, 2H] for the backward: every tile holds all of it; tile t writes the context rows of key tiles t, t + NRT, ...
    for (int kt = tile; kt * 16 < Nk; kt += ntile) {
      const int nr = min(16, Nk - kt * 16);
      Hh* kg = L.kv + (kbase + kt * 16) * 2 * EH;
      for (int id = tid; id < nr * 32; id += NWAVE * 64) {
        const int r = id / 32, c = (id % 32) * 8;
        const Hh* src = c < EH ? sK + (kt * 16 + r) * XS + c : sV + (kt * 16 + r) * XS + (c - EH);
        *(h16x8<Hh>*)(kg + (long long)r * 2 * EH + c) = *(const h16x8<Hh>*)src;
      }
    }
    dd.site = L.site_cattn; dd.p = p.p_attn;
    const DropState dsc = drop_init(dd);
    rs_attn<NKT>(sQ, sK, sV, sS, sPd, cbias, Nk, NKT * 16, nullptr, 0.f, 0.f, r0, nq, Nq, L.Pc, L.Pdc, sg.ldpc, samp, p.scale, dsc, tid, w, lane);
    copy_tile<false>(sQ, XS, L.cctx + row0 * EH, EH, nq, EH, tid);
    {
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) acc = emma(lfrag(sQ, XS, 0, ks * 32, lane), woc[ks], acc);
      dd.site = L.site_co; dd.p = p.p_hidden;
      const DropState dsh = drop_init(dd);
      add_norm16(acc, pb_oc, pg_c, pe_c, sA, red, sO, L.rstd_c, nq, row0, p.eps, dsh, w, lane);
    }
    __syncthreads();                              // sO = c
    copy_tile<false>(sO, XS, L.c + row0 * EH, EH, nq, EH, tid);
    // ================= FFN =================
    h16x8<Hh> w2[16];
    {
      f32x4 acc[4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const h16x8<Hh> a = lfrag(sO, XS, 0, ks * 32, lane);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = emma(a, w1[ct][ks], acc[ct]);
      }
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) w2[ks] = gfrag(L.W2, EI, w * 16, ks * 32, lane);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const int col = (4 * w + ct) * 16 + c16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float zv = acc[ct][r] + pb_ffn[ct];
          sZ[(4 * g + r) * GS + col] = from_f<Hh>(zv);
          sG[(4 * g + r) * GS + col] = from_f<Hh>(gelu_fast(zv));
        }
      }
    }
    __syncthreads();
    copy_tile<false>(sZ, GS, L.z + row0 * EI, EI, nq, EI, tid);
    copy_tile<false>(sG, GS, L.g + row0 * EI, EI, nq, EI, tid);
    {
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) acc = emma(lfrag(sG, GS, 0, ks * 32, lane), w2[ks], acc);
      dd.site = L.site_out; dd.p = p.p_hidden;
      const DropState dsh = drop_init(dd);
      add_norm16(acc, pb_2, pg_2, pe_2, sO, red, sA, L.rstd_o, nq, row0, p.eps, dsh, w, lane);
    }
    __syncthreads();
    copy_tile<true>(sA, XS, L.out + row0 * EH, EH, nq, EH, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && l + 1 < sg.nlayers) __hip_atomic_fetch_add((gu32_t*)(cnt + (long long)samp * 6 + l), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <typename Hh>
__global__ __launch_bounds__(512) void xencoder_rs_kernel(XParamsT<Hh> p, int nt0, int nt1) {
  extern __shared__ __attribute__((aligned(16))) unsigned char enc_smem[];
  int b = blockIdx.x, s = 0, nt = nt0;
  unsigned* cnt = p.sync + 4;
  if (b >= p.seg[0].nsamp * nt0) { b -= p.seg[0].nsamp * nt0; s = 1; nt = nt1; cnt += 6ll * p.seg[0].nsamp; }
  const XSegT<Hh>& sg = p.seg[s];
  const int samp = b / nt, tile = b - samp * nt;
  // two instantiations only (see encoder_mix_kernel: more bodies in one kernel push the parameter block into scratch memory)
  if (nt <= 3) xenc_rs_body<3>(p, sg, samp, tile, nt, cnt, p.sync, enc_smem);
  else xenc_rs_body<5>(p, sg, samp, tile, nt, cnt, p.sync, enc_smem);
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
  if (p.sync) {           // row-split form when every tile of the launch is resident at once (one workgroup per CU)
    const int nt0 = (p.seg[0].Nq + 15) / 16, nt1 = p.nseg > 1 ? (p.seg[1].Nq + 15) / 16 : 1;
    const int ns1 = p.nseg > 1 ? p.seg[1].nsamp : 0;
    const long long words = 4 + 6ll * (p.seg[0].nsamp + ns1);
    if (p.sync_words < words || ((uintptr_t)p.sync & 15)) return MAGIC_ERR_ARG;
    static int ncu = 0;
    if (!ncu) { hipDeviceProp_t pr; int d = 0; (void)hipGetDevice(&d); ncu = (hipGetDeviceProperties(&pr, d) == hipSuccess) ? pr.multiProcessorCount : 256; }
    const int grid = p.seg[0].nsamp * nt0 + ns1 * nt1;
    if (grid <= ncu && grid > blocks) {
      static bool rs_attr = false;
      if (!rs_attr) {
        (void)hipFuncSetAttribute((const void*)xencoder_rs_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)enc_rs_lds_bytes());
        (void)hipFuncSetAttribute((const void*)xencoder_rs_kernel<f16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)enc_rs_lds_bytes());
        rs_attr = true;
      }
      if (hipMemsetAsync(p.sync, 0, (size_t)((words + 3) / 4 * 4) * sizeof(unsigned), (hipStream_t)stream) != hipSuccess) return MAGIC_ERR_LAUNCH;
      if (dtype == DT_BF16) hipLaunchKernelGGL(xencoder_rs_kernel<bf16>, dim3(grid), dim3(512), enc_rs_lds_bytes(), (hipStream_t)stream, p, nt0, nt1);
      else { XParamsT<f16> pf; static_assert(sizeof(pf) == sizeof(p), "layout"); memcpy(&pf, &p, sizeof(pf)); hipLaunchKernelGGL(xencoder_rs_kernel<f16>, dim3(grid), dim3(512), enc_rs_lds_bytes(), (hipStream_t)stream, pf, nt0, nt1); }
      return launch_status();
    }
    if (hipMemsetAsync(p.sync, 0, 16, (hipStream_t)stream) != hipSuccess) return MAGIC_ERR_LAUNCH;
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
