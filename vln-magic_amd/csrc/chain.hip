// Forward-only ROW CHAIN of a post-LN transformer block at the frozen teacher's width (H = 256, FFN 1024; gfx950, bf16 / fp16):
//
//   y1   = LayerNorm(in Wa^T + ba + res)                          (BertSelfOutput / the cross-attention output block)
//   y2   = LayerNorm(gelu(y1 W1^T + bi) W2^T + bo2 + y1)          (BertIntermediate + BertOutput; optional)
//   proj = y_last Wp^T + bp,  [M, H | 2H | 3H]                    (the NEXT attention's Q | K | V projection; optional)
//
// i.e. everything of a block that is per token, between two attention products, as ONE launch instead of five or six
// (dense+LN, FFN1 GEMM, FFN2 GEMM, LayerNorm, QKV GEMM).  The teacher of the MAKD step (kdl.train_teacher = false,
// pretrain_src/config/r2r_magic_pretrain.json:62-87) runs no backward and no dropout, so nothing is saved for one: the FFN
// pre-activation and the GELU output never leave the CU.
//
// One 512-thread workgroup owns 32 rows (two 16-row MFMA tiles).  Weights are not staged through LDS: every weight element is used by
// exactly two MFMAs of the workgroup, so each lane loads its B-fragment (8 consecutive k of one weight row, 16 bytes) from L2 straight
// into registers (the scheme of csrc/encoder.hip at H = 128, where the fragments of a whole stage fit the register file; at H = 256 a
// workgroup streams 1.6 MB of weights, as ONE sequence of 8-fragment chunks through a ring of CNB register buffers that keeps CNB - 1
// chunks -- 8 KB per wave each -- in flight across the stage boundaries: the stream is bound by the 64 B/clk a CU takes from L2 at ~2 us of
// loaded latency).  LDS per workgroup 118 KB, one workgroup per CU.  Round 4: CNB 4 -> 3 (227 -> 187 registers): the same 0.71-0.73 ms for
// the teacher's forward alone, the overlapped training step 8 us faster (profiles/micro/r04_ab_chain_ring.txt) -- the register file a
// teacher workgroup leaves free is what the student's small kernels on the same CU run in.
// ALL FOUR WEIGHT MATRICES ARE READ IN FRAGMENT ORDER (magic_pack_frag_spans below; the frozen teacher packs them once).
// Two problems can share one launch (text || panorama encoder, global || local co-attention encoder):
// group.hpp KIND_CHAIN.
#include "enc_common.hpp"
#include "group.hpp"
#include <cstdlib>

#define CH 256
#define CI 1024
#ifndef CHAIN_QUARTERS
#define CHAIN_QUARTERS 0   // 1: the 32-row kernel's FFN one 256-column quarter of the intermediate at a time (81 KB of LDS instead of 129 KB; measured: no gain, profiles/micro/r06_ab_chain_quarters.txt)
#endif
#ifdef CHAIN_LDS_SW
// round 6 A/B (profiles/micro/r06_ab_chain_lds_sw.txt): XOR-swizzled images for chain_fwd_kernel's A operands.  With the padded pitch (33 / 129 slots) every
// 16-lane group of a ds_read_b128 fragment read -- lanes {0-3, 12-15 (k group 0), 20-27 (k group 1)} etc. -- has ONE lane pair on the same 16-byte slot
// (rows r and r' = r + 8 - 16 g differ by the slot the k group adds): 2 cycles per group instead of 1, the kernel's 0.36 LDS-conflict share.  Rows of
// exactly 256 / 1024 elements with 16-byte chunk c of row r stored at c ^ (r & 15) put the 16 lanes of every group on 16 distinct slots.
#define KP 256        // (the 32-row kernel's pitches; the 64-row kernel keeps the padded CP below)
#define KG 1024
#define CADDR(row, col, pitch) ((row) * (pitch) + (((((col) >> 3) ^ ((row) & 15))) << 3) + ((col) & 7))
#else
#define KP 264
#define KG 1032
#define CADDR(row, col, pitch) ((row) * (pitch) + (col))
#endif
#define CP 264        // row pitch of the [16][256] images: 528 B = 33 16-byte slots
#define CG 1032       // row pitch of the [16][1024] GELU image: 2064 B = 129 slots
// A fragment of 16 rows x 32 k from a (possibly swizzled) image
template <typename Hh> __device__ __forceinline__ h16x8<Hh> cfrag(const Hh* s, int pitch, int row0, int k0, int lane) {
  return *(const h16x8<Hh>*)(s + CADDR(row0 + (lane & 15), k0 + 8 * (lane >> 4), pitch));
}
// cooperative copy of rows x cols from a (possibly swizzled) LDS image to global rows (16-byte vectors)
template <typename Hh> __device__ __forceinline__ void ccopy_out(const Hh* s, int pitch, Hh* g, long long ldg, int rows, int cols, int tid) {
  const int cpr = cols / 8;
  for (int id = tid; id < rows * cpr; id += NWAVE * 64) {
    const int r = id / cpr, c = (id % cpr) * 8;
    *(h16x8<Hh>*)(g + (long long)r * ldg + c) = *(const h16x8<Hh>*)(s + CADDR(r, c, pitch));
  }
}

struct ChainParams {
  int M, ld_in, Np, pad_;
  const void* in; const void* res;                                           // [M, H] (pitch ld_in), [M, H]
  const void* Wa; const float* ba; const float* g1; const float* b1; void* y1;   // y1 may be NULL (not stored)
  const void* W1; const float* bi; const void* W2; const float* bo2; const float* g2; const float* b2; void* y2;   // W1 NULL: no FFN
  const void* Wp; const float* bp; void* proj;                               // Wp NULL: no projection; [Np, H] -> [M, Np]
  float eps; int pad2_;
};

#define CRT 2          // 16-row tiles per workgroup
#define CROWS (16 * CRT)
#ifndef CNB
#define CNB 3          // (4 measured 8 us per training step SLOWER next to the student's stream: 227 instead of 187 registers, same time alone)
#endif
//                     weight-fragment chunks (8 fragments = 8 KB per wave each) in the register ring: CNB - 1 are in flight ahead of the MFMAs

// LayerNorm over CROWS rows x 256 columns held as CRT x 2 16x16 accumulator tiles per wave (columns (2w + ct) * 16 + c16)
template <typename Hh>
__device__ __forceinline__ void chain_norm(f32x4 (&acc)[CRT][2], const float* sPar, const Hh* sR,
                                           float* red, Hh* sOut, const int nq, const float eps, const int w, const int lane) {
  const int g = lane >> 4, c16 = lane & 15;
  float bv[2], gv[2], btv[2];           // bias | gamma | beta of this lane's two columns: staged in LDS by the prologue (see chain_body)
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int col = (2 * w + ct) * 16 + c16;
    bv[ct] = sPar[col]; gv[ct] = sPar[CH + col]; btv[ct] = sPar[2 * CH + col];
  }
  float s[CRT][4];
#pragma unroll
  for (int rt = 0; rt < CRT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = rt * 16 + 4 * g + r;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) acc[rt][ct][r] += bv[ct] + to_f(sR[CADDR(row, (2 * w + ct) * 16 + c16, KP)]);
      s[rt][r] = g16_sum(acc[rt][0][r] + acc[rt][1][r]);
    }
  if (c16 == 0) {
#pragma unroll
    for (int rt = 0; rt < CRT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[w * CROWS + rt * 16 + 4 * g + r] = s[rt][r];
  }
  __syncthreads();
  float mean[CRT][4];
#pragma unroll
  for (int rt = 0; rt < CRT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float t = 0.f;
#pragma unroll
      for (int ww = 0; ww < NWAVE; ++ww) t += red[ww * CROWS + rt * 16 + 4 * g + r];
      mean[rt][r] = t * (1.0f / CH);
      const float d0 = acc[rt][0][r] - mean[rt][r], d1 = acc[rt][1][r] - mean[rt][r];
      s[rt][r] = g16_sum(d0 * d0 + d1 * d1);
    }
  if (c16 == 0) {
#pragma unroll
    for (int rt = 0; rt < CRT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[NWAVE * CROWS + w * CROWS + rt * 16 + 4 * g + r] = s[rt][r];
  }
  __syncthreads();
#pragma unroll
  for (int rt = 0; rt < CRT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = rt * 16 + 4 * g + r;
      float t = 0.f;
#pragma unroll
      for (int ww = 0; ww < NWAVE; ++ww) t += red[NWAVE * CROWS + ww * CROWS + rr];
      const float rstd = rsqrtf(t * (1.0f / CH) + eps);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
        sOut[CADDR(rr, (2 * w + ct) * 16 + c16, KP)] = (rr < nq) ? from_f<Hh>((acc[rt][ct][r] - mean[rt][r]) * rstd * gv[ct] + btv[ct]) : (Hh)0.0f;
    }
}

template <typename Hh>
__device__ __forceinline__ void chain_rows_in(const Hh* src, long long ld, int nq, Hh* dst, int tid) {
  // CROWS rows x 256 columns, 16-byte chunks; rows >= nq are zeros
  typedef __attribute__((ext_vector_type(4))) unsigned u4;
  for (int id = tid; id < CROWS * (CH / 8); id += NWAVE * 64) {
    const int r = id / (CH / 8), c = (id % (CH / 8)) * 8;
    const u4 v = r < nq ? *(const u4*)(src + r * ld + c) : (u4){0u, 0u, 0u, 0u};
    *(u4*)(dst + CADDR(r, c, KP)) = v;
  }
}

// The workgroup's weights as ONE stream of chunks of 8 fragments per wave, in the order the stages consume them:
//   0..1   stage 1  (column tile 2w + i of Wa, 8 k-steps)                10..17 stage 2b (k-steps 4c..4c+3 of column tiles 2w, 2w + 1 of W2)
//   2..9   stage 2a (column tile 8w + i of W1, 8 k-steps)                18..23 stage 3  (column tile w nct + j of Wp, 8 k-steps; j < nct)
// (weights do not depend on data: the ring keeps loading across the stage boundaries and their barriers)
// B fragment (nt, ks) of a weight matrix kept in FRAGMENT ORDER (magic_pack_frag_spans): the 64 lanes' 16 bytes are one contiguous KB.
// (The row-major form -- lane l reads 16 bytes of weight row l & 15 -- touches 16 cache lines per quarter-wave: measured 12 B/clk per CU.)
template <typename Hh> __device__ __forceinline__ h16x8<Hh> pfrag(const Hh* __restrict__ Wf, const int K, const int nt, const int ks, const int lane) {
  return *(const h16x8<Hh>*)(Wf + ((long long)(nt * (K >> 5) + ks) * 64 + lane) * 8);
}

template <typename Hh>
__device__ __forceinline__ void chain_load_chunk(h16x8<Hh> (&b)[8], const int cid, const ChainParams& p, const int w, const int lane, const int nct) {
  if (cid < 2) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) b[ks] = pfrag((const Hh*)p.Wa, CH, 2 * w + cid, ks, lane);
#if CHAIN_QUARTERS
  } else if (cid < 18) {
    // quarter q of the intermediate (columns 256 q ..): two W1 chunks (this wave's column tiles 16 q + 2 w, + 1), then two W2 chunks (k-steps 8 q .. 8 q + 7)
    const int q = (cid - 2) >> 2, r = (cid - 2) & 3;
    if (r < 2) {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) b[ks] = pfrag((const Hh*)p.W1, CH, 16 * q + 2 * w + r, ks, lane);
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        b[ks] = pfrag((const Hh*)p.W2, CI, 2 * w, 8 * q + 4 * (r - 2) + ks, lane);
        b[4 + ks] = pfrag((const Hh*)p.W2, CI, 2 * w + 1, 8 * q + 4 * (r - 2) + ks, lane);
      }
    }
#else
  } else if (cid < 10) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) b[ks] = pfrag((const Hh*)p.W1, CH, 8 * w + cid - 2, ks, lane);
  } else if (cid < 18) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      b[ks] = pfrag((const Hh*)p.W2, CI, 2 * w, 4 * (cid - 10) + ks, lane);
      b[4 + ks] = pfrag((const Hh*)p.W2, CI, 2 * w + 1, 4 * (cid - 10) + ks, lane);
    }
#endif
  } else {
    // UNCONDITIONAL (a load the compiler cannot count makes every later wait a full drain): tiles past the projection's width re-read its
    // last tile, a chain without a projection reads Wa's fragments -- valid addresses, unused data
    const Hh* Wq = p.Wp ? (const Hh*)p.Wp : (const Hh*)p.Wa;
    const int j = cid - 18, nt = p.Wp ? w * nct + (j < nct ? j : nct - 1) : 2 * w;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) b[ks] = pfrag(Wq, CH, nt, ks, lane);
  }
}

#ifdef CHAIN_TIMING
__device__ long long chain_ticks[16];          // wall_clock64 (100 MHz) marks of tile 0 (profiles/micro/chain_timing.hip)
#define CH_MARK(i) do { if (tile == 0 && tid == 0) chain_ticks[i] = wall_clock64(); } while (0)
#else
#define CH_MARK(i)
#endif
template <typename Hh, bool FFN>
__device__ __forceinline__ void chain_body(const ChainParams& p, const int tile, unsigned char* smem) {
#if CHAIN_QUARTERS
  // 81 KB (round 6; 129 KB with the whole GELU image): sY1 | sG | sRes | sIn.  The projection image [CROWS][Np + 8] (<= 49.7 KB) takes the three images that
  // are dead by then: sY1 | sG | sRes after an FFN (its A operand is sIn), sG | sRes | sIn without one (its A operand is sY1).
  Hh* sY1 = (Hh*)smem;                      // [CROWS][CP]
  Hh* sG = sY1 + CROWS * KP;                // [CROWS][CP]  GELU output of ONE 256-column quarter of the intermediate
  Hh* sRes = sG + CROWS * KP;               // [CROWS][CP]  residual of stage 1
  Hh* sIn = sRes + CROWS * KP;              // [CROWS][CP]  stage-1 input; later the block output y2
  float* red = (float*)(sIn + CROWS * KP);  // [2][8][CROWS]
  Hh* sProj = FFN ? sY1 : sG;
#else
  Hh* sIn = (Hh*)smem;                      // [CROWS][CP]  stage-1 input; later the block output y2
  Hh* sRes = sIn + CROWS * KP;              // [CROWS][CP]  residual of stage 1
  Hh* sY1 = sRes + CROWS * KP;              // [CROWS][CP]
  Hh* sG = sY1 + CROWS * KP;                // [CROWS][CG]  GELU output; later the projection image [CROWS][Np + 8]
  float* red = (float*)(sG + CROWS * KG);   // [2][8][CROWS]
  Hh* sProj = sG;
#endif
  float* sPar = red + 2 * NWAVE * CROWS;    // ba | g1 | b1 | bo2 | g2 | b2 (256 each) | bi (1024) | bp (<= 768): every small parameter, staged ONCE --
                                            // a global load between the weight chunks makes the compiler drain the whole ring (vmcnt(0)) before its use
  const int tid = threadIdx.x, lane0 = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int lane = lane0;
  asm volatile("" : "+v"(lane));
  const int g = lane >> 4, c16 = lane & 15;
  const int row0 = tile * CROWS, nq = min(CROWS, p.M - row0);
  const int nct = p.Wp ? (p.Np >> 7) : 0;
  constexpr int NS = FFN ? 24 : 8;          // chunks in this variant's stream; SEQ(s): stream position -> chunk id
#define SEQ(s) (FFN ? (s) : ((s) < 2 ? (s) : (s) + 16))
#define AHEAD(s) do { if ((s) + CNB - 1 < NS) chain_load_chunk<Hh>(ring[((s) + CNB - 1) % CNB], SEQ((s) + CNB - 1), p, w, lane, nct); } while (0)
  CH_MARK(0);
  chain_rows_in((const Hh*)p.in + (long long)row0 * p.ld_in, p.ld_in, nq, sIn, tid);
  chain_rows_in((const Hh*)p.res + (long long)row0 * CH, CH, nq, sRes, tid);
  if (tid < CH) {
    sPar[tid] = p.ba[tid]; sPar[CH + tid] = p.g1[tid]; sPar[2 * CH + tid] = p.b1[tid];
    if (FFN) { sPar[3 * CH + tid] = p.bo2[tid]; sPar[4 * CH + tid] = p.g2[tid]; sPar[5 * CH + tid] = p.b2[tid]; }
  }
  if (FFN) { sPar[6 * CH + tid] = p.bi[tid]; sPar[6 * CH + 512 + tid] = p.bi[512 + tid]; }
  for (int i = tid; i < nct * 128; i += NWAVE * 64) sPar[6 * CH + CI + i] = p.bp[i];
  h16x8<Hh> ring[CNB][8];
#pragma unroll
  for (int s = 0; s < CNB - 1; ++s) chain_load_chunk<Hh>(ring[s], SEQ(s), p, w, lane, nct);
  __syncthreads();
  CH_MARK(1);
  // ================= 1: y1 = LayerNorm(in Wa^T + ba + res) =================
  {
    f32x4 acc[CRT][2];
#pragma unroll
    for (int rt = 0; rt < CRT; ++rt) { acc[rt][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[rt][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      AHEAD(i);
      KSTEP_FENCE();
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int rt = 0; rt < CRT; ++rt) acc[rt][i] = emma(cfrag(sIn, KP, rt * 16, ks * 32, lane), ring[i % CNB][ks], acc[rt][i]);
      KSTEP_FENCE();
    }
    chain_norm(acc, sPar, sRes, red, sY1, nq, p.eps, w, lane);
  }
  __syncthreads();
  CH_MARK(2);
  if (p.y1) ccopy_out(sY1, KP, (Hh*)p.y1 + (long long)row0 * CH, CH, nq, CH, tid);
  const Hh* sLast = sY1;
  if constexpr (FFN) {
#if CHAIN_QUARTERS
    // ================= 2: y2 = LayerNorm(gelu(y1 W1^T + bi) W2^T + bo2 + y1), one 256-column QUARTER of the intermediate at a time =================
    // (round 6: the GELU image of a quarter is 17 KB instead of 66 KB for the whole intermediate -- the workgroup's LDS drops from 129 to 81 KB, which is what
    // lets a chain workgroup share a CU with a 68 KB row-block backward workgroup of the student instead of taking turns with it; the second product
    // accumulates in registers over the quarters, as chain64_body does)
    {
      f32x4 acc2[CRT][2];
#pragma unroll
      for (int rt = 0; rt < CRT; ++rt) { acc2[rt][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc2[rt][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // ---- 2a: this wave's two column tiles (16 q + 2 w, + 1) of gelu(y1 W1^T + bi) -> columns 32 w .. 32 w + 31 of the quarter's image
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          AHEAD(2 + 4 * q + ct);
          const int lcol = (2 * w + ct) * 16 + c16;
          const float bfc = sPar[6 * CH + 256 * q + lcol];
          KSTEP_FENCE();
          f32x4 acc[CRT];
#pragma unroll
          for (int rt = 0; rt < CRT; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int rt = 0; rt < CRT; ++rt) acc[rt] = emma(cfrag(sY1, KP, rt * 16, ks * 32, lane), ring[(2 + 4 * q + ct) % CNB][ks], acc[rt]);
#pragma unroll
          for (int rt = 0; rt < CRT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) sG[CADDR(rt * 16 + 4 * g + r, lcol, KP)] = from_f<Hh>(gelu_fast(acc[rt][r] + bfc));
          KSTEP_FENCE();
        }
        __syncthreads();
        // ---- 2b: y2 += g_q W2[:, 256 q ..]^T: 8 k-steps in two chunks of 4 x 2 column tiles
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
          AHEAD(4 + 4 * q + ch);
          KSTEP_FENCE();
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int rt = 0; rt < CRT; ++rt) {
              const h16x8<Hh> a = cfrag(sG, KP, rt * 16, (4 * ch + ks) * 32, lane);
              acc2[rt][0] = emma(a, ring[(4 + 4 * q + ch) % CNB][ks], acc2[rt][0]);
              acc2[rt][1] = emma(a, ring[(4 + 4 * q + ch) % CNB][4 + ks], acc2[rt][1]);
            }
          KSTEP_FENCE();
        }
        if (q < 3) __syncthreads();             // every wave is done with this quarter's image before the next one overwrites it
      }
      CH_MARK(3);
      chain_norm(acc2, sPar + 3 * CH, sY1, red, sIn, nq, p.eps, w, lane);
    }
#else
    // ================= 2a: g = gelu(y1 W1^T + bi): 8 column tiles per wave =================
    {
      // (the A fragments come from LDS at every use: holding the 16 of a stage in registers next to the ring spilled, and a scratch
      // access between the weight chunks drains the ring just like any other vector-memory operation)
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) {
        AHEAD(2 + ct);
        const int col = (8 * w + ct) * 16 + c16;
        const float bfc = sPar[6 * CH + col];
        KSTEP_FENCE();
        f32x4 acc[CRT];
#pragma unroll
        for (int rt = 0; rt < CRT; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
          for (int rt = 0; rt < CRT; ++rt) acc[rt] = emma(cfrag(sY1, KP, rt * 16, ks * 32, lane), ring[(2 + ct) % CNB][ks], acc[rt]);
#pragma unroll
        for (int rt = 0; rt < CRT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) sG[CADDR(rt * 16 + 4 * g + r, col, KG)] = from_f<Hh>(gelu_fast(acc[rt][r] + bfc));
        KSTEP_FENCE();
      }
    }
    __syncthreads();
    CH_MARK(3);
    // ================= 2b: y2 = LayerNorm(g W2^T + bo2 + y1): K = 1024 in 8 chunks of 4 k-steps x 2 column tiles =================
    {
      f32x4 acc[CRT][2];
#pragma unroll
      for (int rt = 0; rt < CRT; ++rt) { acc[rt][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[rt][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) {
        AHEAD(10 + ch);
        KSTEP_FENCE();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int rt = 0; rt < CRT; ++rt) {
            const h16x8<Hh> a = cfrag(sG, KG, rt * 16, (4 * ch + ks) * 32, lane);
            acc[rt][0] = emma(a, ring[(10 + ch) % CNB][ks], acc[rt][0]);
            acc[rt][1] = emma(a, ring[(10 + ch) % CNB][4 + ks], acc[rt][1]);
          }
        KSTEP_FENCE();
      }
      chain_norm(acc, sPar + 3 * CH, sY1, red, sIn, nq, p.eps, w, lane);
    }
#endif
    __syncthreads();
    CH_MARK(4);
    ccopy_out(sIn, KP, (Hh*)p.y2 + (long long)row0 * CH, CH, nq, CH, tid);
    sLast = sIn;
  }
  if (nct) {
    // ================= 3: proj = y_last Wp^T + bp, nct = Np / 128 column tiles per wave =================
    constexpr int S3 = FFN ? 18 : 2;
    const int pp = p.Np + 8;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      AHEAD(S3 + j);
      if (j < nct) {
        const int tile_n = w * nct + j;
        const float bpv = sPar[6 * CH + CI + tile_n * 16 + c16];
        KSTEP_FENCE();
        f32x4 acc[CRT];
#pragma unroll
        for (int rt = 0; rt < CRT; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
          for (int rt = 0; rt < CRT; ++rt) acc[rt] = emma(cfrag(sLast, KP, rt * 16, ks * 32, lane), ring[(S3 + j) % CNB][ks], acc[rt]);
#pragma unroll
        for (int rt = 0; rt < CRT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) sProj[(rt * 16 + 4 * g + r) * pp + tile_n * 16 + c16] = from_f<Hh>(acc[rt][r] + bpv);
      }
      KSTEP_FENCE();
    }
    __syncthreads();
    CH_MARK(5);
    copy_out(sProj, pp, (Hh*)p.proj + (long long)row0 * p.Np, p.Np, nq, p.Np, tid);
  }
  CH_MARK(6);
#undef SEQ
#undef AHEAD
}

// (the two problems as an ARRAY in the kernel-argument segment: `pr.p[which]` becomes scalar loads at a computed offset where they are used;
// selecting between two by-value structs kept both in SGPRs and spilled 159 of them into vector registers)
struct ChainPair { ChainParams p[2]; int split; };
#ifndef CHAIN_ATTR
#define CHAIN_ATTR
#endif
template <typename Hh>
__global__ __launch_bounds__(512) CHAIN_ATTR void chain_fwd_kernel(ChainPair pr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char chain_smem[];
  const int which = (int)blockIdx.x < pr.split ? 0 : 1;
  const ChainParams& p = pr.p[which];
  const int tile = which ? blockIdx.x - pr.split : blockIdx.x;
  if (p.W1) chain_body<Hh, true>(p, tile, chain_smem);
  else chain_body<Hh, false>(p, tile, chain_smem);
}


// =============================================================================================================================
// The same chain on 64-ROW tiles with the 32 x 32 x 16 product (round 4).  What a workgroup costs the chip is its CU for the ~27 us it
// takes to pull the block's 1.57 MB of weights through one CU's L2 port, whatever its row count -- so rows per workgroup set the CU time
// of the teacher's forward, and that CU time is what the student's stream pays for (profiles/micro/r04_teacher_contention.txt; a timing
// build with HALF the chain's workgroups ran the overlapped step 45 us faster).  Round 3's 64-row form on the 16 x 16 x 32 product was
// LDS-bound (four A-fragment reads per weight fragment: 57 us per workgroup); on 32 x 32 x 16 a weight fragment (32 weight rows x 16 k)
// meets TWO A fragments (32 rows x 16 k each) -- the LDS traffic per weight byte of today's 32-row form at half the workgroups.
//   * B operand: lane l needs W[32 NT + (l & 31)][16 KS + 8 (l >> 5) .. + 7].  The buffers stay in the 16 x 32 fragment order of
//     magic_pack_frag_spans: that is chunk (l & 15) + 16 (2 (KS & 1) + (l >> 5)) of fragment (2 NT + ((l >> 4) & 1), KS >> 1) -- every
//     quarter-wave still reads 256 contiguous bytes, a chunk of 8 k-steps is two contiguous 4 KB runs.
//   * accumulators: column 32 w + (l & 31) on the lane, rows (i & 3) + 8 (i >> 2) + 4 (l >> 5) in the 16 registers of a 32-row tile.
//   * the FFN goes through LDS one 256-column chunk of the intermediate at a time (GELU image [64][256]); the second product accumulates
//     in registers over the four chunks.  LDS 155 KB, one workgroup per CU.
typedef __attribute__((ext_vector_type(16))) float f32x16;
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
template <typename Hh> __device__ __forceinline__ h16x8<Hh> lfrag32(const Hh* s, int pitch, int row0, int k0, int lane) {
  return *(const h16x8<Hh>*)(s + (row0 + (lane & 31)) * pitch + k0 + 8 * (lane >> 5));
}
template <typename Hh> __device__ __forceinline__ h16x8<Hh> pfrag32(const Hh* __restrict__ Wf, const int K, const int NT, const int KS, const int lane) {
  const int nt16 = 2 * NT + ((lane >> 4) & 1), l2 = (lane & 15) + 16 * (2 * (KS & 1) + (lane >> 5));
  return *(const h16x8<Hh>*)(Wf + ((long long)(nt16 * (K >> 5) + (KS >> 1)) * 64 + l2) * 8);
}
#define C6ROWS 64
#define C6RT 2
#define C6NB 3         // ring depth (chunks of 8 fragments = 8 KB per wave)
// chunk ids: 0..1 stage 1 (Wa tile w, k-steps 8 id ..) | per FFN chunk c: 2 + 4c, 3 + 4c = W1 tile 8c + w (k-steps 0-7, 8-15),
// 4 + 4c, 5 + 4c = W2 tile w (k-steps 16c .. 16c + 7, 16c + 8 .. 16c + 15) | 18 + 2j, 19 + 2j = Wp tile 8j + w (j < Np / 256)
template <typename Hh>
__device__ __forceinline__ void chain64_load_chunk(h16x8<Hh> (&b)[8], const int cid, const ChainParams& p, const int w, const int lane, const int nj) {
  if (cid < 2) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) b[ks] = pfrag32((const Hh*)p.Wa, CH, w, 8 * cid + ks, lane);
  } else if (cid < 18) {
    const int c = (cid - 2) >> 2, q = (cid - 2) & 3;
    if (q < 2) {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) b[ks] = pfrag32((const Hh*)p.W1, CH, 8 * c + w, 8 * q + ks, lane);
    } else {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) b[ks] = pfrag32((const Hh*)p.W2, CI, w, 16 * c + 8 * (q - 2) + ks, lane);
    }
  } else {
    // UNCONDITIONAL, as in chain_load_chunk: passes beyond the projection's width re-read its last tile, a chain without one reads Wa
    const Hh* Wq = p.Wp ? (const Hh*)p.Wp : (const Hh*)p.Wa;
    const int j = (cid - 18) >> 1, hf = (cid - 18) & 1, nt = p.Wp ? 8 * (j < nj ? j : nj - 1) + w : w;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) b[ks] = pfrag32(Wq, CH, nt, 8 * hf + ks, lane);
  }
}
// LayerNorm of 64 rows x 256 columns held as two 32 x 32 accumulator tiles per wave (columns 32 w + (l & 31)): v = acc + bias + residual.
// Two-pass statistics as chain_norm.  A row's 256 columns sit in 16 half-rows of 16 lanes (8 waves x 2): per pass the 16 partials of every
// row go to LDS, ONE workgroup barrier, then every wave folds them for itself (lane L: row L) into a wave-private row of statistics -- no
// second barrier.  red: part1 [16][64] | part2 [16][64] | per-wave statistics [8][64].
template <typename Hh>
__device__ __forceinline__ void chain64_norm(f32x16 (&acc)[C6RT], const float* sPar, const Hh* sR, float* red, Hh* sOut, const float eps,
                                             const int w, const int lane) {
  const int col = 32 * w + (lane & 31), h = lane >> 5, sub = (lane >> 4) & 1;
  const float bv = sPar[col], gv = sPar[CH + col], btv = sPar[2 * CH + col];
  float* part1 = red;
  float* part2 = red + 16 * C6ROWS;
  float* stat = red + 32 * C6ROWS + w * C6ROWS;
#pragma unroll
  for (int rt = 0; rt < C6RT; ++rt)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = rt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
      acc[rt][i] += bv + to_f(sR[row * CP + col]);
      const float s = g16_sum(acc[rt][i]);
      if ((lane & 15) == 0) part1[(2 * w + sub) * C6ROWS + row] = s;
    }
  __syncthreads();
  {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += part1[q * C6ROWS + lane];
    stat[lane] = t * (1.0f / CH);
  }
  wave_lds_sync();
#pragma unroll
  for (int rt = 0; rt < C6RT; ++rt)
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
      const f32x4 m4 = *(const f32x4*)(stat + rt * 32 + 8 * i4 + 4 * h);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 4 * i4 + r, row = rt * 32 + r + 8 * i4 + 4 * h;
        acc[rt][i] -= m4[r];
        const float s = g16_sum(acc[rt][i] * acc[rt][i]);
        if ((lane & 15) == 0) part2[(2 * w + sub) * C6ROWS + row] = s;
      }
    }
  __syncthreads();
  {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += part2[q * C6ROWS + lane];
    wave_lds_sync();                         // (every lane of the wave has read its means before the row is rewritten)
    stat[lane] = rsqrtf(t * (1.0f / CH) + eps);
  }
  wave_lds_sync();
#pragma unroll
  for (int rt = 0; rt < C6RT; ++rt)
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
      const f32x4 r4 = *(const f32x4*)(stat + rt * 32 + 8 * i4 + 4 * h);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 4 * i4 + r, row = rt * 32 + r + 8 * i4 + 4 * h;
        sOut[row * CP + col] = from_f<Hh>(acc[rt][i] * r4[r] * gv + btv);
      }
    }
}
template <typename Hh>
__device__ __forceinline__ void chain64_rows_in(const Hh* src, long long ld, int nq, Hh* dst, int tid) {
  typedef __attribute__((ext_vector_type(4))) unsigned u4;
  for (int id = tid; id < C6ROWS * (CH / 8); id += NWAVE * 64) {
    const int r = id / (CH / 8), c = (id % (CH / 8)) * 8;
    const u4 v = r < nq ? *(const u4*)(src + r * ld + c) : (u4){0u, 0u, 0u, 0u};
    *(u4*)(dst + r * CP + c) = v;
  }
}
template <typename Hh, bool FFN>
__device__ __forceinline__ void chain64_body(const ChainParams& p, const int tile, unsigned char* smem) {
  Hh* sIn = (Hh*)smem;                      // [64][CP]  stage-1 input; later the block output y2
  Hh* sRes = sIn + C6ROWS * CP;             // [64][CP]  residual of stage 1
  Hh* sY1 = sRes + C6ROWS * CP;             // [64][CP]
  Hh* sG = sY1 + C6ROWS * CP;               // [64][CP]  GELU image of one 256-column chunk; later the projection's 256-column staging image
  float* red = (float*)(sG + C6ROWS * CP);  // LayerNorm scratch: see chain64_norm
  float* sPar = red + 40 * C6ROWS;          // ba | g1 | b1 | bo2 | g2 | b2 (256 each) | bi (1024) | bp (<= 768)
  const int tid = threadIdx.x, lane0 = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int lane = lane0;
  asm volatile("" : "+v"(lane));
  const int h = lane >> 5, c32 = lane & 31;
  const int row0 = tile * C6ROWS, nq = min(C6ROWS, p.M - row0);
  const int nj = p.Wp ? (p.Np >> 8) : 0;     // 256-column passes of the projection
  constexpr int NS = FFN ? 24 : 8;
#define SEQ6(s) (FFN ? (s) : ((s) < 2 ? (s) : (s) + 16))
#define AHEAD6(s) do { if ((s) + C6NB - 1 < NS) chain64_load_chunk<Hh>(ring[((s) + C6NB - 1) % C6NB], SEQ6((s) + C6NB - 1), p, w, lane, nj); } while (0)
  CH_MARK(0);
  chain64_rows_in((const Hh*)p.in + (long long)row0 * p.ld_in, p.ld_in, nq, sIn, tid);
  chain64_rows_in((const Hh*)p.res + (long long)row0 * CH, CH, nq, sRes, tid);
  if (tid < CH) {
    sPar[tid] = p.ba[tid]; sPar[CH + tid] = p.g1[tid]; sPar[2 * CH + tid] = p.b1[tid];
    if (FFN) { sPar[3 * CH + tid] = p.bo2[tid]; sPar[4 * CH + tid] = p.g2[tid]; sPar[5 * CH + tid] = p.b2[tid]; }
  }
  if (FFN) { sPar[6 * CH + tid] = p.bi[tid]; sPar[6 * CH + 512 + tid] = p.bi[512 + tid]; }
  for (int i = tid; i < nj * 256; i += NWAVE * 64) sPar[6 * CH + CI + i] = p.bp[i];
  h16x8<Hh> ring[C6NB][8];
#pragma unroll
  for (int s = 0; s < C6NB - 1; ++s) chain64_load_chunk<Hh>(ring[s], SEQ6(s), p, w, lane, nj);
  __syncthreads();
  CH_MARK(1);
  // ================= 1: y1 = LayerNorm(in Wa^T + ba + res) =================
  {
    f32x16 acc[C6RT];
#pragma unroll
    for (int rt = 0; rt < C6RT; ++rt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[rt][i] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      AHEAD6(i);
      KSTEP_FENCE();
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int rt = 0; rt < C6RT; ++rt) acc[rt] = mfma32(lfrag32(sIn, CP, rt * 32, (8 * i + ks) * 16, lane), ring[i % C6NB][ks], acc[rt]);
      KSTEP_FENCE();
    }
    CH_MARK(7);
    chain64_norm(acc, sPar, sRes, red, sY1, p.eps, w, lane);
  }
  __syncthreads();
  CH_MARK(2);
  if (p.y1) copy_out(sY1, CP, (Hh*)p.y1 + (long long)row0 * CH, CH, nq, CH, tid);
  const Hh* sLast = sY1;
  if constexpr (FFN) {
    f32x16 acc2[C6RT];
#pragma unroll
    for (int rt = 0; rt < C6RT; ++rt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc2[rt][i] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      Hh* gbuf = (c & 1) ? sRes : sG;
      // ---- 2a: g[:, 256 c ..] = gelu(y1 W1[256 c ..]^T + bi): this wave's 32 columns
      {
        f32x16 acc[C6RT];
#pragma unroll
        for (int rt = 0; rt < C6RT; ++rt)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[rt][i] = 0.f;
        const float bfc = sPar[6 * CH + 256 * c + 32 * w + c32];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          AHEAD6(2 + 4 * c + q);
          KSTEP_FENCE();
#pragma unroll
          for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int rt = 0; rt < C6RT; ++rt) acc[rt] = mfma32(lfrag32(sY1, CP, rt * 32, (8 * q + ks) * 16, lane), ring[(2 + 4 * c + q) % C6NB][ks], acc[rt]);
          KSTEP_FENCE();
        }
        // the image alternates between two buffers (the stage-1 residual is dead): the one written now was last read by 2b of chunk c - 2,
        // which every wave finished before it passed the barrier of chunk c - 1 -- one barrier per chunk
#pragma unroll
        for (int rt = 0; rt < C6RT; ++rt)
#pragma unroll
          for (int i = 0; i < 16; ++i)
            gbuf[(rt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h) * CP + 32 * w + c32] = from_f<Hh>(gelu_fast(acc[rt][i] + bfc));
      }
      __syncthreads();
      // ---- 2b: acc2 += g[:, 256 c ..] W2[:, 256 c ..]^T
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        AHEAD6(4 + 4 * c + q);
        KSTEP_FENCE();
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
          for (int rt = 0; rt < C6RT; ++rt) acc2[rt] = mfma32(lfrag32(gbuf, CP, rt * 32, (8 * q + ks) * 16, lane), ring[(4 + 4 * c + q) % C6NB][ks], acc2[rt]);
        KSTEP_FENCE();
      }
    }
    CH_MARK(3);
    chain64_norm(acc2, sPar + 3 * CH, sY1, red, sIn, p.eps, w, lane);
    __syncthreads();
    CH_MARK(4);
    copy_out(sIn, CP, (Hh*)p.y2 + (long long)row0 * CH, CH, nq, CH, tid);
    sLast = sIn;
  }
  if (nj) {
    // ================= 3: proj = y_last Wp^T + bp, one 256-column pass at a time (this wave's 32 columns of each) =================
    constexpr int S3 = FFN ? 18 : 2;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      f32x16 acc[C6RT];
#pragma unroll
      for (int rt = 0; rt < C6RT; ++rt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[rt][i] = 0.f;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        AHEAD6(S3 + 2 * j + q);
        KSTEP_FENCE();
        if (j < nj) {
#pragma unroll
          for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int rt = 0; rt < C6RT; ++rt) acc[rt] = mfma32(lfrag32(sLast, CP, rt * 32, (8 * q + ks) * 16, lane), ring[(S3 + 2 * j + q) % C6NB][ks], acc[rt]);
        }
        KSTEP_FENCE();
      }
      if (j < nj) {
        const float bpv = sPar[6 * CH + CI + 256 * j + 32 * w + c32];
        Hh* pbuf = (j & 1) ? sRes : sG;      // alternating staging images: the one written now was copied out two passes ago (one barrier per pass)
#pragma unroll
        for (int rt = 0; rt < C6RT; ++rt)
#pragma unroll
          for (int i = 0; i < 16; ++i)
            pbuf[(rt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h) * CP + 32 * w + c32] = from_f<Hh>(acc[rt][i] + bpv);
        __syncthreads();
        copy_out(pbuf, CP, (Hh*)p.proj + (long long)row0 * p.Np + 256 * j, p.Np, nq, 256, tid);
      }
    }
  }
  CH_MARK(6);
#undef SEQ6
#undef AHEAD6
}
template <typename Hh>
__global__ __launch_bounds__(512) void chain64_fwd_kernel(ChainPair pr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char chain_smem[];
  const int which = (int)blockIdx.x < pr.split ? 0 : 1;
  const ChainParams& p = pr.p[which];
  const int tile = which ? blockIdx.x - pr.split : blockIdx.x;
  if (p.W1) chain64_body<Hh, true>(p, tile, chain_smem);
  else chain64_body<Hh, false>(p, tile, chain_smem);
}
static size_t chain64_lds_bytes() { return (size_t)(4 * C6ROWS * CP) * 2 + (40 * C6ROWS + 6 * CH + CI + 3 * CH) * sizeof(float); }

#if CHAIN_QUARTERS
static size_t chain_lds_bytes() { return (size_t)(4 * CROWS * KP) * 2 + (2 * NWAVE * CROWS + 6 * CH + CI + 3 * CH) * sizeof(float); }
#else
static size_t chain_lds_bytes() { return (size_t)(3 * CROWS * KP + CROWS * KG) * 2 + (2 * NWAVE * CROWS + 6 * CH + CI + 3 * CH) * sizeof(float); }
#endif

static bool chain_valid(const ChainParams& p) {
  if (p.M <= 0 || !p.in || !p.res || !p.Wa || !p.ba || !p.g1 || !p.b1 || p.ld_in < CH || (p.ld_in & 7)) return false;
  if (p.W1 && (!p.bi || !p.W2 || !p.bo2 || !p.g2 || !p.b2 || !p.y2)) return false;
  if (p.Wp && (!p.bp || !p.proj || (p.Np != CH && p.Np != 2 * CH && p.Np != 3 * CH))) return false;
  if (!p.W1 && !p.y1 && !p.Wp) return false;          // nothing would be written
  return true;
}

// rows per workgroup: 1 = by launch size (the default, see launch_chain), 32 = chain_fwd_kernel always, 64 = chain64_fwd_kernel always.
// magic_chain_tile_rows(0) reads the setting, (1 | 32 | 64) sets it; MAGIC_CHAIN_ROWS=32 | 64 in the environment picks a fixed form from the start.
static int chain_rows_cfg = 0;
extern "C" int magic_chain_tile_rows(int rows) {
  if (!chain_rows_cfg) { const char* e = getenv("MAGIC_CHAIN_ROWS"); const int v = e ? atoi(e) : 1; chain_rows_cfg = (v == 32 || v == 64) ? v : 1; }
  if (rows == 1 || rows == 32 || rows == 64) chain_rows_cfg = rows;
  else if (rows != 0) return MAGIC_ERR_ARG;
  return chain_rows_cfg;
}

int launch_chain(int dtype, int variant, const void* pa_, const void* pb_, hipStream_t st) {
  (void)variant;
  ChainPair pr;
  pr.p[0] = *(const ChainParams*)pa_;
  pr.p[1] = pb_ ? *(const ChainParams*)pb_ : pr.p[0];
  // 64-row tiles for launches with more 32-row tiles than the chip has CUs (one workgroup per CU: such a launch runs in two rounds of 26 us,
  // in one of 39 us on 64-row tiles: the text || panorama pairs, 398 tiles at B = 48: 59.5 -> 46.9 us).  A launch that fits in one round
  // only gets longer on 64-row tiles (3840 rows: 29 -> 43 us) -- its CU time falls by a quarter, but the overlapped training step could not
  // tell the two policies apart (1.456 vs 1.457 ms, profiles/micro/r04_ab_chain64_threshold.txt), so the default keeps every launch at its
  // faster form.  MAGIC_CHAIN_64_MIN_TILES overrides the threshold of the by-size setting.
  static int min_tiles = -1;
  if (min_tiles < 0) {
    const char* e = getenv("MAGIC_CHAIN_64_MIN_TILES");
    hipDeviceProp_t pr; int d = 0; (void)hipGetDevice(&d);
    min_tiles = e ? atoi(e) : ((hipGetDeviceProperties(&pr, d) == hipSuccess ? pr.multiProcessorCount : 256) + 1);
  }
  const int t32 = (pr.p[0].M + 31) / 32 + (pb_ ? (pr.p[1].M + 31) / 32 : 0);
  const int cfg = magic_chain_tile_rows(0);
  const int rows = cfg == 1 ? (t32 >= min_tiles ? 64 : 32) : cfg;
  const int ta = (pr.p[0].M + rows - 1) / rows, tb = pb_ ? (pr.p[1].M + rows - 1) / rows : 0;
  pr.split = ta;
  static bool attr_done[3] = {false, false, false};
  if (!attr_done[dtype == DT_BF16 ? DT_BF16 : DT_F16]) {
    if (dtype == DT_BF16) {
      hipFuncSetAttribute((const void*)chain_fwd_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)chain_lds_bytes());
      hipFuncSetAttribute((const void*)chain64_fwd_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)chain64_lds_bytes());
    } else {
      hipFuncSetAttribute((const void*)chain_fwd_kernel<f16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)chain_lds_bytes());
      hipFuncSetAttribute((const void*)chain64_fwd_kernel<f16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)chain64_lds_bytes());
    }
    attr_done[dtype == DT_BF16 ? DT_BF16 : DT_F16] = true;
  }
  const size_t lds = rows == 64 ? chain64_lds_bytes() : chain_lds_bytes();
  if (dtype == DT_BF16) {
    if (rows == 64) hipLaunchKernelGGL(chain64_fwd_kernel<bf16>, dim3(ta + tb), dim3(512), lds, st, pr);
    else hipLaunchKernelGGL(chain_fwd_kernel<bf16>, dim3(ta + tb), dim3(512), lds, st, pr);
  } else {
    if (rows == 64) hipLaunchKernelGGL(chain64_fwd_kernel<f16>, dim3(ta + tb), dim3(512), lds, st, pr);
    else hipLaunchKernelGGL(chain_fwd_kernel<f16>, dim3(ta + tb), dim3(512), lds, st, pr);
  }
  return launch_status();
}

// ---- weights in MFMA-fragment order -------------------------------------------------------------------------------------------
// dst[off ..) of a [rows, cols] row-major matrix at src[off ..): for every 16-row x 32-column fragment (row tile nt, k-step ks) the 64
// lanes' operands back to back -- chunk ((nt (cols / 32) + ks) 64 + l) holds src[(16 nt + (l & 15)) cols + 32 ks + 8 (l >> 4) .. + 7].
// Same calling convention as magic_transpose_spans; 2-byte elements of either 16-bit type.
#define FSP_MAX 160
struct FSpans { long long off[FSP_MAX]; int rows[FSP_MAX]; int cols[FSP_MAX]; int blk0[FSP_MAX + 1]; int n; };
__global__ __launch_bounds__(256) void pack_frag_spans_kernel(const unsigned short* __restrict__ src, unsigned short* __restrict__ dst, FSpans t) {
  typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
  const int id = blockIdx.x;
  int s = 0;
  for (int i = 1; i < t.n; ++i) s += (id >= t.blk0[i]) ? 1 : 0;
  const int C = t.cols[s];
  const long long chunk = (long long)(id - t.blk0[s]) * 256 + threadIdx.x;      // 16-byte chunk of the packed span
  const int l = (int)(chunk & 63);
  const long long frag = chunk >> 6;
  const int kpr = C >> 5, nt = (int)(frag / kpr), ks = (int)(frag % kpr);
  const unsigned short* a = src + t.off[s] + (long long)(16 * nt + (l & 15)) * C + 32 * ks + 8 * (l >> 4);
  *(u16x8*)(dst + t.off[s] + chunk * 8) = *(const u16x8*)a;
}
extern "C" int magic_pack_frag_spans(const void* src, void* dst, int n, const long long* offs, const int* rows, const int* cols, void* stream) {
  if (!src || !dst || n < 0 || (n && (!offs || !rows || !cols)) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return MAGIC_ERR_ARG;
  for (int i = 0; i < n; ++i)
    if (rows[i] <= 0 || cols[i] <= 0 || rows[i] % 16 || cols[i] % 32 || ((long long)rows[i] * cols[i]) % 2048 || offs[i] % 8) return MAGIC_ERR_ARG;
  for (int base = 0; base < n; base += FSP_MAX) {
    FSpans t;
    t.n = n - base < FSP_MAX ? n - base : FSP_MAX;
    int blocks = 0;
    for (int i = 0; i < t.n; ++i) {
      t.off[i] = offs[base + i]; t.rows[i] = rows[base + i]; t.cols[i] = cols[base + i];
      t.blk0[i] = blocks;
      blocks += (int)(((long long)rows[base + i] * cols[base + i]) / 2048);
    }
    t.blk0[t.n] = blocks;
    hipLaunchKernelGGL(pack_frag_spans_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)src, (unsigned short*)dst, t);
  }
  return launch_status();
}

// One launch for a training step's weight layouts: every span's fragment-order copy of W (flag 1 -> dst_f) and of W^T (flag 2 -> dst_tf) from
// the row-major 16-bit shadow the AdamW kernel has just rewritten; 64 x 64 tiles, spans with rows and cols multiples of 64.
struct LSpans { long long off[FSP_MAX]; int rows[FSP_MAX]; int cols[FSP_MAX]; int flags[FSP_MAX]; int tile0[FSP_MAX + 1]; int n; };
__global__ __launch_bounds__(256) void layout_spans_kernel(const unsigned short* __restrict__ src, unsigned short* __restrict__ dst_f,
                                                           unsigned short* __restrict__ dst_tf, LSpans t) {
  typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
  __shared__ unsigned short tile[64][72];
  const int id = blockIdx.x;
  int s = 0;
  for (int i = 1; i < t.n; ++i) s += (id >= t.tile0[i]) ? 1 : 0;
  const int local = id - t.tile0[s], R = t.rows[s], C = t.cols[s], fl = t.flags[s];
  const int tc = C / 64, r0 = (local / tc) * 64, c0 = (local % tc) * 64;
  const unsigned short* a = src + t.off[s];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int ch = threadIdx.x + it * 256, r = ch >> 3, c = (ch & 7) * 8;
    const u16x8 v = *(const u16x8*)(a + (long long)(r0 + r) * C + c0 + c);
    *(u16x8*)&tile[r][c] = v;
    if (fl & 1) {        // W: the 16 bytes this thread holds ARE one lane's operand of fragment ((r0 + r) / 16, (c0 + c) / 32)
      const int nt = (r0 + r) >> 4, rr = (r0 + r) & 15, ks = (c0 + c) >> 5, g = ((c0 + c) & 31) >> 3;
      *(u16x8*)(dst_f + t.off[s] + ((long long)(nt * (C >> 5) + ks) * 64 + g * 16 + rr) * 8) = v;
    }
  }
  if (!(fl & 2)) return;
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {      // W^T [C, R]: rows c0 .. c0 + 63 (4 row tiles), k = r0 .. r0 + 63 (2 k-steps): 8 fragments x 64 lanes
    const int ch = threadIdx.x + it * 256, l = ch & 63, fr = ch >> 6, i = fr >> 1, j = fr & 1, g = l >> 4, nn = l & 15;
    u16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[32 * j + 8 * g + e][16 * i + nn];
    const int nt = (c0 >> 4) + i, ks = (r0 >> 5) + j;
    *(u16x8*)(dst_tf + t.off[s] + ((long long)(nt * (R >> 5) + ks) * 64 + l) * 8) = v;
  }
}
extern "C" int magic_layout_spans(const void* src, void* dst_f, void* dst_tf, int n, const long long* offs, const int* rows, const int* cols,
                                  const int* flags, void* stream) {
  if (!src || n < 0 || (n && (!offs || !rows || !cols || !flags)) || ((uintptr_t)src & 15) || ((uintptr_t)dst_f & 15) || ((uintptr_t)dst_tf & 15)) return MAGIC_ERR_ARG;
  for (int i = 0; i < n; ++i) {
    if (rows[i] <= 0 || cols[i] <= 0 || rows[i] % 64 || cols[i] % 64 || offs[i] % 8 || !(flags[i] & 3)) return MAGIC_ERR_ARG;
    if (((flags[i] & 1) && !dst_f) || ((flags[i] & 2) && !dst_tf)) return MAGIC_ERR_ARG;
  }
  for (int base = 0; base < n; base += FSP_MAX) {
    LSpans t;
    t.n = n - base < FSP_MAX ? n - base : FSP_MAX;
    int tiles = 0;
    for (int i = 0; i < t.n; ++i) {
      t.off[i] = offs[base + i]; t.rows[i] = rows[base + i]; t.cols[i] = cols[base + i]; t.flags[i] = flags[base + i];
      t.tile0[i] = tiles;
      tiles += (rows[base + i] / 64) * (cols[base + i] / 64);
    }
    t.tile0[t.n] = tiles;
    hipLaunchKernelGGL(layout_spans_kernel, dim3(tiles), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)src, (unsigned short*)dst_f,
                       (unsigned short*)dst_tf, t);
  }
  return launch_status();
}

extern "C" int magic_chain_supported(int dtype, int H, int I) { return dtype_is16(dtype) && H == CH && I == CI; }

extern "C" int magic_chain_fwd(int dtype, const void* params, int nbytes, void* stream) {
  if (!params || nbytes != (int)sizeof(ChainParams) || !dtype_is16(dtype)) return MAGIC_ERR_ARG;
  ChainParams p;
  memcpy(&p, params, sizeof(p));
  if (!chain_valid(p)) return MAGIC_ERR_ARG;
  if (group_record(KIND_CHAIN, dtype, 0, &p, sizeof(p))) return MAGIC_OK;
  return launch_chain(dtype, 0, &p, nullptr, (hipStream_t)stream);
}
