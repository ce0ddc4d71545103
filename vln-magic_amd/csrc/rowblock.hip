// Row-block pipeline kernel (forward): a chain of up to 4 per-token linear stages run back to back on a block of 32 rows,
// with every intermediate activation held in LDS.  Everything a transformer block does AFTER its attention product is
// row-local, so the BertSelfOutput -> BertIntermediate -> BertOutput tail (and the NEXT block's Q/K/V projection) is one
// launch instead of four:
//     a   = LayerNorm(dropout(ctx Wo^T + bo) + x)            kind 1
//     g   = gelu(a W1^T + b1)                 (z = pre-act)   kind 2
//     out = LayerNorm(dropout(g W2^T + b2) + a)              kind 1   (residual = output of stage 0, read from LDS)
//     qkv = out Wqkv'^T + bqkv'                               kind 3   (the following layer's projection)
// At H = 128..256 these GEMMs are 1-16 k-tiles long: as separate launches each one pays a launch, a cold first load and a
// drain (~8-10 us in situ); chained in one workgroup the activations never leave the CU and only weights stream from L2.
//
// Geometry: 256 threads = 4 waves, 32 rows x up to 256 output columns per chunk; wave w owns columns [w*WC, (w+1)*WC) of the
// chunk (WC = chunk/4 in {16..64}), both 16-row M tiles.  A operand = LDS activation image [32][width + PAD] (k contiguous),
// B operand = the weight's [N][K] rows staged per k-tile as [chunk][BK + pad]; the next weight tile is prefetched into
// registers under the MFMAs.  LayerNorm statistics are a cross-wave LDS reduction, exactly as in linear_ln_kernel.
// Saved-for-backward tensors (a, rstd, z, g, out, rstd, qkv) are written once, straight from the epilogues.
#include "common.hpp"
#include "group.hpp"
#include "rowblock.hpp"

template <typename T> struct RT;
template <> struct RT<bf16> { typedef bf16x8 vec; static constexpr int VE = 8, BK = 64, SB = 72, KS = 2, PAD = 8; };
template <> struct RT<float> { typedef f32x4 vec; static constexpr int VE = 4, BK = 32, SB = 34, KS = 8, PAD = 4; };

__device__ __forceinline__ bf16x8 rb_afrag(const bf16* s, int stride, int row0, int k, int lane) {
  return *(const bf16x8*)(s + (row0 + (lane & 15)) * stride + k + 8 * (lane >> 4));
}
__device__ __forceinline__ float rb_afrag(const float* s, int stride, int row0, int k, int lane) {
  return s[(row0 + (lane & 15)) * stride + k + (lane >> 4)];
}
__device__ __forceinline__ bf16x8 rb_bfrag(const bf16* s, int n0, int ks, int lane) {
  return *(const bf16x8*)(s + (n0 + (lane & 15)) * RT<bf16>::SB + ks * 32 + 8 * (lane >> 4));
}
__device__ __forceinline__ float rb_bfrag(const float* s, int n0, int ks, int lane) {
  return s[(n0 + (lane & 15)) * RT<float>::SB + ks * 4 + (lane >> 4)];
}
__device__ __forceinline__ f32x4 rb_mma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 rb_mma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

template <typename T>
__device__ __forceinline__ void rb_body(const RbParams& p, const int bid, unsigned char* lds_raw) {
  typedef typename RT<T>::vec vec;
  constexpr int VE = RT<T>::VE, BK = RT<T>::BK, SB = RT<T>::SB, KS = RT<T>::KS, KSTEP = BK / KS;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c16 = lane & 15;
  const int m0 = bid * RB_ROWS, M = p.M;
  // LDS: two narrow activation images, one wide one, the weight tile, the LayerNorm reduction scratch
  T* buf[3];
  int bstr[3] = {p.wn + RT<T>::PAD, p.wn + RT<T>::PAD, p.ww + RT<T>::PAD};
  buf[0] = (T*)lds_raw;
  buf[1] = buf[0] + RB_ROWS * bstr[0];
  buf[2] = buf[1] + RB_ROWS * bstr[1];
  T* sB = buf[2] + RB_ROWS * bstr[2];
  float* red = (float*)(sB + RB_NC * SB);

  // ---- stage-0 input rows -> LDS image (rows >= M are zero)
  {
    const T* X = (const T*)p.X;
    const int vpr = p.K0 / VE;
    for (int id = tid; id < RB_ROWS * vpr; id += 256) {
      const int r = id / vpr, c = (id % vpr) * VE;
      vec z;
#pragma unroll
      for (int e = 0; e < VE; ++e) z[e] = (T)0.0f;
      if (m0 + r < M) z = *(const vec*)(X + (long long)(m0 + r) * p.ldx + c);
      T* d = buf[p.x_buf] + r * bstr[p.x_buf] + c;
      if constexpr (sizeof(T) == 2) { *(vec*)d = z; }
      else { d[0] = z[0]; d[1] = z[1]; d[2] = z[2]; d[3] = z[3]; }
    }
  }
  __syncthreads();

  for (int si = 0; si < p.nstage; ++si) {
    const RbStage& st = p.st[si];
    const T* sA = buf[st.in_buf];
    const int sas = bstr[st.in_buf];
    const T* W = (const T*)st.W;
    const int ktn = st.K / BK;
    for (int nb = 0; nb < st.N; nb += RB_NC) {
      const int ncur = min(RB_NC, st.N - nb);
      const int tpw = ncur >> 6, WC = tpw * 16;         // 16-column tiles per wave, columns per wave
      f32x4 acc[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      vec vb[RB_NC / 32];
      auto loadB = [&](int kt) {
#pragma unroll
        for (int i = 0; i < RB_NC / 32; ++i) {
          const int id = tid + 256 * i, row = id >> 3, cv = id & 7;
          if (row < ncur) vb[i] = *(const vec*)(W + (long long)(nb + row) * st.ldw + kt * BK + cv * VE);
        }
      };
      auto storeB = [&]() {
#pragma unroll
        for (int i = 0; i < RB_NC / 32; ++i) {
          const int id = tid + 256 * i, row = id >> 3, cv = id & 7;
          if (row < ncur) {
            T* d = sB + row * SB + cv * VE;
            if constexpr (sizeof(T) == 2) { *(vec*)d = vb[i]; }
            else { ((float2*)d)[0] = make_float2(vb[i][0], vb[i][1]); ((float2*)d)[1] = make_float2(vb[i][2], vb[i][3]); }
          }
        }
      };
      loadB(0);
      for (int kt = 0; kt < ktn; ++kt) {
        storeB();
        __syncthreads();
        if (kt + 1 < ktn) loadB(kt + 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const auto a0 = rb_afrag(sA, sas, 0, kt * BK + ks * KSTEP, lane), a1 = rb_afrag(sA, sas, 16, kt * BK + ks * KSTEP, lane);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (j < tpw) {
              const auto b = rb_bfrag(sB, w * WC + j * 16, ks, lane);
              acc[0][j] = rb_mma(a0, b, acc[0][j]);
              acc[1][j] = rb_mma(a1, b, acc[1][j]);
            }
        }
        __syncthreads();
      }
      // ---- epilogue of this chunk.  C/D map: col = lane&15, row = 4*(lane>>4) + r of each 16x16 tile
      float bv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = (j < tpw && st.bias) ? st.bias[nb + w * WC + j * 16 + c16] : 0.f;
      if (st.kind == RB_LIN) {
        T* out = (T*)st.out;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = m0 + i * 16 + 4 * g + r;
            if (row < M) {
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (j < tpw) out[(long long)row * st.ldo + nb + w * WC + j * 16 + c16] = from_f<T>(acc[i][j][r] + bv[j]);
            }
          }
      } else if (st.kind == RB_LIN_ACT) {
        T* out = (T*)st.out; T* pre = (T*)st.pre;
        T* so = buf[st.out_buf];
        const int sos = bstr[st.out_buf];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int rl = i * 16 + 4 * g + r, row = m0 + rl;
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (j < tpw) {
                const int col = nb + w * WC + j * 16 + c16;
                const float v = acc[i][j][r] + bv[j];
                const T gv = from_f<T>(gelu_f(v));
                so[rl * sos + col] = gv;
                if (row < M) {
                  if (pre) pre[(long long)row * st.ldpre + col] = from_f<T>(v);
                  out[(long long)row * st.ldo + col] = gv;
                }
              }
          }
      } else {          // RB_LIN_LN: the whole row lives in this chunk (N <= 256)
        const DropState dsn = drop_init(st.drop);
        const int N = st.N;
        float gv[4], btv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int col = w * WC + j * 16 + c16;
          gv[j] = (j < tpw) ? st.gamma[col] : 0.f; btv[j] = (j < tpw) ? st.beta[col] : 0.f;
        }
        float s[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int rl = i * 16 + 4 * g + r, row = m0 + rl;
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (j < tpw) {
                const int col = w * WC + j * 16 + c16;
                float v = acc[i][j][r] + bv[j];
                if (dsn.on) v *= drop_mul(dsn, (unsigned)(row * N + col));
                float rs;
                if (st.res_buf >= 0) rs = to_f(buf[st.res_buf][rl * bstr[st.res_buf] + col]);
                else rs = (st.res && row < M) ? to_f(((const T*)st.res)[(long long)row * st.ldres + col]) : 0.f;
                acc[i][j][r] = v + rs; t += acc[i][j][r];
              }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
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
            mean[i][r] = (red[rr] + red[32 + rr] + red[64 + rr] + red[96 + rr]) / N;
          }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (j < tpw) { const float d = acc[i][j][r] - mean[i][r]; t += d * d; }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
            s[i][r] = t;
          }
        if (c16 == 0) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[w * 32 + i * 16 + 4 * g + r] = s[i][r];
        }
        __syncthreads();
        T* out = (T*)st.out;
        T* so = st.out_buf >= 0 ? buf[st.out_buf] : nullptr;
        const int sos = st.out_buf >= 0 ? bstr[st.out_buf] : 0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int rr = i * 16 + 4 * g + r, row = m0 + rr;
            const float rstd = rsqrtf((red[rr] + red[32 + rr] + red[64 + rr] + red[96 + rr]) / N + st.eps);
            if (st.rstd && row < M && w == 0 && c16 == 0) st.rstd[row] = rstd;
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (j < tpw) {
                const int col = w * WC + j * 16 + c16;
                const T y = from_f<T>((acc[i][j][r] - mean[i][r]) * rstd * gv[j] + btv[j]);
                if (so) so[rr * sos + col] = y;
                if (row < M) out[(long long)row * st.ldo + col] = y;
              }
          }
      }
      __syncthreads();        // LDS activation written by this chunk is visible; sB / red are free again
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void rowblock_fwd_kernel(RbParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rb_lds[];
  rb_body<T>(p, blockIdx.x, rb_lds);
}
template <typename T>
__global__ __launch_bounds__(256) void rowblock_fwd_pair_kernel(RbParams a, RbParams b, int nA) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rb_lds[];
  if ((int)blockIdx.x < nA) rb_body<T>(a, blockIdx.x, rb_lds);
  else rb_body<T>(b, blockIdx.x - nA, rb_lds);
}

static size_t rb_lds_bytes(int dtype, int wn, int ww) {
  if (dtype == DT_BF16) return (size_t)(RB_ROWS * (2 * (wn + 8) + (ww + 8)) + RB_NC * 72) * 2 + 128 * 4;
  return (size_t)(RB_ROWS * (2 * (wn + 4) + (ww + 4)) + RB_NC * 34) * 4 + 128 * 4;
}
#define RB_LDS_MAX (160 * 1024)

// Build the kernel parameter block from the host-visible stage descriptors: validates shapes, assigns LDS images
// (two narrow ones ping-pong, ACT outputs go to the wide one).  Returns MAGIC_ERR_UNSUPPORTED for shapes the host should
// run as separate magic_gemm / magic_linear_ln launches.
static int rb_build(int dtype, int M, const void* X, int ldx, int K0, int nstage, const magic_rb_stage* s,
                    const void* drop_seed, float drop_p, RbParams& p) {
  if (M <= 0 || !X || nstage <= 0 || nstage > RB_MAXST || !s) return MAGIC_ERR_ARG;
  if (dtype != DT_F32 && dtype != DT_BF16) return MAGIC_ERR_ARG;
  if (!drop_args_ok(drop_seed, drop_p)) return MAGIC_ERR_ARG;
  const int ve = dtype == DT_BF16 ? 8 : 4, bk = dtype == DT_BF16 ? 64 : 32;
  if (K0 <= 0 || K0 % bk || ldx % ve || ((uintptr_t)X & 15)) return MAGIC_ERR_ARG;
  int wn = K0, ww = 0;
  for (int i = 0; i < nstage; ++i) {
    if (s[i].kind == RB_LIN_LN) wn = s[i].N > wn ? s[i].N : wn;
    if (s[i].kind == RB_LIN_ACT) ww = s[i].N > ww ? s[i].N : ww;
  }
  if (wn > 256) return MAGIC_ERR_UNSUPPORTED;
  if (ww == 0) ww = 8;
  if (rb_lds_bytes(dtype, wn, ww) > RB_LDS_MAX) return MAGIC_ERR_UNSUPPORTED;
  p = RbParams{};
  p.M = M; p.X = X; p.ldx = ldx; p.K0 = K0; p.nstage = nstage; p.wn = wn; p.ww = ww; p.x_buf = 0;
  int cur = 0, curw = K0;                 // LDS image holding the current stage input, its width
  int obuf[RB_MAXST];
  for (int i = 0; i < nstage; ++i) {
    const magic_rb_stage& h = s[i];
    RbStage& d = p.st[i];
    if (h.kind < RB_LIN_LN || h.kind > RB_LIN) return MAGIC_ERR_ARG;
    if (h.N <= 0 || h.N % 64 || h.K != curw || h.K % bk || !h.W || h.ldw % ve || ((uintptr_t)h.W & 15) || !h.out || h.ldo < h.N) return MAGIC_ERR_ARG;
    if ((long long)M * h.N > 0xFFFFFFFFll) return MAGIC_ERR_ARG;
    d.kind = h.kind; d.N = h.N; d.K = h.K; d.W = h.W; d.ldw = h.ldw; d.bias = h.bias; d.out = h.out; d.ldo = h.ldo;
    d.pre = h.pre; d.ldpre = h.ldpre; d.in_buf = cur; d.out_buf = -1; d.res_buf = -1;
    obuf[i] = -1;
    const bool last = (i == nstage - 1);
    if (h.kind == RB_LIN_LN) {
      if (h.N > 256 || !h.gamma || !h.beta) return MAGIC_ERR_ARG;
      d.gamma = h.gamma; d.beta = h.beta; d.eps = h.eps; d.rstd = h.rstd;
      d.drop = DropDesc{(drop_p > 0.f && h.drop_site) ? (const unsigned*)drop_seed : nullptr, h.drop_site, drop_p};
      if (h.res) { d.res = h.res; d.ldres = h.ldres; }
      else if (h.res_stage >= 0) {
        if (h.res_stage >= i || obuf[h.res_stage] < 0 || p.st[h.res_stage].N != h.N) return MAGIC_ERR_ARG;
        d.res_buf = obuf[h.res_stage];
      }
      // output image: a narrow one that is neither this stage's input nor a residual still needed later
      bool need = !last;
      for (int k = i + 1; k < nstage; ++k) if (!s[k].res && s[k].kind == RB_LIN_LN && s[k].res_stage == i) need = true;
      if (need) {
        int busy[2] = {0, 0};
        if (cur < 2) busy[cur] = 1;
        if (d.res_buf >= 0 && d.res_buf < 2) {
          bool later = false;
          for (int k = i + 1; k < nstage; ++k) if (!s[k].res && s[k].kind == RB_LIN_LN && s[k].res_stage == h.res_stage) later = true;
          // the residual is read element-wise by the thread that overwrites the same element -> its image may be reused
          if (later) busy[d.res_buf] = 1;
        }
        for (int k = i + 1; k < nstage; ++k)
          if (!s[k].res && s[k].kind == RB_LIN_LN && s[k].res_stage >= 0 && s[k].res_stage < i && obuf[s[k].res_stage] >= 0 && obuf[s[k].res_stage] < 2)
            busy[obuf[s[k].res_stage]] = 1;
        const int ob = !busy[0] ? 0 : (!busy[1] ? 1 : -1);
        if (ob < 0) return MAGIC_ERR_UNSUPPORTED;
        d.out_buf = ob; obuf[i] = ob; cur = ob; curw = h.N;
      }
    } else if (h.kind == RB_LIN_ACT) {
      if (last) return MAGIC_ERR_ARG;       // its output feeds the next stage
      d.out_buf = 2; obuf[i] = 2; cur = 2; curw = h.N;
    } else {
      if (!last) return MAGIC_ERR_ARG;      // a plain projection ends the chain (its output is not kept in LDS)
    }
  }
  return MAGIC_OK;
}

// LDS bytes the kernel needs for narrow width wn (stage input / LayerNorm outputs, <= 256) and wide width ww (GELU output);
// the host falls back to separate launches when this exceeds 160 KiB.
extern "C" int magic_rowblock_lds_bytes(int dtype, int wn, int ww) {
  if ((dtype != DT_F32 && dtype != DT_BF16) || wn <= 0 || wn > 256 || ww < 0) return MAGIC_ERR_ARG;
  return (int)rb_lds_bytes(dtype, wn, ww > 0 ? ww : 8);
}

extern "C" int magic_rowblock_fwd(int dtype, int M, const void* X, int ldx, int K0, int nstage, const magic_rb_stage* stages,
                                  const void* drop_seed, float drop_p, void* stream) {
  RbParams p;
  const int rc = rb_build(dtype, M, X, ldx, K0, nstage, stages, drop_seed, drop_p, p);
  if (rc) return rc;
  if (group_record(KIND_RB, dtype, 0, &p, sizeof(p))) return MAGIC_OK;
  return launch_rb(dtype, 0, &p, nullptr, (hipStream_t)stream);
}

int launch_rb(int dtype, int, const void* pa, const void* pb, hipStream_t st) {
  const RbParams& a = *(const RbParams*)pa;
  dim3 block(256);
  const int nA = (a.M + RB_ROWS - 1) / RB_ROWS;
  size_t shm = rb_lds_bytes(dtype, a.wn, a.ww);
  if (!pb) {
#define RB1(TY)                                                                                                                \
    do {                                                                                                                       \
      if (shm > 64 * 1024) (void)hipFuncSetAttribute((const void*)rowblock_fwd_kernel<TY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
      hipLaunchKernelGGL((rowblock_fwd_kernel<TY>), dim3(nA), block, shm, st, a);                                              \
    } while (0)
    if (dtype == DT_BF16) RB1(bf16); else RB1(float);
#undef RB1
    return launch_status();
  }
  const RbParams& b = *(const RbParams*)pb;
  const size_t sb = rb_lds_bytes(dtype, b.wn, b.ww);
  if (sb > shm) shm = sb;
  const int nB = (b.M + RB_ROWS - 1) / RB_ROWS;
#define RB2(TY)                                                                                                                \
  do {                                                                                                                         \
    if (shm > 64 * 1024) (void)hipFuncSetAttribute((const void*)rowblock_fwd_pair_kernel<TY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
    hipLaunchKernelGGL((rowblock_fwd_pair_kernel<TY>), dim3(nA + nB), block, shm, st, a, b, nA);                               \
  } while (0)
  if (dtype == DT_BF16) RB2(bf16); else RB2(float);
#undef RB2
  return launch_status();
}
