// Flat-buffer optimizer kernels.  All student parameters live in ONE contiguous fp32 buffer (and the
// gradients / Adam moments / bf16 shadow weights in congruent buffers), so the optimizer is two launches
// per step instead of ~8 torch kernels per parameter tensor (pretrain_src/optim/adamw.py:53-112), and the
// DDP all-reduce is a few large contiguous chunks.
#include "common.hpp"
#include "group.hpp"

// sum of squares -> out[0] (atomic, block-reduced); 16-byte loads, four in flight per lane.  1024-thread blocks, at most one per CU:
// every block ends in an atomic on the SAME address and those serialise in L2 (~20 ns each: 512 blocks spent 10 of 19 us there)
// step[0] = global_step: drives the learning-rate schedule and ALWAYS advances (the reference sets lr from global_step and counts a step whose
// update GradScaler skipped, train_r2r_magic.py loop / optim/sched.py); step[1] = the optimizer's state step: drives Adam's bias correction and is
// taken back by adamw_kernel when it skips (a skipped optimizer.step() leaves state['step'] alone).  Rounds 4-5 kept ONE word for both, so every skip
// stalled the warm-up / decay by a step.
struct SchedArgs { int* step; float lr0; int warmup, total; float b1, b2; float* lr_ss; };      // step == nullptr: no schedule work
__device__ __forceinline__ void sched_advance(const SchedArgs& a) {
  const int gs = a.step[0];          // global_step before this optimizer step
  a.step[0] = gs + 1;
  const int t = a.step[1] + 1;       // the optimizer's state step after this update
  a.step[1] = t;
  double f = gs < a.warmup ? (double)gs / (double)a.warmup : fmax(0.0, (double)(a.total - gs) / (double)(a.total - a.warmup));
  double lr = (double)a.lr0 * f;
  if (lr <= 0.0) lr = 1e-8;
  const double ss = lr * sqrt(1.0 - pow((double)a.b2, (double)t)) / (1.0 - pow((double)a.b1, (double)t));
  a.lr_ss[0] = (float)lr; a.lr_ss[1] = (float)ss;
}
__global__ __launch_bounds__(1024) void sumsq_kernel(long long n, const float* __restrict__ g, float* out, SchedArgs sa) {
  __shared__ float red[16];
  if (sa.step && blockIdx.x == 0 && threadIdx.x == 0) sched_advance(sa);      // the optimizer's schedule rides along (nothing here reads it)
  float s = 0.f;
  const long long n4 = n >> 2, stride = (long long)gridDim.x * 1024;
  for (long long i = (long long)blockIdx.x * 1024 + threadIdx.x; i < n4; i += 4 * stride) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = (i + u * stride < n4) ? ((const float4*)g)[i + u * stride] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 4; ++u) s += v[u].x * v[u].x + v[u].y * v[u].y + v[u].z * v[u].z + v[u].w * v[u].w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[(n4 << 2) + threadIdx.x]; s += v * v; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += red[w];
    atomicAdd(out, t);
  }
}

// HF AdamW (adamw.py:84-110): m,v update; p -= step_size * m / (sqrt(v) + eps); then p -= lr*wd*p.
// Gradient clipping folded in: g *= min(1, max_norm / (sqrt(sumsq) + 1e-6)) (clip_grad_norm_ semantics),
// and the gradient pre-scale `gscale` (e.g. 1/world_size after a sum all-reduce).  Optionally refreshes
// the bf16 shadow copy used by the MFMA kernels.
template <typename Hh>
__global__ __launch_bounds__(256) void adamw_kernel(long long n, float* p, float* g, float* m, float* v, Hh* shadow,
                                                    float lr, float b1, float b2, float eps, float wd, float step_size,
                                                    const float* sumsq, float max_norm, float gscale, const float* lr_ss, long long n_decay, int zero_g,
                                                    unsigned* overflow, float* scale_state, int* sched_step, int decay_first) {
  if (lr_ss) { lr = lr_ss[0]; step_size = lr_ss[1]; }     // device-side schedule (HIP-graph replay)
  // dynamic loss scale (fp16 storage; loss.hip step_rng_kernel owns the rule): the gradient buffer holds S x the gradient -- read 1 / S here, leave
  // `pending` = 1 (updated) or 2 (skipped) for the next step's prologue.  Nothing else reads scale_state between the loss kernels and that prologue.
  if (scale_state) gscale *= scale_state[1];
  const bool bad = sumsq && !isfinite(sumsq[0]);
  if (scale_state && blockIdx.x == 0 && threadIdx.x == 0) scale_state[3] = bad ? 2.f : 1.f;
  float clip = gscale;
  if (bad) {
    // GradScaler.step semantics: a skipped optimizer.step() does not advance the optimizer's STATE step -- take back the advance the schedule launch
    // made for this step (its bias-correction scalars were never used); global_step (sched_step[0], the lr schedule) stays advanced
    if (sched_step && blockIdx.x == 0 && threadIdx.x == 0) sched_step[1] -= 1;
    // an overflowed / NaN gradient (fp16 storage under a static gradient scale: an activation gradient past 65504 becomes inf): SKIP the
    // update as amp.GradScaler.step does (train_r2r_magic.py:370-371) -- weights and moments untouched, the gradient consumed (zeroed) so
    // the next step starts clean -- and count it where the host can see it.  Without this, clip = 0 and 0 x inf = NaN poisons p, m, v for good.
    if (overflow && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(overflow, 1u);
    if (zero_g)
      for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) g[i] = 0.f;
    return;
  }
  if (sumsq && max_norm > 0.f) {
    const float nrm = sqrtf(sumsq[0]) * gscale;
    clip = gscale * fminf(1.f, max_norm / (nrm + 1e-6f));
  }
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float gi = g[i] * clip;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    // decay_first: torch.optim.AdamW's order (the navigator's optimizer, map_nav_src/r2r/agent_base.py:122-137): p *= 1 - lr wd, THEN the Adam update;
    // otherwise the pretraining AdamW's (adamw.py:109-110): the update, then the decay of the updated value
    float pi = p[i];
    const bool dec = wd > 0.f && i < n_decay;               // [0, n_decay): the decayed group (optim/misc.py:13-22), the rest: biases / LayerNorm
    if (dec && decay_first) pi -= lr * wd * pi;
    pi -= step_size * mi / (sqrtf(vi) + eps);
    if (dec && !decay_first) pi -= lr * wd * pi;
    m[i] = mi; v[i] = vi; p[i] = pi;
    if (shadow) shadow[i] = (Hh)pi;
    if (zero_g) g[i] = 0.f;             // the next step's accumulators start from zero: saves its 40 MB fill launch
  }
}

template <typename Hh>
__global__ __launch_bounds__(256) void cast_f32_h16_kernel(long long n, const float* x, Hh* y) {
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = ((const float4*)x)[i];
    h16x4<Hh> o; o[0] = (Hh)v.x; o[1] = (Hh)v.y; o[2] = (Hh)v.z; o[3] = (Hh)v.w;
    ((h16x4<Hh>*)y)[i] = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) y[(n4 << 2) + threadIdx.x] = (Hh)x[(n4 << 2) + threadIdx.x];
}

template <typename Hh>
__global__ __launch_bounds__(256) void cast_h16_f32_kernel(long long n, const Hh* x, float* y) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = (float)x[i];
}

template <typename T>
__global__ __launch_bounds__(256) void add_kernel(long long n, const T* x, T* y) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = from_f<T>(to_f(y[i]) + to_f(x[i]));
}

// y += x_0 + ... + x_{n-1} (n <= 8 tensors of y's shape): fp32 sum, one rounding.  The key / value input gradients of the cross-modal blocks
// (3 blocks x 2 encoders, all with respect to the same text rows) are computed side by side into separate buffers and folded here.
struct AddN { const void* x[8]; int n; };
template <typename T>
__global__ __launch_bounds__(256) void add_n_kernel(long long n8, long long n, T* __restrict__ y, AddN a) {
  constexpr int VE = 16 / (int)sizeof(T);
  typedef __attribute__((ext_vector_type(VE))) T vec_t;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    vec_t v = *(const vec_t*)(y + i * VE);
    float acc[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] = to_f(v[e]);
    for (int j = 0; j < a.n; ++j) {
      const vec_t u = *(const vec_t*)((const T*)a.x[j] + i * VE);
#pragma unroll
      for (int e = 0; e < VE; ++e) acc[e] += to_f(u[e]);
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) v[e] = from_f<T>(acc[e]);
    *(vec_t*)(y + i * VE) = v;
  }
  if (blockIdx.x == 0)
    for (long long i = n8 * VE + threadIdx.x; i < n; i += 256) {
      float acc = to_f(y[i]);
      for (int j = 0; j < a.n; ++j) acc += to_f(((const T*)a.x[j])[i]);
      y[i] = from_f<T>(acc);
    }
}

static inline int nblocks(long long n, int per) {
  long long b = (n + per - 1) / per;
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

extern "C" int magic_sumsq(long long n, const float* g, float* out, void* stream) {
  if (n <= 0 || ((uintptr_t)g & 15)) return MAGIC_ERR_ARG;
  const int nb = nblocks(n, 4096);
  hipLaunchKernelGGL(sumsq_kernel, dim3(nb > 256 ? 256 : nb), dim3(1024), 0, (hipStream_t)stream, n, g, out, SchedArgs{});
  return launch_status();
}
// the same launch also advances the device-side schedule (magic_sched_step's arithmetic; `out` must have been zeroed earlier in the step)
extern "C" int magic_sumsq_sched(long long n, const float* g, float* out, int* step, float lr0, int warmup, int total, float b1, float b2,
                                 float* lr_ss, void* stream) {
  if (n <= 0 || ((uintptr_t)g & 15) || !step || !lr_ss || warmup <= 0 || total <= warmup) return MAGIC_ERR_ARG;
  const int nb = nblocks(n, 4096);
  hipLaunchKernelGGL(sumsq_kernel, dim3(nb > 256 ? 256 : nb), dim3(1024), 0, (hipStream_t)stream, n, g, out, SchedArgs{step, lr0, warmup, total, b1, b2, lr_ss});
  return launch_status();
}

extern "C" int magic_adamw(long long n, float* p, float* g, float* m, float* v, void* shadow, int shadow_dtype,
                           float lr, float b1, float b2, float eps, float wd, float step_size,
                           const float* sumsq, float max_norm, float gscale, const float* lr_ss, long long n_decay, int zero_grad,
                           unsigned* overflow, float* scale_state, int* sched_step, int decay_first, void* stream) {
  if (n <= 0 || (shadow && !dtype_is16(shadow_dtype))) return MAGIC_ERR_ARG;
  if (scale_state && !sumsq) return MAGIC_ERR_ARG;        // the skip decision needs the gradient norm
  if (n_decay < 0) n_decay = n;                    // the whole range is one group
  if (shadow && shadow_dtype == DT_F16)
    hipLaunchKernelGGL(adamw_kernel<f16>, dim3(nblocks(n, 256)), dim3(256), 0, (hipStream_t)stream, n, p, g, m, v, (f16*)shadow, lr, b1, b2, eps, wd,
                       step_size, sumsq, max_norm, gscale, lr_ss, n_decay, zero_grad, overflow, scale_state, sched_step, decay_first);
  else
    hipLaunchKernelGGL(adamw_kernel<bf16>, dim3(nblocks(n, 256)), dim3(256), 0, (hipStream_t)stream, n, p, g, m, v, (bf16*)shadow, lr, b1, b2, eps, wd,
                       step_size, sumsq, max_norm, gscale, lr_ss, n_decay, zero_grad, overflow, scale_state, sched_step, decay_first);
  return launch_status();
}

// dtype16 = DT_BF16 | DT_F16: the 16-bit side of the conversion; to16 != 0: fp32 -> 16-bit, else 16-bit -> fp32
extern "C" int magic_cast(int dtype16, int to16, long long n, const void* x, void* y, void* stream) {
  if (n <= 0 || !dtype_is16(dtype16)) return MAGIC_ERR_ARG;
  const hipStream_t st = (hipStream_t)stream;
  if (to16) {
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 7)) return MAGIC_ERR_ARG;
    if (dtype16 == DT_BF16) hipLaunchKernelGGL(cast_f32_h16_kernel<bf16>, dim3(nblocks(n, 1024)), dim3(256), 0, st, n, (const float*)x, (bf16*)y);
    else hipLaunchKernelGGL(cast_f32_h16_kernel<f16>, dim3(nblocks(n, 1024)), dim3(256), 0, st, n, (const float*)x, (f16*)y);
  } else {
    if (dtype16 == DT_BF16) hipLaunchKernelGGL(cast_h16_f32_kernel<bf16>, dim3(nblocks(n, 256)), dim3(256), 0, st, n, (const bf16*)x, (float*)y);
    else hipLaunchKernelGGL(cast_h16_f32_kernel<f16>, dim3(nblocks(n, 256)), dim3(256), 0, st, n, (const f16*)x, (float*)y);
  }
  return launch_status();
}

extern "C" int magic_add(int dtype, long long n, const void* x, void* y, void* stream) {
  if (n <= 0) return MAGIC_ERR_ARG;
  if (dtype == DT_BF16) hipLaunchKernelGGL(add_kernel<bf16>, dim3(nblocks(n, 256)), dim3(256), 0, (hipStream_t)stream, n, (const bf16*)x, (bf16*)y);
  else if (dtype == DT_F16) hipLaunchKernelGGL(add_kernel<f16>, dim3(nblocks(n, 256)), dim3(256), 0, (hipStream_t)stream, n, (const f16*)x, (f16*)y);
  else hipLaunchKernelGGL(add_kernel<float>, dim3(nblocks(n, 256)), dim3(256), 0, (hipStream_t)stream, n, (const float*)x, (float*)y);
  return launch_status();
}

extern "C" int magic_add_n(int dtype, long long n, int count, const void* const* xs, void* y, void* stream) {
  if (n <= 0 || count < 1 || count > 8 || !xs || !y || ((uintptr_t)y & 15)) return MAGIC_ERR_ARG;
  AddN a;
  a.n = count;
  for (int j = 0; j < 8; ++j) {
    a.x[j] = j < count ? xs[j] : nullptr;
    if (j < count && (!xs[j] || ((uintptr_t)xs[j] & 15))) return MAGIC_ERR_ARG;
  }
  const int ve = dtype_is16(dtype) ? 8 : 4;
  const long long n8 = n / ve;
  const int nb = nblocks(n8 > 0 ? n8 : 1, 256);
  if (dtype == DT_BF16) hipLaunchKernelGGL(add_n_kernel<bf16>, dim3(nb), dim3(256), 0, (hipStream_t)stream, n8, n, (bf16*)y, a);
  else if (dtype == DT_F16) hipLaunchKernelGGL(add_n_kernel<f16>, dim3(nb), dim3(256), 0, (hipStream_t)stream, n8, n, (f16*)y, a);
  else if (dtype == DT_F32) hipLaunchKernelGGL(add_n_kernel<float>, dim3(nb), dim3(256), 0, (hipStream_t)stream, n8, n, (float*)y, a);
  else return MAGIC_ERR_ARG;
  return launch_status();
}

extern "C" int magic_abi_version(void) { return 1; }

// device/runtime probe used by the host loader to fail loudly when no gfx950 device is present
extern "C" int magic_device_info(int* cu_count, int* clock_khz, char* arch, int arch_len) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return MAGIC_ERR_LAUNCH;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return MAGIC_ERR_LAUNCH;
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (clock_khz) *clock_khz = prop.clockRate;
  if (arch && arch_len > 0) {
    int i = 0;
    for (; i < arch_len - 1 && prop.gcnArchName[i]; ++i) arch[i] = prop.gcnArchName[i];
    arch[i] = 0;
  }
  return MAGIC_OK;
}

// dz = dy * act'(z)   (kind 1: GELU(erf), 2: ReLU) -- MLM transform head, where no GEMM sits between LN and the activation
template <typename T>
__global__ __launch_bounds__(256) void dact_kernel(long long n, const T* dy, const T* z, T* dz, int kind) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float zz = to_f(z[i]);
    const float d = kind == 1 ? dgelu_f(zz) : (zz > 0.f ? 1.f : 0.f);
    dz[i] = from_f<T>(to_f(dy[i]) * d);
  }
}

extern "C" int magic_dact(int dtype, int kind, long long n, const void* dy, const void* z, void* dz, void* stream) {
  if (n <= 0 || (kind != 1 && kind != 2)) return MAGIC_ERR_ARG;
  if (dtype == DT_BF16) hipLaunchKernelGGL(dact_kernel<bf16>, dim3(nblocks(n, 256)), dim3(256), 0, (hipStream_t)stream, n, (const bf16*)dy, (const bf16*)z, (bf16*)dz, kind);
  else if (dtype == DT_F16) hipLaunchKernelGGL(dact_kernel<f16>, dim3(nblocks(n, 256)), dim3(256), 0, (hipStream_t)stream, n, (const f16*)dy, (const f16*)z, (f16*)dz, kind);
  else hipLaunchKernelGGL(dact_kernel<float>, dim3(nblocks(n, 256)), dim3(256), 0, (hipStream_t)stream, n, (const float*)dy, (const float*)z, (float*)dz, kind);
  return launch_status();
}

// Device-side optimizer schedule so a captured HIP graph can be replayed: t = ++step; lr = lr0 * warmup_linear(t-1)
// (pretrain_src/optim/sched.py:17-30, <=0 -> 1e-8); step_size = lr * sqrt(1-b2^t) / (1-b1^t) (adamw.py:97-100).
__global__ void sched_step_kernel(int* step, float lr0, int warmup, int total, float b1, float b2, float* lr_ss, float* zero_me) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (zero_me) zero_me[0] = 0.f;   // the gradient-norm accumulator of the step that starts here (saves a fill launch)
  sched_advance(SchedArgs{step, lr0, warmup, total, b1, b2, lr_ss});
}

extern "C" int magic_sched_step(int* step, float lr0, int warmup, int total, float b1, float b2, float* lr_ss, float* zero_me, void* stream) {
  if (!step || !lr_ss || warmup <= 0 || total <= warmup) return MAGIC_ERR_ARG;
  hipLaunchKernelGGL(sched_step_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step, lr0, warmup, total, b1, b2, lr_ss, zero_me);
  return launch_status();
}

// ---- pair-grouping of launches (group.hpp) ---------------------------------------------------------------------------
GroupState& group_state() {
  static thread_local GroupState g = {};
  return g;
}

extern "C" int magic_group_begin(void) {
  GroupState& g = group_state();
  if (g.active) return MAGIC_ERR_ARG;
  g.active = true; g.n = 0;
  return MAGIC_OK;
}

static int launch_one(const GroupRec& r, const GroupRec* other, hipStream_t st) {
  const void* pb = other ? (const void*)other->blob : nullptr;
  switch (r.kind) {
    case KIND_GEMM: return launch_gemm(r.dtype, r.variant, r.blob, pb, st);
    case KIND_ATTN_FWD: return launch_attn_fwd(r.dtype, r.variant, r.blob, pb, st);
    case KIND_ATTN_BWD: return launch_attn_bwd(r.dtype, r.variant, r.blob, pb, st);
    case KIND_LLN: return launch_lln(r.dtype, r.variant, r.blob, pb, st);
    case KIND_LNB: return launch_lnb(r.dtype, r.variant, r.blob, pb, st);
    case KIND_LLB: return launch_llb(r.dtype, r.variant, r.blob, pb, st);
    case KIND_CHAIN: return launch_chain(r.dtype, r.variant, r.blob, pb, st);
    case KIND_LNF: return launch_lnf(r.dtype, r.variant, r.blob, pb, st);
    default: return MAGIC_ERR_ARG;
  }
}

// launches what was recorded since magic_group_begin(): records are independent by contract, so they are partitioned by
// (kind, dtype, variant); a class of GEMMs becomes ONE grouped launch (up to 8 problems), other kinds launch as pairs.
extern "C" int magic_group_end(void* stream) {
  GroupState& g = group_state();
  if (!g.active) return MAGIC_ERR_ARG;
  g.active = false;
  hipStream_t st = (hipStream_t)stream;
  int rc = MAGIC_OK;
  bool used[GROUP_CAP] = {};
  for (int i = 0; i < g.n && rc == MAGIC_OK; ++i) {
    if (used[i]) continue;
    int idx[GROUP_CAP], m = 0;
    for (int j = i; j < g.n; ++j)
      if (!used[j] && g.rec[j].kind == g.rec[i].kind && g.rec[j].dtype == g.rec[i].dtype && g.rec[j].variant == g.rec[i].variant) { idx[m++] = j; used[j] = true; }
    if (g.rec[i].kind == KIND_GEMM && m > 2) {
      const void* ps[GROUP_CAP];
      for (int k = 0; k < m; ++k) ps[k] = g.rec[idx[k]].blob;
      rc = launch_gemm_n(g.rec[i].dtype, g.rec[i].variant, ps, m, st);
    } else {
      for (int k = 0; k < m && rc == MAGIC_OK; k += 2) rc = launch_one(g.rec[idx[k]], k + 1 < m ? &g.rec[idx[k + 1]] : nullptr, st);
    }
  }
  g.n = 0;
  return rc;
}
