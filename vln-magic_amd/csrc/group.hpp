// Grouping of launches: between magic_group_begin() and magic_group_end(stream) the groupable entry points
// (magic_gemm, magic_attn_fwd/bwd, magic_linear_ln, magic_ln_bwd, magic_chain_fwd) RECORD their validated parameter blocks instead of
// launching; magic_group_end() then issues ONE kernel that serves both problems (block id < split -> problem A, else B)
// when the two records are compatible, or launches them one after the other otherwise.  Used by the host to run two
// independent same-shaped sub-networks (global || local co-attention encoders, text || panorama encoders) in lockstep
// with half the launches.  State is thread-local: begin / calls / end happen on one host thread.
#pragma once
#include <hip/hip_runtime.h>
#include <cstring>

enum GroupKind { KIND_NONE = 0, KIND_GEMM = 1, KIND_ATTN_FWD = 2, KIND_ATTN_BWD = 3, KIND_LLN = 4, KIND_LNB = 5, KIND_LLB = 7, KIND_CHAIN = 8, KIND_LNF = 9 };

struct GroupRec { int kind, dtype, variant; alignas(16) unsigned char blob[1024]; };
#define GROUP_CAP 8
struct GroupState { bool active; int n; GroupRec rec[GROUP_CAP]; };

GroupState& group_state();

// returns true if the call was recorded (caller must return MAGIC_OK without launching)
static inline bool group_record(int kind, int dtype, int variant, const void* params, size_t bytes) {
  GroupState& g = group_state();
  if (!g.active || g.n >= GROUP_CAP || bytes > sizeof(g.rec[0].blob)) return false;
  GroupRec& r = g.rec[g.n++];
  r.kind = kind; r.dtype = dtype; r.variant = variant;
  memcpy(r.blob, params, bytes);
  return true;
}

// per-kind launchers (pb == nullptr -> single problem); defined next to their kernels
int launch_gemm(int dtype, int layout, const void* pa, const void* pb, hipStream_t st);
int launch_gemm_n(int dtype, int layout, const void* const* ps, int n, hipStream_t st);     // n <= 8 problems, one launch
int launch_attn_fwd(int dtype, int variant, const void* pa, const void* pb, hipStream_t st);
int launch_attn_bwd(int dtype, int variant, const void* pa, const void* pb, hipStream_t st);
int launch_lln(int dtype, int ht, const void* pa, const void* pb, hipStream_t st);
int launch_lnb(int dtype, int nit, const void* pa, const void* pb, hipStream_t st);
int launch_llb(int dtype, int ht, const void* pa, const void* pb, hipStream_t st);
int launch_chain(int dtype, int variant, const void* pa, const void* pb, hipStream_t st);
int launch_lnf(int dtype, int H, const void* pa, const void* pb, hipStream_t st);
