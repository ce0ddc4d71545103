// Parameter blocks of the row-block pipeline kernel (rowblock.hip).
#pragma once
#include "common.hpp"

#define RB_ROWS 32
#define RB_NC 256          // output columns per chunk (4 waves x <= 64)
#define RB_MAXST 4
#define RB_LIN_LN 1        // y = LayerNorm(dropout(x W^T + b) + residual)
#define RB_LIN_ACT 2       // y = gelu(x W^T + b), optionally keeping the pre-activation
#define RB_LIN 3           // y = x W^T + b   (last stage only)

// host-visible stage descriptor (include/magic_hip.h declares the same struct)
struct magic_rb_stage {
  int kind, N, K;
  const void* W; int ldw; const float* bias;
  const void* res; int ldres; int res_stage;
  const float* gamma; const float* beta; float eps; float* rstd; unsigned drop_site;
  void* out; int ldo; void* pre; int ldpre;
};

struct RbStage {
  int kind, N, K, ldw, ldo, ldpre, ldres, in_buf, out_buf, res_buf;
  const void* W; const float* bias; void* out; void* pre; const void* res;
  const float* gamma; const float* beta; float* rstd; float eps; DropDesc drop;
};
struct RbParams {
  int M, ldx, K0, nstage, wn, ww, x_buf;
  const void* X;
  RbStage st[RB_MAXST];
};
