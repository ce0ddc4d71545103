/* Native host helpers of the navigator planner (host/nav_plan.py, host/graph_map.py): the two loops that were a third of the planner's time
 * as Python.  Plain C (no GPU): the same arithmetic in the same order as the Python forms they replace, which stay as the reference
 * implementation (tests/test_navplan_cpu.py runs both).  Built by the Makefile into ../_magic_hostplan.so, bound with ctypes.
 *
 *  mp_hops_row   : len(FloydGraph.path(names[i], names[j])) for every j -- the recursive reconstruction over the `via` table of
 *                  map_nav_src/r2r/speaker_utils.py:527-546 (FloydGraph.path), memoised per call.
 *  mp_dtw_cands  : the nDTW expert's inner loops (map_nav_src/r2r/agent.py:356-363 with r2r/eval_utils.py cal_dtw): the DTW row of the walked
 *                  prefix extended by each candidate's connecting path; returns each candidate's last table entry.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define MP_MAXN 512

static int hops_rec(const int32_t* via, int ld, int a, int b, int16_t* memo, int n) {
  if (a == b) return 0;
  int16_t* m = memo + (size_t)a * n + b;
  if (*m >= 0) return *m;
  const int k = via[(size_t)a * ld + b];
  const int v = k < 0 ? 1 : hops_rec(via, ld, a, k, memo, n) + hops_rec(via, ld, k, b, memo, n);
  *m = (int16_t)v;
  return v;
}

/* out[j] = number of viewpoints after i up to and including j on the reconstructed path; memo: n*n int16 scratch (caller-owned) */
int mp_hops_row(const int32_t* via, int ld, int n, int i, int64_t* out, int16_t* memo) {
  if (n <= 0 || n > MP_MAXN || i < 0 || i >= n) return -1;
  memset(memo, 0xFF, (size_t)n * n * sizeof(int16_t));
  for (int j = 0; j < n; ++j) out[j] = hops_rec(via, ld, i, j, memo, n);
  return 0;
}

/* D: [nd, nd] shortest distances of the scan (row-major doubles); row: DTW row of the walked prefix, G + 1 entries (row[0] of the table's
 * first row is 0, the rest inf; later rows start with inf); ref: the G reference-path nodes; candidate c extends the prefix by
 * nodes[offs[c] .. offs[c+1]).  out_last[c] = last entry of the extended row.  tmp: 2 * (G + 1) doubles of scratch. */
int mp_dtw_cands(const double* D, int nd, const double* row, int G, const int32_t* ref, const int32_t* nodes, const int32_t* offs, int ncand,
                 double* out_last, double* tmp) {
  if (G <= 0 || ncand < 0) return -1;
  double* a = tmp;
  double* b = tmp + (G + 1);
  for (int c = 0; c < ncand; ++c) {
    memcpy(a, row, (size_t)(G + 1) * sizeof(double));
    for (int q = offs[c]; q < offs[c + 1]; ++q) {
      const double* dp = D + (size_t)nodes[q] * nd;
      double left = INFINITY;
      b[0] = INFINITY;
      for (int j = 1; j <= G; ++j) {
        const double up = a[j], diag = a[j - 1];
        double best = up < diag ? up : diag;
        if (left < best) best = left;
        left = b[j] = dp[ref[j - 1]] + best;
      }
      double* t = a; a = b; b = t;
    }
    out_last[c] = a[G];
  }
  return 0;
}

/* the same recurrence for ONE extension, returning the whole row (the walked prefix's own update) */
int mp_dtw_extend(const double* D, int nd, const double* row, int G, const int32_t* ref, const int32_t* nodes, int nnodes, double* out_row, double* tmp) {
  if (G <= 0) return -1;
  double* a = tmp;
  double* b = tmp + (G + 1);
  memcpy(a, row, (size_t)(G + 1) * sizeof(double));
  for (int q = 0; q < nnodes; ++q) {
    const double* dp = D + (size_t)nodes[q] * nd;
    double left = INFINITY;
    b[0] = INFINITY;
    for (int j = 1; j <= G; ++j) {
      const double up = a[j], diag = a[j - 1];
      double best = up < diag ? up : diag;
      if (left < best) best = left;
      left = b[j] = dp[ref[j - 1]] + best;
    }
    double* t = a; a = b; b = t;
  }
  memcpy(out_row, a, (size_t)(G + 1) * sizeof(double));
  return 0;
}
