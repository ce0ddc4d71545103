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
/* bumped whenever an exported signature changes; host/hostplan.py refuses a library that reports another number (a stale .so called through ctypes
 * with an older mp_plan_nav argument list would write through the wrong pointers without any error).  MP_SRC_ID: sha256 prefix of this file, set by
 * the Makefile; the binding compares it with the source it sits next to. */
#define MP_ABI_VERSION 2
#ifndef MP_SRC_ID
#define MP_SRC_ID "unset"
#endif
int mp_abi_version(void) { return MP_ABI_VERSION; }
const char* mp_src_id(void) { return MP_SRC_ID; }

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

/* ---- the planner's per-sample loops (host/nav_plan.py NavPlanner.begin_nav / end_step) ------------------------------------------------------
 * The second half of a step's plan walks every episode's map: visited / unvisited split in arrival order, step ids, pair distances, the
 * embedding sources of every map token (agent.py:905-924: visited -> the visiting step's fused row, unvisited -> mean of the views that showed
 * it), the local -> global fusion map ([LINEAGE] DUET navigation forward) and the inputs of the position features (agent.py:175-251,290-328).
 * As Python over ~30 nodes x 16 episodes it was 2 ms per step -- more than the GPU needs for the step's forward.  The state stays in the numpy
 * arrays of host/graph_map.FloydGraph / GraphMap and of the planner (registered here by pointer); one call plans the whole batch.
 * Arithmetic and ORDER of every entry equal the Python form (tests/test_navplan_cpu.py runs both against the oracle). */
#include <stdlib.h>

typedef struct {
  double* d; int32_t* via; uint8_t* seen; int ld;          /* FloydGraph._d / _via / _seen (row pitch ld) */
  double* pos;                                             /* GraphMap.pos_by_id [cap][3] */
  int64_t* step;                                           /* node_step_ids by dense id */
  int64_t* fused; int32_t* vcount; int64_t* vrows; int vmax;   /* embedding bookkeeping by dense id */
} MpEpisode;
typedef struct { int B; MpEpisode* ep; int16_t* memo; } MpBatch;

void* mp_batch_new(int B) {
  MpBatch* h = (MpBatch*)calloc(1, sizeof(MpBatch));
  if (!h) return 0;
  h->B = B;
  h->ep = (MpEpisode*)calloc((size_t)B, sizeof(MpEpisode));
  h->memo = (int16_t*)malloc((size_t)MP_MAXN * MP_MAXN * sizeof(int16_t));
  if (!h->ep || !h->memo) { free(h->ep); free(h->memo); free(h); return 0; }
  return h;
}
void mp_batch_free(void* hh) {
  MpBatch* h = (MpBatch*)hh;
  if (!h) return;
  free(h->ep); free(h->memo); free(h);
}
int mp_batch_set(void* hh, int i, double* d, int32_t* via, uint8_t* seen, int ld, double* pos, int64_t* step, int64_t* fused, int32_t* vcount,
                 int64_t* vrows, int vmax) {
  MpBatch* h = (MpBatch*)hh;
  if (!h || i < 0 || i >= h->B) return -1;
  MpEpisode* e = &h->ep[i];
  e->d = d; e->via = via; e->seen = seen; e->ld = ld; e->pos = pos; e->step = step; e->fused = fused; e->vcount = vcount; e->vrows = vrows; e->vmax = vmax;
  return 0;
}

/* GraphMap.update_graph for one episode: edges cur -> candidates (straight-line distance of the positions already written to `pos`), then
 * FloydGraph.update(cur): relax every pair through cur, mark it visited (speaker_utils.py:511-526) */
int mp_update_graph(void* hh, int i, int n, int cur, const int32_t* cid, int nc) {
  MpBatch* h = (MpBatch*)hh;
  if (!h || i < 0 || i >= h->B || n <= 0 || n > MP_MAXN) return -1;
  MpEpisode* e = &h->ep[i];
  const int ld = e->ld;
  const double* p = e->pos + 3 * (size_t)cur;
  for (int c = 0; c < nc; ++c) {
    const int j = cid[c];
    const double* q = e->pos + 3 * (size_t)j;
    const double dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
    const double dis = sqrt(dx * dx + dy * dy + dz * dz);
    if (dis < e->d[(size_t)cur * ld + j]) {
      e->d[(size_t)cur * ld + j] = e->d[(size_t)j * ld + cur] = dis;
      e->via[(size_t)cur * ld + j] = e->via[(size_t)j * ld + cur] = -1;
    }
  }
  for (int a = 0; a < n; ++a) {
    const double dak = e->d[(size_t)a * ld + cur];
    for (int b = 0; b < n; ++b) {
      if (a == b) continue;
      const double c2 = dak + e->d[(size_t)cur * ld + b];
      if (c2 < e->d[(size_t)a * ld + b]) { e->d[(size_t)a * ld + b] = c2; e->via[(size_t)a * ld + b] = cur; }
    }
  }
  e->seen[cur] = 1;
  return 0;
}

int mp_plan_nav(void* hh, int t, int K, int V, const int32_t* n_nodes, const uint8_t* ended, const int32_t* ci, const int32_t* start,
                const int32_t* cid, const int32_t* coff, long long base, long long fused0, long long prev_cls0,
                int64_t* step_ids, float* pair, uint8_t* visited, int32_t* fsrc, uint8_t* bw, int32_t* order, int32_t* nv_out,
                double* tp, double* gd, int64_t* hp, int32_t* seg_off,
                int64_t* coo_o, int64_t* coo_s, float* coo_w, int coo_cap, int32_t* ncoo_out) {
  MpBatch* h = (MpBatch*)hh;
  if (!h) return -1;
  const int B = h->B, Vp = V + 2;
  int nco = 0, tot = 0;
  for (int i = 0; i < B; ++i) {
    MpEpisode* e = &h->ep[i];
    const int n = n_nodes[i], ld = e->ld, nc = coff[i + 1] - coff[i];
    const int32_t* c_i = cid + coff[i];
    if (n <= 0 || n + 2 > K || n > MP_MAXN || nc > V) return -2;
    const int cur = ci[i];
    /* (a) embedding bookkeeping of a live episode: the current viewpoint's fused row, the views that show still-unvisited neighbours */
    if (!ended[i]) {
      e->fused[cur] = fused0 + i;
      e->vcount[cur] = 0;
      for (int j = 0; j < nc; ++j) {
        const int k = c_i[j];
        if (!e->seen[k]) {
          if (e->vcount[k] >= e->vmax) return -3;
          e->vrows[(size_t)k * e->vmax + e->vcount[k]++] = base + (long long)i * V + j;
        }
      }
    }
    /* (b) token order: visited nodes, then unvisited, each in arrival order */
    int32_t* ord = order + (size_t)i * K;
    int nv = 0, m = 0;
    for (int k = 0; k < n; ++k) if (e->seen[k]) ord[m++] = k;
    nv = m;
    for (int k = 0; k < n; ++k) if (!e->seen[k]) ord[m++] = k;
    nv_out[i] = nv;
    uint8_t* vis = visited + (size_t)i * K;
    for (int k = 1; k < 2 + nv; ++k) vis[k] = 1;
    int64_t* sid = step_ids + (size_t)i * K;
    for (int k = 0; k < n; ++k) sid[2 + k] = e->step[ord[k]];
    /* (c) inputs of the position features: every map token, the start viewpoint, every candidate -- position, graph distance, hop count */
    seg_off[i] = tot;
    if (mp_hops_row(e->via, ld, n, cur, hp + tot, h->memo) != 0) return -4;       /* hops to every node, dense-id order (scratch: reordered below) */
    {
      int64_t hrow[MP_MAXN];
      memcpy(hrow, hp + tot, (size_t)n * sizeof(int64_t));
      for (int q = 0; q < n + 1 + nc; ++q) {
        const int k = q < n ? ord[q] : (q == n ? start[i] : c_i[q - n - 1]);
        tp[3 * (size_t)(tot + q)] = e->pos[3 * (size_t)k]; tp[3 * (size_t)(tot + q) + 1] = e->pos[3 * (size_t)k + 1]; tp[3 * (size_t)(tot + q) + 2] = e->pos[3 * (size_t)k + 2];
        gd[tot + q] = k == cur ? 0.0 : e->d[(size_t)cur * ld + k];
        hp[tot + q] = hrow[k];
      }
    }
    tot += n + 1 + nc;
    /* (d) pair distances between the map tokens (diagonal 0) */
    float* pr = pair + (size_t)i * K * K;
    for (int a = 0; a < n; ++a)
      for (int b = 0; b < n; ++b) pr[(size_t)(2 + a) * K + 2 + b] = a == b ? 0.0f : (float)e->d[(size_t)ord[a] * ld + ord[b]];
    /* (e) embedding sources */
    if (nco + 2 + n * 2 > coo_cap) return -5;
    if (t > 0) {
      coo_o[nco] = (long long)i * K + 1; coo_s[nco] = prev_cls0 + i; coo_w[nco++] = 1.0f;
      coo_o[nco] = (long long)B * K + (long long)i * Vp + 1; coo_s[nco] = prev_cls0 + i; coo_w[nco++] = 1.0f;
    }
    for (int k = 0; k < nv; ++k) { coo_o[nco] = (long long)i * K + 2 + k; coo_s[nco] = e->fused[ord[k]]; coo_w[nco++] = 1.0f; }
    for (int k = nv; k < n; ++k) {
      const int nid = ord[k], cnt = e->vcount[nid];
      if (cnt <= 0) return -6;
      if (nco + cnt > coo_cap) return -5;
      const float w = (float)(1.0 / (double)cnt);
      for (int r = 0; r < cnt; ++r) { coo_o[nco] = (long long)i * K + 2 + k; coo_s[nco] = e->vrows[(size_t)nid * e->vmax + r]; coo_w[nco++] = w; }
    }
    /* (f) local -> global logit fusion: views of visited nodes are "backtrack" views; an unvisited node takes the LAST view that shows it */
    int32_t* fs = fsrc + (size_t)i * K;
    uint8_t* bwr = bw + (size_t)i * Vp;
    fs[0] = 0;
    {
      int32_t loc_of[MP_MAXN];
      for (int k = 0; k < n; ++k) loc_of[k] = -2;
      for (int j = 0; j < nc; ++j) {
        if (e->seen[c_i[j]]) bwr[2 + j] = 1;
        else loc_of[c_i[j]] = j + 2;
      }
      for (int k = nv; k < n; ++k) fs[2 + k] = loc_of[ord[k]];
    }
  }
  seg_off[B] = tot;
  *ncoo_out = nco;
  return 0;
}
