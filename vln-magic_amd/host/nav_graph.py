"""Greedy navigation (inference) with ONE HIP graph per decision step.

`GMapNavAgent.rollout` in test mode (feedback 'argmax', map_nav_src/r2r/agent.py:722-1160 under `torch.no_grad()`) issues ~150
kernel launches per step; with eager launches the step is host-bound at every batch size (2.0 ms at B = 1).  Here the step --
feature gather, panorama encoder, log append, map / viewpoint token gather, both cross-modal encoders, heads, fusion, softmax /
argmax -- is captured ONCE with static shapes and replayed:

  * views padded to 37 and map tokens to `Kmax` per episode (`NavPlanner(pad_V=, pad_K=)`): padded tokens are masked in every
    attention and get -inf logits, so the valid logits are unchanged (tests/test_rollout_gpu.py compares with the eager loop);
  * every plan array of a step lives at a fixed offset of ONE pinned staging buffer -> ONE device buffer (sizes are the maxima; the
    CSR entry arrays are simply longer than the entries in use);
  * the embedding log is written by index (`index_copy_` with device row indices), so the growing row offsets are data, not
    launch arguments;
  * the instruction is padded to `Lmax` and encoded eagerly once per episode batch; its K/V cache is copied into the static buffer;
  * the step's actions and stop probabilities come back in one small pinned copy.
"""
import numpy as np
import torch

from .lib import capture as _capture
from . import ops as O
from .nav_plan import NavPlanner


class GreedyNavigator:
    def __init__(self, model, feature_table, batch_size, Lmax=80, Kmax=48, Tmax=15, V=37, nnz_per_token=6):
        self.m, self.table, self.B = model, feature_table, batch_size
        self.dev = feature_table.device
        self.Lmax, self.Kmax, self.Tmax, self.V, self.Vp = Lmax, Kmax, Tmax, V, V + 2
        B, K, Vp, H = batch_size, Kmax, V + 2, model.net.H
        self.n_out = B * K + B * Vp
        self.nnz_max = B * K * nnz_per_token + B * Vp
        self.log_rows = Tmax * B * (V + 2)
        f32, i32, i64, u8 = np.float32, np.int32, np.int64, np.uint8
        spec = [("vp_rows", (B,), i32), ("view_order", (B, V), i32), ("loc_fts", (B, V, 7), f32), ("nav_types", (B, V), i32),
                ("view_lens", (B,), i32), ("pano_masks", (B, V), u8), ("gmap_logit_masks", (B, K), u8), ("gmap_step_ids", (B, K), i32), ("gmap_pos_fts", (B, K, 7), f32),
                ("gmap_pair_dists", (B, K, K), f32), ("gmap_visited_masks", (B, K), u8), ("gmap_masks", (B, K), u8),
                ("vp_pos_fts", (B, Vp, 14), f32), ("vp_nav_masks", (B, Vp), u8), ("vp_masks", (B, Vp), u8),
                ("fsrc", (B, K), i32), ("bw", (B, Vp), u8), ("csr_ptr", (self.n_out + 1,), i32), ("csr_idx", (self.nnz_max,), i32),
                ("csr_w", (self.nnz_max,), f32), ("rows_pe", (B * V,), i64), ("rows_pf", (B,), i64), ("rows_cls", (B,), i64)]
        off, self.layout = 0, {}
        for name, shape, dt in spec:
            off = (off + 15) & ~15
            n = int(np.prod(shape)) * np.dtype(dt).itemsize
            self.layout[name] = (off, n, shape, dt)
            off += n
        self.stage = torch.empty(off, dtype=torch.uint8, pin_memory=True)
        self.stage_np = self.stage.numpy()
        self.dbuf = torch.empty(off, dtype=torch.uint8, device=self.dev)
        self.d = {}
        for name, (o, n, shape, dt) in self.layout.items():
            self.d[name] = self.dbuf[o:o + n].view(getattr(torch, np.dtype(dt).name)).view(shape)
        self.log = torch.zeros(self.log_rows, H, dtype=model.net.dtype, device=self.dev)
        nl = 2 * model.net.cfg.num_x_layers
        self.txt_kv = torch.zeros(nl, B * Lmax, 2 * H, dtype=model.net.dtype, device=self.dev)
        self.txt_embeds = torch.zeros(B, Lmax, H, dtype=model.net.dtype, device=self.dev)
        self.txt_masks = torch.zeros(B, Lmax, dtype=torch.bool, device=self.dev)
        self.out_dev = torch.zeros(2, B, dtype=torch.float32, device=self.dev)          # [action, stop probability]
        self.out_host = torch.zeros(2, B, dtype=torch.float32, pin_memory=True)
        self.graph = None
        self.stream = torch.cuda.Stream(device=self.dev)

    # ---- one decision step on the static buffers (captured once) -------------------------------------------------
    def _step(self):
        d, m, B, K, V, Vp = self.d, self.m, self.B, self.Kmax, self.V, self.Vp
        H = m.net.H
        fts = torch.empty(B, V, self.table.shape[2], dtype=self.table.dtype, device=self.dev)
        O.view_gather(self.table, d["vp_rows"], d["view_order"], fts)
        pe, pm, pf, pa = m("panorama", dict(view_img_fts=fts, loc_fts=d["loc_fts"], nav_types=d["nav_types"], view_lens=d["view_lens"],
                                            already_dropout=True, pano_masks=d["pano_masks"].view(torch.bool)))
        self.log.index_copy_(0, d["rows_pe"], pe.reshape(B * V, H))
        self.log.index_copy_(0, d["rows_pf"], pf)
        g = torch.empty(self.n_out, H, dtype=self.log.dtype, device=self.dev)
        O.csr_gather(self.log, d["csr_ptr"], d["csr_idx"], d["csr_w"], g, self.n_out, H)
        lens = ([self.Lmax] * B, [K] * B, [Vp] * B)                    # only used for FLOP accounting
        outs = m("navigation", dict(gmap_img_embeds=g[:B * K].view(B, K, H), vp_img_embeds=g[B * K:].view(B, Vp, H),
                                    txt_embeds=self.txt_embeds, txt_kv=self.txt_kv, txt_masks=self.txt_masks,
                                    gmap_masks=d["gmap_masks"].view(torch.bool), vp_masks=d["vp_masks"].view(torch.bool), gmap_step_ids=d["gmap_step_ids"],
                                    gmap_logit_masks=d["gmap_logit_masks"],
                                    gmap_pos_fts=d["gmap_pos_fts"], gmap_pair_dists=d["gmap_pair_dists"],
                                    gmap_visited_masks=d["gmap_visited_masks"].view(torch.bool), gmap_vpids=None, vp_pos_fts=d["vp_pos_fts"],
                                    vp_nav_masks=d["vp_nav_masks"].view(torch.bool), vp_cand_vpids=None, host_lens=lens, fusion=(d["fsrc"], d["bw"])))
        self.log.index_copy_(0, d["rows_cls"], outs["cls_embeds"])
        logits = outs["fused_logits"]
        self.out_dev[0].copy_(logits.argmax(1).float())
        self.out_dev[1].copy_(torch.softmax(logits, 1)[:, 0])
        self.last_logits = logits
        self.out_host.copy_(self.out_dev, non_blocking=True)

    def _fill(self, plan):
        st = self.stage_np
        B, V = self.B, self.V
        arrays = {k: plan[k] for k in ("vp_rows", "view_order", "loc_fts", "nav_types", "view_lens", "gmap_step_ids", "gmap_pos_fts",
                                       "gmap_pair_dists", "gmap_visited_masks", "gmap_masks", "vp_pos_fts", "vp_nav_masks", "vp_masks",
                                       "fsrc", "bw")}
        ptr, idx, w = plan["csr"]
        if len(idx) > self.nnz_max or plan["K"] != self.Kmax or plan["V"] != V or plan["log_rows"] > self.log_rows:
            raise ValueError(f"episode exceeds the captured static shapes (K {plan['K']}/{self.Kmax}, V {plan['V']}/{V}, "
                             f"nnz {len(idx)}/{self.nnz_max}, log rows {plan['log_rows']}/{self.log_rows})")
        arrays.update(pano_masks=np.arange(V)[None] < np.asarray(plan["view_lens"])[:, None],
                      gmap_logit_masks=~np.asarray(plan["gmap_visited_masks"], bool) & np.asarray(plan["gmap_masks"], bool))
        arrays.update(csr_ptr=ptr, rows_pe=np.arange(plan["log_base"], plan["log_base"] + B * V, dtype=np.int64),
                      rows_pf=np.arange(plan["log_fused"], plan["log_fused"] + B, dtype=np.int64),
                      rows_cls=np.arange(plan["log_cls"], plan["log_cls"] + B, dtype=np.int64))
        for k, a in arrays.items():
            o, n, shape, dt = self.layout[k]
            a = np.ascontiguousarray(a)
            if a.dtype == np.bool_:
                a = a.view(np.uint8)
            st[o:o + n] = a.astype(dt, copy=False).reshape(-1).view(np.uint8)
        for k, a in (("csr_idx", idx), ("csr_w", w)):
            o, n, shape, dt = self.layout[k]
            b = np.ascontiguousarray(a).astype(dt, copy=False).reshape(-1).view(np.uint8)
            st[o:o + len(b)] = b

    @torch.no_grad()
    def run(self, env, obs, record=False):
        """greedy episodes for one batch; returns dict(traj, n_steps, decisions[, steps])"""
        m, B, dev = self.m, self.B, self.dev
        assert len(obs) == B
        m.eval()
        m.store.sync_shadow()
        pl = NavPlanner(env, obs, feedback="argmax", max_action_len=self.Tmax, train=False, pad_V=self.V, pad_K=self.Kmax)
        lang = pl.language()
        L = lang["txt_ids"].shape[1]
        if L > self.Lmax:
            raise ValueError(f"instruction of {L} tokens, static buffers hold {self.Lmax}")
        ids = np.zeros((B, self.Lmax), np.int64)
        ids[:, :L] = lang["txt_ids"]
        with torch.cuda.stream(self.stream):
            tmask = torch.arange(self.Lmax, device=dev)[None] < torch.as_tensor(lang["txt_lens"], device=dev)[:, None]
            txt, _ = m("language", dict(txt_ids=torch.from_numpy(ids).to(dev), txt_masks=tmask))
            self.txt_embeds.copy_(txt)
            self.txt_kv.copy_(m.text_kv(txt))
            self.txt_masks.copy_(tmask)
            self.log.zero_()
        steps, stop_probs, decisions = [], [], 0
        for t in range(self.Tmax):
            plan = pl.begin_step()
            decisions += int((~pl.ended).sum())
            self._fill(plan)
            with torch.cuda.stream(self.stream):
                self.dbuf.copy_(self.stage, non_blocking=True)
                if self.graph is None:                       # first step ever: run it eagerly once (allocator warm-up), then capture
                    self._step()
                    self.stream.synchronize()
                    self.log.zero_()
                    g = torch.cuda.CUDAGraph()
                    with _capture(g, stream=self.stream, capture_error_mode="relaxed"):
                        self._step()
                    self.graph = g
                self.graph.replay()
            self.stream.synchronize()
            a = self.out_host[0].numpy().astype(np.int64)
            stop_probs.append(self.out_host[1].numpy().copy())
            if record:
                steps.append(dict(logits=self.last_logits.detach().float().cpu()[:, :int(plan["gmap_lens"].max())], vpids=plan["gmap_vpids"]))
            done = pl.end_step(a)
            if record:
                steps[-1]["actions"] = list(pl.actions)
            if done:
                break
        traj = pl.finish(stop_probs)
        return dict(traj=traj, n_steps=len(stop_probs), decisions=decisions, steps=steps)
