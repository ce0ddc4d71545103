"""Mint golden vectors from the importable parts of the reference (authoring container only).

Run:  python tests/golden/mint_golden.py      (needs /root/reference; writes tests/golden/*.pt)
The reference's Python never ships; only the small tensors written here are committed.
Import recipe: SURVEY.md Appendix F.  What is minted:
  makd_primitives.pt  kd_loss / mse_loss / exponential_decay / invert_normalized_losses, both flavours
                      (pretrain_src/optim/kd_loss.py, map_nav_src/utils/kd_loss.py)
  makd_agent.pt       GMapNavAgent.compute_kd_losses (map_nav_src/r2r/agent.py:546-719) on fixed inputs
  adamw.pt            pretrain_src/optim/adamw.py AdamW.step x3 + optim/sched.py get_lr_sched table
  collate.pt          pretrain_src/data/tasks.py {mlm,sap,cfp}_collate on our synthetic samples
  ops.pt              map_nav_src/utils/ops.py pad_tensors / gen_seq_masks
  nav_loop.pt         map_nav_src/r2r/speaker_utils.py FloydGraph; map_nav_src/r2r/agent.py _language_variable,
                      _panorama_feature_variable_do, _nav_gmap_variable, _nav_vp_variable_mem, _teacher_action
  gmap_pos.pt         pretrain_src/data/dataset.py get_gmap_pos_fts (incl. the [stop] node's row) on a synthetic scan graph
  zdict.pt            map_nav_src/r2r/data_utils.py LoadZdict (read_*_tsv, load_img_tensor, load_instr_tensor) and
                      map_nav_src/utils/data.py KMeansPicker (read_tim_tsv, seeded K-means + random_pick_front_features) on
                      synthetic TSV rows generated here
Every step runs from a scratch working directory (the reference's parsers create output directories relative to the cwd).
"""
import importlib.util
import os
import sys
import types
from collections import defaultdict
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import numpy.ma  # noqa: E402  (import before aliasing np.bool, else numpy.ma breaks)
try:
    import scipy.sparse  # noqa
    import sklearn.cluster  # noqa
except Exception:
    pass
np.bool = bool
np.int = int
sys.dont_write_bytecode = True


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def stub(names):
    for n in names:
        try:
            __import__(n)
        except Exception:
            sys.modules[n] = types.ModuleType(n)


def mint_primitives():
    P = load_by_path("ref_kd_pre", f"{REF}/pretrain_src/optim/kd_loss.py")
    N = load_by_path("ref_kd_nav", f"{REF}/map_nav_src/utils/kd_loss.py")
    g = torch.Generator().manual_seed(7)
    B, K = 6, 9
    s = torch.randn(B, K, generator=g) * 2
    t = torch.randn(B, K, generator=g) * 2
    for b in range(B):
        s[b, K - 1 - (b % 3)] = float("-inf")
        t[b, K - 1 - (b % 3)] = float("-inf")
    w = torch.rand(B, generator=g)
    fs, ft = torch.randn(B, 5, 8, generator=g), torch.randn(B, 5, 8, generator=g)
    fa, fb = torch.randn(B, 2, 5, 5, generator=g), torch.randn(B, 2, 5, 5, generator=g)
    wbad = torch.rand(B + 1, generator=g)
    out = dict(s=s, t=t, w=w, fs=fs, ft=ft, fa=fa, fb=fb, wbad=wbad, cases={})
    c = out["cases"]
    for T in (1, 2):
        c[f"pre_kd_T{T}"] = P.kd_loss(s, t, temperature=T)
        c[f"pre_kd_T{T}_w"] = P.kd_loss(s, t, temperature=T, t_sample_weights=w)
        for lt in ("sum", "mean"):
            c[f"nav_kd_T{T}_{lt}"] = N.kd_loss(s, t, temperature=T, loss_type=lt)
            c[f"nav_kd_T{T}_{lt}_w"] = N.kd_loss(s, t, temperature=T, t_sample_weights=w, loss_type=lt)
    c["pre_mse"] = P.mse_loss(fs, ft)
    c["pre_mse_w"] = P.mse_loss(fs, ft, t_sample_weights=w)
    c["pre_mse_wbad"] = P.mse_loss(fs, ft, t_sample_weights=wbad)        # silent fallback :15-16
    c["pre_mse4_w"] = P.mse_loss(fa, fb, t_sample_weights=w)
    for lt in ("sum", "mean"):
        c[f"nav_mse_{lt}"] = N.mse_loss(fs, ft, loss_type=lt)
        c[f"nav_mse_{lt}_w"] = N.mse_loss(fs, ft, t_sample_weights=w, loss_type=lt)
        c[f"nav_mse4_{lt}_w"] = N.mse_loss(fa, fb, t_sample_weights=w, loss_type=lt)
    try:
        N.mse_loss(fs, ft, t_sample_weights=wbad)
        c["nav_mse_wbad_raises"] = torch.tensor(0)
    except ValueError:
        c["nav_mse_wbad_raises"] = torch.tensor(1)
    losses = torch.tensor([0.0, 1.0, 2.0, 0.3])
    out["losses"] = losses
    c["exp_decay_0.7"] = N.exponential_decay(losses, 0.7)
    c["exp_decay_pre_0.7"] = P.exponential_decay(losses, 0.7)
    c["invert_norm"] = N.invert_normalized_losses(losses)
    torch.save(out, os.path.join(HERE, "makd_primitives.pt"))
    print("makd_primitives:", len(c), "cases")


def mint_agent():
    sys.path.insert(0, f"{REF}/map_nav_src")
    stub(["MatterSim", "line_profiler", "jsonlines", "h5py", "spacy", "nltk", "tensorboardX", "progressbar"])
    models = types.ModuleType("models")
    for sub, attrs in (("graph_utils", ["GraphMap"]), ("model", ["VLNBert", "Critic"]), ("ops", ["pad_tensors_wgrad"])):
        m = types.ModuleType(f"models.{sub}")
        for a in attrs:
            setattr(m, a, object)
        sys.modules[f"models.{sub}"] = m
        setattr(models, sub, m)
    sys.modules["models"] = models
    import utils.kd_loss as K
    K.dkd_loss = None
    from r2r.agent import GMapNavAgent

    g = torch.Generator().manual_seed(11)
    B, L, Kn, V, Hs, Ht, hs, ht = 4, 12, 7, 9, 32, 64, 2, 4
    rnd = lambda *sh: torch.randn(*sh, generator=g)
    sm = lambda x: torch.softmax(x, -1)

    def outputs(H, h):
        o = defaultdict(lambda: None)
        o["txt_embeds"] = rnd(B, L, H)
        o["txt_attns"] = sm(rnd(B, h, L, L))
        o["pano_embeds"] = rnd(B, V, H)
        o["pano_fused_embeds"] = rnd(B, H)
        o["img_attns"] = sm(rnd(B, V, V))
        o["nav_outs"] = dict(gmap_embeds=rnd(B, Kn, H), vp_embeds=rnd(B, V + 2, H),
                             gmap_attns=sm(rnd(B, h, Kn, L)), vp_attns=sm(rnd(B, h, V + 2, L)))
        lg = rnd(B, Kn)
        lg[:, 1] = float("-inf")
        lg[0, 3] = float("-inf")
        o["nav_logits"] = lg
        return o

    s_out, t_out = outputs(Hs, hs), outputs(Ht, ht)
    t_out["sample_weights"] = torch.rand(B, generator=g)
    s_out["sample_weights"] = torch.rand(B, generator=g)
    torch.manual_seed(3)
    names = ["txt_emb_w", "kdl_img_w", "kdl_avg_img_w", "global_cross_w", "local_cross_w"]
    heads = {n: nn.Linear(Hs, Ht) for n in names}
    student_inner = SimpleNamespace(**heads)
    rw = torch.softmax(rnd(5) / 4.0, -1) * 5
    # 'learned_weight' (agent.py:583-586, ...): five raw scalars on s_model = the LEARNER's inner model (student for t2s, teacher for s2t);
    # drawn from their own generator so the cases minted in earlier rounds keep their values
    g2 = torch.Generator().manual_seed(23)
    lw_names = ["kdl_txt_weight", "kdl_img_weight", "kdl_global_weight", "kdl_local_weight", "kdl_predict_weight"]
    lw_student = {n: torch.randn(1, generator=g2) for n in lw_names}
    lw_teacher = {n: torch.randn(1, generator=g2) for n in lw_names}
    for n in lw_names:
        setattr(student_inner, n, nn.Parameter(lw_student[n].clone()))
    teacher_inner = SimpleNamespace(**{n: nn.Parameter(lw_teacher[n].clone()) for n in lw_names})
    nav_targets = torch.tensor([2, 0, 4, -100])
    plain = lambda o: {k: (dict(v) if isinstance(v, dict) else v) for k, v in o.items()}
    fx = dict(s_out=plain(s_out), t_out=plain(t_out), heads={n: (h.weight.data.clone(), h.bias.data.clone()) for n, h in heads.items()},
              rw=rw, nav_targets=nav_targets, cases={}, learned_student=lw_student, learned_teacher=lw_teacher)

    def run(t, role, mode, loss_type="sum"):
        args = SimpleNamespace(kd_loss_type=loss_type, kd_ability_types=["txt", "img", "global", "local", "action"],
                               train_kdl_noFeat=False, train_kdl_noAttn=False, train_kdl_noLogit=False,
                               kdl_temperature=2.0, kdl_adaptive_ability_weight=mode is not None,
                               kdl_adaptive_ability_weight_type=mode, kdl_logit_loss="kd",
                               kdl_dkd_alpha=1.0, kdl_dkd_beta=1.0, ignoreid=-100)
        me = SimpleNamespace(args=args, vln_bert=SimpleNamespace(vln_bert=student_inner),
                             teacher_vln_bert=SimpleNamespace(vln_bert=teacher_inner),
                             kdl_feat_loss=K.mse_loss, kdl_attn_loss=K.mse_loss, kdl_logit_loss=K.kd_loss)
        acc = defaultdict(float)
        if role == "t2s":
            res = GMapNavAgent.compute_kd_losses(me, t, s_out, t_out, acc, nav_targets, role="t2s", softmax_weights=rw)
        else:   # ICoD reverse: the teacher is the learner, the real student's heads project the target
            res = GMapNavAgent.compute_kd_losses(me, t, t_out, s_out, acc, nav_targets, role="s2t", softmax_weights=rw)
        return {k: torch.as_tensor(float(v)) for k, v in res.items()}

    for t in (0, 1):
        for mode in ("RW", None):
            for lt in ("sum", "mean"):
                fx["cases"][f"t2s_t{t}_{mode}_{lt}"] = run(t, "t2s", mode, lt)
        fx["cases"][f"s2t_t{t}_RW"] = run(t, "s2t", "RW")
        for lt in ("sum", "mean"):
            fx["cases"][f"t2s_t{t}_learned_weight_{lt}"] = run(t, "t2s", "learned_weight", lt)
        fx["cases"][f"s2t_t{t}_learned_weight"] = run(t, "s2t", "learned_weight")
    torch.save(fx, os.path.join(HERE, "makd_agent.pt"))
    print("makd_agent:", list(fx["cases"]))
    sys.path.remove(f"{REF}/map_nav_src")
    for k in [k for k in sys.modules if k.split(".")[0] in ("utils", "r2r", "models")]:
        del sys.modules[k]


def mint_pretrain_side():
    sys.path.insert(0, f"{REF}/pretrain_src")
    stub(["pynvml", "jsonlines", "h5py", "nltk", "lmdb", "msgpack_numpy", "tensorboardX", "easydict", "progressbar"])
    sys.modules["msgpack_numpy"].patch = lambda: None
    from optim.adamw import AdamW
    from optim.sched import get_lr_sched
    g = torch.Generator().manual_seed(5)
    p0 = [torch.randn(7, 5, generator=g), torch.randn(5, generator=g)]
    grads = [[torch.randn(7, 5, generator=g), torch.randn(5, generator=g)] for _ in range(3)]
    params = [nn.Parameter(p.clone()) for p in p0]
    opt = AdamW([{"params": [params[0]], "weight_decay": 0.01}, {"params": [params[1]], "weight_decay": 0.0}],
                lr=5e-5, betas=(0.9, 0.98))
    hist = []
    lrs = [5e-5, 4e-5, 1e-3]
    for st in range(3):
        for grp in opt.param_groups:
            grp["lr"] = lrs[st]
        for p, gr in zip(params, grads[st]):
            p.grad = gr.clone()
        opt.step()
        hist.append([p.data.clone() for p in params])
    opts = SimpleNamespace(learning_rate=5e-5, warmup_steps=10000, num_train_steps=200000)
    steps = [0, 1, 5000, 9999, 10000, 10001, 100000, 199999, 200000, 250000]
    torch.save(dict(p0=p0, grads=grads, lrs=lrs, hist=hist, sched_steps=steps,
                    sched=[get_lr_sched(s, opts) for s in steps]), os.path.join(HERE, "adamw.pt"))
    print("adamw ok")

    # collate: feed OUR synthetic samples to the reference collates
    import magic_amd
    from magic_amd.host import synth
    from data import tasks as T
    rng = np.random.default_rng([99, 0])
    import random
    pyrng = random.Random(99)
    samples = [synth.make_sample(rng, pyrng, uid=i, min_len=5, max_len=12, min_steps=2, max_steps=4) for i in range(3)]
    fx = {"samples": samples}
    ref_items = []
    for s in samples:
        it = dict(s)
        ref_items.append(it)
    sap = T.sap_collate([dict(x) for x in ref_items])
    cfp_items = [dict(x, extra_heads=None) for x in ref_items]
    cfp = T.cfp_collate(cfp_items)
    mrng = np.random.default_rng(5)
    mlm_items = []
    for s in samples:
        ids, labels = synth.random_word_mask(s["txt_ids"], mrng, 50265)
        it = {k: v for k, v in s.items() if k not in ("local_act_labels", "global_act_labels")}
        it["txt_ids"], it["txt_labels"] = ids, labels
        mlm_items.append(it)
    mlm = T.mlm_collate([dict(x) for x in mlm_items])
    keep = lambda b: {k: v for k, v in b.items() if torch.is_tensor(v) or isinstance(v, (list, type(None)))}
    fx.update(sap=keep(sap), cfp=keep(cfp), mlm=keep(mlm), mlm_items=mlm_items)
    torch.save(fx, os.path.join(HERE, "collate.pt"))
    print("collate ok", sorted(sap.keys()))
    sys.path.remove(f"{REF}/pretrain_src")
    for k in [k for k in sys.modules if k.split(".")[0] in ("utils", "data", "optim", "parser")]:
        del sys.modules[k]


def mint_mrc():
    """collate_mrc.pt: the reference's MRC item construction (_get_img_mask / _mask_img_feat, tasks.py:168-181, used by
    MrcDataset.__getitem__ :205-221) and mrc_collate (:263-310) on our synthetic samples, plus validate_mrc's two
    numbers (KL sum and soft-target accuracy, train_r2r_magic.py:470-489) restated from the reference primitives."""
    sys.path.insert(0, f"{REF}/pretrain_src")
    stub(["pynvml", "jsonlines", "h5py", "nltk", "lmdb", "msgpack_numpy", "tensorboardX", "easydict", "progressbar"])
    sys.modules["msgpack_numpy"].patch = lambda: None
    import random
    import magic_amd  # noqa: F401
    from magic_amd.host import synth
    from data import tasks as T
    rng = np.random.default_rng([99, 0])
    pyrng = random.Random(99)
    samples = [synth.make_sample(rng, pyrng, uid=i, min_len=5, max_len=12, min_steps=2, max_steps=4) for i in range(3)]
    items = []
    mrng = np.random.default_rng(11)
    for s in samples:
        it = {k: v for k, v in s.items() if k not in ("local_act_labels", "global_act_labels")}
        nv = it["traj_view_img_fts"][-1].shape[0]
        m = mrng.random(nv) < 0.15                       # the draw order synth.collate('mrc') uses
        if not m.any():
            m[int(mrng.integers(0, nv))] = True
        m = torch.from_numpy(m)
        views = list(it["traj_view_img_fts"])
        views[-1] = T._mask_img_feat(views[-1], m)       # reference masking primitive
        it["traj_view_img_fts"] = views
        it["vp_view_mrc_masks"] = m
        it["vp_view_probs"] = torch.softmax(torch.from_numpy(mrng.standard_normal((nv, 1000)).astype(np.float32)) * 2, -1)
        it["vp_angles"] = None
        items.append(it)
    batch = T.mrc_collate([dict(x) for x in items])
    keep = {k: v for k, v in batch.items() if torch.is_tensor(v) or isinstance(v, (list, type(None)))}
    # reference target extraction (tasks.py:183-187) = what the model is expected to return as view_targets
    targets = T._get_targets(batch["vp_view_probs"], batch["vp_view_mrc_masks"])
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(targets.shape[0], 1000, generator=g)
    import torch.nn.functional as F
    kl = F.kl_div(F.log_softmax(logits, dim=-1), targets, reduction="sum")            # validate_mrc :484-485
    n_correct = (logits.max(dim=-1)[1] == targets.max(dim=-1)[1]).sum().item()          # compute_accuracy_for_soft_targets :470-474
    ref = torch.load(os.path.join(HERE, "collate.pt"), weights_only=False)["samples"]      # same seeds -> same samples: not stored twice
    assert all(torch.equal(a["txt_ids"], b["txt_ids"]) and torch.equal(a["traj_view_img_fts"][-1], b["traj_view_img_fts"][-1])
               for a, b in zip(samples, ref))
    torch.save(dict(seed=11, mrc=keep, targets=targets, logits=logits, kl_sum=float(kl), n_correct=int(n_correct)),
               os.path.join(HERE, "collate_mrc.pt"))
    print("collate_mrc ok", sorted(keep.keys()), float(kl), n_correct)
    sys.path.remove(f"{REF}/pretrain_src")
    for k in [k for k in sys.modules if k.split(".")[0] in ("utils", "data", "optim", "parser")]:
        del sys.modules[k]


def mint_ragged():
    """collate_ragged.pt: sap_collate / mlm_collate (tasks.py:392-451, :110-176) on samples whose panoramas have 36 OR 37 view
    tokens (two candidates in one view): pins the padding of the view dimension and vp_lens / traj_vp_view_lens."""
    sys.path.insert(0, f"{REF}/pretrain_src")
    stub(["pynvml", "jsonlines", "h5py", "nltk", "lmdb", "msgpack_numpy", "tensorboardX", "easydict", "progressbar"])
    sys.modules["msgpack_numpy"].patch = lambda: None
    import random
    import magic_amd  # noqa: F401
    from magic_amd.host import synth
    from data import tasks as T
    rng = np.random.default_rng([77, 0])
    pyrng = random.Random(77)
    samples = [synth.make_sample(rng, pyrng, uid=i, min_len=5, max_len=9, min_steps=2, max_steps=4, dup_view_prob=0.5, img_dim=16)
               for i in range(4)]
    lens = sum([[x.shape[0] for x in s["traj_view_img_fts"]] for s in samples], [])
    assert 36 in lens and 37 in lens, lens
    sap = T.sap_collate([dict(x) for x in samples])
    keep = {k: v for k, v in sap.items() if torch.is_tensor(v) or isinstance(v, (list, type(None)))}
    torch.save(dict(seed=77, sap=keep), os.path.join(HERE, "collate_ragged.pt"))
    print("collate_ragged ok", lens, tuple(sap["traj_view_img_fts"].shape), tuple(sap["vp_pos_fts"].shape), sap["vp_lens"].tolist())
    sys.path.remove(f"{REF}/pretrain_src")
    for k in [k for k in sys.modules if k.split(".")[0] in ("utils", "data", "optim", "parser")]:
        del sys.modules[k]


def mint_ingest():
    """ingest.pt: R2RTextPathData.get_traj_pano_fts (pretrain_src/data/dataset.py:729-772) run UNBOUND on synthetic candidate
    tables and a small feature store; plus get_view_rel_angles / get_angle_fts tables (data/common.py:77-103)."""
    sys.path.insert(0, f"{REF}/pretrain_src")
    stub(["pynvml", "jsonlines", "h5py", "nltk", "lmdb", "msgpack_numpy", "tensorboardX", "easydict", "progressbar"])
    sys.modules["msgpack_numpy"].patch = lambda: None
    from data import dataset as DS
    from data.common import get_angle_fts, get_view_rel_angles
    rng = np.random.default_rng(17)
    D = 16
    scan = "scanA"
    vps = [f"vp{i}" for i in range(7)]
    store = {f"{scan}_{v}": rng.standard_normal((36, D + 5)).astype(np.float32) for v in vps}       # wider than image_feat_size
    cands = {}
    for i, v in enumerate(vps):
        n = int(rng.integers(1, 6))
        d = {}
        for j in range(n):
            vidx = int(rng.integers(0, 36)) if not (i == 3 and j == 1) else list(d.values())[0][0]   # vp3: two candidates share one view
            d[f"c{i}_{j}"] = [vidx, float(rng.uniform(0.5, 5)), float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-0.2, 0.2))]
        cands[f"{scan}_{v}"] = d
    fake = SimpleNamespace(scanvp_cands=cands, all_point_rel_angles=[get_view_rel_angles(baseViewId=i) for i in range(36)],
                           args=SimpleNamespace(correct_heading=False), angle_feat_size=4,
                           get_scanvp_feature=lambda sc, vp: store[f"{sc}_{vp}"][:, :D])
    paths = [vps[0:3], vps[2:7], vps[3:4]]
    outs = []
    for path in paths:
        f, loc, nav, cv, last = DS.R2RTextPathData.get_traj_pano_fts(fake, scan, path, 0.0, 0.0)
        outs.append(dict(fts=[torch.from_numpy(np.asarray(x)) for x in f], loc=[torch.from_numpy(np.asarray(x)) for x in loc],
                         nav=[list(x) for x in nav], cand=cv, last=torch.from_numpy(np.asarray(last))))
    torch.save(dict(D=D, scan=scan, vps=vps, store={k: torch.from_numpy(v) for k, v in store.items()}, cands=cands, paths=paths, outs=outs,
                    rel=[torch.from_numpy(get_view_rel_angles(baseViewId=i)) for i in (0, 12, 23)],
                    ang_in=torch.from_numpy(rng.uniform(-3, 3, (9, 2)).astype(np.float32))),
               os.path.join(HERE, "ingest.pt"))
    fx = torch.load(os.path.join(HERE, "ingest.pt"), weights_only=False)
    a = fx["ang_in"].numpy()
    fx["ang_fts4"] = torch.from_numpy(get_angle_fts(a[:, 0], a[:, 1], 4))
    fx["ang_fts8"] = torch.from_numpy(get_angle_fts(a[:, 0], a[:, 1], 8))
    torch.save(fx, os.path.join(HERE, "ingest.pt"))
    print("ingest ok", [len(o["fts"]) for o in outs], [tuple(x.shape) for x in outs[1]["fts"]])
    sys.path.remove(f"{REF}/pretrain_src")
    for k in [k for k in sys.modules if k.split(".")[0] in ("utils", "data", "optim", "parser")]:
        del sys.modules[k]


def mint_gmap_pos():
    """gmap_pos.pt: R2RTextPathData.get_gmap_pos_fts (pretrain_src/data/dataset.py:553-575) run UNBOUND on a synthetic scan graph: the 7
    position features of every map node relative to the current viewpoint, INCLUDING the [stop] node (vpid None), whose row the reference
    builds from rel_angles [0, 0] and rel_dists [0, 0, 0] -> [sin 0, cos 0, sin 0, cos 0, 0, 0, 0]."""
    sys.path.insert(0, f"{REF}/pretrain_src")
    stub(["pynvml", "jsonlines", "h5py", "nltk", "lmdb", "msgpack_numpy", "tensorboardX", "easydict", "progressbar"])
    sys.modules["msgpack_numpy"].patch = lambda: None
    from data import dataset as DS
    rng = np.random.default_rng(23)
    scan = "scanB"
    vps = [f"vp{i}" for i in range(6)]
    pos = {v: rng.uniform(-8, 8, 3).astype(np.float64) for v in vps}
    dist = {a: {b: float(np.linalg.norm(pos[a] - pos[b]) * 1.2) for b in vps} for a in vps}
    paths = {a: {b: [a] + [f"m{k}" for k in range(int(rng.integers(0, 4)))] + ([b] if b != a else []) for b in vps} for a in vps}
    fake = SimpleNamespace(graphs={scan: SimpleNamespace(nodes={v: {"position": pos[v]} for v in vps})},
                           shortest_distances={scan: dist}, shortest_paths={scan: paths}, angle_feat_size=4)
    cases = []
    for cur, ids, h, e in (("vp2", [None, "vp0", "vp1", "vp2", "vp4"], 0.0, 0.0), ("vp5", [None, "vp5", "vp3"], 0.7, -0.2), ("vp0", [None], 1.1, 0.3)):
        out = DS.R2RTextPathData.get_gmap_pos_fts(fake, scan, cur, ids, h, e)
        cases.append(dict(cur=cur, ids=ids, heading=h, elevation=e, out=torch.from_numpy(np.asarray(out))))
    torch.save(dict(scan=scan, vps=vps, pos={k: torch.from_numpy(v) for k, v in pos.items()}, dist=dist,
                    path_len={a: {b: len(paths[a][b]) for b in vps} for a in vps}, max_dist=DS.MAX_DIST, max_step=DS.MAX_STEP, cases=cases),
               os.path.join(HERE, "gmap_pos.pt"))
    print("gmap_pos ok", [tuple(c["out"].shape) for c in cases], cases[0]["out"][0].tolist())
    sys.path.remove(f"{REF}/pretrain_src")
    for k in [k for k in sys.modules if k.split(".")[0] in ("utils", "data", "optim", "parser")]:
        del sys.modules[k]


def mint_gmap_inputs():
    """gmap_inputs.pt: R2RTextPathData.get_gmap_inputs (pretrain_src/data/dataset.py:520-552) and get_vp_pos_fts (:555-565) run UNBOUND on a synthetic
    scan: a walk that revisits the frontier (a candidate seen at step 0 is visited at step 2), both act_visited_node settings."""
    sys.path.insert(0, f"{REF}/pretrain_src")
    stub(["pynvml", "jsonlines", "h5py", "nltk", "lmdb", "msgpack_numpy", "tensorboardX", "easydict", "progressbar"])
    sys.modules["msgpack_numpy"].patch = lambda: None
    from data import dataset as DS
    rng = np.random.default_rng(31)
    scan = "scanC"
    vps = [f"vp{i}" for i in range(9)]
    pos = {v: rng.uniform(-8, 8, 3).astype(np.float64) for v in vps}
    dist = {a: {b: float(np.linalg.norm(pos[a] - pos[b]) * 1.3) for b in vps} for a in vps}
    paths = {a: {b: [a] + [f"m{k}" for k in range(int(rng.integers(0, 3)))] + ([b] if b != a else []) for b in vps} for a in vps}
    cand_lists = {"vp0": ["vp1", "vp3", "vp2"], "vp1": ["vp0", "vp2", "vp4"], "vp2": ["vp1", "vp5", "vp0"], "vp3": ["vp0", "vp6"],
                  "vp4": ["vp1", "vp7"], "vp5": ["vp2", "vp8", "vp4"], "vp6": ["vp3"], "vp7": ["vp4"], "vp8": ["vp5"]}
    cands = {f"{scan}_{v}": {c: [0, 1.0, 0.0, 0.0] for c in cs} for v, cs in cand_lists.items()}
    cases = []
    for avn in (False, True):
        fake = SimpleNamespace(graphs={scan: SimpleNamespace(nodes={v: {"position": pos[v]} for v in vps})}, scanvp_cands=cands,
                               shortest_distances={scan: dist}, shortest_paths={scan: paths}, angle_feat_size=4, act_visited_node=avn)
        fake.get_gmap_pos_fts = lambda *a, _f=fake: DS.R2RTextPathData.get_gmap_pos_fts(_f, *a)
        for path, h, e in ((["vp0"], 0.0, 0.0), (["vp0", "vp1", "vp2"], 0.4, -0.1), (["vp0", "vp1", "vp2", "vp5", "vp4"], 2.1, 0.2)):
            ids, steps, vis, posf, pair = DS.R2RTextPathData.get_gmap_inputs(fake, scan, path, h, e)
            cand_ids = cand_lists[path[-1]]
            vpf = DS.R2RTextPathData.get_vp_pos_fts(fake, scan, path[0], path[-1], cand_ids, h, e, 36)
            cases.append(dict(act_visited_node=avn, path=path, heading=h, elevation=e, ids=ids, steps=list(steps), vis=list(vis),
                              pos=torch.from_numpy(np.asarray(posf)), pair=torch.from_numpy(np.asarray(pair)), cand=cand_ids,
                              vp_pos=torch.from_numpy(np.asarray(vpf))))
    torch.save(dict(scan=scan, vps=vps, pos={k: torch.from_numpy(v) for k, v in pos.items()}, dist=dist, cand_lists=cand_lists,
                    path_len={a: {b: len(paths[a][b]) for b in vps} for a in vps}, max_dist=DS.MAX_DIST, max_step=DS.MAX_STEP, cases=cases),
               os.path.join(HERE, "gmap_inputs.pt"))
    print("gmap_inputs ok", [(c["act_visited_node"], len(c["ids"]), tuple(c["vp_pos"].shape)) for c in cases])
    sys.path.remove(f"{REF}/pretrain_src")
    for k in [k for k in sys.modules if k.split(".")[0] in ("utils", "data", "optim", "parser")]:
        del sys.modules[k]


def mint_nav_loop():
    """nav_loop.pt: the reference's own FloydGraph (map_nav_src/r2r/speaker_utils.py:501-546) on a fixed edge/update script,
    and GMapNavAgent._language_variable / _panorama_feature_variable_do / _nav_gmap_variable / _nav_vp_variable_mem /
    _teacher_action (map_nav_src/r2r/agent.py:63-373) driven unbound over a teacher-forced walk through our synthetic stepper,
    with oracle/rollout_ref.RefGraphMap standing in for the withheld GraphMap.  Tensor.cuda is made the identity for this
    script (no GPU in the authoring container; the methods only use it to place their results)."""
    sys.path.insert(0, f"{REF}/map_nav_src")
    stub(["MatterSim", "line_profiler", "jsonlines", "h5py", "spacy", "nltk", "tensorboardX", "progressbar"])
    from oracle import rollout_ref as R
    from magic_amd.host.synth_env import SynthNavEnv
    models = types.ModuleType("models")
    for sub, attrs in (("graph_utils", ["GraphMap"]), ("model", ["VLNBert", "Critic"]), ("ops", ["pad_tensors_wgrad"])):
        m = types.ModuleType(f"models.{sub}")
        for a in attrs:
            setattr(m, a, object)
        sys.modules[f"models.{sub}"] = m
        setattr(models, sub, m)
    sys.modules["models"] = models
    import utils.kd_loss as K
    K.dkd_loss = None
    import r2r.agent as A
    A.pad_tensors_wgrad = lambda ts: R.pad_rows(ts)
    torch.Tensor.cuda = lambda self, *a, **k: self
    argv, sys.argv = sys.argv, [sys.argv[0], "--mode", "train", "--root_dir", "/tmp/none"]   # the module parses argv on import (:14-15)
    try:
        F_ = load_by_path("ref_speaker_utils", f"{REF}/map_nav_src/r2r/speaker_utils.py").FloydGraph
    finally:
        sys.argv = argv
    # -- FloydGraph script
    rng = np.random.default_rng(5)
    names = [f"n{i}" for i in range(12)]
    fg, script, answers = F_(), [], []
    for step in range(36):
        x, y = (int(v) for v in rng.choice(len(names), 2, replace=False))
        d = float(rng.uniform(0.5, 6))
        fg.add_edge(names[x], names[y], d)
        script.append(("edge", x, y, d))
        if step % 3 == 2:
            k = int(rng.integers(len(names)))
            if names[k] in fg._dis:
                fg.update(names[k])
                script.append(("update", k, 0, 0.0))
        known = [n for n in names if n in fg._dis]
        answers.append([(names.index(u), names.index(v), float(fg.distance(u, v)),
                         [names.index(q) for q in fg.path(u, v)] if fg.distance(u, v) < 9e7 else None, fg.visited(u))
                        for u in known for v in known])
    out = dict(floyd=dict(names=names, script=script, answers=answers))
    # -- agent builders over a teacher-forced walk
    feat = 16
    env = SynthNavEnv(batch_size=4, n_scans=2, nodes_per_scan=30, feat_dim=feat, seed=21, instr_len=(5, 11), path_hops=(2, 4))
    obs = env.reset()
    args = SimpleNamespace(image_feat_size=feat, act_visited_nodes=False, enc_full_graph=True, ignoreid=-100, expert_policy="spl",
                           fusion="dynamic")
    me = SimpleNamespace(args=args, env=env)
    g = torch.Generator().manual_seed(9)
    H = 8
    gmaps = [R.RefGraphMap(ob["viewpoint"]) for ob in obs]
    for gm, ob in zip(gmaps, obs):
        gm.update_graph(ob)
    traj = [dict(path=[[ob["viewpoint"]]]) for ob in obs]
    ended = np.array([False] * len(obs))
    last = None
    keep = lambda ob: {k: (v if k != "candidate" else [dict(c) for c in v]) for k, v in ob.items()}
    steps = []
    out["lang"] = dict(obs=[keep(o) for o in obs])
    lv = A.GMapNavAgent._language_variable(me, obs, None, None)
    out["lang"]["txt_ids"], out["lang"]["txt_masks"] = lv["txt_ids"], lv["txt_masks"]
    for t in range(5):
        for i, gm in enumerate(gmaps):
            if not ended[i]:
                gm.node_step_ids[obs[i]["viewpoint"]] = t + 1
        pano = A.GMapNavAgent._panorama_feature_variable_do(me, obs)
        B, V = pano["view_img_fts"].shape[:2]
        pe, pf = torch.randn(B, V, H, generator=g), torch.randn(B, H, generator=g)
        for i, gm in enumerate(gmaps):
            if ended[i]:
                continue
            gm.update_node_embed(obs[i]["viewpoint"], pf[i], rewrite=True)
            for j, cv in enumerate(pano["cand_vpids"][i]):
                if not gm.graph.visited(cv):
                    gm.update_node_embed(cv, pe[i, j])
        nav = A.GMapNavAgent._nav_gmap_variable(me, obs, gmaps, last, teacher=False)
        nav.update(A.GMapNavAgent._nav_vp_variable_mem(me, obs, gmaps, pe, pano["cand_vpids"], pano["view_lens"], pano["nav_types"], last))
        tgt_il = A.GMapNavAgent._teacher_action(me, obs, nav["gmap_vpids"], ended, visited_masks=nav["gmap_visited_masks"],
                                                imitation_learning=True, t=t, traj=traj)
        tgt_spl = A.GMapNavAgent._teacher_action(me, obs, nav["gmap_vpids"], ended, visited_masks=nav["gmap_visited_masks"],
                                                 imitation_learning=False, t=t, traj=traj)
        args.expert_policy = "ndtw"
        tgt_ndtw = A.GMapNavAgent._teacher_action(me, obs, nav["gmap_vpids"], ended, visited_masks=nav["gmap_visited_masks"],
                                                  imitation_learning=False, t=t, traj=traj)
        args.expert_policy = "spl"
        steps.append(dict(obs=[keep(o) for o in obs], ended=ended.copy(), pe=pe, pf=pf, last=last,
                          pano={k: v for k, v in pano.items() if v is not None},
                          nav={k: v for k, v in nav.items() if v is not None},
                          tgt_il=tgt_il, tgt_spl=tgt_spl, tgt_ndtw=tgt_ndtw, traj=[dict(path=[list(p) for p in x["path"]]) for x in traj]))
        last = torch.randn(B, H, generator=g)
        acts, hops = [], [None] * B
        for i, ob in enumerate(obs):
            gt = ob["gt_path"]
            if ended[i] or t >= len(gt) - 1:
                acts.append(None)
            else:
                acts.append(gt[t + 1])
                traj[i]["path"].append(gmaps[i].graph.path(ob["viewpoint"], gt[t + 1]))
                hops[i] = traj[i]["path"][-2][-1] if len(traj[i]["path"][-1]) == 1 else traj[i]["path"][-1][-2]
        env.step(acts, hops)
        obs = env._get_obs()
        for i, ob in enumerate(obs):
            if not ended[i]:
                gmaps[i].update_graph(ob)
        ended = np.logical_or(ended, np.array([a is None for a in acts]))
        if ended.all():
            break
    out["steps"] = steps
    out["env"] = dict(shortest={n: sc.sdist for n, sc in env.scans.items()}, vps={n: sc.vps for n, sc in env.scans.items()},
                      nxt={n: sc.nxt for n, sc in env.scans.items()})
    torch.save(out, os.path.join(HERE, "nav_loop.pt"))
    print("nav_loop: floyd", len(script), "ops;", len(steps), "agent steps")


def mint_zdict():
    """zdict.pt: the reference's dictionary loaders on synthetic rows (features are random fp32 vectors written in the reference's
    own TSV/base64 format by host/zdict.py's writers; the fixture stores those inputs and what the reference parsed from them)."""
    import tempfile
    sys.path.insert(0, f"{REF}/map_nav_src")
    stub(["MatterSim", "line_profiler", "jsonlines", "h5py", "spacy", "nltk", "tensorboardX", "progressbar", "sklearnex"])
    import magic_amd  # noqa: F401
    from magic_amd.host import zdict as Z
    DU = load_by_path("ref_data_utils", f"{REF}/map_nav_src/r2r/data_utils.py")
    UD = load_by_path("ref_utils_data", f"{REF}/map_nav_src/utils/data.py")
    rng = np.random.default_rng(11)
    img_rows = [(f"room{i}", rng.standard_normal(12).astype(np.float32), float(p)) for i, p in enumerate(rng.dirichlet(np.ones(5)))]
    txt_rows = []
    for i, p in enumerate(rng.dirichlet(np.ones(7))):
        txt_rows.append(("direction" if i % 3 else "landmark", f"tok{i}", rng.standard_normal(8).astype(np.float32), float(p)))
    n = 40
    txt, vp, gm = (rng.standard_normal((n, 6)).astype(np.float32) + 3 * (np.arange(n)[:, None] % 4) for _ in range(3))
    out = dict(img_rows=img_rows, txt_rows=txt_rows, tim=(txt, vp, gm), n_clusters=4, seed=123)
    with tempfile.TemporaryDirectory() as d:
        fi, ft, fm = os.path.join(d, "img.tsv"), os.path.join(d, "txt.tsv"), os.path.join(d, "tim.tsv")
        Z.write_img_tsv(fi, img_rows)
        Z.write_instr_tsv(ft, txt_rows)
        Z.write_tim_tsv(fm, txt, vp, gm)
        ref = DU.LoadZdict(fi, ft)
        a, b = ref.load_all_zdicts()
        out["read_img"] = [(r["roomtype"], torch.from_numpy(r["feature"].copy()), r["pz"]) for r in a]
        out["read_instr"] = [(r["token_type"], r["token"], torch.from_numpy(r["feature"].copy()), r["pz"]) for r in b]
        real_cuda = torch.Tensor.cuda
        torch.Tensor.cuda = lambda self, *a_, **k_: self           # the loaders hard-code .cuda() (data_utils.py:88-89,:113-118)
        try:
            out["img_tensor"] = ref.load_img_tensor()
            out["instr_tensor"] = ref.load_instr_tensor()
            np.random.seed(5)
            out["instr_tensor_random"] = ref.load_instr_tensor(is_random=True)
        finally:
            torch.Tensor.cuda = real_cuda
        np.random.seed(out["seed"])
        picker = UD.KMeansPicker(fm, n_clusters=out["n_clusters"])
        out["tim_read"] = tuple(torch.from_numpy(x.copy()) for x in picker.read_tim_tsv(fm))
        out["labels"] = {k: torch.from_numpy(picker.kmeans_model_dict[k].labels_.astype(np.int64)) for k in picker.feat_dicts}
        picked = picker.random_pick_front_features()
        out["picked"] = {k: torch.from_numpy(np.array(v)) for k, v in picked.items()}
    torch.save(out, os.path.join(HERE, "zdict.pt"))
    print("zdict:", {k: tuple(v.shape) for k, v in out["picked"].items()})


def mint_ops():
    O = load_by_path("ref_ops", f"{REF}/map_nav_src/utils/ops.py")
    g = torch.Generator().manual_seed(2)
    ts = [torch.randn(n, 3, generator=g) for n in (2, 5, 1)]
    lens = torch.tensor([2, 5, 1, 0])
    torch.save(dict(ts=ts, padded=O.pad_tensors(ts), lens=lens, masks=O.gen_seq_masks(lens),
                    masks8=O.gen_seq_masks(lens, max_len=8)), os.path.join(HERE, "ops.pt"))
    print("ops ok")


if __name__ == "__main__":
    import tempfile
    os.chdir(tempfile.mkdtemp(prefix="mint_golden_"))      # nothing the reference's modules create lands in the repo
    if "--zdict-only" in sys.argv:
        mint_zdict()
        sys.exit(0)
    if "--mrc-only" in sys.argv:
        mint_mrc()
        sys.exit(0)
    if "--ragged-only" in sys.argv:
        mint_ragged()
        sys.exit(0)
    if "--nav-loop-only" in sys.argv:
        mint_nav_loop()
        sys.exit(0)
    if "--agent-only" in sys.argv:
        mint_agent()
        sys.exit(0)
    if "--ingest-only" in sys.argv:
        mint_ingest()
        sys.exit(0)
    if "--pretrain-side-only" in sys.argv:
        mint_pretrain_side()
        sys.exit(0)
    if "--gmap-pos-only" in sys.argv:
        mint_gmap_pos()
        sys.exit(0)
    if "--gmap-inputs-only" in sys.argv:
        mint_gmap_inputs()
        sys.exit(0)
    mint_primitives()
    mint_ops()
    mint_zdict()
    mint_agent()
    mint_pretrain_side()
