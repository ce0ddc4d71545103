"""Data side of the causal-intervention inputs: the back-door z-dictionaries and the front-door feature
dictionary, in the reference's on-disk formats, delivered as DEVICE tensors in the shapes the model's blocks consume
(host/causal.py).  Pinned bit-exactly to the reference's own loaders (tests/golden/zdict.pt, tests/test_zdict_cpu.py).

  ZDict             map_nav_src/r2r/data_utils.py:45-120 `LoadZdict`: TSV rows, features as base64 of raw little-endian fp32,
                    `pz` = the prior P(z) of the entry.  Image dictionary fields (roomtype, feature, pz); instruction dictionary
                    fields (token_type in {direction, landmark}, token, feature, pz).
  FrontDoorPicker   map_nav_src/utils/data.py:436-513 `KMeansPicker`: TSV of (path_id, txt_feats, vp_feats, gmap_feats) rows =
                    the `extract_cfp_features` vectors of the training paths; K-means per feature kind (scikit-learn, the
                    reference's own dependency), then ONE random member per cluster, clusters in label order, drawn with
                    numpy's global generator exactly as the reference does (so a seeded run reproduces its pick).
"""
import base64
import csv
from collections import defaultdict

import numpy as np
import torch

IMG_FIELDS = ["roomtype", "feature", "pz"]                       # data_utils.py:47
TXT_FIELDS = ["token_type", "token", "feature", "pz"]            # data_utils.py:48
TIM_FIELDS = ["path_id", "txt_feats", "vp_feats", "gmap_feats"]  # utils/data.py:438


def _b64(a):
    return str(base64.b64encode(np.ascontiguousarray(a, dtype=np.float32)), "utf-8")


def _f32(s):
    return np.frombuffer(base64.b64decode(s), dtype=np.float32)


class ZDict:
    def __init__(self, img_zdict_file=None, txt_zdict_file=None, device="cuda"):
        self.img_zdict_file, self.txt_zdict_file, self.device = img_zdict_file, txt_zdict_file, device

    def _rows(self, path, fields):
        with open(path, "rt") as f:
            for item in csv.DictReader(f, delimiter="\t", fieldnames=fields):
                item["feature"] = _f32(item["feature"])
                item["pz"] = float(item["pz"])
                yield item

    def read_img_tsv(self):
        return list(self._rows(self.img_zdict_file, IMG_FIELDS))

    def read_instr_tsv(self):
        return list(self._rows(self.txt_zdict_file, TXT_FIELDS))

    def load_all_zdicts(self):
        return self.read_img_tsv(), self.read_instr_tsv()

    def load_img_tensor(self):
        """{'img_features' [Nz, D] fp32, 'img_pzs' [Nz] fp64} on the device (data_utils.py:77-90)"""
        rows = self.read_img_tsv()
        return {"img_features": torch.from_numpy(np.array([r["feature"] for r in rows])).to(self.device),
                "img_pzs": torch.from_numpy(np.array([r["pz"] for r in rows])).to(self.device)}

    def load_instr_tensor(self, is_random=False):
        """direction and landmark entries as separate dictionaries (data_utils.py:92-120); is_random replaces every feature by
        np.random.random of its shape (the reference's ablation switch), drawn in file order from numpy's global generator"""
        out = {"direction": ([], []), "landmark": ([], [])}
        for r in self._rows(self.txt_zdict_file, TXT_FIELDS):
            if r["token_type"] not in out:
                continue
            ft = np.random.random(r["feature"].shape).astype(np.float32) if is_random else r["feature"]
            out[r["token_type"]][0].append(ft)
            out[r["token_type"]][1].append(r["pz"])
        t = lambda a: torch.from_numpy(np.array(a)).to(self.device)
        lm = {"instr_landmark_features": t(out["landmark"][0]), "instr_landmark_pzs": t(out["landmark"][1])}
        if not out["direction"][0]:          # a dictionary without direction entries (REVERIE): landmark keys only (data_utils.py:117-120)
            return lm
        return {"instr_direction_features": t(out["direction"][0]), "instr_direction_pzs": t(out["direction"][1]), **lm}


def write_img_tsv(path, rows):
    """rows: iterable of (roomtype, feature fp32[D], pz)"""
    with open(path, "wt") as f:
        w = csv.DictWriter(f, delimiter="\t", fieldnames=IMG_FIELDS)
        for name, ft, pz in rows:
            w.writerow({"roomtype": name, "feature": _b64(ft), "pz": pz})


def write_instr_tsv(path, rows):
    """rows: iterable of (token_type, token, feature fp32[D], pz) -- the format `update_z_dict(save_file=True)` writes (agent.py:1316-1338)"""
    with open(path, "wt") as f:
        w = csv.DictWriter(f, delimiter="\t", fieldnames=TXT_FIELDS)
        for tt, tok, ft, pz in rows:
            w.writerow({"token_type": tt, "token": tok, "feature": _b64(ft), "pz": pz})


def write_tim_tsv(path, txt, vp, gmap):
    with open(path, "wt") as f:
        w = csv.DictWriter(f, delimiter="\t", fieldnames=TIM_FIELDS)
        for i in range(len(txt)):
            w.writerow({"path_id": i, "txt_feats": _b64(txt[i]), "vp_feats": _b64(vp[i]), "gmap_feats": _b64(gmap[i])})


class FrontDoorPicker:
    KINDS = ("txt_feats", "vp_feats", "gmap_feats")

    def __init__(self, front_feat_file, n_clusters=24, kmeans_models=None):
        """front_feat_file: TSV written by the `extract_cfp_features` pass (agent.py:1535-1541); n_clusters: front_n_clusters
        (parser.py:142, 24 in the shipped scripts); kmeans_models: optional dict kind -> fitted model (the reference's joblib path)"""
        from sklearn.cluster import KMeans          # the reference's own dependency for this step (utils/data.py:24)
        self.n_clusters = n_clusters
        txt, vp, gm = self.read_tim_tsv(front_feat_file)
        self.feat_dicts = {"txt_feats": txt, "vp_feats": vp, "gmap_feats": gm}
        self.kmeans_model_dict = dict(kmeans_models or {})
        for k in self.KINDS:
            if k not in self.kmeans_model_dict:
                km = KMeans(n_clusters=n_clusters)
                km.fit(self.feat_dicts[k])
                self.kmeans_model_dict[k] = km

    @staticmethod
    def read_tim_tsv(path):
        cols = ([], [], [])
        with open(path, "rt") as f:
            for item in csv.DictReader(f, delimiter="\t", fieldnames=TIM_FIELDS):
                for c, k in zip(cols, FrontDoorPicker.KINDS):
                    c.append(_f32(item[k]))
        return tuple(np.array(c) for c in cols)

    def random_pick_front_features(self):
        """kind -> list of n_clusters vectors: one uniformly drawn member of every cluster, clusters in ascending label order"""
        out = defaultdict(list)
        for k in self.KINDS:
            labels = self.kmeans_model_dict[k].labels_
            for lab in np.unique(labels):
                idx = np.where(labels == lab)[0]
                out[k].append(self.feat_dicts[k][np.random.choice(idx)])
        return out

    def device_tensors(self, batch_size, device="cuda", picked=None):
        """the three front-door inputs of one rollout, each [B, n_clusters, D] -- the dictionary repeated over the batch as the
        agent does (agent.py:1212-1227)"""
        picked = picked or self.random_pick_front_features()
        rep = lambda a: torch.from_numpy(np.array(a)).to(device).unsqueeze(0).expand(batch_size, -1, -1).contiguous()
        return rep(picked["txt_feats"]), rep(picked["vp_feats"]), rep(picked["gmap_feats"])
