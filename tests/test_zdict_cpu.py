"""Data side of the causal-intervention inputs against fixtures minted from the reference's own loaders
(tests/golden/mint_golden.py::mint_zdict -> zdict.pt): `LoadZdict` (map_nav_src/r2r/data_utils.py:45-120) and `KMeansPicker`
(map_nav_src/utils/data.py:436-513) run on synthetic TSV rows; host/zdict.py must parse and pick bit-identically."""
import os

import numpy as np
import pytest
import torch

import magic_amd  # noqa: F401
from magic_amd.host import zdict as Z

G = torch.load(os.path.join(os.path.dirname(__file__), "golden", "zdict.pt"), weights_only=False)


@pytest.fixture()
def files(tmp_path):
    fi, ft, fm = str(tmp_path / "img.tsv"), str(tmp_path / "txt.tsv"), str(tmp_path / "tim.tsv")
    Z.write_img_tsv(fi, G["img_rows"])
    Z.write_instr_tsv(ft, G["txt_rows"])
    Z.write_tim_tsv(fm, *G["tim"])
    return fi, ft, fm


def test_zdict_loaders_match_reference(files):
    fi, ft, _ = files
    zd = Z.ZDict(fi, ft, device="cpu")
    img, ins = zd.load_all_zdicts()
    assert len(img) == len(G["read_img"]) and len(ins) == len(G["read_instr"])
    for r, (name, feat, pz) in zip(img, G["read_img"]):
        assert r["roomtype"] == name and r["pz"] == pz and np.array_equal(r["feature"], feat.numpy())
    for r, (tt, tok, feat, pz) in zip(ins, G["read_instr"]):
        assert (r["token_type"], r["token"], r["pz"]) == (tt, tok, pz) and np.array_equal(r["feature"], feat.numpy())
    got = zd.load_img_tensor()
    for k, v in G["img_tensor"].items():
        assert got[k].dtype == v.dtype and torch.equal(got[k], v), k
    got = zd.load_instr_tensor()
    for k, v in G["instr_tensor"].items():
        assert got[k].dtype == v.dtype and torch.equal(got[k], v), k
    np.random.seed(5)
    got = zd.load_instr_tensor(is_random=True)
    for k, v in G["instr_tensor_random"].items():
        assert torch.equal(got[k], v), k
    # direction / landmark split and the priors
    n_dir = sum(1 for r in G["txt_rows"] if r[0] == "direction")
    assert got["instr_direction_features"].shape[0] == n_dir and got["instr_landmark_features"].shape[0] == len(G["txt_rows"]) - n_dir
    assert abs(float(zd.load_img_tensor()["img_pzs"].sum()) - 1.0) < 1e-9


def test_front_door_picker_matches_reference(files):
    pytest.importorskip("sklearn")
    fm = files[2]
    np.random.seed(G["seed"])
    pk = Z.FrontDoorPicker(fm, n_clusters=G["n_clusters"])
    for a, b in zip(pk.read_tim_tsv(fm), G["tim_read"]):
        assert np.array_equal(a, b.numpy())
    for k, lab in G["labels"].items():
        assert np.array_equal(pk.kmeans_model_dict[k].labels_, lab.numpy()), k
    picked = pk.random_pick_front_features()
    for k, v in G["picked"].items():
        assert np.array_equal(np.array(picked[k]), v.numpy()), k
    t, v, g = pk.device_tensors(3, device="cpu", picked=picked)
    assert t.shape == (3, G["n_clusters"], 6) and torch.equal(t[0], t[2]) and torch.equal(v[1], G["picked"]["vp_feats"])
