"""Model configuration for the MAGIC cross-modal transformer.

Mirrors the attribute names of the reference's HF ``PretrainedConfig`` JSON
(/root/reference/pretrain_src/config/r2r_magic_model_config.json) and the teacher/student
attribute surgery of pretrain_src/train_r2r_magic.py:125-160, so that a config object built by
the reference driver can be handed to our model unchanged (any object with these attributes works).
"""
import copy
import json
from types import SimpleNamespace

# size family: SURVEY A.3 (run_r2r_kdl_valid.sh:85-94)
FAMILY = {"S": 128, "M": 256, "B": 384, "L": 768}

_DEFAULTS = dict(
    hidden_size=768, intermediate_size=3072, num_attention_heads=12,
    num_l_layers=6, num_x_layers=3, num_pano_layers=2,
    layer_norm_eps=1e-12, max_position_embeddings=514, max_action_steps=100,
    type_vocab_size=1, vocab_size=50265, hidden_act="gelu",
    hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, pred_head_dropout_prob=0.1,
    initializer_range=0.02,
    use_lang2visn_attn=True, graph_sprels=True, glocal_fuse=True,
    image_feat_size=768, image_prob_size=1000, angle_feat_size=4,
    adaptive_pano_fusion=True, cfp_temperature=1.0,
    role="student", kd=False, teacher_hidden_size=None,
)


def make_config(hidden_size=128, mlp_ratio=4, role="student", teacher_hidden_size=None, **over):
    """Build a config for one model of the S/M/B/L family: heads = H/64, FFN = H*mlp_ratio
    (train_r2r_magic.py:142-143,156-157)."""
    d = dict(_DEFAULTS)
    d.update(hidden_size=hidden_size, intermediate_size=int(hidden_size * mlp_ratio),
             num_attention_heads=int(hidden_size // 64), role=role,
             teacher_hidden_size=teacher_hidden_size, kd=teacher_hidden_size is not None)
    d.update(over)
    return SimpleNamespace(**d)


def teacher_student_from_json(path, kdl=None):
    """Reproduce train_r2r_magic.py:103-160: returns (teacher_cfg, student_cfg) from the reference's
    model-config JSON. ``kdl`` is the ``kdl`` block of the pretrain JSON (r2r_magic_pretrain.json:62-87)."""
    with open(path) as f:
        base = json.load(f)
    full = dict(_DEFAULTS)
    full.update(base)
    kd = bool(kdl and kdl.get("knowledge_distillation", False))
    teacher = dict(full)
    for k, v in full.items():
        if k.startswith("teacher_"):
            teacher[k[8:]] = v
    teacher["intermediate_size"] = int(teacher["hidden_size"] * teacher.get("mlp_ratio", 4))
    teacher["num_attention_heads"] = int(teacher["hidden_size"] // 64)
    teacher.update(role="teacher", kd=kd)
    student = dict(full)
    for k, v in full.items():
        if k.startswith("student_"):
            if kd:
                student["teacher_" + k[8:]] = teacher[k[8:]]
            student[k[8:]] = v
    student["intermediate_size"] = int(student["hidden_size"] * student.get("mlp_ratio", 4))
    student["num_attention_heads"] = int(student["hidden_size"] // 64)
    student.update(role="student", kd=kd, kdl=copy.deepcopy(kdl))
    return SimpleNamespace(**teacher), SimpleNamespace(**student)


def cfg_get(cfg, name, default=None):
    v = getattr(cfg, name, None)
    if v is None:
        return _DEFAULTS.get(name, default) if default is None else default
    return v
