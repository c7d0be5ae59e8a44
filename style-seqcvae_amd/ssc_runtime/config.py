"""Config with the reference's keys, defaults, yaml merge, dotted CLI overrides, validation and freezing
(updown-baseline/updown/config.py:4-154), on PyYAML (yacs is not a dependency here)."""
import ast
import copy
from typing import Any, List, Optional

import yaml


class _Node(dict):
    """Attribute-access dict (the part of yacs.CfgNode the code base uses)."""

    def __init__(self, d=None):
        super().__init__()
        object.__setattr__(self, "_frozen", False)
        for k, v in (d or {}).items():
            self[k] = _Node(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        if object.__getattribute__(self, "_frozen"):
            raise AttributeError(f"config is frozen; cannot set {k}")
        self[k] = v

    def freeze(self):
        object.__setattr__(self, "_frozen", True)
        for v in self.values():
            if isinstance(v, _Node):
                v.freeze()

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, _Node) else v) for k, v in self.items()}


def _defaults() -> dict:
    # key set and default values: updown-baseline/updown/config.py:6-104
    return {
        "LOG_TO_FILE": True, "CHECKPOINT_EVERY_N_EPOCHS": 10, "PRINT_EVERY_N_BATCHES": 100, "RANDOM_SEED": 0,
        "DATA": {
            "VOCABULARY": "data/vocabulary",
            "TRAIN_FEATURES": "data/coco_train2017_vg_detector_features_adaptive.h5",
            "INFER_FEATURES": "data/nocaps_val_vg_detector_features_adaptive.h5",
            "TRAIN_CAPTIONS": "data/coco/captions_train2017.json",
            "INFER_CAPTIONS": "data/nocaps/nocaps_val_image_info.json",
            "SENTICAP_CAPTIONS": "", "DO_LOAD_COCO": True, "DO_LOAD_SENTICAP": False, "SENTICAP_SENTIMENT": "",
            "EXPERT_CAPTIONS": "", "COCO_ATTRIBS_OBJS": "", "REMOVE_SAMPLES_WITHOUT_ATTRIBS": False,
            "USE_OBJ_ATT_PREDS": False, "ATT_PRED_THRESH": 0.3, "MAX_CAPTION_LENGTH": 20,
            "CBS": {"INFER_BOXES": "data/nocaps_val_oi_detector_boxes.json", "CLASS_HIERARCHY": "data/cbs/class_hierarchy.json",
                    "WORDFORMS": "", "WORDFORMS_ATTRIBS": "", "NMS_THRESHOLD": 0.85, "MAX_GIVEN_OBJECTS": 2,
                    "MAX_GIVEN_CONSTRAINTS": 3, "MAX_WORDS_PER_CONSTRAINT": 3},
        },
        "MODEL": {
            "IMAGE_FEATURE_SIZE": 2048, "EMBEDDING_SIZE": 1000, "HIDDEN_SIZE": 1200, "ATTENTION_PROJECTION_SIZE": 768,
            "BEAM_SIZE": 5, "USE_CBS": False, "CBS_SIMPLE": True, "MIN_CONSTRAINTS_TO_SATISFY": 2,
            "PRIOR_MODE": "AG", "DO_USE_CLUSTER_VECTOR": True, "FC_LAYER_PER_ATTRIB": True, "NUM_LSTM_LAYERS": 1,
            "LSTM_DROPOUT": 0.1, "Z_SPACE": 150, "SENTIMENT_VAE": 0, "SENTI_PRIOR_MULTIP": 1.0,
            "LATENT_EMBEDDING_MULTIP": 1.0, "KLD_WEIGHT": 750, "N_Z_SAMPLES": 0, "STATE_MACHINE_PER_Z_SAMPLE": False,
            "LATENT_EMBEDDING": "glove", "PRIOR_STD": 1.0, "SIMPLE_VAE": True, "DO_USE_KLD_ANNEALING": False,
            "KLD_DECREASING": False, "KLD_INITIAL_WEIGHT": 2.0, "KLD_ANNEALING_PER_EPOCH": 0.25,
            "KLD_N_EPOCHS_BEFORE_RESET": 4,
        },
        "OPTIM": {
            "BATCH_SIZE": 150, "NUM_ITERATIONS": 70000, "LR": 0.015, "MOMENTUM": 0.9, "LR_DECAY_EVERY_N": 7,
            "LR_DECAY": 0.5, "LR_DECAY_START_EPOCH": 10, "WEIGHT_DECAY": 0.001, "CLIP_GRADIENTS": 12.5,
            "EPOCH_START_DECODER_TRAINING": 40000, "BEFORE_UPDATE_DECODER_EVERY": 30,
        },
    }


def _merge(dst: dict, src: dict, path=""):
    for k, v in src.items():
        if k not in dst:
            raise KeyError(f"Non-existent config key: {path}{k}")
        if isinstance(dst[k], dict):
            if not isinstance(v, dict):
                raise ValueError(f"{path}{k} must be a mapping")
            _merge(dst[k], v, path + k + ".")
        else:
            dst[k] = _coerce(v, dst[k], path + k)


def _coerce(value: Any, old: Any, key: str):
    if isinstance(value, str) and not isinstance(old, str):
        try:
            value = ast.literal_eval(value)
        except (ValueError, SyntaxError):
            pass
    if isinstance(old, bool) or old is None:
        return value
    if isinstance(old, float) and isinstance(value, int):
        return float(value)
    if isinstance(old, int) and isinstance(value, float) and not isinstance(old, bool):
        return value  # yacs allows int -> float replacement for numeric keys (KLD_WEIGHT: 750 vs 750.0)
    if type(value) is not type(old):
        raise ValueError(f"Type mismatch for config key {key}: {type(old).__name__} vs {type(value).__name__}")
    return value


class Config(object):
    def __init__(self, config_file: Optional[str] = None, config_override: List[Any] = []):
        tree = _defaults()
        if config_file is not None:
            with open(config_file) as f:
                _merge(tree, yaml.safe_load(f) or {})
        if len(config_override) % 2 != 0:
            raise ValueError("config_override must be a list of key value pairs")
        for k, v in zip(config_override[0::2], config_override[1::2]):
            node = tree
            parts = k.split(".")
            for p in parts[:-1]:
                node = node[p]
            if parts[-1] not in node:
                raise KeyError(f"Non-existent config key: {k}")
            node[parts[-1]] = _coerce(v, node[parts[-1]], k)
        self._C = _Node(tree)
        self._validate()
        self._C.freeze()

    def dump(self, file_path: str):
        with open(file_path, "w") as f:
            yaml.safe_dump(self._C.to_dict(), f, default_flow_style=False)

    def _validate(self):
        # updown-baseline/updown/config.py:129-140
        if self._C.MODEL.USE_CBS:
            assert self._C.MODEL.EMBEDDING_SIZE in (300, 600), (
                "Word embeddings must be initialized with fixed GloVe Embeddings (300/600 dim) for CBS decoding; "
                f"found MODEL.EMBEDDING_SIZE {self._C.MODEL.EMBEDDING_SIZE}")
        assert self._C.MODEL.MIN_CONSTRAINTS_TO_SATISFY <= self._C.DATA.CBS.MAX_GIVEN_CONSTRAINTS, \
            "Satisfying more constraints than maximum specified is not possible."

    def __getattr__(self, attr: str):
        return getattr(self.__dict__["_C"], attr)

    def __str__(self):
        d = self._C.to_dict()
        return "\n".join(yaml.safe_dump({k: d[k]}, default_flow_style=False) for k in ("RANDOM_SEED", "DATA", "MODEL", "OPTIM"))

    def __repr__(self):
        return repr(self._C.to_dict())
