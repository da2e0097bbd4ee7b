"""DLC project / model config layout (kept so existing projects drop in).

  read_config, GetModelFolder, GetTrainingSetFolder   DLC/utils/auxiliaryfunctions.py:139-157,304-315
  default_config, load_config (cfg_from_file)         PET/default_config.py:16-59, PET/config.py:39-55
  get_train_config                                    DGP/utils_model.py:88-110

Unlike the reference, load_config returns a FRESH attribute-dict per call instead of mutating one
process-global EasyDict (PET/config.py:14) -- DGP hyper-parameters therefore cannot leak between steps.
"""
from __future__ import annotations

import copy
import os
from pathlib import Path

import yaml


class AttrDict(dict):
    """dict with attribute access (stands in for easydict.EasyDict)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return AttrDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


# PET/default_config.py:16-59
DEFAULT_CONFIG = dict(
    stride=8.0, weigh_part_predictions=False, weigh_negatives=False, fg_fraction=0.25,
    weigh_only_present_joints=False, mean_pixel=[123.68, 116.779, 103.939], shuffle=True,
    snapshot_prefix="./snapshot", log_dir="log", global_scale=1.0, location_refinement=False,
    locref_stdev=7.2801, locref_loss_weight=1.0, locref_huber_loss=True, optimizer="sgd",
    intermediate_supervision=False, intermediate_supervision_layer=12, regularize=False, weight_decay=0.0001,
    mirror=False, crop_pad=0, scoremap_dir="test", batch_size=1, dataset_type="default", deterministic=False,
    crop=False, cropratio=0.25, minsize=100, leftwidth=400, rightwidth=400, topheight=400, bottomheight=400,
)


def read_config(configname):
    """Project config.yaml -> dict.  Missing file raises FileNotFoundError like the reference."""
    path = Path(configname)
    if not os.path.exists(path):
        raise FileNotFoundError("Config file is not found. Please make sure that the file exists and/or there are "
                                "no unnecessary spaces in the path of the config file!")
    with open(path, "r") as f:
        return yaml.load(f, Loader=yaml.SafeLoader)


def GetModelFolder(trainFraction, shuffle, cfg) -> Path:
    return Path("dlc-models/iteration-" + str(cfg["iteration"]) + "/" + cfg["Task"] + cfg["date"] + "-trainset" +
                str(int(trainFraction * 100)) + "shuffle" + str(shuffle))


def GetTrainingSetFolder(cfg) -> Path:
    return Path(os.path.join("training-datasets", "iteration-" + str(cfg["iteration"]),
                             "UnaugmentedDataSet_" + cfg["Task"] + cfg["date"]))


def _merge(a: dict, b: dict):
    for k, v in a.items():
        if isinstance(v, dict) and isinstance(b.get(k), dict):
            _merge(v, b[k])
        else:
            b[k] = v


def load_config(filename="pose_cfg.yaml") -> AttrDict:
    """pose_cfg.yaml merged over the DLC defaults; snapshot_prefix = <train dir>/snapshot (PET/config.py:39-51)."""
    with open(filename, "r") as f:
        y = yaml.load(f, Loader=yaml.SafeLoader) or {}
    cfg = AttrDict(copy.deepcopy(DEFAULT_CONFIG))
    y["snapshot_prefix"] = str(filename).split("pose_cfg.yaml")[0] + "snapshot"
    _merge(y, cfg)
    return cfg


def get_train_config(cfg, shuffle=0) -> AttrDict:
    """Resolve <project>/dlc-models/iteration-i/<Task><date>-trainset<pct>shuffle<k>/train/pose_cfg.yaml
    (DGP/utils_model.py:88-110).  Note: like the reference, TrainingFraction is indexed by `iteration`."""
    project_path = cfg["project_path"]
    train_fraction = cfg["TrainingFraction"][cfg["iteration"]]
    modelfolder = os.path.join(project_path, str(GetModelFolder(train_fraction, shuffle, cfg)))
    path_train_config = Path(modelfolder) / "train" / "pose_cfg.yaml"
    try:
        dlc_cfg = load_config(str(path_train_config))
    except FileNotFoundError:
        raise FileNotFoundError("It seems the model for shuffle %s and trainFraction %s does not exist."
                                % (shuffle, train_fraction))
    dlc_cfg.video_path = cfg.get("video_path")
    dlc_cfg.project_path = cfg["project_path"]
    return dlc_cfg


def skeleton_matrix(cfg):
    """S0 [n_limbs, nj] in {+1,-1,0} from the project's `skeleton` pairs (DGP/models/fitdgp.py:607-617)."""
    import numpy as np
    parts = list(cfg["bodyparts"])
    limbs = cfg.get("skeleton") or []
    S0 = np.zeros((len(limbs), len(parts)))
    for i, (a, b) in enumerate(limbs):
        S0[i, parts.index(a)] = 1
        S0[i, parts.index(b)] = -1
    return S0
