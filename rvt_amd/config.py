"""Config objects for the backbone.  The reference composes them with Hydra/OmegaConf
(config/model/maxvit_yolox/default.yaml + config/experiment/<ds>/<size>.yaml + config/modifier.py);
neither package exists in this image, so this module provides
  * ``AttrDict`` — attribute + ``.get`` access, the only DictConfig behaviour the backbone relies on
    (an actual omegaconf DictConfig works just as well);
  * ``backbone_config(size, dataset)`` — the same keys/values as the reference YAML tree;
  * ``modify_backbone_config`` — the runtime derivation of ``in_res_hw`` and ``partition_size``
    (reference config/modifier.py:28-41).
"""
from __future__ import annotations

import math
from typing import Tuple


class AttrDict(dict):
    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        for k, v in list(self.items()):
            self[k] = _wrap(v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = _wrap(v)


def _wrap(v):
    if isinstance(v, dict) and not isinstance(v, AttrDict):
        return AttrDict(v)
    return v


# dataset resolutions after the optional 2x down-sampling (reference data/utils/spatial.py; config/dataset/*.yaml)
DATASET_HW = {'gen1': (240, 304), 'gen4': (360, 640)}
PARTITION_SPLIT_32 = {'gen1': 1, 'gen4': 2}          # config/experiment/gen1/default.yaml:42, default.yaml:14
EMBED = {'tiny': (32, 32), 'small': (48, 24), 'base': (64, 32)}   # (embed_dim, dim_head): config/experiment/*/{tiny,small,base}.yaml


def backbone_config(size: str = 'base', dataset: str = 'gen4') -> AttrDict:
    embed_dim, dim_head = EMBED[size]
    cfg = AttrDict({
        'name': 'MaxViTRNN',
        'compile': {'enable': False, 'args': {'mode': 'reduce-overhead'}},
        'input_channels': 20,
        'enable_masking': False,
        'partition_split_32': PARTITION_SPLIT_32[dataset],
        'embed_dim': embed_dim,
        'dim_multiplier': [1, 2, 4, 8],
        'num_blocks': [1, 1, 1, 1],
        'T_max_chrono_init': [4, 8, 16, 32],
        'stem': {'patch_size': 4},
        'stage': {
            'downsample': {'type': 'patch', 'overlap': True, 'norm_affine': True},
            'attention': {'use_torch_mha': False, 'partition_size': None, 'dim_head': dim_head,
                          'attention_bias': True, 'mlp_activation': 'gelu', 'mlp_gated': False, 'mlp_bias': True,
                          'mlp_ratio': 4, 'drop_mlp': 0, 'drop_path': 0, 'ls_init_value': 1e-5},
            'lstm': {'dws_conv': False, 'dws_conv_only_hidden': True, 'dws_conv_kernel_size': 3,
                     'drop_cell_update': 0},
        },
    })
    return modify_backbone_config(cfg, DATASET_HW[dataset])


def modify_backbone_config(cfg, dataset_hw: Tuple[int, int]):
    """in_res_hw = dataset (h,w) rounded up to a multiple of 32*partition_split_32; partition_size = in_res/(32*split)."""
    split = cfg.partition_split_32
    assert split in (1, 2, 4)
    mult = 32 * split
    mdl_hw = tuple(math.ceil(x / mult) * mult for x in dataset_hw)
    cfg.in_res_hw = mdl_hw
    part = tuple(x // mult for x in mdl_hw)
    assert (mdl_hw[0] // 32) % part[0] == 0 and (mdl_hw[1] // 32) % part[1] == 0
    cfg.stage.attention.partition_size = part
    return cfg
