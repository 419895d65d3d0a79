"""Enums / aliases of the reference's batch contract (data/utils/types.py:13-55), restated."""
from enum import Enum, auto
from typing import Dict, List, Optional, Tuple

import torch


class DataType(Enum):
    EV_REPR = auto()
    FLOW = auto()
    IMAGE = auto()
    OBJLABELS = auto()
    OBJLABELS_SEQ = auto()
    IS_PADDED_MASK = auto()
    IS_FIRST_SAMPLE = auto()
    TOKEN_MASK = auto()


class DatasetSamplingMode(str, Enum):
    RANDOM = 'random'
    STREAM = 'stream'
    MIXED = 'mixed'


class Mode(Enum):
    TRAIN = auto()
    VAL = auto()
    TEST = auto()


LstmState = Optional[Tuple[torch.Tensor, torch.Tensor]]
LstmStates = List[LstmState]
FeatureMap = torch.Tensor
BackboneFeatures = Dict[int, torch.Tensor]
