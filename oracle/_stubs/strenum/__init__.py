"""Stand-in for the `strenum` package (Python 3.10 has no enum.StrEnum).

TEST INFRASTRUCTURE ONLY — lets /root/reference/data/utils/types.py:3-6 import.
"""
from enum import Enum


class StrEnum(str, Enum):
    def __str__(self):
        return str(self.value)
