"""Parse the generated constant tables (oracle/mobi_tables.h) into numpy arrays for table-identity tests."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path=None):
    src = open(path or os.path.join(ROOT, "oracle", "mobi_tables.h")).read()
    out = {}
    for m in re.finditer(r"static const (uint8_t|uint16_t) (\w+)((?:\[\d+\])+) = \{(.*?)\};", src, flags=re.S):
        dims = [int(x) for x in re.findall(r"\[(\d+)\]", m.group(3))]
        nums = [int(x, 0) for x in re.findall(r"0x[0-9A-Fa-f]+|\d+", m.group(4))]
        out[m.group(2)] = np.array(nums, dtype=np.int64).reshape(dims)
    return out
