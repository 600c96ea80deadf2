#!/usr/bin/env python3
"""Extracts the 819 real BRISK2 node descriptors (48 bytes each) of the reference's DBoW2
vocabulary resources/small_voc.yml.gz (k=9, L=3) into tests/golden/small_voc_desc.bin.

Run in the build container only (needs /root/reference); the output is data, not source.
"""
import gzip
import os
import re
import sys

import numpy as np

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/resources/small_voc.yml.gz"
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                   "small_voc_desc.bin")
text = gzip.open(src, "rt").read()
rows = [np.array(m.split(), dtype=np.uint8) for m in re.findall(r'descriptor:"([0-9 ]+)"', text)]
rows = [r for r in rows if len(r) == 48]
arr = np.stack(rows)
print(arr.shape)
arr.tofile(dst)
