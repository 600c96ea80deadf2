#!/usr/bin/env python3
"""Extracts the TREE of the reference's DBoW2 vocabulary resources/small_voc.yml.gz (k = 9, L = 3:
819 nodes below the root, 729 words) into tests/golden/small_voc_tree.npz: per node (index =
nodeId, 0 = root) parent, weight, word id (-1 for inner nodes) and the 48-byte descriptor.

Run in the build container only (needs /root/reference); the output is data, not source."""
import gzip
import os
import re
import sys

import numpy as np

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/resources/small_voc.yml.gz"
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                   "small_voc_tree.npz")
text = gzip.open(src, "rt").read()
k = int(re.search(r"\bk: (\d+)", text).group(1))
L = int(re.search(r"\bL: (\d+)", text).group(1))
scoring = int(re.search(r"scoringType: (\d+)", text).group(1))
weighting = int(re.search(r"weightingType: (\d+)", text).group(1))
nodes = re.findall(r"nodeId:(\d+), parentId:(\d+), weight:([0-9.eE+-]+),\s*descriptor:\"([0-9 ]+)\"", text)
n = max(int(a) for a, _, _, _ in nodes) + 1
parent = np.full(n, -1, dtype=np.int32)
weight = np.zeros(n, dtype=np.float64)
desc = np.zeros((n, 48), dtype=np.uint8)
for nid, pid, w, d in nodes:
    nid = int(nid)
    parent[nid] = int(pid)
    weight[nid] = float(w)
    desc[nid] = np.array(d.split(), dtype=np.uint8)
word = np.full(n, -1, dtype=np.int32)
for wid, nid in re.findall(r"wordId:(\d+), nodeId:(\d+)", text):
    word[int(nid)] = int(wid)
print(k, L, n, (word >= 0).sum(), scoring, weighting)
np.savez_compressed(dst, k=k, L=L, scoring=scoring, weighting=weighting, parent=parent, weight=weight,
                    word=word, desc=desc)
