#!/bin/bash
# GPU box: A/B of OKVFE_DESC_BLOCKS (workgroups per image of describe_kernel); rebuilds k_describe.hip per value
R=$PWD
for nb in "$@"; do
  (cd $R/okvis2_amd/csrc && touch k_describe.hip && make -j8 EXTRA=-DOKVFE_DESC_BLOCKS=$nb > /dev/null 2>&1)
  echo "== OKVFE_DESC_BLOCKS=$nb"
  bash $R/tools/kstats.sh 2>&1 | grep -E "describe_kernel|^fps"
done
(cd $R/okvis2_amd/csrc && touch k_describe.hip && make -j8 > /dev/null 2>&1)
