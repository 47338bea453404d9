#!/bin/bash
# stage-set probe for several builds: tools/experiments/r5_stagesets.sh LIB_A [LIB_B ...]
cd $GRAFT_REPO_ROOT
for L in "$@"; do echo "== $L"; VISLAM_HIP_LIB=$GRAFT_REPO_ROOT/$L timeout -k 10 300 python tools/stage_sets_probe.py 2>/dev/null | cut -c1-250; done
