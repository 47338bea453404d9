#!/bin/bash
# headline, headline + results download, S-752P for several builds, alternating: tools/experiments/r5_d2h_ab.sh LIB_A LIB_B ...
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for L in "$@"; do
  export VISLAM_HIP_LIB=$GRAFT_REPO_ROOT/$L
  a=$(timeout -k 10 200 python tools/step_time.py 15 60 2>/dev/null | tail -1 | sed 's/.*step, //; s/ frames.*//')
  b=$(timeout -k 10 200 python tools/step_time.py 15 60 d2h 2>/dev/null | tail -1 | sed 's/.*step, //; s/ frames.*//')
  echo "$(basename $L) headline $a | + results download $b"
done; done
