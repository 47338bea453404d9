#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2; do for L in "$@"; do VISLAM_HIP_LIB=$GRAFT_REPO_ROOT/$L timeout -k 10 200 python tools/single_frame_probe.py 200 2>/dev/null | tail -1; done; done
cat > /tmp/cal.xml <<'XML'
XML
python - <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import re
src = open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "bench.py")).read()
xml = src.split('CAL_XML = """', 1)[1].split('"""', 1)[0]
open("/tmp/cal.xml", "w").write(xml)
PY
timeout -k 10 200 vi-slam_amd/lib/addframe_bench /tmp/cal.xml 200 6 | grep "^{"
