#!/bin/bash
# shader clock and package power while the headline pipeline runs (rocm-smi sampled beside bench.py): tools/clocks_under_load.sh [STAGE_MASK ...]
# (stage masks as bench.py --stages: 1 detect, 3 + matcher, 7 + pose, 15 + Camera::Update = the headline; default 15)
cd $GRAFT_REPO_ROOT
echo "== idle"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -4
for S in ${@:-15}; do
timeout -k 10 200 python bench.py --no-legs --no-cpu-baseline --steps 300 --warmup 5 --stages $S > gpurun_out/clocks_bench.json 2>/dev/null &
BP=$!
sleep 12
for i in 1 2 3 4; do echo "== stages $S under load, sample $i"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | head -3; sleep 1.5; done
wait $BP
python -c "
import json
j=json.loads([l for l in open('gpurun_out/clocks_bench.json') if l.startswith('{')][-1]); print('stages $S bench', round(j['value']), 'frames/s')"
done
