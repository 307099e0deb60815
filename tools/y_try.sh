#!/bin/bash
set -eo pipefail
O=gpurun_out/yt; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_ebe.py -x -q > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
for r in 1 2; do
  timeout -k 10 300 python3 bench.py --steps 5 --warmup 1 --no-cpu --no-mg > $O/bx.json 2> $O/bx.err
  python3 -c "import json;d=json.load(open('$O/bx.json'));print('brick', round(d['value']/1e6,2), d['config']['ms_per_cg_iteration'], d['roofline']['avg_launch_ms'])"
done
TAG="sheared" timeout -k 10 120 python3 tools/ebe_apply_time.py 149 40 sheared 2>&1 | tail -1
