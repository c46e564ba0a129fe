#!/bin/bash
# A/B of ONE library under settings of ONE environment knob, alternating on ONE GPU box, three rounds:
#   tools/r03_ab_env.sh FHEVC_HADAMARD_FORM valu mfma
cd "$GRAFT_REPO_ROOT"
VAR=$1; shift
for round in 1 2 3; do
  for v in "$@"; do
    env $VAR=$v timeout -k 10 200 python3 bench.py --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --no-host-path --no-stages --no-variants > /tmp/ab.log 2>/tmp/ab.err || { tail -5 /tmp/ab.err; exit 1; }
    python3 -c "
import json;d=json.loads(open('/tmp/ab.log').read().strip().splitlines()[-1]);print('$VAR=$v round $round: ms/step %.4f  cnn ms %.4f  CTU/s %.4g' % (d['ms_per_step'],d['roofline']['avg_launch_ms'], d['value']))"
  done
done
