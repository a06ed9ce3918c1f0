#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r4_run6; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $R
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" | tee -a $out/pytest.log; tail -6 $out/pytest.log
grep -q "pytest rc 0" $out/pytest.log || exit 1
timeout -k 10 600 python3 bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc $?"; python3 -c "
import json; d=json.load(open('$out/bench.json'))
print('value', d['value'], 'ref_order', d['reference_order'], 'cpp_api', d['cpp_api'])
print({k: d['roofline'].get(k) for k in ('frac','frac_rocprof','frac_measured','frac_timed','frac_timed_measured')})
"
