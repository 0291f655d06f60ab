#!/bin/bash
# Same-box comparison of the ROUND-1 tree (copied to _r1tree/ with its own library) and the current tree on the 13B launch-per-operator
# path: is the "13B regression" (460 tok/s in round 1, ~400 since) code or boxes?
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"])'
for i in 1 2; do
  (cd _r1tree && timeout 400 python bench.py --model 13B --steps 64 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$P" "r1-tree 13B")
  timeout 400 python bench.py --model 13B --steps 64 --no-cpu-baseline --no-tp 2>/dev/null | tail -1 | python -c "$P" "r4-tree 13B"
done
(cd _r1tree && timeout 400 python bench.py --steps 64 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$P" "r1-tree 7B-int4")
MI355_FUSED=0 timeout 400 python bench.py --steps 64 --no-cpu-baseline --no-tp 2>/dev/null | tail -1 | python -c "$P" "r4-tree 7B-int4-launch-path"
