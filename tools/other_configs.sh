#!/bin/bash
# GPU box: the BASELINE configs other than the bench default, one lean bench.py line each -> "args Mrays/s ms_per_iteration" (profiles/r3_other_configs.txt)
cd $GRAFT_REPO_ROOT
while read -r args; do
  python3 bench.py --lean 1 --cpu-seconds 0 --bvh-scene "" --steps 20 --warmup 5 $args 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d['value'], d['ms_per_iteration'])" "$args"
done <<'CFG'
--scene mis_test.txt --mode mis --res 1024 1024 --depth 16
--scene metallic_analytic.txt --mode mis --res 1024 1024 --depth 16
--scene obj_test.txt --mode mis
--scene bunny_cornell.txt
--scene bunny_cornell.txt --mode mis
--scene bunny_cornell.txt --res 2048 2048 --depth 16
--scene cornell.txt --res 400 400 --depth 4
--scene cornell.txt --mode mis
--scene cornell.txt --mode direct
--scene mis_test2.txt --mode direct
--scene textureEnv_test.txt --mode mis
CFG
