cd $GRAFT_REPO_ROOT
grep -h "fw_version\|sdma_fw_version\|gfx_target_version\|unique_id" /sys/class/kfd/kfd/topology/nodes/*/properties 2>/dev/null | sort | uniq -c | head -8
cat /sys/module/amdgpu/version 2>/dev/null
for i in 1 2 3; do for w in new old; do if [ $w = old ]; then B=tools/ab/old_tree/bench.py; else B=bench.py; fi; python $B --no-cpu-baseline --no-other-dtype --no-roofline --steps 600 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$w', round(d['ms_per_step'],4))"; done; done
