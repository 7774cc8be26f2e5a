B="--steps 20 --warmup 5 --no-cpu-baseline --rollout-k 0 --image-envs 0 --numpy-api 0 --no-configs --steady-seconds 0.3 --kernel-timing-launches 1"
run() { echo "== $1"; env $1 python bench.py $B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.4g ms/step %.5f block_ms med %.4f min %.4f loop %.5f' % (j['value'], j['ms_per_step'], j['block_ms']['median'], j['block_ms']['min'], j['loop_issue']['ms_per_step']))"; }
run "X=1"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"
run "HSA_ENABLE_INTERRUPT=0"
run "AMD_DIRECT_DISPATCH=0"
run "GPU_MAX_HW_QUEUES=1"
run "X=2"
