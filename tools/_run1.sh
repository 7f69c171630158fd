for o in "" "lane_max_est=10" "lane_waves_per_cu=16" "lane_node_cap=96"; do
AVK_OPTS=$o timeout 120 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-parity --no-secondary --boundary-calls 0 --watchdog-seconds 60 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('[$o]', round(d['ms_per_step'],3), d['config']['workspace_tiers'], d['config']['lane_kernel_regions'])"
done
