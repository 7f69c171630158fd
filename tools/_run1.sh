for o in "lane_head_stream=0" "lane_head_stream=0,order_guard=1" "lane_head_stream=1,order_guard=1" "lane_head_stream=0,order_guard=1,lane_head_width=32"; do
AVK_OPTS=$o timeout 120 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-parity --no-secondary --boundary-calls 0 --watchdog-seconds 60 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('[$o]', round(d['ms_per_step'],3))"
done
