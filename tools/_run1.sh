for o in "lane_max_est=14" "lane_max_est=12" "lane_max_est=10" "lane_max_est=14,lane_head_width=32"; do
AVK_OPTS=$o timeout 120 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-parity --no-secondary --boundary-calls 0 --watchdog-seconds 60 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('[$o]', round(d['ms_per_step'],3), d['config']['workspace_tiers'], d['config']['lane_kernel_regions'])"
done
AVK_OPTS=lane_max_est=14 bash tools/lane_timeline.sh tlg
python tools/show_timeline.py gpurun_out/prof_tlg/stats/tlg_results.db
