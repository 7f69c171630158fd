cd /root/repo
[ -f /tmp/w100.bin ] || python tools/dump_workload.py 1.0 /tmp/w100.bin > /dev/null
run() { printf "%-52s " "$1"; for k in 1 2 3; do timeout 120 .scratch/first_step_probe /tmp/w100.bin 50 1 25 "$1" 2>&1 | tail -1 | sed 's/.*finished \([0-9.]*\) ms later.*/\1/' | tr '\n' ' '; done; echo; }
run "lane_kernel=1"
run "hbm_ed_cap=48"
run "hbm_ed_cap=128"
run "hbm_ed_cap=512"
run "lane_kernel=1"
