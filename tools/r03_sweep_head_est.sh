cd /root/repo
[ -f /tmp/w100.bin ] || python tools/dump_workload.py 1.0 /tmp/w100.bin > /dev/null
run() { printf "%-52s " "$1"; timeout 120 .scratch/first_step_probe /tmp/w100.bin 50 1 25 "$1" 2>&1 | tail -1 | sed 's/.*finished \([0-9.]*\) ms later.*/\1 ms for 50 steps/'; }
for rep in 1 2; do for v in 1 2 3 4 6 8; do run "lane_head_est=$v"; done; done
