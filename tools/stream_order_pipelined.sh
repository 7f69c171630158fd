#!/bin/bash
# the asynchronous boundary (two genomes in flight) and the synchronous call under AVK_STREAM_ORDER candidates, fresh processes: tools/stream_order_pipelined.sh order...
for o in "$@"; do
  echo -n "order $o: pipelined "
  for rep in 1 2; do AVK_STREAM_ORDER=$o python tools/boundary_pipelined.py 1.0 24 2>&1 | grep "genomes back" | sed "s/.*: \([0-9.]*\) ms per genome/\1/" | tr "\n" " "; done
  echo -n " call "
  AVK_STREAM_ORDER=$o python tools/gpu_boundary_time.py 1 30 2>&1 | grep regions | sed "s/.*median \([0-9.]*\) .*/\1/" | tr "\n" " "
  echo -n " resident "
  AVK_STREAM_ORDER=$o python tools/gpu_enqueue_time.py 1 2>&1 | grep regions | tail -1 | sed "s/.*(\([0-9.]*\) ms per step)$/\1/"
done
