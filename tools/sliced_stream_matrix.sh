#!/bin/bash
# which streams should carry the sliced mode's collectives?  in-process replicas on one GPU, tools/sliced_costs.py
for cs in 0 1 4; do for cp in 0 -1; do for rp in pipe equal; do
  [ $cs = 0 ] && [ $cp = -1 ] && continue
  echo "== comm_streams=$cs comm_prio=$cp round_prio=$rp"
  IMT_SLICED_COMM_STREAMS=$cs IMT_SLICED_COMM_PRIO=$cp IMT_SLICED_ROUND_PRIO=$rp timeout -k 10 120 python tools/sliced_costs.py ${WORLDS:-1 2} 2>&1 | grep "^world" | sed -e 's/ through imt_sliced_step//' -e 's/(.*); host/; host/'
done; done; done
