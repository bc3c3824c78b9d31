#!/bin/bash
# search kernel: the irrelevance bound tested on 96 or 128 bits (PRS_PREFILTER_96_LIMIT = largest bound tested on 96 bits) on the side legs
for lim in 32 0 64; do
  for leg in tum kitti_real euroc; do
    PRS_PREFILTER_96_LIMIT=$lim python tools/bench_leg.py $leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('limit $lim %-10s %8.0f frames/s  search %.2f ms  gn %.2f  parity %s' % ('$leg', d['value'], d['ms_per_kernel']['align_kernel (search)'], d['ms_per_kernel']['gn_kernel'], d['parity']['correspondences_bit_exact']))"
  done
done
