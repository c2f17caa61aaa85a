#!/bin/bash
# ON THE GPU BOX: round 2's library (built from commit 7eaace9 into bitnetmcu_amd/libbitnetmcu_hip_r02ref.so, not tracked) against
# the current one, alternating processes on one box, three rounds.  Did the round-3 changes of the dual kernel (counter block per
# stream, leave protocol, one more kernel argument) cost the headline anything?
cd "$(dirname "$0")/.." || exit 1
for r in 1 2 3; do
  for lib in libbitnetmcu_hip_r02ref.so libbitnetmcu_hip.so; do
    BNM_AB_LIBRARY=$PWD/bitnetmcu_amd/$lib timeout 200 python profiles/old_new_ab.py 2>/dev/null | grep '^{'
  done
done
