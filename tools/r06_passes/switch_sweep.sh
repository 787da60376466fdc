#!/bin/bash
# Round 6: the GPU suite under every run-time switch a user may flip (the alternative paths must stay green after the round's kernel edits)
cd "$GRAFT_REPO_ROOT"
for sw in "CPN_HIP_GRAPH=0" "CPN_S1F=0" "CPN_PAIR=0" "CPN_BRIDGE=0" "CPN_HOIST=0" "CPN_S1Q=0" "CPN_BLPHASE=0"; do
  echo "== $sw"; env $sw timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -2
done
