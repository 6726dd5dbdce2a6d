#!/bin/bash
# A/B of one attention switch on one box:  bash tools/archive/exp_attn_env.sh CA_ATTN_WPE3   (kernel times at the models' shapes,
# interleaved, then the attention parity tests with the switch on)
V=$1
for i in 1 2; do
  echo "== $V=0"; env $V=0 python tools/dev_attn_perf.py 2>/dev/null | grep -v "^$"
  echo "== $V=1"; env $V=1 python tools/dev_attn_perf.py 2>/dev/null | grep -v "^$"
done
env $V=1 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -k "attention" 2>&1 | tail -2
