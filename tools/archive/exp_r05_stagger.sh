#!/bin/bash
# Experiment: kernel X with every second workgroup of an XCD starting late (CA_X_STAGGER_US) - do the launches gain when
# the CUs' store bursts stop coinciding?  FFN1 forward plain / + GELU + dropout (two outputs), same box, two rounds.
for round in 1 2; do
for us in 0 3 6 12 22; do
  for epi in 0 1; do
    echo -n "stagger ${us} us r$round: "; CA_X_STAGGER_US=$us python tools/dev_gemm_perf.py 3992 7680 1920 0 0 20 3 0 0 $epi 2>&1 | tail -1
  done
done; done
