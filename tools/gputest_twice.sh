#!/bin/bash
# End-of-round check on the GPU box: the whole -m gpu suite, no -x, twice back to back in one lease (ordering / stream races
# show up in the second run).  Output -> gpurun_out/gputest.log (copied to profiles/gputest_<round>_end.log).
mkdir -p gpurun_out
{
echo "# gpurun: python -m pytest tests -q -m gpu, whole suite, no -x, run twice back to back in one lease"
echo "## run A"; timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -12
echo "## run B"; timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -12
} > gpurun_out/gputest.log
tail -3 gpurun_out/gputest.log
