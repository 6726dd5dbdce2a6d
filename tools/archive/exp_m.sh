#!/bin/bash
# Kernel M (128x128 tile on 8 waves) against S / L / the automatic choice: the Whisper decoder's rows (M = 904) and the
# N = d GEMMs of the d = 1024 / 1280 models at M = 3992.
cd "$(dirname "$0")/.."
python tools/archive/dev_dec_gemm.py 904 2>&1 | grep -v amdgpu
python tools/archive/dev_dec_gemm.py 3992 2>&1 | grep -v amdgpu
