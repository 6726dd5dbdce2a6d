(cd _base && python bench.py --no-cpu-baseline --steps 6 --warmup 3 --gemm-breakdown 2>&1 | grep -E "TFLOP/s|^\{" | cut -c1-150 | head -8)
echo ---- new
python bench.py --no-cpu-baseline --no-also --steps 6 --warmup 3 --gemm-breakdown 2>&1 | grep -E "TFLOP/s|^\{" | cut -c1-150 | head -8
