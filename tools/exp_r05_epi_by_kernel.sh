for shp in "3992 7680 1920" "12000 5120 1280" "31936 4096 1024"; do
for force in 1 2 3; do
for epi in 0 1; do
python tools/dev_gemm_perf.py $shp 0 0 20 $force 0 0 $epi 2>&1 | tail -1
done; done; done
