for rep in 1 2; do
FOCUS_GEMM_LDS_EPI=1 python tools/gemm_sweep.py 0 2>&1 | grep variant | sed 's/^/LDS1 /'
FOCUS_GEMM_LDS_EPI=0 python tools/gemm_sweep.py 0 2>&1 | grep variant | sed 's/^/LDS0 /'
done
FOCUS_GEMM_LDS_EPI=0 python tools/gemm_sweep.py 4 6 2>&1 | grep variant | sed 's/^/LDS0 /'
