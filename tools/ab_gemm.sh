for rep in 1 2; do
FOCUS_GEMM_WS=1 FOCUS_GEMM_WS_BM=256 python tools/gemm_sweep.py 0 2>&1 | grep variant | sed 's/^/WS256 /'
FOCUS_GEMM_WS=1 FOCUS_GEMM_WS_BM=128 python tools/gemm_sweep.py 0 2>&1 | grep variant | sed 's/^/WS128 /'
FOCUS_GEMM_WS=0 python tools/gemm_sweep.py 0 2>&1 | grep variant | sed 's/^/UNI   /'
done
