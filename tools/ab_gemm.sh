for rep in 1 2; do
FOCUS_GEMM_WS=1 python tools/gemm_sweep.py 0 2>&1 | grep variant | sed 's/^/WS1 /'
FOCUS_GEMM_WS=0 python tools/gemm_sweep.py 0 2>&1 | grep variant | sed 's/^/WS0 /'
done
