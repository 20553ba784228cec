mkdir -p gpurun_out/r4h
echo "== matrix prod: regular (prio)"; python3 tools/mbench.py --check 3 2>&1 | grep -E "matrix_|max rel"
echo "== matrix prod: no prio"; KPAL_HIP_LIBRARY=build/variants/lib_mnoprio.so python3 tools/mbench.py --check 0 2>&1 | grep -E "matrix_"
echo "== matrix sum: regular (prio)"; python3 tools/mbench.py --check 3 --metric sum 2>&1 | grep -E "matrix_|max rel"
echo "== matrix sum: no prio"; KPAL_HIP_LIBRARY=build/variants/lib_mnoprio.so python3 tools/mbench.py --check 0 --metric sum 2>&1 | grep -E "matrix_"
echo "== k=12"; bash tools/ab_run.sh gpurun_out/r4h/abr_k12 --k 12 --reads 100000000 --steps 10 --warmup 2 | grep -v mnoprio
echo "== k=15"; bash tools/ab_run.sh gpurun_out/r4h/abr_k15 --k 15 --reads 100000000 --steps 4 --warmup 1 | grep -v mnoprio
