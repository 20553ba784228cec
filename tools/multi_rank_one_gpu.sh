#!/bin/bash
# On a one-GPU box: bench.py --gpus N with N real ranks that share the GPU (KPAL_BENCH_SHARED_GPU=1) and the test stand-in for RCCL
# (tests/native/fake_rccl.cpp) behind the library communicator -- worlds of 2, 3, 4 and 8, weak / strong, k = 9 / 12 / 13, the
# bin-range merge as the headline.  Prints one summary per run (profiles/r5/multi_rank_one_gpu.log); says nothing about speed.
hipcc -O2 -shared -fPIC -o /tmp/libfake_rccl.so tests/native/fake_rccl.cpp || exit 1
export KPAL_RCCL_LIBRARY=/tmp/libfake_rccl.so KPAL_BENCH_SHARED_GPU=1
run() { name=$1; shift; n=$1; shift; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29700 + RANDOM % 200)) bench.py --gpus $n --steps 2 --warmup 1 --no-cpu "$@" > gpurun_out/multi_rank/multi_$name.json 2> gpurun_out/multi_rank/multi_$name.err; echo "== $name rc=$?"; python - <<PY
import json
try:
    l=json.loads([x for x in open("gpurun_out/multi_rank/multi_$name.json") if x.startswith("{")][0])
    print(l["reduce_mode"], l["n_gpus"], l["scaling"], l["merged_equals_single_stream"], round(l["ms_per_step"],2), l.get("fallback_reason"), l.get("library_rccl_error"))
    for k,v in l["extra"].items(): print("   ", k, round(v["ms_per_step"],2), v["merged_equals_single_stream"], v["checksum_ok"])
except Exception as e:
    print("no line:", e); print(open("gpurun_out/multi_rank/multi_$name.err").read()[-1500:])
PY
}
mkdir -p gpurun_out/multi_rank
run w4 4 --reads 3000000
run w2_k13_range 2 --reads 3000000 --k 13 --range-merge
run w2_strong 2 --reads 5000001 --strong
run w3 3 --reads 2000000
run w8_k9 8 --reads 500000 --k 9
run w2_torch_u32_overlap 2 --reads 3000000 --reduce-via torch --reduce u32 --overlap-reduce
run w4_torch_overlap 4 --reads 2000000 --reduce-via torch --overlap-reduce
