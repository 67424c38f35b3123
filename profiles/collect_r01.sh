#!/bin/bash
# Run on the GPU box from the repo root (via gpurun): kernel trace + the HBM-traffic counter passes for bench.py's
# default workload (C3).  Counters go in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r01
mkdir -p $OUT
python3 bench.py --steps 5 --warmup 1 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1; echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_write.log 2>&1; echo "write rc=$?"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_BUBBLE_sum --output-format csv -d $OUT/pmc_rdreq -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_rdreq.log 2>&1; echo "rdreq rc=$?"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/pmc_l2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_l2.log 2>&1; echo "l2 rc=$?"
