#!/bin/bash
# Run on the GPU box from the repo root (via gpurun): for one bench.py workload, the bench line itself, a kernel trace
# with stats, and the counter passes (each in its own run: --kernel-trace/--stats never together with --pmc).
#   profiles/collect_r04.sh <workload> [extra bench.py flags]
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
W=${1:-c3}; shift
OUT=gpurun_out/r04/$W
mkdir -p $OUT
B="python3 bench.py --workload $W $@"
$B --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
# (the traced and counted runs skip the two-directional sibling: its launches carry the headline kernel's name)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B --steps 3 --warmup 1 --no-cpu-baseline --no-two-directional-row > $OUT/trace.log 2>&1; echo "trace rc=$?"
P="--steps 1 --warmup 0 --no-cpu-baseline --no-two-directional-row"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B $P > $OUT/pmc_fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B $P > $OUT/pmc_write.log 2>&1; echo "write rc=$?"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/pmc_l2 -- $B $P > $OUT/pmc_l2.log 2>&1; echo "l2 rc=$?"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/sq1 -- $B $P > $OUT/sq1.log 2>&1; echo "sq1 rc=$?"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/sq2 -- $B $P > $OUT/sq2.log 2>&1; echo "sq2 rc=$?"
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq3 -- $B $P > $OUT/sq3.log 2>&1; echo "sq3 rc=$?"
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/sq4 -- $B $P > $OUT/sq4.log 2>&1; echo "sq4 rc=$?"
# what the counters were collected on: a content hash of the library's sources as they are on this box (bench.py refuses to
# quote the counters once the sources differ)
cat all-pairs-similarity_amd/csrc/*.hip all-pairs-similarity_amd/csrc/*.hpp include/apss.h | sha256sum | cut -d' ' -f1 > $OUT/csrc.sha256
