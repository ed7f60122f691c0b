#!/bin/bash
# rocprofv3 passes over the bench workload.  Usage: tools/profile.sh <tag>   (outputs under gpurun_out/prof_<tag>/)
# Counter passes are separate runs with --kernel-trace only (MI355X_MICROARCH.md, HBM / rocprofv3 section).
tag=${1:-run}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp
export TMPDIR=/tmp
# BENCH_ARGS: another workload than configs[2], e.g. BENCH_ARGS="--blocks 9 --channels 128 --batch 32" (configs[1])
cmd="python $root/bench.py --steps 3 --warmup 1 --lean $BENCH_ARGS"
# 40 timed passes (+ 5 warm-up): the AVERAGE launch duration of the dominant kernel in kernel_stats.csv is what the bench line's
# roofline.launch_us must agree with (12 launches with one outlier did not)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python $root/bench.py --steps 40 --warmup 5 --lean $BENCH_ARGS > $out/trace.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/pmc_sq -o p --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- $cmd > $out/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/pmc_fetch -o p --pmc FETCH_SIZE GRBM_GUI_ACTIVE -- $cmd > $out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/pmc_write -o p --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- $cmd > $out/pmc_write.log 2>&1
# round 6: what a CU pulls through the fabric ("fabric bytes per CU" wants a counter file).  Separate passes; a
# counter name this rocprofv3 does not know fails its own pass only (see the .log beside it)
rocprofv3 --kernel-trace --output-format csv -d $out/pmc_tcp -o p --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum -- $cmd > $out/pmc_tcp.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/pmc_tcc -o p --pmc TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum -- $cmd > $out/pmc_tcc.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/pmc_wait -o p --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM -- $cmd > $out/pmc_wait.log 2>&1
tail -1 $out/trace.log
