#!/bin/bash
# Runs on the GPU box (via gpurun): smoke, GPU parity tests, kernel lab, bench.
# A step that times out / is killed stops the script (no further GPU work).
set -u
mkdir -p gpurun_out
step() {  # step <seconds> <logfile> <cmd...>
    local secs=$1 log=$2; shift 2
    echo "=== $* (limit ${secs}s)" | tee -a gpurun_out/steps.log
    timeout -k 10 "$secs" "$@" > "gpurun_out/$log" 2>&1
    local rc=$?
    echo "    rc=$rc" | tee -a gpurun_out/steps.log
    tail -n 12 "gpurun_out/$log"
    if [ $rc -ge 124 ]; then echo "step killed/timed out: stopping"; exit $rc; fi
    return 0
}
: > gpurun_out/steps.log
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | sort | uniq -c | head -4 | tee -a gpurun_out/steps.log
nproc | tee -a gpurun_out/steps.log
for s in "$@"; do
  case $s in
    smoke)  step 300 smoke.log python -c "import __graft_entry__ as g; g.smoke()" ;;
    tests)  step 900 pytest_gpu.log python -m pytest tests -x -q -m gpu ;;
    testsall) step 900 pytest_gpu.log python -m pytest tests -q -m gpu ;;
    lab)    step 600 lab.log python tools/lab_csr.py --out gpurun_out/lab.json ;;
    other)  step 900 lab_other.log python tools/lab_other.py ${OTHER_ARGS:-} ;;
    profother) export TMPDIR=/tmp
            step 600 prof_other.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_other -o other -- python3 tools/lab_other.py ${OTHER_ARGS:-} ;;
    prof45) export TMPDIR=/tmp
            step 400 prof4.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof4 -o c4 -- python3 bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline
            step 600 prof5.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof5 -o c5 -- python3 bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline
            ;;
    pmc)    export TMPDIR=/tmp
            step 400 pmc1.log rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc1 -o p -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline
            step 400 pmc2.log rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d gpurun_out/pmc2 -o p -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline
            step 400 pmc3.log rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc3 -o p -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline
            ;;
    hugecsc) step 1000 lab_huge_csc.log python tools/lab_huge_csc.py ${HUGE_ARGS:-} ;;
    hugecoo) step 1000 lab_huge_coo.log python tools/lab_huge_coo.py ${HUGE_ARGS:-} ;;
    huge)   step 1000 lab_huge.log python tools/lab_huge.py ${HUGE_ARGS:-} ;;
    longrows) step 900 lab_longrows.log python tools/lab_longrows.py ;;
    zoo)    step 900 lab_zoo.log python tools/lab_zoo.py ;;
    fem)    step 600 lab_fem.log python tools/lab_fem.py ;;
    small)  step 600 lab_small.log python tools/lab_small.py ${SMALL_ARGS:-} ;;
    align)  step 600 lab_align.log python tools/lab_align.py ${ALIGN_ARGS:-} ;;
    pmc4)   export TMPDIR=/tmp
            step 400 pmc4a.log rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc4a -o p -- python3 bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline --copies 1
            step 400 pmc4b.log rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/pmc4b -o p -- python3 bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline --copies 1
            ;;
    pmc45)  export TMPDIR=/tmp
            step 400 pmc5f.log rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc5f -o p -- python3 bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline
            step 400 pmc5w.log rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc5w -o p -- python3 bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline
            step 400 pmc4f.log rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc4f -o p -- python3 bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline --copies 1
            step 400 pmc4w.log rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc4w -o p -- python3 bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline --copies 1
            ;;
    ab)     step 600 lab_ab.log python tools/lab_ab.py $AB_ARGS ;;
    bench)  step 400 bench.log python bench.py ;;
    rehearse2) step 600 rehearse2.log python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --same-device --config 2 --exchange allgather ;;
    rehearse2h) step 600 rehearse2h.log python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29632 bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --same-device --config 2 --exchange auto ;;
    rehearse2e) step 600 rehearse2e.log python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29634 bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --same-device --config 2 --extras ;;
    rehearse4e) step 600 rehearse4e.log python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29635 bench.py --gpus 4 --steps 5 --warmup 2 --backend gloo --same-device --config 2 ;;
    rehearse1) step 600 rehearse1.log python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29633 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline ;;
    bench1) step 400 bench1.log python bench.py --config 1 ;;
    bench4) step 400 bench4.log python bench.py --config 4 ;;
    bench5) step 600 bench5.log python bench.py --config 5 ;;
    bench2) step 400 bench2.log python bench.py --config 2 ;;
    prof)   # per-kernel time (stats) and, in separate passes, the HBM counters
            export TMPDIR=/tmp
            step 400 prof_stats.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -o bench -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline
            step 400 prof_fetch.log rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline
            step 400 prof_write.log rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline
            ;;
    variants) # VARIANTS="name ..." under spalinalg_amd/lib_var/: COO parity tests + lab per variant ("main" = the shipped library)
            for v in ${VARIANTS:-main}; do
              if [ "$v" = main ]; then unset SPAL_HIP_LIB; else export SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/$v/libspal_hip.so; fi
              step 300 var_${v}_tests.log python -m pytest tests/test_gpu_csc_coo.py -x -q -m gpu -k "${VARIANT_TESTS:-coo or assembl}"
              step 300 var_${v}_lab.log python tools/lab_other.py ${VARIANT_LAB:-coo}
            done
            unset SPAL_HIP_LIB ;;
    *) echo "unknown step $s" ;;
  esac
done
