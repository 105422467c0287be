export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/hot1 -- python3 tools/gpu_perf.py 1279 137 10 1 8 > gpurun_out/hot1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/hot2 -- python3 tools/gpu_perf.py 1279 137 10 1 8 > gpurun_out/hot2.log 2>&1
