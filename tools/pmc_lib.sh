# PMC passes for a given engine library: bash tools/pmc_lib.sh <lib.so>
export TMPDIR=/tmp; R=$PWD; cd /tmp
export MM_AMD_LIB=$R/$1 MM_BENCH_NOCHECK=1
rm -rf $R/gpurun_out/pmc1 $R/gpurun_out/pmc2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc1 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/pmc2 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc2.log 2>&1
rm -rf $R/gpurun_out/pmc3
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU --output-format csv -d $R/gpurun_out/pmc3 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc3.log 2>&1
