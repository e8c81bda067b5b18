# PMC passes around an arbitrary python script: bash tools/pmc_cmd.sh <script.py> [args...]   (env passes through)
export TMPDIR=/tmp; R=$PWD; cd /tmp
rm -rf $R/gpurun_out/pmcA $R/gpurun_out/pmcB
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $R/gpurun_out/pmcA -- python3 $R/$1 ${@:2} > $R/gpurun_out/pmcA.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS_ATOMIC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $R/gpurun_out/pmcB -- python3 $R/$1 ${@:2} > $R/gpurun_out/pmcB.log 2>&1
