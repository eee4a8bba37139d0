cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d /root/repo/gpurun_out/pmcA -o a -- python3 /root/repo/bench.py --config c3 --steps 10 --warmup 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM --output-format csv -d /root/repo/gpurun_out/pmcB -o b -- python3 /root/repo/bench.py --config c3 --steps 10 --warmup 2 > /dev/null 2>&1
ls /root/repo/gpurun_out/pmcA /root/repo/gpurun_out/pmcB
