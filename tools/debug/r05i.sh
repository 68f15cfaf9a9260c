mkdir -p gpurun_out/r05i
for k in "" adj16; do
  BARTRT_KERNEL=$k bash tools/pmc_pass.sh a$k "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES" --walkers 1 | tee -a gpurun_out/r05i/pmc.jsonl
  BARTRT_KERNEL=$k bash tools/pmc_pass.sh b$k "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" --walkers 1 | tee -a gpurun_out/r05i/pmc.jsonl
  BARTRT_KERNEL=$k bash tools/pmc_pass.sh c$k "SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA TA_BUSY_avr" --walkers 1 | tee -a gpurun_out/r05i/pmc.jsonl
done
