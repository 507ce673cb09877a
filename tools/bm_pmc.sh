# development aid: SQ counters of the bitmask kernels (mode from $VX_BM_MODE) on the prior-box candidates
export VX_BM_MODE=${VX_BM_MODE:-2}
for c in retinaface ssd; do
  bash tools/prof_pmc.sh sq_$c "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" tools/diag_nms_prior.py $c | grep -E "vx_k_bm|vx_k_nms_image|vx_k_part"
done
