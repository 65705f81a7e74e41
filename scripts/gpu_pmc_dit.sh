export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rm -rf /tmp/pmc_dit
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_dit -o pmc -- python $R/scripts/bench_dit.py 16 3 fp16 > $R/gpurun_out/r3_pmc_dit.log 2>&1
echo rc=$?
python $R/scripts/pmc_summary.py pmc $(find /tmp/pmc_dit -name "*counter_collection.csv") > $R/gpurun_out/r3_pmc_sq_dit_fp16.json 2> $R/gpurun_out/r3_pmc_dit.err
python - <<PY
import json
d=json.load(open("$R/gpurun_out/r3_pmc_sq_dit_fp16.json"))
for k,v in d["kernels"].items():
    if not any(t in k for t in ("gemm_hh","flash_attn_hh","gemm_f16_mfma","ln_modulate")): continue
    g=v.get("GRBM_GUI_ACTIVE",{}).get("mean",0); m=v.get("SQ_VALU_MFMA_BUSY_CYCLES",{}).get("mean",0)
    bc=v.get("SQ_LDS_BANK_CONFLICT",{}).get("mean",0); ia=v.get("SQ_LDS_IDX_ACTIVE",{}).get("mean",1)
    print(f"{k[:60]:60s} disp {v['GRBM_GUI_ACTIVE']['dispatches']:5d} mfma_busy {m/(g*128) if g else 0:.3f} lds_conflict_share {bc/ia if ia else 0:.3f}")
PY
