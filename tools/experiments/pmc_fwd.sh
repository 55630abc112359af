#!/bin/bash
# rocprofv3 SQ counters of the split forward kernel (training launch, 4081 x 128 points) for both tile schedules
# (BENERF_FWD_PIPE=0 / 1): MFMA / VALU / LDS / VMEM activity, wait cycles, instruction counts.  Run on the GPU box.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for PIPE in 0 1; do
O=gpurun_out/pmc_fwd$PIPE; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM_WR" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"; do
i=$((i+1))
BENERF_FWD_PIPE=$PIPE timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p -- python tools/experiments/time_fwd_modes.py > /dev/null 2>&1
done
PIPE=$PIPE python - <<'P'
import csv,glob,collections,os
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_fwd%s/p*/**/*counter_collection.csv" % os.environ["PIPE"], recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        name = "fwd_save" if "mlp_fwd_split_kernel<1, true" in k else "fwd_infer" if "mlp_fwd_split_kernel<1, false" in k else None
        if name: acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name,d in acc.items():
    print("PIPE=%s %s" % (os.environ["PIPE"], name))
    for c,v in sorted(d.items()): print("   %-34s %14.0f  (n=%d)"%(c, sum(v)/len(v), len(v)))
P
find $O -name "*.csv" | xargs rm -f
done
