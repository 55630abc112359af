cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_dx; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM_WR" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"; do
i=$((i+1))
timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p -- python tools/experiments/time_mlp_kernels.py 4081 128 3 > /dev/null 2>&1
done
python - <<'P'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_dx/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        name = ("dx" if ("mlp_bwd_split" in k or "mlp_bwd_f16" in k) else "fwd (training launch)" if ("mlp_fwd_split_kernel<1, 2" in k or "mlp_fwd_split_kernel<1, 1" in k)
                else "fwd (inference launch)" if "mlp_fwd_split_kernel<1, 0" in k else "dw_big" if ("dw_split_big" in k or "dw_f16_big" in k)
                else "dw_small" if ("dw_split_small" in k or "dw_f16_small" in k) else None)
        if name: acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name,d in sorted(acc.items()):
    m = {c: sum(v)/len(v) for c,v in d.items()}
    print(name)
    for c,v in sorted(m.items()): print("   %-34s %14.0f  (n=%d)"%(c, v, len(d[c])))
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:     # per-wave-cycle fractions (SQ_WAVE_CYCLES: resident-wave cycles summed over the chip; SQ_BUSY_CYCLES x SIMDs: cycles the kernel ran)
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"):
            if c in m: print("   %-34s %14.3f  of the waves' cycles" % (c + " / WAVE_CYCLES", m[c] / wc))
    if "SQ_INSTS_MFMA" in m:
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU"):
            if c in m: print("   %-34s %14.2f  per MFMA instruction" % (c, m[c] / m["SQ_INSTS_MFMA"]))
P
find $O -name "*.csv" | xargs rm -f
