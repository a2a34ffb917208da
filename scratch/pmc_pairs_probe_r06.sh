# Where does beam_expand_pairs_kernel<4,2> wait?  Counter probe (counters only, one pass per set) on configs[3].
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_pairs_r06
rm -rf $out && mkdir -p $out
rocprofv3 -L > $out/counters_avail.txt 2>&1
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_VMEM_RD" \
           "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU" \
           "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $out -o p_$tag -- python scratch/cfg_beam.py cfg4 > $out/p_$tag.log 2>&1
  echo "$tag rc=$?"
done
find $out -name "*.db" -delete
python - <<'PY'
import csv, glob, collections
vals = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/pmc_pairs_r06/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for name in ("beam_expand_pairs_kernel", "beam_boxes_kernel", "beam_emit_kernel"):
            if name in k and "Li4ELi2" in k.replace("<4, 2>", "Li4ELi2") or (name in k and ("<4, 2>" in k or "<4, 3>" in k)):
                vals[name][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in vals.items():
    print(k, {c: f"{v:.4g}" for c, v in sorted(d.items())})
PY
du -sh $out
