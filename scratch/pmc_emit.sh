cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_emit; rm -rf $out; mkdir -p $out
for set in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $set --output-format csv -d $out -o $tag -- python scratch/cfg_beam.py cfg4 > /dev/null 2>&1
done
python - <<'P'
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/pmc_emit/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "beam_emit" in k or "expand_clustered_last" in k or "trace_filter_pair" in k:
            tot[k.split("(")[0][:70]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in tot.items():
    print(k)
    for c, x in sorted(v.items()):
        print("   %-24s %.4g" % (c, x))
P
